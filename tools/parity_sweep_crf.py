#!/usr/bin/env python
"""Randomised parity sweep of krahenbuhl2013.CRF() on an MI355X against the CPU oracle over random map sizes (both the
LDS-resident and the global-memory path), label counts, scale factors and unary kinds (probabilities / log-probabilities)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import krahenbuhl2013
from dsrg_amd import synthetic as S
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = 0.0
rows = []
for it in range(n):
    rng = np.random.default_rng(20_000 + it)
    H, W = (int(rng.integers(1, 71)), int(rng.integers(1, 71))) if it % 2 == 0 else (int(rng.integers(60, 180)), int(rng.integers(60, 220)))
    C = int(rng.choice([2, 3, 7, 21, 21, 21, 33]))
    scale = float(rng.choice([1.0, 3.0, 12.0]))
    kind = ["smooth", "noise", "dark_corner"][it % 3]
    img = S.make_images(rng, 1, size=max(H, W, 8), kind=kind)[0, :, :H, :W] + S.MEAN_PIXEL[:, None, None]
    im = np.ascontiguousarray(np.transpose(img, (1, 2, 0)))
    logits = S.make_logits(rng, 1, C, H, W, gain=float(rng.uniform(2, 40)), sigma=float(rng.uniform(1, 10)))
    un = np.ascontiguousarray(np.transpose(np.maximum(O.softmax_forward(logits)[0], 1e-5), (1, 2, 0)))
    if rng.random() < 0.5:
        un = np.log(un)
    want = O.CRF(im, un, scale_factor=scale)
    got = krahenbuhl2013.CRF(im, un, scale_factor=scale)
    d = float(np.abs(got - want).max()) if np.isfinite(got).all() else float("inf")
    agree = float((got.argmax(2) == want.argmax(2)).mean())
    rows.append((d, H, W, C, scale, kind, agree))
    worst = max(worst, d)
rows.sort(reverse=True)
for r in rows[:int(os.environ.get("SWEEP_SHOW", "5"))]:
    print("max|dQ| %.2e at %dx%d C=%d scale=%g %s (argmax agreement %.5f)" % r)
print("%d CRF() calls: worst max|dQ| %.2e, worst argmax agreement %.5f" % (n, worst, min(r[6] for r in rows)))

# ---- the batched objects (dsrg_crf_create_batch): random shapes and batch sizes, every image against the oracle AND against the
# one-image object of the same path (bit for bit)
from dsrg_amd.crf import CRF_device_batch, DenseCRF
nb_cases = max(4, n // 4)
worst_b, unequal = 0.0, 0
for it in range(nb_cases):
    rng = np.random.default_rng(40_000 + it)
    H, W, B = int(rng.integers(20, 140)), int(rng.integers(20, 160)), int(rng.integers(2, 9))
    C = int(rng.choice([2, 5, 21, 21]))
    scale = float(rng.choice([1.0, 1.0, 3.0]))
    ims, uns = [], []
    for k in range(B):
        kind = ["smooth", "noise", "dark_corner"][(it + k) % 3]
        img = S.make_images(rng, 1, size=max(H, W, 8), kind=kind)[0, :, :H, :W] + S.MEAN_PIXEL[:, None, None]
        ims.append(np.ascontiguousarray(np.transpose(img, (1, 2, 0))).astype(np.uint8))
        logits = S.make_logits(rng, 1, C, H, W, gain=float(rng.uniform(2, 40)), sigma=float(rng.uniform(1, 10)))
        uns.append(np.ascontiguousarray(np.transpose(np.log(np.maximum(O.softmax_forward(logits)[0], 1e-5)), (1, 2, 0))).astype(np.float32))
    ti, tu = torch.from_numpy(np.stack(ims)).cuda(), torch.from_numpy(np.stack(uns)).cuda()
    got = CRF_device_batch(ti, tu, scale_factor=scale)
    for k in range(B):
        crf = DenseCRF(W, H, C, nimages=1)
        crf.set_unary_energy((-tu[k]).contiguous())
        crf.add_pairwise_energy(10, 80.0 / scale, 80.0 / scale, 13, 13, 13, 3, 3.0 / scale, 3.0 / scale, ti[k].contiguous())
        one = crf.inference(10, out=torch.empty((H, W, C), dtype=torch.float32, device="cuda"))
        unequal += int(not torch.equal(got[k], one))
        if k in (0, B - 1):
            worst_b = max(worst_b, float(np.abs(got[k].cpu().numpy() - O.CRF(ims[k], uns[k], scale_factor=scale)).max()))
print("%d batched calls (2..8 images, 20..160 pixels a side): %d images differ from their single-image result; worst max|dQ| vs the "
      "oracle %.2e" % (nb_cases, unequal, worst_b))
