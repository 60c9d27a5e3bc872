#!/usr/bin/env python
"""Randomised parity sweep over SHAPES on an MI355X (parity_sweep.py fixes 8 x 21 x 41 x 41 and varies the data):
  srg     seeded region growing alone: random batch, label count (2..96), map size, adversarial marginals (values exactly at
          the thresholds, cues of several classes on one pixel, images without background) — bit-exact
  filter  the lattice normalisation and one DenseKernel::filter application on both kernels, random map sizes of the
          LDS-resident path, label counts 1..96, scales 1/3/12 — <= 2 ulp
  fused   the whole supervision step on random (B, C, H, W) against the oracle layer by layer
usage: parity_sweep_shapes.py [n_per_part] [parts]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops, synthetic as S
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
parts = sys.argv[2].split(",") if len(sys.argv) > 2 else ["srg", "filter", "fused"]
dev = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).cuda().to(dt) if dt else torch.from_numpy(np.ascontiguousarray(a)).cuda()
bad = 0


def ulps(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return int(np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)).max())


def random_labels_cues(rng, B, C, H, W):
    labels = np.zeros((B, 1, 1, C), np.float32)
    cues = np.zeros((B, C, H, W), np.float32)
    for b in range(B):
        npres = int(rng.integers(1, min(C, 7) + 1))
        pres = rng.choice(C, size=npres, replace=False)
        if rng.random() < 0.7:
            pres[0] = 0                                           # background usually present, not always
        pres = np.unique(pres)
        labels[b, 0, 0, pres] = 1.0
        for c in rng.choice(C, size=min(C, npres + 2), replace=False):    # cues of absent classes too: copied, never grown
            for _ in range(int(rng.integers(0, 4))):
                h, w = int(rng.integers(1, max(2, H // 3 + 1))), int(rng.integers(1, max(2, W // 3 + 1)))
                y, x = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
                cues[b, c, y:y + h, x:x + w] = 1.0
    return labels, cues


if "srg" in parts:
    diff = total = 0
    for it in range(n):
        rng = np.random.default_rng(30_000 + it)
        B, C = int(rng.integers(1, 5)), int(rng.choice([2, 3, 5, 21, 21, 40, 64, 65, 81, 96]))
        H, W = (int(rng.integers(1, 100)), int(rng.integers(1, 100))) if it % 5 else (int(rng.integers(100, 230)), int(rng.integers(100, 230)))
        labels, cues = random_labels_cues(rng, B, C, H, W)
        refined = np.empty((B, C, H, W))
        for b in range(B):
            z = S.make_logits(rng, 1, C, max(H, 2), max(W, 2), gain=float(rng.uniform(5, 80)), sigma=float(rng.uniform(0.7, 6)))[0, :, :H, :W]
            e = np.exp(z.astype(np.float64) - z.max(0, keepdims=True))
            refined[b] = e / e.sum(0, keepdims=True)
        # a tenth of the pixels sit exactly on a threshold (the comparisons are strict in the reference)
        m = rng.random((B, C, H, W)) < 0.1
        refined[m] = rng.choice([0.99, 0.85, np.nextafter(0.99, 1), np.nextafter(0.85, 1), np.nextafter(0.85, 0)], size=int(m.sum()))
        th1, th2 = (0.99, 0.85) if it % 4 else (float(rng.uniform(0.5, 0.99)), float(rng.uniform(0.3, 0.9)))
        want = O.srg_grow_batch(labels, cues, refined, th1, th2)
        got = ops.srg_grow(dev(labels), dev(cues), dev(refined, torch.float64), th1, th2).cpu().numpy()
        d = int((got != want).sum())
        if d:
            print("SRG MISMATCH seed %d: B=%d C=%d %dx%d th %.3f/%.3f: %d elements" % (30_000 + it, B, C, H, W, th1, th2, d))
        diff += d; total += want.size
    print("srg: %d random batches, %d of %d elements differ" % (n, diff, total))
    bad += diff

if "filter" in parts:
    worst = 0
    for it in range(n):
        rng = np.random.default_rng(40_000 + it)
        B, C = int(rng.integers(1, 4)), int(rng.choice([1, 2, 3, 4, 7, 21, 21, 33, 81, 96]))
        H, W = int(rng.integers(1, 71)), int(rng.integers(1, 71))
        scale = float(rng.choice([1.0, 3.0, 12.0]))
        kind = ["smooth", "noise", "dark_corner"][it % 3]
        N = H * W
        img = S.make_images(rng, B, size=max(H, W, 8), kind=kind)[:, :, :H, :W] + S.MEAN_PIXEL[None, :, None, None]
        im_u8 = np.ascontiguousarray(np.transpose(img, (0, 2, 3, 1))).astype(np.uint8)
        q = O.softmax_forward(S.make_logits(rng, B, C, max(H, 2), max(W, 2), gain=8.0)[:, :, :H, :W])
        try:
            ctx = ops.Context(B, C, H, W)
        except Exception as e:                                      # maps beyond the LDS-resident path have no filter_once
            print("filter: seed %d %dx%d skipped: %s" % (40_000 + it, H, W, str(e)[:100]))
            continue
        ops.crf_meanfield(dev(q), dev(im_u8), 1, scale, ctx=ctx)
        got = {k: ctx.filter_once(k, dev(q)).cpu().numpy() for k in (0, 1)}
        for b in range(B):
            oc = O.DenseCRF(W, H, C)
            oc.add_pairwise_energy(10, 80 / scale, 80 / scale, 13, 13, 13, 3, 3 / scale, 3 / scale, im_u8[b].ravel())
            qlf = np.ascontiguousarray(np.transpose(q[b].reshape(C, N), (1, 0)))
            for k in (0, 1):
                u1 = ulps(ctx.lattice_norm(k, b if k == 1 else 0), oc.lattice_norm(k))
                u2 = ulps(np.transpose(got[k][b].reshape(C, N), (1, 0)), oc.kernel_filter(k, qlf))
                if max(u1, u2) > 2:
                    print("FILTER MISMATCH seed %d: B=%d C=%d %dx%d scale %g %s image %d kernel %d: norm %d ulp, filter %d ulp"
                          % (40_000 + it, B, C, H, W, scale, kind, b, k, u1, u2))
                    bad += 1
                worst = max(worst, u1, u2)
        del ctx
    print("filter: %d random shapes, worst %d ulp" % (n, worst))

if "fused" in parts:
    worst_q = worst_l = worst_g = 0.0
    flips = total = 0
    for it in range(n):
        rng = np.random.default_rng(50_000 + it)
        B, C = int(rng.integers(1, 7)), int(rng.choice([2, 3, 5, 21, 21, 21, 30, 64, 81, 96]))
        H, W = int(rng.integers(2, 66)), int(rng.integers(2, 66))
        size = 8 * (max(H, W) - 1) + 1
        images = np.ascontiguousarray(S.make_images(rng, B, size=size, kind=["smooth", "noise", "dark_corner"][it % 3])[:, :, :8 * (H - 1) + 1, :8 * (W - 1) + 1])
        logits = S.make_logits(rng, B, C, H, W, gain=float(rng.uniform(2, 60)), sigma=float(rng.uniform(1, 8)))
        labels, cues = random_labels_cues(rng, B, C, H, W)
        try:
            losses, grad, blobs = ops.supervision_step(dev(logits), dev(images), dev(labels), dev(cues), want_blobs=True)
        except Exception as e:                                      # a shape the path refuses must say so, not crash later
            print("fused: seed %d B=%d C=%d %dx%d refused: %s" % (50_000 + it, B, C, H, W, str(e)[:120]))
            continue
        probs = O.softmax_forward(logits)
        refined, logq = O.crf_refine_batch(probs, images, 12.0, 10)
        seeds = O.srg_grow_batch(labels, cues, refined)
        dq = float(np.abs(np.exp(blobs["logq"].cpu().numpy()) - refined).max())
        got = blobs["seeds"].cpu().numpy()
        f = int((got != seeds).sum())
        if dq > 1e-4 or f:
            print("FUSED MISMATCH seed %d: B=%d C=%d %dx%d: max|dQ| %.2e, %d seed elements" % (50_000 + it, B, C, H, W, dq, f))
            bad += 1
        worst_q = max(worst_q, dq); flips += f; total += seeds.size
        if not f:
            l1, g1 = O.seed_loss(probs, seeds)
            l2, g2, g3 = O.constrain_loss(probs, logq)
            want = O.softmax_backward(logits, g1 + g2 + O.crf_layer_backward(refined, g3))
            el = max(abs(losses[0].item() - l1) / max(1, abs(l1)), abs(losses[1].item() - l2) / max(1, abs(l2)))
            eg = float(np.abs(grad.cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-30))
            if el > 1e-4 or eg > 1e-3:
                print("FUSED LOSS/GRAD seed %d: B=%d C=%d %dx%d: loss %.2e grad %.2e" % (50_000 + it, B, C, H, W, el, eg))
                bad += 1
            worst_l, worst_g = max(worst_l, el), max(worst_g, eg)
    print("fused: %d random shapes: max|dQ| %.2e, seed elements differing %d of %d, worst relative loss error %.2e, "
          "worst gradient error / max|grad| %.2e" % (n, worst_q, flips, total, worst_l, worst_g))
print("SWEEP", "FAILED (%d)" % bad if bad else "clean")
sys.stdout.flush()
os._exit(1 if bad else 0)
