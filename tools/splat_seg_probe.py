#!/usr/bin/env python
"""the global-memory CRF's splat with rows cut into segments (DSRG_SPLAT_SEG entries each, summed per segment and combined) against
whole rows (one ordered sum per vertex, the reference's order exactly): milliseconds per image on a natural-like, a flat and a
two-colour image — the flat ones have the longest rows a lattice can have"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import synthetic as S
from dsrg_amd.crf import CRF_device
from oracle import oracle as O
H, W, C = 321, 321, 21
rng = np.random.default_rng(5)
logits = S.make_logits(rng, 1, C, H, W, gain=12.0, sigma=12.0)[0]
e = np.exp(logits - logits.max(0, keepdims=True))
un_np = np.ascontiguousarray(np.log(np.maximum(e / e.sum(0, keepdims=True), 1e-5)).transpose(1, 2, 0).astype(np.float32))
un = torch.from_numpy(un_np).cuda()
imgs = {"smooth": (S.make_images(rng, 1, size=H)[0] + S.MEAN_PIXEL[:, None, None]).transpose(1, 2, 0).astype(np.uint8),
        "flat": np.full((H, W, 3), 117, np.uint8)}
two = np.full((H, W, 3), 40, np.uint8); two[:, W // 2:] = 200
imgs["two-colour"] = two
out = []
for name, im_np in imgs.items():
    im = torch.from_numpy(np.ascontiguousarray(im_np)).cuda()
    for _ in range(3):
        q = CRF_device(im, un, scale_factor=1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        q = CRF_device(im, un, scale_factor=1.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    want = O.CRF(im_np, un_np, scale_factor=1.0)
    out.append("%s %.2f ms (max|dQ| %.1e)" % (name, ms, float(np.abs(q.cpu().numpy() - want).max())))
print("DSRG_SPLAT_SEG=%s: " % os.environ.get("DSRG_SPLAT_SEG", "default") + "; ".join(out))
