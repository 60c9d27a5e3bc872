#!/usr/bin/env python
"""single cases of tools/parity_sweep_crf.py by sweep index: max|dQ| vs the oracle per iteration count"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import krahenbuhl2013
from dsrg_amd import synthetic as S
from oracle import oracle as O
for it in [int(a) for a in sys.argv[1:]]:
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
    out = []
    for iters in (0, 1, 2, 10):
        want = O.CRF(im, un, maxiter=iters, scale_factor=scale)
        got = krahenbuhl2013.CRF(im, un, maxiter=iters, scale_factor=scale)
        out.append("%d it: %.2e" % (iters, float(np.abs(got - want).max())))
    print("case %d: %dx%d C=%d scale=%g %s | %s" % (it, H, W, C, scale, kind, "; ".join(out)))
