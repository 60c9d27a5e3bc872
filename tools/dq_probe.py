#!/usr/bin/env python
"""Where do the HIP and oracle CRF marginals differ most?  Prints every batch of the parity sweep whose worst pixel
exceeds 2e-5, with the two label columns of that pixel (diagnostic for tools/parity_sweep.py)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops, synthetic as S
from oracle import oracle as O
B, C, H, W = 8, 21, 41, 41
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
kinds = ["smooth", "noise", "dark_corner"]
for it in range(60):
    rng = np.random.default_rng(10_000 + it)
    images = S.make_images(rng, B, kind=kinds[it % 3])
    gain, sigma = float(rng.uniform(2, 40)), float(rng.uniform(1, 8))
    logits = S.make_logits(rng, B, C, H, W, gain=gain, sigma=sigma)
    probs = O.softmax_forward(logits)
    refined, logq = O.crf_refine_batch(probs.copy(), images, 12.0, 10)
    r2, lq2 = ops.crf_refine(dev(probs.copy()), dev(images))
    r2 = r2.cpu().numpy()
    d = np.abs(r2 - refined)
    if d.max() > 2e-5:
        idx = np.unravel_index(d.argmax(), d.shape)
        b, c, y, x = idx
        print("batch %d kind %s gain %.1f sigma %.1f: max|dQ| %.3e at %s  oracle %.6f hip %.6f ; n(>1e-5)=%d; per-image max %s" % (
            it, kinds[it % 3], gain, sigma, d.max(), idx, refined[idx], r2[idx], int((d > 1e-5).sum()),
            np.array2string(d.reshape(B, -1).max(1), precision=1)))
        print("   pixel column oracle:", np.array2string(refined[b, :, y, x], precision=5, max_line_width=250))
        print("   pixel column hip   :", np.array2string(r2[b, :, y, x], precision=5, max_line_width=250))
