#!/usr/bin/env python
"""Randomised parity sweep on an MI355X: many seeded batches through the fused supervision step against the CPU oracle.
Reports the worst CRF marginal difference, the number of seed pixels that differ, and the worst loss / gradient error."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops, synthetic as S
from oracle import oracle as O

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, C, H, W = 8, 21, 41, 41
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
worst_q = worst_l = worst_g = 0.0
flips = total = 0
kinds = ["smooth", "noise", "dark_corner"]
for it in range(n_batches):
    rng = np.random.default_rng(10_000 + it)
    images = S.make_images(rng, B, kind=kinds[it % 3])
    logits = S.make_logits(rng, B, C, H, W, gain=float(rng.uniform(2, 40)), sigma=float(rng.uniform(1, 8)))
    labels, cues = S.make_labels_cues(rng, B, C, H, W)
    losses, grad, blobs = ops.supervision_step(dev(logits), dev(images), dev(labels), dev(cues), want_blobs=True)
    probs = O.softmax_forward(logits)
    refined, logq = O.crf_refine_batch(probs, images, 12.0, 10)
    seeds = O.srg_grow_batch(labels, cues, refined)
    worst_q = max(worst_q, float(np.abs(np.exp(blobs["logq"].cpu().numpy()) - refined).max()))
    got = blobs["seeds"].cpu().numpy()
    flips += int((got != seeds).sum()); total += seeds.size
    if (got == seeds).all():
        l1, g1 = O.seed_loss(probs, seeds)
        l2, g2, g3 = O.constrain_loss(probs, logq)
        want = O.softmax_backward(logits, g1 + g2 + O.crf_layer_backward(refined, g3))
        worst_l = max(worst_l, abs(losses[0].item() - l1) / max(1, abs(l1)), abs(losses[1].item() - l2) / max(1, abs(l2)))
        worst_g = max(worst_g, float(np.abs(grad.cpu().numpy() - want).max() / np.abs(want).max()))
print("%d batches of %d images: max|dQ| %.2e, seed pixels differing %d of %d, worst relative loss error %.2e, "
      "worst gradient error / max|grad| %.2e" % (n_batches, B, worst_q, flips, total, worst_l, worst_g))
