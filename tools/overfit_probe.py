#!/usr/bin/env python
"""Sanity run: the full train-s step repeated on ONE synthetic batch must drive both losses down (gradient chain,
Caffe-style SGD and bf16 autocast all in the loop).  Prints the losses every 20 steps."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import synthetic as S
from dsrg_amd.trainer import DSRGTrainer

torch.manual_seed(0)
dev = torch.device("cuda", 0)
b = S.make_batch(7, 8)
images, labels, cues = (torch.from_numpy(b[k]).to(dev) for k in ("images", "labels", "cues"))
tr = DSRGTrainer(dev)
hist = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    losses = tr.step(images, labels, cues)
    if it % 20 == 0 or it < 3:
        l = [float(x) for x in losses.detach().cpu()]
        hist.append(l)
        print("step %4d  loss-Seed %.4f  loss-Constrain %.4f" % (it, l[0], l[1]), flush=True)
assert all(np.isfinite(h).all() for h in hist), "non-finite loss"
print("seed loss %.4f -> %.4f, constrain loss %.4f -> %.4f" % (hist[0][0], hist[-1][0], hist[0][1], hist[-1][1]))
