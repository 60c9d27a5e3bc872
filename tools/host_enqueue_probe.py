#!/usr/bin/env python
"""host time to enqueue a train step against its wall time (is the launch sequence ahead of the GPU?):  python tools/host_enqueue_probe.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from dsrg_amd import synthetic as S
from dsrg_amd.trainer import DSRGTrainer
dev = torch.device("cuda", 0)
tr = DSRGTrainer(dev, seed=0)
b = S.make_batch(1, 16)
d = lambda a: torch.from_numpy(a).to(dev)
im, la, cu = d(b["images"]), d(b["labels"]), d(b["cues"])
for _ in range(8):
    tr.step(im, la, cu)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.step(im, la, cu)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms per step, wall %.2f ms per step" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
