#!/usr/bin/env python
"""host time to enqueue a train step, phase by phase, against its wall time (is the launch sequence ahead of the GPU?):
python tools/host_enqueue_probe.py"""
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
# the same step, phase by phase (host clock only; nothing waits for the device)
from dsrg_amd.ops import crf_prepare
acc = {}
def tick(k, t):
    acc[k] = acc.get(k, 0.0) + time.perf_counter() - t
torch.cuda.synchronize()
w0 = time.perf_counter()
for _ in range(20):
    t = time.perf_counter(); tr.opt.zero_grad(); x = im.contiguous(memory_format=torch.channels_last)
    main = torch.cuda.current_stream(); tr.side.wait_stream(main)
    with torch.cuda.stream(tr.side):
        crf_prepare(im, cu.shape[1], cu.shape[2], cu.shape[3])
    tick("prepare", t); t = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = tr.model(x)
    logits = logits.float().contiguous()
    tick("forward", t); t = time.perf_counter()
    torch.cuda.current_stream().wait_stream(tr.side)
    total, losses = tr.loss_fn(logits, im, la, cu, prepared=True)
    tick("loss", t); t = time.perf_counter()
    total.backward()
    tick("backward", t); t = time.perf_counter()
    tr.opt.step()
    tick("sgd", t)
w1 = time.perf_counter()
torch.cuda.synchronize()
w2 = time.perf_counter()
print("phases (host ms per step):", {k: round(v / 20 * 1e3, 3) for k, v in acc.items()}, "sum %.2f, wall %.2f" % ((w1 - w0) / 20 * 1e3, (w2 - w0) / 20 * 1e3))
