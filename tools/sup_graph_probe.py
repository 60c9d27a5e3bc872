#!/usr/bin/env python
"""Supervision step launched eagerly vs replayed from a hipGraph (torch.cuda.CUDAGraph around ops.supervision_step):
what do the ~30 kernel boundaries of a step cost when the host is out of the way?  usage: sup_graph_probe.py [B]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops, synthetic as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
b = S.make_batch(1000, B)
d = lambda a: torch.from_numpy(a).cuda()
logits, images, labels, cues = d(b["logits"]), d(b["images"]), d(b["labels"]), d(b["cues"])
ctx = ops.get_context(B, 21, 41, 41)
for _ in range(5):
    out = ops.supervision_step(logits, images, labels, cues, ctx=ctx)
torch.cuda.synchronize()
def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
eager = timeit(lambda: ops.supervision_step(logits, images, labels, cues, ctx=ctx))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        ops.supervision_step(logits, images, labels, cues, ctx=ctx)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    losses, grad, _ = ops.supervision_step(logits, images, labels, cues, ctx=ctx)
torch.cuda.synchronize()
ref = ops.supervision_step(logits, images, labels, cues, ctx=ctx)
g.replay(); torch.cuda.synchronize()
same = torch.equal(ref[0], losses) and torch.equal(ref[1], grad)
graph = timeit(g.replay)
print("B %d: eager %.4f ms/step, hipGraph replay %.4f ms/step, identical results: %s" % (B, eager, graph, same))
sys.stdout.flush(); os._exit(0)
