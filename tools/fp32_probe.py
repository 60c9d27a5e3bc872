#!/usr/bin/env python
"""ms per train-s step at batch 16 for the backbone precisions / conv routes under discussion (bench.py's fp32 leg):
  bf16 autocast (default) | fp32 with the GEMM-route convolutions | fp32 with MIOpen everywhere"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import synthetic as S  # noqa: E402
from dsrg_amd.backbone import VGG16ASPP  # noqa: E402
from dsrg_amd.trainer import DSRGTrainer  # noqa: E402

device = torch.device("cuda", 0)
b = S.make_batch(1000, 16)
d = lambda a: torch.from_numpy(a).to(device)
images, labels, cues = d(b["images"]), d(b["labels"]), d(b["cues"])
for name, amp, gemm in [("bf16 gemm", torch.bfloat16, True), ("fp32 gemm-route", None, True), ("fp32 miopen", None, False)]:
    tr = DSRGTrainer(device, amp_dtype=amp, net=VGG16ASPP(gemm_convs=gemm))
    for _ in range(3):
        tr.step(images, labels, cues)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 8
    for _ in range(n):
        l = tr.step(images, labels, cues)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-18s %8.2f ms/step %8.1f images/s  losses %s" % (name, dt * 1e3, 16 / dt, [float(x) for x in l]), flush=True)
    del tr
    torch.cuda.empty_cache()
