#!/usr/bin/env python
"""Time the VGG16-ASPP fwd+bwd under a few PyTorch-ROCm settings (plumbing, not the product)."""
import sys, os, time, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd.backbone import VGG16ASPP, count_flops_per_image

B = 16
x0 = torch.randn(B, 3, 321, 321, device="cuda")
g0 = torch.randn(B, 21, 41, 41, device="cuda")
fl = count_flops_per_image() * 3 * B
for bench, cl, dt in itertools.product([False, True], [True, False], [torch.bfloat16, torch.float16, None]):
    torch.backends.cudnn.benchmark = bench
    net = VGG16ASPP().cuda()
    x = x0
    if cl:
        net = net.to(memory_format=torch.channels_last); x = x0.contiguous(memory_format=torch.channels_last)
    def step():
        for p in net.parameters(): p.grad = None
        with torch.autocast("cuda", dtype=dt, enabled=dt is not None):
            y = net(x)
        y.float().backward(g0)
    try:
        for _ in range(4): step()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(8): step()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 8 * 1e3
        print("benchmark=%s channels_last=%s dtype=%s : %.2f ms/step  %.0f TFLOP/s" % (bench, cl, dt, ms, fl / ms / 1e9), flush=True)
    except Exception as e:
        print("benchmark=%s channels_last=%s dtype=%s : FAILED %s" % (bench, cl, dt, str(e)[:100]), flush=True)
    del net
