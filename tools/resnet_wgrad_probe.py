#!/usr/bin/env python
"""weight gradient of the narrow ResNet layers (res2 / res3, batch 10 of 513 x 513): this repo's implicit-GEMM weight-gradient kernel
(partly empty 256 x 256 tiles) against the library's (aten convolution_backward -> MIOpen), microseconds per call"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops
cl = torch.channels_last
def t(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
print("%-28s %10s %10s %10s" % ("layer", "own us", "library us", "merged us"))
for name, cin, cout, hw, k, d in [("res2 c1 256->64", 256, 64, 129, 1, 1), ("res2 c3 64->256", 64, 256, 129, 1, 1), ("res2 c1 64->64", 64, 64, 129, 1, 1),
                                  ("res3 c1 512->128", 512, 128, 65, 1, 1), ("res3 c3 128->512", 128, 512, 65, 1, 1), ("res3 c2 128->128 3x3", 128, 128, 65, 3, 1),
                                  ("res4 c1 1024->256", 1024, 256, 65, 1, 1), ("res4 c2 256->256 3x3 d2", 256, 256, 65, 3, 2)]:
    x = torch.randn(10, cin, hw, hw, device="cuda").bfloat16().contiguous(memory_format=cl)
    g = torch.randn(10, cout, hw, hw, device="cuda").bfloat16().contiguous(memory_format=cl)
    w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=cl)
    wb = w.bfloat16()
    pd = ops.pack_conv_weight(w, for_dgrad=True)
    p = d * (k // 2)
    own = t(lambda: ops.conv_igemm_wgrad([x], [g], [d], k))
    lib = t(lambda: torch.ops.aten.convolution_backward(g, x, wb, None, [1, 1], [p, p], [d, d], False, [0, 0], 1, [False, True, False]))
    dg = t(lambda: ops.conv_igemm([g], [pd], None, [d], k, False))
    mg = t(lambda: ops.conv_igemm_backward_residual(g, pd, x, d, k))
    print("%-28s %10.1f %10.1f %10.1f   (data gradient alone %.1f)" % (name, own, lib, mg, dg))
