#!/usr/bin/env python
"""Forward time of the 41x41 backbone layers: MIOpen conv vs explicit NHWC im2col + hipBLASLt GEMM."""
import sys, os, time
import torch, torch.nn.functional as F
B, H, W = 16, 41, 41
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def im2col_gemm(x, wmat, bias, k, dil):
    # x: (B,H,W,C) bf16 contiguous; wmat: (k*k*C, Cout)
    if k == 1:
        a = x.reshape(-1, x.shape[-1])
    else:
        p = dil
        xp = F.pad(x, (0, 0, p, p, p, p))
        cols = [xp[:, dy * dil:dy * dil + H, dx * dil:dx * dil + W, :] for dy in range(3) for dx in range(3)]
        a = torch.cat(cols, dim=-1).reshape(-1, 9 * x.shape[-1])
    return torch.addmm(bias, a, wmat).view(B, H, W, -1)
for (cin, cout, k, dil) in [(256, 512, 3, 1), (512, 512, 3, 1), (512, 512, 3, 2), (512, 1024, 3, 6), (512, 1024, 3, 24), (1024, 1024, 1, 1), (1024, 21, 1, 1)]:
    conv = torch.nn.Conv2d(cin, cout, k, padding=dil * (k // 2), dilation=dil).cuda().to(memory_format=torch.channels_last).bfloat16()
    x = torch.randn(B, cin, H, W, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xn = x.permute(0, 2, 3, 1).contiguous()
    wmat = conv.weight.permute(2, 3, 1, 0).reshape(k * k * cin, cout).contiguous()
    bias = conv.bias
    with torch.no_grad():
        ref = conv(x).permute(0, 2, 3, 1)
        out = im2col_gemm(xn, wmat, bias, k, dil)
        err = (ref.float() - out.float()).abs().max().item()
        t1 = timeit(lambda: conv(x)); t2 = timeit(lambda: im2col_gemm(xn, wmat, bias, k, dil))
    fl = 2 * B * H * W * cin * cout * k * k
    print("cin %4d cout %4d k%d dil%2d: miopen %.3f ms (%.0f TF) | im2col+gemm %.3f ms (%.0f TF) | maxdiff %.3g" % (
        cin, cout, k, dil, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, err), flush=True)
