#!/usr/bin/env python
"""Layer-level A/B (round 6): a FLOAT32 convolution — the arithmetic of the reference's Caffe backbone — three ways at batch 16, 41x41:
  split    the implicit-GEMM kernel in split mode (dsrg_conv_igemm_split_f32): operands as three bf16 planes, six bf16 products per
           multiply-add on the fp32 accumulators; time of the kernel alone and with the two operand splits (torch ops in this prototype)
  fp32     today's float32 leg: NHWC im2col (HIP) + hipBLASLt float32 GEMM (backbone._im2col_gemm); time with and without the im2col
  bf16     the bf16 implicit-GEMM launch of the headline, for scale
Errors are max |y - y64| / max |y64| against a float64 convolution of the same float32 operands."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops
from dsrg_amd.backbone import _im2col_gemm
CL = torch.channels_last


def timed(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def conv64(x, w, b, d):
    """float64 reference, 4 images at a time (unfold + matmul on the GPU)"""
    B, cin, H, W = x.shape
    out = []
    w2 = w.double().reshape(w.shape[0], -1)
    for i in range(0, B, 4):
        cols = F.unfold(x[i:i + 4].double(), w.shape[2], dilation=d, padding=d * (w.shape[2] // 2))      # (b, cin k k, H W)
        out.append((w2 @ cols + b.double().view(1, -1, 1)).view(-1, w.shape[0], H, W))
    return torch.cat(out)


def main():
    B, H, W = 16, 41, 41
    rounds = 5
    print("%-26s %9s %9s %9s %9s %9s | %9s %9s %9s | %7s %7s" % ("layer (batch 16, 41x41)", "split us", "+splits", "fp32 us", "mm only", "bf16 us",
                                                                  "err split", "err fp32", "err bf16", "x mm", "x route"))
    for name, cin, cout, d in (("conv4_2 512->512 d1", 512, 512, 1), ("fc6_1 512->1024 d6", 512, 1024, 6), ("conv5_1 512->512 d2", 512, 512, 2)):
        g = torch.Generator(device="cuda").manual_seed(cin + cout + d)
        x = torch.relu(torch.randn(B, cin, H, W, device="cuda", generator=g)).contiguous(memory_format=CL)
        w = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (2.0 / (cin * 9)) ** 0.5).contiguous(memory_format=CL)
        b = torch.randn(cout, device="cuda", generator=g) * 0.1
        y64 = conv64(x, w, b, d)
        scale = float(y64.abs().max())
        # split mode: whole call (splits + kernel) and the kernel alone on prepared operands
        y_s = ops.conv_igemm_split(x, w, b, d, False)
        err_s = float((y_s.double() - y64).abs().max()) / scale
        x3 = ops.split3_bf16(x.permute(0, 2, 3, 1).contiguous().float(), 3)
        w0, w1, w2 = [ops.pack_conv_weight(p) for p in ops.split3_bf16(w.float(), 0).split(cout, 0)]
        wv = torch.cat([w0, w1, w0, w2, w0, w1], dim=1).contiguous()
        y = torch.empty((B, cout, H, W), dtype=torch.float32, device="cuda", memory_format=CL)
        from dsrg_amd import _lib
        P = ops._ptr
        k_only = lambda: ops.check(_lib.lib().dsrg_conv_igemm_split_f32(P(x3), P(wv), P(b), P(y), d, B, H, W, cin, cout, 3, 0, ops._stream()))   # noqa: E731
        whole = lambda: ops.conv_igemm_split(x, w, b, d, False)                                                                                    # noqa: E731
        # float32 route of today
        y_f = _im2col_gemm(x, w, b, d, False)
        err_f = float((y_f.double() - y64).abs().max()) / scale
        route = lambda: _im2col_gemm(x, w, b, d, False)                                                                                            # noqa: E731
        cols = ops.im2col3x3_nhwc(x.permute(0, 2, 3, 1).contiguous(), d)
        wmat = w.permute(2, 3, 1, 0).reshape(9 * cin, cout).contiguous()
        o2 = torch.empty(B * H * W, cout, device="cuda")
        mm = lambda: torch.addmm(b, cols, wmat, out=o2)                                                                                            # noqa: E731
        # bf16 launch
        xb, pk = x.bfloat16().contiguous(memory_format=CL), ops.pack_conv_weight(w.bfloat16())
        y_b = ops.conv_igemm([xb], [pk], [b], [d], 3, False)[0]
        err_b = float((y_b.double() - y64).abs().max()) / scale
        bf = lambda: ops.conv_igemm([xb], [pk], [b], [d], 3, False)                                                                                # noqa: E731
        fns = {"k": k_only, "w": whole, "r": route, "m": mm, "b": bf}
        for f in fns.values():
            f(); f()
        t = {k: [] for k in fns}
        for _ in range(rounds):
            for k, f in fns.items():
                t[k].append(timed(f))
        m = {k: float(np.median(v)) for k, v in t.items()}
        print("%-26s %9.1f %9.1f %9.1f %9.1f %9.1f | %9.2e %9.2e %9.2e | %7.2f %7.2f" % (name, m["k"], m["w"], m["r"], m["m"], m["b"], err_s, err_f, err_b,
                                                                                        m["m"] / m["k"], m["r"] / m["k"]), flush=True)
    print("x mm = hipBLASLt float32 GEMM alone / split kernel alone; x route = im2col + GEMM / split kernel alone")


if __name__ == "__main__":
    main()
