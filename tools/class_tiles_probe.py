#!/usr/bin/env python
"""A/B on one box: the dilated fc6_k launches with round 5's row-aligned pixel tiles (variant 8) against the class-ordered tiles of
round 6 (variant 3 = default where they pay, 9 = forced), and variant 6 (every K-step) for scale.  Forward and masked data gradient,
the four branches in one launch and each dilation on its own; microseconds, median of interleaved rounds."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops
CL = torch.channels_last


def timed(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 41
    rounds = 5
    variants = (6, 8, 9, 3)
    print("%-36s %10s %10s %10s %10s %8s" % ("launch (batch %d, %dx%d)" % (B, H, W), "all steps", "row tiles", "class", "default", "cls/row"))
    for what, cin, cout in (("forward 512->1024", 512, 1024), ("data gradient 1024->512", 1024, 512)):
        for dils in ([6, 12, 18, 24], [6], [12], [18], [24]):
            n = len(dils)
            xs = [torch.randn(B, cin, H, W, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(n)]
            ws = [ops.pack_conv_weight((torch.randn(cout, cin, 3, 3, device="cuda") * 0.02).bfloat16()) for _ in range(n)]
            if what.startswith("data"):
                ys = [torch.relu(torch.randn(B, cout, H, W, device="cuda")).bfloat16().contiguous(memory_format=CL) for _ in range(n)]
                run = lambda: ops.conv_igemm_dgrad(xs, ws, ys, dils, 3, 2.0)                    # noqa: E731
            else:
                run = lambda: ops.conv_igemm(xs, ws, [None] * n, dils, 3, True, 0.5, 7, stream_k=False)      # noqa: E731
            t = {v: [] for v in variants}
            for v in variants:
                ops.set_igemm_variant(v); run(); run()
            for _ in range(rounds):
                for v in variants:
                    ops.set_igemm_variant(v)
                    t[v].append(timed(run))
            m = {v: np.median(t[v]) for v in variants}
            print("%-36s %10.1f %10.1f %10.1f %10.1f %8.3f" % ("%s d=%s" % (what, dils), m[6], m[8], m[9], m[3], m[9] / m[8]), flush=True)
    ops.set_igemm_variant(-1)


if __name__ == "__main__":
    main()
