#!/usr/bin/env python
"""A/B on one box: the dilated fc6_k launches with and without the border-tap / border-row skipping of round 5
(conv_igemm.hip; variant 6 = every K-step multiplied, as in round 4).  Forward, data gradient and weight gradient, the four
branches in one launch and each dilation on its own; microseconds, median of interleaved rounds."""
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
    B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 41, 41
    rounds = 5
    print("%-34s %10s %10s %8s" % ("launch (batch %d, 41x41)" % B, "all steps", "skipping", "ratio"))
    for what, cin, cout in (("forward 512->1024", 512, 1024), ("data gradient 1024->512", 1024, 512)):
        for dils in ([6, 12, 18, 24], [6], [12], [18], [24], [1]):
            n = len(dils)
            xs = [torch.randn(B, cin, H, W, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(n)]
            ws = [ops.pack_conv_weight((torch.randn(cout, cin, 3, 3, device="cuda") * 0.02).bfloat16()) for _ in range(n)]
            run = lambda: ops.conv_igemm(xs, ws, [None] * n, dils, 3, False, stream_k=False)      # noqa: E731
            t = {6: [], 3: []}
            for v in (6, 3):
                ops.set_igemm_variant(v); run(); run()
            for _ in range(rounds):
                for v in (6, 3):
                    ops.set_igemm_variant(v)
                    t[v].append(timed(run))
            a, b = np.median(t[6]), np.median(t[3])
            print("%-34s %10.1f %10.1f %8.3f" % ("%s d=%s" % (what, dils), a, b, b / a), flush=True)
    for dils in ([6, 12, 18, 24], [6], [12], [18], [24], [1]):
        n = len(dils)
        x = torch.randn(B, 512, H, W, device="cuda").bfloat16().contiguous(memory_format=CL)
        gs = [torch.randn(B, 1024, H, W, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(n)]
        run = lambda: ops.conv_igemm_wgrad([x] * n, gs, dils, 3)                             # noqa: E731
        t = {6: [], 7: [], 3: []}
        for v in (6, 7, 3):
            ops.set_igemm_variant(v); run(); run()
        for _ in range(rounds):
            for v in (6, 7, 3):
                ops.set_igemm_variant(v)
                t[v].append(timed(run))
        a, b, c = np.median(t[6]), np.median(t[3]), np.median(t[7])
        print("%-34s %10.1f %10.1f %8.3f   (dead steps of the flat order skipped: %.1f)" % ("weight gradient d=%s" % dils, a, b, b / a, c), flush=True)
    ops.set_igemm_variant(-1)


if __name__ == "__main__":
    main()
