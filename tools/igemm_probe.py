#!/usr/bin/env python
"""A/B on one box: the implicit-GEMM convolution (csrc/conv_igemm.hip) against the route it replaces (NHWC im2col kernel +
hipBLASLt GEMM with fused bias / ReLU, solutions picked by TunableOp) for the wide layers of the backbone at batch 16.
Prints one line per layer: microseconds (median of the rounds; variants interleaved inside every round) and TFLOP/s.

  python tools/igemm_probe.py [--rounds 5] [--iters 10] [--batch 16]
"""
import argparse
import os
import sys

os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/dsrg_tunableop_probe.csv")
os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops                                                           # noqa: E402
from dsrg_amd.backbone import _im2col_gemm                                         # noqa: E402

CL = torch.channels_last


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16)
    args = ap.parse_args()
    B = args.batch
    layers = [  # name, H, W, cin, cout, k, dilations (len = groups)
        ("conv4_2 512->512 d1", 41, 41, 512, 512, 3, [1]),
        ("conv5_1 512->512 d2", 41, 41, 512, 512, 3, [2]),
        ("conv4_1 256->512 d1", 41, 41, 256, 512, 3, [1]),
        ("conv4_1 dgrad 512->256", 41, 41, 512, 256, 3, [1]),
        ("fc6 512->1024 d12", 41, 41, 512, 1024, 3, [12]),
        ("fc6 x4 (one launch)", 41, 41, 512, 1024, 3, [6, 12, 18, 24]),
        ("fc6 dgrad 1024->512 d12", 41, 41, 1024, 512, 3, [12]),
        ("fc6 dgrad x4", 41, 41, 1024, 512, 3, [6, 12, 18, 24]),
        ("fc7 1024->1024 1x1", 41, 41, 1024, 1024, 1, [1]),
        ("fc7 x4 (one launch)", 41, 41, 1024, 1024, 1, [1, 1, 1, 1]),
        ("conv3_2 256->256 81x81", 81, 81, 256, 256, 3, [1]),
        ("conv3_1 128->256 81x81", 81, 81, 128, 256, 3, [1]),
    ]
    print("%-28s %9s %9s %9s %9s | %8s %8s | %s" % ("layer", "variant", "default", "im2col+mm", "mm only", "TF/s ig", "TF/s old", "max err"))
    for name, H, W, cin, cout, k, dils in layers:
        n = len(dils)
        torch.manual_seed(1)
        xs = [torch.randn(B, cin, H, W, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(n)]
        ws = [(torch.randn(cout, cin, k, k, device="cuda") * (2.0 / (cin * k * k)) ** 0.5).bfloat16() for _ in range(n)]
        bs = [torch.randn(cout, device="cuda") for _ in range(n)]
        packed = [ops.pack_conv_weight(w) for w in ws]
        bsb = [b.bfloat16() for b in bs]

        sk = [True]

        def run_ig():
            return ops.conv_igemm(xs, packed, bs, dils, k, True, stream_k=sk[0])

        def run_old():
            return [_im2col_gemm(xs[g], ws[g], bsb[g], dils[g], True) for g in range(n)]

        cols = [ops.im2col3x3_nhwc(xs[g].permute(0, 2, 3, 1), dils[g]) if k == 3 else xs[g].permute(0, 2, 3, 1).reshape(-1, cin) for g in range(n)]
        wm = [ws[g].permute(2, 3, 1, 0).reshape(k * k * cin, cout) for g in range(n)]
        outs = [torch.empty(B * H * W, cout, device="cuda", dtype=torch.bfloat16) for _ in range(n)]

        def run_mm():
            for g in range(n):
                torch._addmm_activation(bsb[g], cols[g], wm[g], out=outs[g])

        # correctness of the first group against fp32 torch (bias rounded as the old route rounds it: bf16)
        want = torch.relu(F.conv2d(xs[0].float(), ws[0].float(), bs[0], padding=dils[0] * (k // 2), dilation=dils[0]))
        got = run_ig()[0]
        err = float((got.float() - want).abs().max() / want.abs().max())
        for fn in (run_ig, run_old, run_mm):
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        t = {"ig1": [], "ig0": [], "old": [], "mm": []}
        for _ in range(args.rounds):
            sk[0] = False
            ops.set_igemm_variant(int(os.environ.get("PROBE_VARIANT", "4")))     # 4: stream-K wherever legal (needs the scratch)
            sk[0] = os.environ.get("PROBE_VARIANT", "4") == "4"
            t["ig1"].append(timed(run_ig, args.iters))
            ops.set_igemm_variant(-1)
            sk[0] = False
            t["ig0"].append(timed(run_ig, args.iters))
            t["old"].append(timed(run_old, args.iters))
            t["mm"].append(timed(run_mm, args.iters))
        ops.set_igemm_variant(-1)
        med = {k_: float(np.median(v)) for k_, v in t.items()}
        flops = 2.0 * B * H * W * cin * k * k * cout * n
        print("%-28s %9.1f %9.1f %9.1f %9.1f | %8.0f %8.0f | %.2e" % (
            name, med["ig1"], med["ig0"], med["old"], med["mm"], flops / min(med["ig1"], med["ig0"]) / 1e6,
            flops / med["old"] / 1e6, err), flush=True)
    wgrad_probe(B, args.rounds, args.iters)
    absorb_probe(B, args.rounds, args.iters)


def absorb_probe(B, rounds, iters):
    """the data gradient with the ReLU (+ Dropout) backward and bias gradient of the layer below in its store against the plain
    data gradient followed by ops.relu_bwd_bias"""
    layers = [("conv4_3 -> conv4_2 out", 41, 41, 512, 512, 3, [1]), ("conv3_3 -> conv3_2 out", 81, 81, 256, 256, 3, [1]),
              ("fc7 x4 -> fc6 out", 41, 41, 1024, 1024, 1, [1] * 4)]
    print("%-28s %9s %9s %9s" % ("fused backward", "fused", "separate", "dgrad only"))
    for name, H, W, cf, cb, k, dils in layers:
        n = len(dils)
        torch.manual_seed(3)
        gs = [torch.randn(B, cf, H, W, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(n)]
        ys = [torch.relu(torch.randn(B, cb, H, W, device="cuda")).bfloat16().contiguous(memory_format=CL) for _ in range(n)]
        packs = [ops.pack_conv_weight((torch.randn(cf, cb, k, k, device="cuda") * 0.02), for_dgrad=True) for _ in range(n)]

        def fused():
            return ops.conv_igemm_dgrad(gs, packs, ys, dils, k, 2.0)

        def plain():
            return ops.conv_igemm(gs, packs, None, dils, k, False, stream_k=False)

        def separate():
            return [ops.relu_bwd_bias(g, y, 2.0) for g, y in zip(plain(), ys)]
        for fn in (fused, separate, plain):
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        t = {"f": [], "s": [], "p": []}
        for _ in range(rounds):
            t["f"].append(timed(fused, iters)); t["s"].append(timed(separate, iters)); t["p"].append(timed(plain, iters))
        print("%-28s %9.1f %9.1f %9.1f" % (name, np.median(t["f"]), np.median(t["s"]), np.median(t["p"])), flush=True)


def wgrad_probe(B, rounds, iters):
    """the implicit weight gradient against im2col + g^T @ cols (what backbone._ConvFn.backward runs today)"""
    layers = [("conv4_2 512->512", 41, 41, 512, 512, 3, [1]), ("conv4_1 256->512", 41, 41, 256, 512, 3, [1]),
              ("fc6 512->1024 d12", 41, 41, 512, 1024, 3, [12]), ("fc6 x4 (one launch)", 41, 41, 512, 1024, 3, [6, 12, 18, 24]),
              ("fc7 1024->1024 1x1", 41, 41, 1024, 1024, 1, [1]), ("fc7 x4 (one launch)", 41, 41, 1024, 1024, 1, [1] * 4),
              ("conv3_2 256->256 81x81", 81, 81, 256, 256, 3, [1])]
    print("%-28s %9s %9s %9s | %8s %8s | %s" % ("weight gradient", "igemm", "im2col+mm", "mm only", "TF/s ig", "TF/s old", "max err"))
    for name, H, W, cin, cout, k, dils in layers:
        n = len(dils)
        torch.manual_seed(2)
        xs = [torch.randn(B, cin, H, W, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(n)]
        gs = [torch.randn(B, cout, H, W, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(n)]

        def run_ig():
            return ops.conv_igemm_wgrad(xs, gs, dils, k)

        def cols_of(g_):
            return ops.im2col3x3_nhwc(xs[g_].permute(0, 2, 3, 1), dils[g_]) if k == 3 else xs[g_].permute(0, 2, 3, 1).reshape(-1, cin)

        def run_old():
            return [torch.mm(gs[g_].permute(0, 2, 3, 1).reshape(-1, cout).t(), cols_of(g_)) for g_ in range(n)]
        cols = [cols_of(g_) for g_ in range(n)]

        def run_mm():
            return [torch.mm(gs[g_].permute(0, 2, 3, 1).reshape(-1, cout).t(), cols[g_]) for g_ in range(n)]
        want = run_old()[0].float().view(cout, k, k, cin).permute(0, 3, 1, 2)
        got = run_ig()[0]
        err = float((got - want).abs().max() / want.abs().max())
        for fn in (run_ig, run_old, run_mm):
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        t = {"ig": [], "ig3": [], "old": [], "mm": []}
        for _ in range(rounds):
            ops.set_igemm_variant(3)
            t["ig3"].append(timed(run_ig, iters))
            ops.set_igemm_variant(1)
            t["ig"].append(timed(run_ig, iters))
            t["old"].append(timed(run_old, iters))
            t["mm"].append(timed(run_mm, iters))
        med = {k_: float(np.median(v)) for k_, v in t.items()}
        flops = 2.0 * B * H * W * cin * k * k * cout * n
        print("%-28s %9.1f %9.1f %9.1f | %8.0f %8.0f | %.2e  (staggered %.1f)" % (name, med["ig"], med["old"], med["mm"], flops / med["ig"] / 1e6,
                                                                  flops / med["old"] / 1e6, err, med["ig3"]), flush=True)


if __name__ == "__main__":
    main()
