#!/usr/bin/env python
"""The full train-s step repeated on ONE synthetic batch, bf16-autocast backbone and float32 backbone (the reference's Caffe
arithmetic) side by side: same initial weights, same batch, same solver.  Both losses must go down, and the two trajectories
must stay together (gradient chain, Caffe-style SGD, fp32 master weights under bf16 products all in the loop).

  overfit_probe.py [steps=300] [batch=8] [--dropout P (default: 0 and 0.5, one table each)] [--init kaiming|default] [--lr X]

With Dropout on the legs draw different masks (the bf16 leg in the convolution epilogues from a counter-based generator, the
float32 leg from torch's), so only the Dropout-off table compares arithmetic; the Dropout-on table shows both still train."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import synthetic as S
from dsrg_amd.backbone import VGG16ASPP
from dsrg_amd.trainer import DSRGTrainer


def trajectory(dev, amp, dropout, steps, batch, init, lr):
    torch.manual_seed(0)
    net = VGG16ASPP(dropout=dropout)
    if init == "kaiming":
        from grad_fidelity import kaiming_
        kaiming_(net)
    tr = DSRGTrainer(dev, amp_dtype=amp, seed=0, net=net)
    if lr is not None:
        tr.opt.base_lr = lr
    images, labels, cues = batch
    hist = []
    for _ in range(steps):
        hist.append(tr.step(images, labels, cues).detach())
    out = torch.stack(hist).cpu().numpy()
    del tr
    torch.cuda.empty_cache()
    return out


def main():
    argv = [a for a in sys.argv[1:]]
    opt = lambda k, d: (argv[argv.index(k) + 1] if k in argv else d)                   # noqa: E731
    pos = [a for i, a in enumerate(argv) if not a.startswith("--") and (i == 0 or not argv[i - 1].startswith("--"))]
    steps = int(pos[0]) if pos else 300
    B = int(pos[1]) if len(pos) > 1 else 8
    drops = [float(opt("--dropout", "0"))] if "--dropout" in argv else [0.0, 0.5]
    init, lr = opt("--init", "default"), (float(opt("--lr", "0")) or None)
    dev = torch.device("cuda", 0)
    b = S.make_batch(7, B)
    batch = tuple(torch.from_numpy(b[k]).to(dev) for k in ("images", "labels", "cues"))
    for p in drops:
        h16 = trajectory(dev, torch.bfloat16, p, steps, batch, init, lr)
        h32 = trajectory(dev, None, p, steps, batch, init, lr)
        assert np.isfinite(h16).all() and np.isfinite(h32).all(), "non-finite loss"
        t16, t32 = h16.sum(1), h32.sum(1)
        gap = np.abs(t16 - t32) / np.maximum(np.abs(t32), 1e-12)
        print("== %d steps, batch %d, Dropout %.2f, init %s, base_lr %s ==" % (steps, B, p, init, lr or "solver-s (5e-4)"))
        print("%6s | %12s %12s | %12s %12s | %s" % ("step", "Seed bf16", "Seed fp32", "Constr bf16", "Constr fp32", "rel gap of the total"))
        for it in sorted(set(list(range(0, steps, max(1, steps // 15))) + [0, 1, 2, steps - 1])):
            print("%6d | %12.5f %12.5f | %12.5f %12.5f | %.3e" % (it, h16[it, 0], h32[it, 0], h16[it, 1], h32[it, 1], gap[it]))
        print("total loss %.4f -> %.4f (bf16), %.4f -> %.4f (fp32); max rel gap %.3e, final rel gap %.3e, mean over the last "
              "tenth %.3e" % (t16[0], t16[-1], t32[0], t32[-1], gap.max(), gap[-1], gap[-max(1, steps // 10):].mean()))
        assert t16[-1] < t16[0] and t32[-1] < t32[0], "a leg did not reduce its loss"


if __name__ == "__main__":
    main()
