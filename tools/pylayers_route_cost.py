#!/usr/bin/env python
"""What the Caffe Python-layer route costs next to the fused device-resident step: the five drop-in
`pylayers` classes driven the way Caffe's Net::ForwardBackward drives them — numpy blobs, every layer call stages its bottoms
host -> HBM and its tops HBM -> host, the CRF runs in CRFLayer AND in DSRGLayer — against ONE dsrg_supervision_step on
device tensors.  Same batch (16 synthetic images), same results.   usage: pylayers_route_cost.py [B]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pylayers
from dsrg_amd import ops, synthetic as S


class Blob(object):
    def __init__(self, a):
        self.data = np.ascontiguousarray(a, np.float32)
        self.diff = np.zeros_like(self.data)

    def reshape(self, *s):
        if tuple(s) != self.data.shape:
            self.data = np.zeros(s, np.float32)
            self.diff = np.zeros(s, np.float32)


B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
b = S.make_batch(1000, B)
zeros = lambda: np.zeros_like(b["logits"])
fc8, images, labels, cues = Blob(b["logits"]), Blob(b["images"]), Blob(b["labels"]), Blob(b["cues"])
probs, crf_log, seeds, l_seed, l_con = Blob(zeros()), Blob(zeros()), Blob(zeros()), Blob(np.zeros(1)), Blob(np.zeros(1))
layers = [(pylayers.SoftmaxLayer(), [fc8], [probs]), (pylayers.CRFLayer(), [probs, images], [crf_log]),
          (pylayers.DSRGLayer(), [labels, probs, cues, images], [seeds]),
          (pylayers.BalancedSeedLossLayer(), [probs, seeds], [l_seed]), (pylayers.ConstrainLossLayer(), [probs, crf_log], [l_con])]
for lay, bot, top in layers:
    lay.param_str = "{'th1': 0.99, 'th2': 0.85}"
    lay.setup(bot, top)
    lay.reshape(bot, top)


def caffe_step():
    for lay, bot, top in layers:
        lay.forward(bot, top)
    for blob in (probs, crf_log, fc8):
        blob.diff[...] = 0
    # Caffe sums the diffs a blob receives from its consumers (implicit Split layer): each backward writes its bottom diffs,
    # the driver accumulates
    acc_p, acc_lq = np.zeros_like(probs.data), np.zeros_like(probs.data)
    layers[4][0].backward(layers[4][2], [True, True], layers[4][1]); acc_p += probs.diff; acc_lq += crf_log.diff
    layers[3][0].backward(layers[3][2], [True, False], layers[3][1]); acc_p += probs.diff
    layers[2][0].backward(layers[2][2], [False, True, False, False], layers[2][1])
    crf_log.diff[...] = acc_lq
    layers[1][0].backward(layers[1][2], [True, False], layers[1][1]); acc_p += probs.diff
    probs.diff[...] = acc_p
    layers[0][0].backward(layers[0][2], [True], layers[0][1])
    return float(l_seed.data[0]), float(l_con.data[0]), fc8.diff.copy()


d = lambda a: torch.from_numpy(a).cuda()
lg, im, lb, cu = d(b["logits"]), d(b["images"]), d(b["labels"]), d(b["cues"])
ctx = ops.get_context(B, 21, 41, 41)
for _ in range(3):
    ref = caffe_step()
    losses, grad, _ = ops.supervision_step(lg, im, lb, cu, ctx=ctx)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    caffe_step()
t_layers = (time.perf_counter() - t0) / n * 1e3
t0 = time.perf_counter()
for _ in range(200):
    ops.supervision_step(lg, im, lb, cu, ctx=ctx)
torch.cuda.synchronize()
t_fused = (time.perf_counter() - t0) / 200 * 1e3
gd = np.abs(grad.cpu().numpy() - ref[2]).max() / max(np.abs(ref[2]).max(), 1e-30)
# where the route's time goes: the same step with the layers' cost accounting on (dsrg_amd.layers.profile; the device is waited
# for after every upload and kernel group so that each millisecond is booked where it is spent — the step itself gets slower)
from dsrg_amd import layers as _L
_L.profile = {}
np_ = 10
for _ in range(np_):
    caffe_step()
prof, _L.profile = _L.profile, None
cats = ["digest", "pin", "h2d", "kernels", "d2h", "sync", "total"]
print("per-layer cost of the Caffe route, ms per step (accounting on: every phase waited for):")
print("%-32s " % "layer call" + " ".join("%8s" % c for c in cats) + "    other")
tot = {c: 0.0 for c in cats}
for name in ["SoftmaxLayer.forward", "CRFLayer.forward", "DSRGLayer.forward", "BalancedSeedLossLayer.forward", "ConstrainLossLayer.forward",
             "ConstrainLossLayer.backward", "BalancedSeedLossLayer.backward", "CRFLayer.backward", "SoftmaxLayer.backward"]:
    row = [prof.get((name, c), 0.0) / np_ * 1e3 for c in cats]
    for c, v in zip(cats, row):
        tot[c] += v
    print("%-32s " % name + " ".join("%8.3f" % v for v in row) + " %8.3f" % (row[-1] - sum(row[:-1])))
print("%-32s " % "sum" + " ".join("%8.3f" % tot[c] for c in cats) + " %8.3f" % (tot["total"] - sum(tot[c] for c in cats[:-1])))
print("pinned host buffers: %d of %d seen; resident device copies: %d" % (sum(1 for e in _L._pinned.values() if e[2]), len(_L._pinned), len(_L._resident)))
print("B %d: Caffe Python-layer route (numpy blobs, 5 layers fwd+bwd, CRF twice) %.2f ms per step; fused device-resident "
      "dsrg_supervision_step %.3f ms; losses %.6f/%.6f vs %.6f/%.6f, max rel. gradient difference %.1e" % (
          B, t_layers, t_fused, ref[0], ref[1], float(losses[0]), float(losses[1]), gd))
sys.stdout.flush(); os._exit(0)
