#!/usr/bin/env python
"""Phase timeline of mf_persistent_kernel (debug hook dsrg_debug_set_filter_trace; 64 stamps of the 100 MHz wall clock per
workgroup): per iteration the wait for the marginals, the Gaussian filter, the bilateral filter, the hand-off stores and
the pixel-parallel half.   python tools/persist_trace.py [B]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops, synthetic as S, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
b = S.make_batch(1000, B)
d = lambda a: torch.from_numpy(a).cuda()
logits, images, labels, cues = d(b["logits"]), d(b["images"]), d(b["labels"]), d(b["cues"])
ctx = ops.get_context(B, 21, 41, 41)
for _ in range(3):
    ops.supervision_step(logits, images, labels, cues, ctx=ctx)
torch.cuda.synchronize()
nblk = 1024
buf = torch.zeros(nblk * 64, dtype=torch.int64, device="cuda")
L = _lib.lib()
L.dsrg_debug_set_filter_trace.argtypes = [ctypes.c_void_p]
L.dsrg_debug_set_filter_trace(ctypes.c_void_p(buf.data_ptr()))
ops.supervision_step(logits, images, labels, cues, ctx=ctx)
torch.cuda.synchronize()
L.dsrg_debug_set_filter_trace(None)
t = buf.cpu().numpy().reshape(nblk, 64)
t = t[t[:, 0] > 0].astype(np.float64) / 100.0            # us
t0 = t[:, 0].min()
print("workgroups %d; start spread %.2f us; Q0 phase mean %.2f max %.2f us" % (len(t), t[:, 0].max() - t0, (t[:, 1] - t[:, 0]).mean(),
                                                                                  (t[:, 1] - t[:, 0]).max()))
names = ["wait Q", "gauss", "bilat", "V store", "update(+wait V)"]
tot = np.zeros(5)
for it in range(10):
    base = 2 + it * 6
    prev = t[:, 1] if it == 0 else t[:, 2 + (it - 1) * 6 + 4]
    row = []
    for k in range(5):
        cur = t[:, base + k]
        dt = cur - prev
        row.append("%s %.2f/%.2f" % (names[k], dt.mean(), dt.max()))
        tot[k] += dt.mean()
        prev = cur
    if it in (0, 1, 5, 9):
        print("it %2d (mean/max us): %s" % (it + 1, "  ".join(row)))
print("sum over 10 iterations (mean us per workgroup):", "  ".join("%s %.1f" % (n, v) for n, v in zip(names, tot)))
end = t[:, 2 + 9 * 6 + 4]
print("kernel span first start -> last end: %.1f us; per-workgroup total mean %.1f max %.1f" % (end.max() - t0, (end - t[:, 0]).mean(),
                                                                                              (end - t[:, 0]).max()))
sys.stdout.flush(); os._exit(0)
