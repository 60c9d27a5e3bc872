#!/usr/bin/env python
"""Per-workgroup phase timeline of mf_filter_kernel (debug hook dsrg_debug_set_filter_trace)."""
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
nblk = 4096
buf = torch.zeros(nblk * 32, dtype=torch.int64, device="cuda")
L = _lib.lib()
L.dsrg_debug_set_filter_trace.argtypes = [ctypes.c_void_p]
L.dsrg_debug_set_filter_trace(ctypes.c_void_p(buf.data_ptr()))
ops.supervision_step(logits, images, labels, cues, ctx=ctx)   # the 10 launches overwrite each other: last one stays
torch.cuda.synchronize()
L.dsrg_debug_set_filter_trace(None)
t = buf.cpu().numpy().reshape(nblk, 32)
bil = t[t[:, 0] > 0][:, 0:16]
gau = t[t[:, 16] > 0][:, 16:32]
t0 = min([x[:, 0].min() for x in (bil, gau) if len(x)])
print("bilateral blocks", len(bil), "gaussian blocks", len(gau))
for kind, sel, nb in (("bilateral", bil, 6), ("gaussian", gau, 3)):
    if not len(sel):
        continue
    print("%s: M mean %.0f  start (us) min %.2f max %.2f  end max %.2f" % (
        kind, sel[:, 12].mean(), (sel[:, 0].min() - t0) / 100.0, (sel[:, 0].max() - t0) / 100.0, (sel[:, 11].max() - t0) / 100.0))
    stamps = [1, 2, 3, 4] + [5 + j for j in range(nb)] + [11]
    prev = sel[:, 0]
    for nm, si in zip(["inq", "prod", "rowsum", "valwr"] + ["blur%d" % j for j in range(nb)] + ["slice"], stamps):
        dtt = (sel[:, si] - prev) / 100.0
        print("  %-7s mean %.2f  max %.2f us" % (nm, dtt.mean(), dtt.max()))
        prev = sel[:, si]
    print("  total   mean %.2f max %.2f us" % (((sel[:, 11] - sel[:, 0]) / 100.0).mean(), ((sel[:, 11] - sel[:, 0]) / 100.0).max()))
import os as _os
sys.stdout.flush(); _os._exit(0)
