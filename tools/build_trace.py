#!/usr/bin/env python
"""Per-phase timeline of lattice_build_kernel<5> (debug hook dsrg_debug_set_build_trace)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops, synthetic as S, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kind = sys.argv[2] if len(sys.argv) > 2 else "smooth"
b = S.make_batch(1000, B, image_kind=kind)
d = lambda a: torch.from_numpy(a).cuda()
logits, images, labels, cues = d(b["logits"]), d(b["images"]), d(b["labels"]), d(b["cues"])
ctx = ops.get_context(B, 21, 41, 41)
for _ in range(3):
    ops.supervision_step(logits, images, labels, cues, ctx=ctx)
torch.cuda.synchronize()
buf = torch.zeros(64 * 16 + 1024 * 8, dtype=torch.int64, device="cuda")
L = _lib.lib()
L.dsrg_debug_set_build_trace.argtypes = [ctypes.c_void_p]
L.dsrg_debug_set_build_trace(ctypes.c_void_p(buf.data_ptr()))
ops.supervision_step(logits, images, labels, cues, ctx=ctx)
torch.cuda.synchronize()
L.dsrg_debug_set_build_trace(None)
raw = buf.cpu().numpy()
t = raw[:64 * 16].reshape(64, 16)[:B]
names = ["embed", "insert", "ids", "vid+neigh", "csr count/scan", "csr fill/sort", "csr write", "norm splat", "norm blur", "norm slice"]
prev = t[:, 0]
for i, nm in enumerate(names):
    if not (t[:, i + 1] > 0).all() or (nm.startswith("norm") and not (t[:, 8] > 0).all()):
        continue                             # the in-kernel norm pass runs only when the build is not split over kernels
    dt = (t[:, i + 1] - prev) / 100.0
    print("  %-15s mean %7.2f  max %7.2f us" % (nm, dt.mean(), dt.max()))
    prev = t[:, i + 1]
if (t[:, 11] > 0).all():
    print("  (csr write: sorted entries -> first/extras lists %.2f us, phantom rows + extras count %.2f us)" % (
        ((t[:, 11] - t[:, 6]) / 100.0).mean(), ((t[:, 7] - t[:, 11]) / 100.0).mean()))
print("  (split build: the three csr phases run in the neighbour launch's extra workgroup — the first of them includes the\n"
      "   kernel boundary and the re-load of the entries' vertex ids; 'total' is the build kernel alone)")
print("  total mean %.2f max %.2f us;  M:" % (((t[:, 10] - t[:, 0]) / 100.0).mean(), ((t[:, 10] - t[:, 0]) / 100.0).max()), ctx.lattice_sizes(B))
nt = raw[64 * 16:].reshape(1024, 8)
nt = nt[nt[:, 0] > 0]
if len(nt):
    print("lattice_neigh_kernel: %d workgroups, start spread %.2f us (wave 0 of each workgroup)" % (len(nt), (nt[:, 0].max() - nt[:, 0].min()) / 100.0))
    prev = nt[:, 0]
    for i, nm in enumerate(["stage table+keys", "n1 look-ups", "n2 look-ups", "stores+tail", "flag"]):
        dt = (nt[:, i + 1] - prev) / 100.0
        print("  %-17s mean %7.2f  max %7.2f us" % (nm, dt.mean(), dt.max()))
        prev = nt[:, i + 1]
    print("  first start -> last end %.2f us" % ((nt[:, 5].max() - nt[:, 0].min()) / 100.0))
sys.stdout.flush(); os._exit(0)
