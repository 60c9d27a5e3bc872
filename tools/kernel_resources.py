#!/usr/bin/env python
"""Register / spill / LDS table of every kernel in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
usage: tools/kernel_resources.py dsrg_amd/csrc/meanfield.hip [name-filter]"""
import os, re, subprocess, sys
src = os.path.abspath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, cwd=os.path.dirname(os.path.abspath(src)) or ".").stderr.decode()
rows, cur = [], None
for ln in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
try:
    import subprocess as sp
    dem = sp.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + [r["name"] for r in rows], stdout=sp.PIPE).stdout.decode().splitlines()
except Exception:
    dem = [r["name"] for r in rows]
print("%-78s %5s %5s %6s %6s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "sSpill", "vSpill", "scratch", "occ", "LDS"))
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("void dsrg::", "")
    if flt and flt not in d:
        continue
    print("%-78s %5s %5s %6s %6s %7s %4s %7s" % (d[:78], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
                                                  r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS Size")))
