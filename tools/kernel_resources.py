#!/usr/bin/env python3
"""Compact table of hipcc's -Rpass-analysis=kernel-resource-usage for one translation unit:
   tools/kernel_resources.py krotov.jl_amd/csrc/kr_mf64.hip [filter]
(VGPRs + AGPRs share one 512-entry file per lane on gfx950; ScratchSize > 0 means spills.)"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-c", "-o", "/dev/null",
                      src] + [x for x in sys.argv[3:]] + [ "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        m = re.search(r":\d+:\d+: remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if not rows:
    sys.exit("no kernels reported; compiler said:\n" + out[-3000:])
demangle = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
print("%-64s %5s %5s %8s %7s %4s" % ("kernel", "VGPR", "AGPR", "scratch", "LDS", "occ"))
for r, d in zip(rows, demangle):
    d = re.sub(r"\(KrSweepArgs\)|void ", "", d)
    if flt and flt not in d:
        continue
    print("%-64s %5s %5s %8s %7s %4s" % (d[:64], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"),
                                          r.get("LDS Size [bytes/block]", "?"), r.get("Occupancy [waves/SIMD]", "?")))
