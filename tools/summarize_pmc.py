"""Summarise rocprofv3 output of tools/profile.sh: per-kernel time stats and per-launch HBM bytes
(FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes for wide coalesced reads on
gfx950; WRITE_SIZE taken as is; both counters are in KiB-units of 1024 B... see the guide)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
ks = os.path.join(out, "kernel_stats.csv")
if os.path.exists(ks):
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    with open(ks) as f:
        for row in csv.reader(f):
            print(",".join(row))
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(out, "pmc_fetch" if name == "FETCH_SIZE" else "pmc_write")
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter file for", name)
        continue
    acc = defaultdict(lambda: [0.0, 0])
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != name:
                continue
            k = row.get("Kernel_Name", "?").split("(")[0]
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
    print("== %s per launch (raw counter units; KiB per rocprofv3 docs) ==" % name)
    for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print("%-60s launches=%d raw_per_launch=%.1f" % (k[:60], n, v / max(n, 1)))
