#!/usr/bin/env python3
"""Developer tool: profiles/rNN_traffic_all.json (what bench.py quotes as roofline.traffic) from the committed rocprofv3 PMC summaries
profiles/rNN_bench_c{2,3,4,5}_summary.txt (tools/profile.sh; FETCH_SIZE and WRITE_SIZE in separate passes).  Correction as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE reports 1/2 of wide coalesced 16-B/lane reads -> doubled;
WRITE_SIZE exact; both in KiB.      python tools/make_traffic_json.py r03"""
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {"C2": (16, 500, 32), "C3": (64, 1000, 256), "C4": (64, 1000, 1024), "C5": (256, 2000, 64)}
dominant = {"C2": "kr_sweep_update", "C3": "kr_sweep_update", "C4": "kr_mf_update", "C5": "kr_rowsplit_pair_sweep"}
out = []
for cfg, (N, NT, K) in shapes.items():
    path = os.path.join(root, "profiles", "%s_bench_%s_summary.txt" % (tag, cfg.lower()))
    if not os.path.exists(path):
        continue
    sect, vals = None, {}
    for line in open(path):
        if line.startswith("== FETCH_SIZE"):
            sect = "fetch"
        elif line.startswith("== WRITE_SIZE"):
            sect = "write"
        elif line.startswith("=="):
            sect = None
        m = re.match(r"(?:void )?(\S.*?)\s+launches=(\d+) raw_per_launch=([\d.]+)", line)
        if sect and m and dominant[cfg] in m.group(1):
            name, n, v = m.group(1).strip(), int(m.group(2)), float(m.group(3))
            if cfg == "C5" and ", true" not in name:     # <.., UPDATE = true[, MULTI]>: the update sweep, not the known-pulse sweeps
                continue
            # the caller's launches only: the CAL = 1 instantiations are kr_create's short calibration sweeps
            if re.search(r"<[^>]*>", name) and cfg in ("C3", "C4"):
                args = [x.strip() for x in re.search(r"<([^>]*)>", name).group(1).split(",")]
                cal = args[4] if cfg == "C3" else args[4]
                if cal == "1":
                    continue
            if (sect, "n") not in vals or n >= vals[(sect, "n")]:
                vals[(sect, "n")], vals[sect], vals["kernel"] = n, v, name
    if "fetch" not in vals or "write" not in vals:
        continue
    traffic = int((2.0 * vals["fetch"] + vals["write"]) * 1024)
    alg = K * NT * 32 * N
    out.append({"source": "profiles/%s_bench_%s_summary.txt (tools/profile.sh %s_%s %s: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, "
                          "separate passes, command: python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --config %s)" % (tag, cfg.lower(), tag, cfg, cfg, cfg),
                "kernel": vals["kernel"], "config": cfg, "fetch_size_raw_KiB_per_launch": vals["fetch"], "write_size_KiB_per_launch": vals["write"],
                "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE reads 1/2 of wide coalesced 16-B/lane reads on gfx950 -> doubled; WRITE_SIZE exact.",
                "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round(traffic / alg, 3)})
with open(os.path.join(root, "profiles", "%s_traffic_all.json" % tag), "w") as f:
    json.dump(out, f, indent=1)
for e in out:
    print(e["config"], e["kernel"], e["traffic_bytes_per_launch"], e["algorithmic_bytes_per_launch"], e["traffic_over_algorithmic"])
