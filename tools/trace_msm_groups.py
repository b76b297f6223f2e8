#!/usr/bin/env python3
"""Per-MSM kernel time from a rocprofv3 --kernel-trace CSV taken with TKMK_MSM_STREAMS=1 (one stream: the kernels of one MSM are consecutive):
the LAST S seconds are cut at every k_digits launch; for each group: accumulate, sort (digits, hist, scan, scatter), tail (combine, reduce) and the
workgroups of the accumulate launch (= lanes / 256: how many entries it had).  usage: python tools/trace_msm_groups.py <dir> <seconds>"""
import csv
import glob
import json
import os
import sys

d, last = sys.argv[1], float(sys.argv[2])
path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:40], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0))
              for r in csv.DictReader(open(path)))
t_end = max(e for _, e, _, _ in rows)
rows = [r for r in rows if r[0] >= t_end - int(last * 1e9)]
SORT = ("k_digits", "k_hist", "k_scan", "k_scatter", "k_bstart", "k_count_entries")
TAIL = ("k_combine", "k_reduce")
groups, cur = [], None
for s, e, n, g in rows:
    if n.startswith("k_digits"):
        cur = {"at_ms": round((s - rows[0][0]) / 1e6, 2), "accumulate_ms": 0.0, "sort_ms": 0.0, "tail_ms": 0.0, "acc_lanes": 0, "tail_kernels": {}}
        groups.append(cur)
    if cur is None:
        continue
    dt = (e - s) / 1e6
    if n.startswith("k_accumulate"):
        cur["accumulate_ms"] += dt
        cur["acc_lanes"] = g
    elif n.startswith(SORT):
        cur["sort_ms"] += dt
    elif n.startswith(TAIL):
        cur["tail_ms"] += dt
        cur["tail_kernels"][n] = round(cur["tail_kernels"].get(n, 0.0) + dt, 3)
for g in groups:
    for k in ("accumulate_ms", "sort_ms", "tail_ms"):
        g[k] = round(g[k], 3)
print(json.dumps({"msm_launches": len(groups), "accumulate_ms": round(sum(g["accumulate_ms"] for g in groups), 2), "sort_ms": round(sum(g["sort_ms"] for g in groups), 2),
                  "tail_ms": round(sum(g["tail_ms"] for g in groups), 2), "groups": groups}, indent=1))
