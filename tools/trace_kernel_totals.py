#!/usr/bin/env python3
"""Per-kernel totals over the LAST S seconds of a rocprofv3 --kernel-trace CSV (a steady-state proof at the end of a run, without the
one-time kernels of staging and open that --stats mixes in).  usage: python tools/trace_kernel_totals.py <dir> <seconds> [divide_by]"""
import collections
import csv
import glob
import json
import os
import sys

d, last = sys.argv[1], float(sys.argv[2])
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:44]) for r in csv.DictReader(open(path))]
t_end = max(e for _, e, _ in rows)
rows = [r for r in rows if r[0] >= t_end - int(last * 1e9)]
tot, cnt = collections.Counter(), collections.Counter()
for s, e, n in rows:
    tot[n] += e - s
    cnt[n] += 1
out = {"window_s": last, "kernels": len(rows), "sum_ms": round(sum(tot.values()) / 1e6 / div, 2),
       "by_kernel_ms": {k: [round(v / 1e6 / div, 3), cnt[k]] for k, v in tot.most_common(30)}}
print(json.dumps(out, indent=1))
