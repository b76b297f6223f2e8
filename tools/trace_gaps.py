#!/usr/bin/env python3
"""Device idle time inside the LAST S seconds of a rocprofv3 --kernel-trace CSV: union of the kernels' [start, end] intervals over all
streams, the idle gaps between them (largest first, with the kernels on either side).  usage: python tools/trace_gaps.py <dir> <seconds>"""
import csv
import glob
import json
import os
import sys

d, last = sys.argv[1], float(sys.argv[2])
path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:40]) for r in csv.DictReader(open(path)))
t_end = max(e for _, e, _ in rows)
rows = [r for r in rows if r[0] >= t_end - int(last * 1e9)]
busy, gaps = 0, []
cur_s, cur_e, cur_name = rows[0]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_name, n, (cur_e - rows[0][0]) / 1e6))
        cur_s, cur_e, cur_name = s, e, n
    elif e > cur_e:
        cur_e, cur_name = e, n
busy += cur_e - cur_s
span = cur_e - rows[0][0]
gaps.sort(reverse=True)
print(json.dumps({"window_s": last, "span_ms": span / 1e6, "busy_ms": busy / 1e6, "idle_ms": (span - busy) / 1e6, "gaps": len(gaps),
                  "gaps_over_50us_ms": sum(g[0] for g in gaps if g[0] > 50000) / 1e6,
                  "largest": [{"ms": round(g[0] / 1e6, 3), "after": g[1], "before": g[2], "at_ms": round(g[3], 2)} for g in gaps[:40]]}, indent=1))
