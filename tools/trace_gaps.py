"""Timeline summary of a rocprofv3 --kernel-trace CSV: how much of the wall a GPU is busy (union of kernel intervals), the largest idle
gaps with the kernels either side, and how much of the busy time has only latency-bound MSM tail kernels in flight.
usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [--last-seconds S | S] [--copies]"""
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    last_s = float(sys.argv[sys.argv.index("--last-seconds") + 1]) if "--last-seconds" in sys.argv else None
    if last_s is None and len(sys.argv) > 2 and sys.argv[2].replace(".", "", 1).isdigit():
        last_s = float(sys.argv[2])   # short form: the window as the second argument
    path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:40]))
    rows.sort()
    t_end = max(e for _, e, _ in rows)
    if last_s is not None:
        rows = [r for r in rows if r[0] >= t_end - int(last_s * 1e9)]
    t0, t1 = rows[0][0], max(e for _, e, _ in rows)
    busy, cur_s, cur_e, gaps, last_name = 0, rows[0][0], rows[0][1], [], rows[0][2]
    for s, e, name in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_name, name, (cur_e - t0) / 1e6))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        if e >= cur_e:
            last_name = name
    busy += cur_e - cur_s
    gaps.sort(reverse=True)
    # time during which an accumulate kernel is in flight
    acc = [(s, e) for s, e, n in rows if "accumulate" in n]
    acc_busy, cs, ce = 0, None, None
    for s, e in acc:
        if cs is None:
            cs, ce = s, e
        elif s > ce:
            acc_busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if cs is not None:
        acc_busy += ce - cs
    out = {"kernels": len(rows), "span_ms": (t1 - t0) / 1e6, "busy_union_ms": busy / 1e6, "idle_ms": (t1 - t0 - busy) / 1e6,
           "accumulate_in_flight_ms": acc_busy / 1e6, "gaps_over_50us": sum(1 for g in gaps if g[0] > 50e3),
           "idle_in_gaps_over_50us_ms": sum(g[0] for g in gaps if g[0] > 50e3) / 1e6,
           "top_gaps": [{"us": g[0] / 1e3, "after": g[1], "before": g[2], "at_ms": round(g[3], 2)} for g in gaps[:40]]}
    if "--copies" in sys.argv:      # device copies over 30 us with the kernels launched around them (which host operation they belong to)
        big = []
        for i, (s, e, n) in enumerate(rows):
            if "copyBuffer" in n and e - s > 30e3:
                prev = next((rows[j][2] for j in range(i - 1, -1, -1) if "copyBuffer" not in rows[j][2]), "")
                nxt = next((rows[j][2] for j in range(i + 1, len(rows)) if "copyBuffer" not in rows[j][2]), "")
                big.append({"us": (e - s) / 1e3, "at_ms": round((s - t0) / 1e6, 2), "prev": prev, "next": nxt})
        out["copies_over_30us"] = {"count": len(big), "total_ms": sum(b["us"] for b in big) / 1e3, "list": big}
        small = [(e - s) for s, e, n in rows if "copyBuffer" in n and e - s <= 30e3]
        out["copies_under_30us"] = {"count": len(small), "total_ms": sum(small) / 1e6}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
