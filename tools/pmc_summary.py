#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per launch for every kernel.
usage: pmc_summary.py OUT.json NAME=DIR [NAME=DIR ...]   (DIR = the -d directory of one --pmc pass)"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"((?:\w+::)*\w+(?:<[^(]*>)?)\(", name)
    return m.group(1) if m else name.split("(")[0]


def summarise(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = short(row["Kernel_Name"])
            key = (k, row["Dispatch_Id"])
            acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
    per = {}
    for (k, _), v in acc.items():
        per.setdefault(k, []).append(v)
    return {k: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for k, v in per.items()}


if __name__ == "__main__":
    out = {}
    for a in sys.argv[2:]:
        name, d = a.split("=", 1)
        out[name] = summarise(d, name)
    json.dump(out, open(sys.argv[1], "w"), indent=1)
