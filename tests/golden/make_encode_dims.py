#!/usr/bin/env python3
"""Regenerates tests/golden/encode_dims.json from the reference tree (run in the build container only).

The one output-shaped record the reference holds for the commitment path: the (x_degree + 1) x (y_degree + 1) box every
encode_poly of a production-shape proof ran its MSM over — the `msm=AxB` column of "Encode Details (by variable)" in
packages/backend/prove/optimization/timing.local.cpu.current.md:241-261, written by the reference's own timing build
(prove/src/lib.rs `timing` feature, events of category "encode") — together with the setup parameters of that run.
Cross-checked here against the machine-readable twin (timing.local.cpu.current.json, category "encode") and against the CUDA run's
reports (timing.remote.no-output-optimize-size.repeat.cuda.{md,json}): all four must agree.
The reports are parsed as text / JSON data; nothing from the reference is executed."""
import json
import os
import re

OPT = "/root/reference/packages/backend/prove/optimization"


def from_md(path):
    text = open(path).read()
    sec = text[text.index("## Encode Details (by variable)"):]
    rows = {}
    for m in re.finditer(r"^\| (\w+) \| (\w+) \| [0-9.]+ s \| msm=(\d+)x(\d+) \|$", sec, re.M):
        rows[m.group(2)] = {"module": m.group(1), "x": int(m.group(3)), "y": int(m.group(4))}
    return rows


def from_json(path):
    doc = json.load(open(path))
    rows = {}
    for e in doc["events"]:
        if e["category"] == "encode":
            module, _, name = e["name"].split(".")
            (size,) = e["sizes"]
            assert size["label"] == "msm"
            rows[name] = {"module": module, "x": size["dims"][0], "y": size["dims"][1]}
    return rows, doc["setup_params"]


cpu_md = from_md(OPT + "/timing.local.cpu.current.md")
cpu_js, sp = from_json(OPT + "/timing.local.cpu.current.json")
cuda_md = from_md(OPT + "/timing.remote.no-output-optimize-size.repeat.cuda.md")
cuda_js, sp_cuda = from_json(OPT + "/timing.remote.no-output-optimize-size.repeat.cuda.json")
assert cpu_md == cpu_js == cuda_md == cuda_js and sp == sp_cuda and len(cpu_md) == 19
out = {"source": "packages/backend/prove/optimization/timing.local.cpu.current.md:241-261 (= .json category 'encode' = the CUDA run's reports)",
       "setup_params": sp, "boxes": cpu_md}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "encode_dims.json"), "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
print("wrote %d boxes" % len(cpu_md))
