#!/usr/bin/env python3
"""Micro-benchmarks of the arithmetic primitives on the GPU box (diagnostics; prints one JSON line)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))
import tkmk  # noqa: E402

tkmk.set_device(0)
out = {}
blocks = 256 * 8
threads = blocks * 256
for name, kind, iters, ops_per_iter in (("fr_mul", 0, 2000, 1), ("fq_mul", 1, 1000, 1), ("mad_u64_u32", 2, 4000, 8),
                                        ("g1_add_mixed", 3, 100, 1), ("fr_add_sub", 4, 4000, 2), ("fq_sqr", 5, 1000, 1),
                                        ("fr_mul_unsat_9x29", 6, 2000, 1), ("fr_butterfly_unsat_9x29", 7, 1000, 1),
                                        ("fr_butterfly_saturated", 8, 1000, 1)):
    ms = tkmk.diag_bench(kind, iters, blocks, reps=3)
    rate = threads * iters * ops_per_iter / (ms * 1e-3)
    out[name] = {"ms": round(ms, 3), "ops_per_s": rate}
out["fr_mul_mads_per_s"] = out["fr_mul"]["ops_per_s"] * 136
out["fq_mul_mads_per_s"] = out["fq_mul"]["ops_per_s"] * 300
print(json.dumps(out))
