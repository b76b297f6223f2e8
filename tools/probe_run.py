#!/usr/bin/env python3
"""Issue-rate probes of single VALU instructions on gfx950 (8 independent chains per lane)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))
import tkmk
tkmk.set_device(0)
names = {100: "v_add_co_u32", 101: "v_addc_co_u32", 102: "v_add_u32", 103: "v_add3_u32", 104: "v_mov_b32", 105: "v_cndmask_b32",
         106: "v_mul_lo_u32", 107: "v_mul_hi_u32", 108: "v_mad_u32_u24", 109: "v_lshl_add_u64", 110: "v_mad_u64_u32",
         111: "mad_u64+addc pair", 112: "v_fma_f64", 113: "v_mad_i32_i24", 114: "v_alignbit_b32", 115: "v_mad_u32_u16",
         116: "v_cmp + 8 cndmask (per instr, 9 instrs)", 117: "v_and + v_addc pair", 118: "v_sub_u32", 119: "v_and_b32", 120: "v_lshrrev_b32",
         121: "v_lshrrev_b64", 122: "v_and_or_b32", 123: "v_bfe_u32", 124: "v_lshl_add_u32", 125: "v_mul_u32_u24"}
blocks, iters = 256 * 8, 4000
out = {}
for occ_blocks in (256 * 8,):
    for kind, name in names.items():
        ms = tkmk.diag_bench(kind, iters, occ_blocks, reps=3)
        lane_ops = occ_blocks * 256 * iters * 8 / (ms * 1e-3)
        # cycles per wave-instruction per SIMD at an assumed 2.4 GHz: 1024 SIMDs
        cyc = 2.4e9 * 1024 * 64 / lane_ops
        out.setdefault(name, {})["blocks%d" % occ_blocks] = {"lane_ops_per_s": lane_ops, "cyc_per_wave_instr_at_2.4GHz": round(cyc, 2)}
print(json.dumps(out, indent=1))
