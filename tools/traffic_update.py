#!/usr/bin/env python3
"""profiles/traffic.json entry for k_accumulate_chunks from one tools/profile_round.sh output directory:
FETCH_SIZE (KiB, separate --pmc pass) x the 96-byte gather-probe calibration of the same session (sector bytes / raw counter) + WRITE_SIZE.
usage: traffic_update.py gpurun_out/prof_TAG KEY 'workload text' profiles/NAME_OF_COMMITTED_PMC.json"""
import json
import sys

d, key, workload, committed = sys.argv[1:5]
pm = json.load(open(d + "/pmc.json"))
probe = json.load(open(d + "/pmc_probe.json"))["FETCH_SIZE"]
plain = json.load(open(d + "/gather_probe_under_pmc.json"))
K = "tk_msm_bls12_381::k_accumulate_chunks"
raw = pm["FETCH_SIZE"][K]["mean_per_launch"] * 1024
wr = pm["WRITE_SIZE"][K]["mean_per_launch"] * 1024
probe_raw = probe["k_gather_probe<6>"]["mean_per_launch"] * 1024          # 96-byte rows = 6 x 16-byte loads
sectors = plain["row_96"]["bytes_in_64B_sectors"] + plain["row_96"]["index_bytes_per_launch"]
cal = sectors / probe_raw
t = json.load(open("profiles/traffic.json"))
t[key] = {"kernel": "k_accumulate_chunks", "workload": workload, "launches": pm["FETCH_SIZE"][K]["launches"], "fetch_bytes_raw": raw,
          "calibration_sector_bytes_over_raw": cal, "fetch_bytes_calibrated": raw * cal, "write_bytes": wr, "hbm_bytes_per_launch": raw * cal + wr,
          "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes with TKMK_MSM_STREAMS=1 (%s, tools/profile_round.sh); FETCH_SIZE (KiB) calibrated "
                  "with the 96-byte gather probe of the same session" % committed}
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(t[key], indent=1))
