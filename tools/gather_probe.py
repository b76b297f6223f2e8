"""Known-bytes gather probe for the FETCH_SIZE calibration (profiles/traffic.json): gathers n rows of 96 bytes
(and, for comparison, 64 / 128 bytes) at uniformly random rows of a table much larger than the 256 MiB Infinity Cache, with the
access pattern of k_accumulate_chunks (one lane = one row = consecutive 16-byte loads).  Run once plainly (prints the launch time and
the exact byte / sector / line counts of the index list) and once under `rocprofv3 --pmc FETCH_SIZE` (tools/profile_round.sh does
both); FETCH_SIZE per launch / the known counts says what the counter tallies for this pattern."""
import ctypes
import json
import sys
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))


def main():
    import tkmk
    tkmk.set_device(0)
    rows_log, n_log = 25, 26                                  # 2^25 rows (3 GiB at 96 B), 2^26 gathers
    rng = np.random.default_rng(7)
    out = {}
    for row_bytes in (96, 64, 128):
        rows, n = 1 << rows_log, 1 << n_log
        table = tkmk.DeviceBuffer(rows * row_bytes)
        tkmk.lib().tkmk_memset(tkmk._p(table), 0x11, ctypes.c_size_t(rows * row_bytes))
        idx = rng.integers(0, rows, n, dtype=np.uint32)
        d_idx = tkmk.DeviceBuffer.from_host(idx.view(np.uint8))
        ms = ctypes.c_float()
        tkmk._check(tkmk.lib().tkmk_diag_gather_probe(tkmk._p(table), row_bytes, tkmk._p(d_idx), ctypes.c_uint64(n), 3, ctypes.byref(ms)),
                    "tkmk_diag_gather_probe")
        off = idx.astype(np.uint64) * row_bytes
        sectors = int(((off + row_bytes - 1) // 64 - off // 64 + 1).sum())
        lines = int(((off + row_bytes - 1) // 128 - off // 128 + 1).sum())
        out["row_%d" % row_bytes] = {"rows_in_table": rows, "gathers_per_launch": n, "launches": 4, "useful_bytes_per_launch": n * row_bytes,
                                     "index_bytes_per_launch": 4 * n, "bytes_in_64B_sectors": 64 * sectors, "bytes_in_128B_lines": 128 * lines,
                                     "launch_ms": ms.value, "useful_GBps": n * row_bytes / (ms.value * 1e-3) / 1e9}
        table.free()
        d_idx.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
