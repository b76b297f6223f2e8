import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tokamak-zk-evm_amd'))
import tkmk
tkmk.set_device(0)
for mb in (1, 32, 256, 1024):
    n = mb << 20
    b = tkmk.DeviceBuffer(n); b.free()
    t0 = time.perf_counter()
    for _ in range(20):
        b = tkmk.DeviceBuffer(n); b.free()
    print(mb, "MiB alloc+free ms", (time.perf_counter() - t0) / 20 * 1e3)
