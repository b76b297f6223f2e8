import sys, time, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tokamak-zk-evm_amd'))
import tkmk, oracle
tkmk.set_device(0)
p = oracle.g1_random_bases(3, 8)
ones = np.zeros(32*8, np.uint8); ones[0::32]=1
tkmk.msm(ones, p)
t=time.perf_counter()
for _ in range(20): tkmk.msm(ones, p)
print("8-point msm ms", (time.perf_counter()-t)/20*1e3)
