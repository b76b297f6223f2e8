import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))
import tkmk, oracle
tkmk.set_device(0)
n = 300
s = oracle.fr_random(77, n); p = oracle.g1_random_bases(78, n)
want = oracle.g1_msm(s, p)
for c in [int(x) for x in sys.argv[1:]]:
    got = tkmk.projective_to_affine_bytes(tkmk.msm(s, p, c=c))
    print("c", c, "ok" if (got == want).all() else "MISMATCH", flush=True)
