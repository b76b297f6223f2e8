"""Builds tests/hostcheck/libhostcheck.so: the product's __host__ __device__ templates compiled for the HOST (test-only).
Three translation units in parallel (the unsaturated curve code is fully unrolled and slow to compile)."""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tokamak-zk-evm_amd", "csrc")
SO = os.path.join(HERE, "libhostcheck.so")
UNITS = ("hostcheck.cpp", "hostcheck_unsat_bls12_381.cpp", "hostcheck_unsat_bn254.cpp")


def stale():
    deps = glob.glob(os.path.join(HERE, "*.cpp")) + glob.glob(os.path.join(HERE, "*.h")) + \
        [os.path.join(CSRC, f) for f in ("ff.h", "ec.h", "field_params.h", "ntt_plan.h", "ffu.h", "ec_u.h")]
    return not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps)


def build(force=False):
    """returns the path of the library, or None when hipcc is missing"""
    if not force and not stale():
        return SO
    if shutil.which("hipcc") is None:
        return None
    objs, procs = [], []
    for u in UNITS:
        o = os.path.join(HERE, u[:-4] + ".o")
        objs.append(o)
        procs.append(subprocess.Popen(["hipcc", "-O1", "-fPIC", "-c", "--offload-host-only", "-I" + CSRC, os.path.join(HERE, u), "-o", o]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hostcheck compile failed")
    subprocess.run(["hipcc", "-shared", "-fPIC", "--offload-host-only", "-o", SO] + objs, check=True)
    for o in objs:
        os.remove(o)
    return SO
