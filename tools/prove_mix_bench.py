#!/usr/bin/env python3
"""Hot-path replay of one `prove` on the production shape (n = 4096, s_max = 256, m_I = 4096: 2^20 constraint slots).

NOT a prover: it issues, on synthetic data resident in HBM, the MSM / NTT / polynomial-op sequence the reference's
prove0..prove4 issue through ICICLE, with the shapes recorded in the reference's own timing report
(packages/backend/prove/optimization/timing.local.cpu.current.md:243-261 for the 19 encode_poly MSM sizes; SURVEY.md
§3.1 for the per-round NTT / division / evaluation list).  Protocol glue (Keccak transcript, JSON / rkyv IO, sparse
R1CS witness evaluation, the prove1 running product) is not included.  Prints one JSON line.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))
import numpy as np  # noqa: E402
import tkmk  # noqa: E402
from tkmk.poly import DensePolynomialExt as P  # noqa: E402
from tkmk.sigma import Sigma1  # noqa: E402

tkmk.set_device(0)
S_MAX = 256
for a in sys.argv[1:]:
    if a.startswith("--s-max="):
        S_MAX = int(a.split("=")[1])    # 256 = production library; 1024 = the "2^22-constraint" shape of SURVEY.md section 8d cfg 4
n, s_max, m_i = 4096, S_MAX, 4096
tkmk.init_ntt_domain_for_size(4 * max(m_i, n) * 2 * s_max)            # libs/src/utils/mod.rs:51-58
rs_x, rs_y = max(2 * n, 2 * m_i), 2 * s_max                            # xy_powers: 8192 x 512 = 2^22 points

# CRS: any 2^22 curve points do for timing; generated once on the device
h = tkmk.fr_random_device(1, rs_x * rs_y)
g = np.frombuffer(bytes.fromhex(
    "bbc622db0af03afbef1a7af93fe8556c58ac1b173f3a4ea105b974974f8c68c30faca94f8c63952694d79731a7d3f117"
    "e1e7c5462923aa0ce48a88a244c73cd0edb3042ccb18db00f60ad0d595e0f5fce48a1d74ed309ea0f1a0aae381f4b308"), np.uint8).copy()
crs = tkmk.g1_batch_scalar_mul_device(h, g, rs_x * rs_y)
sigma = Sigma1(crs, rs_x, rs_y)


MULTI = "--sequential-commits" not in sys.argv     # default: independent commits of one round go through tkmk_msm_multi


def commits(shapes, seed):
    """the encode_poly MSMs of one round: scalars of each shape against the resident CRS"""
    scal = [tkmk.fr_random_device(seed, xs * ys) for xs, ys in shapes]
    if MULTI:
        tkmk.msm_multi([(sc, crs, xs * ys) for sc, (xs, ys) in zip(scal, shapes)])
    else:
        for sc, (xs, ys) in zip(scal, shapes):
            tkmk.msm(sc, crs, msm_size=xs * ys)


def run():
    t = {}
    def tick(name, t0):
        tkmk.synchronize()
        t[name] = t.get(name, 0.0) + time.perf_counter() - t0

    one = np.zeros(32, np.uint8); one[0] = 1
    chi, zeta = (np.frombuffer(tkmk.fr_random_device(5 + k, 1).to_host(), np.uint8).copy() for k in range(2))
    # ---- init: bXY, u/v/w, s0/s1 : 6 iNTT 4096x256; binding commitments (gathered MSMs, sizes ~ wires of used placements)
    t0 = time.perf_counter()
    ev = [tkmk.fr_random_device(10 + k, n * s_max) for k in range(6)]
    polys = [P.from_rou_evals(e, n, s_max) for e in ev]
    tick("init.intt", t0)
    t0 = time.perf_counter()
    commits(((128, 1), (728, 1), (2048 * s_max, 1), (2048 * s_max, 1)), 20)     # A_free, O_pub, O_mid, O_prv (order of magnitude)
    tick("init.binding_msm", t0)
    u, v, w = polys[1], polys[2], polys[3]
    # ---- prove0: p0 = u*v - w (2 fwd + 1 inv NTT at 8192x512), div_by_vanishing_opt, 6 commits
    t0 = time.perf_counter()
    p0 = (u * v) - w
    tick("prove0.poly", t0)
    t0 = time.perf_counter()
    q_ax, q_ay = p0.div_by_vanishing_opt(n, s_max)
    tick("prove0.div_by_vanishing", t0)
    t0 = time.perf_counter()
    commits(((n + 1, s_max + 1), (n + 1, s_max + 1), (n + 1, s_max + 1), (n, s_max), (n, s_max), (n + 3, s_max + 3)), 30)   # U,V,W,Q_AX,Q_AY,B
    tick("prove0.encode", t0)
    # ---- prove1: 2 fwd NTT 4096x256, batched division, 2 transposes, 1 iNTT, 1 commit
    t0 = time.perf_counter()
    f = polys[4].to_rou_evals()
    gg = polys[5].to_rou_evals()
    quot = tkmk.vec_div(f, gg)
    tr = tkmk.transpose(quot, n, s_max)
    tr2 = tkmk.transpose(tr, s_max, n)
    r_poly = P.from_rou_evals(tr2, n, s_max)
    tick("prove1.poly", t0)
    t0 = time.perf_counter()
    commits(((n + 1, s_max + 1),), 31)
    tick("prove1.encode", t0)
    # ---- prove2: 2 scale_coeffs, 3 Lagrange iNTTs, p_comb fused on 16384x512 (7 leaf NTTs + ~15 pointwise + 1 inverse),
    #              div_by_vanishing_opt, 2 commits
    t0 = time.perf_counter()
    r_wx = r_poly.scale_coeffs_x(chi)
    r_wxy = r_wx.scale_coeffs_y(zeta)
    lag = [P.from_rou_evals(tkmk.fr_random_device(40 + k, n * s_max), n, s_max) for k in range(3)]
    dx, dy = 4 * m_i, 2 * s_max
    leaves = [polys[0], r_poly, r_wx, r_wxy] + lag
    evs = []
    for lf in leaves:
        c = lf.clone()
        c.resize(dx, dy)
        evs.append(tkmk.bintt(c.poly, dx, dy, out=c.poly))
    acc = evs[0]
    for k in range(15):
        other = evs[1 + k % 6]
        acc = tkmk.vec_mul(acc, other, out=acc) if k % 3 == 0 else (tkmk.vec_add(acc, other, out=acc) if k % 3 == 1 else tkmk.vec_sub(acc, other, out=acc))
    p_comb = P.from_rou_evals(acc, dx, dy)
    tick("prove2.poly", t0)
    t0 = time.perf_counter()
    q_cx, q_cy = p_comb.div_by_vanishing_opt(m_i, s_max)
    tick("prove2.div_by_vanishing", t0)
    t0 = time.perf_counter()
    commits(((2 * m_i, 2 * s_max - 1), (2 * m_i - 1, s_max + 1)), 32)
    tick("prove2.encode", t0)
    # ---- prove3: 4 bivariate evaluations, 2 scale_coeffs
    t0 = time.perf_counter()
    for pl in (v, r_poly, r_wx, r_wxy):
        pl.eval(chi, zeta)
    r_poly.scale_coeffs_x(chi)
    r_poly.scale_coeffs_y(zeta)
    tick("prove3.poly", t0)
    # ---- prove4: 5 div_by_ruffini, a few _mul at 8192x512, 9 commits
    t0 = time.perf_counter()
    big = P.from_coeffs(tkmk.fr_random_device(50, 2 * m_i * 2 * s_max), 2 * m_i, 2 * s_max)
    for k in range(5):
        big.div_by_ruffini(chi, zeta)
    for k in range(3):
        _ = u * v
    tick("prove4.poly", t0)
    t0 = time.perf_counter()
    commits(((4825, s_max + 2), (n + 1, 2 * s_max - 1), (n + 2, 2 * s_max - 1), (2 * m_i - 1, 2 * s_max - 1), (1, s_max), (1, s_max), (1, 2 * s_max - 2),
             (1, 2 * s_max - 2), (127, 1)), 33)
    tick("prove4.encode", t0)
    return t


run()                      # warm-up (arena growth, code objects)
t0 = time.perf_counter()
sections = run()
total = time.perf_counter() - t0
slots = n * s_max
print(json.dumps({"commits": "tkmk_msm_multi per round" if MULTI else "sequential bls12_381_msm", "workload": "prove hot-path replay, n=%d s_max=%d m_I=%d (2^%d constraint slots%s), synthetic data" % (n, s_max, m_i, (n * s_max).bit_length() - 1, ", production shape" if s_max == 256 else ""),
                  "total_s": total, "constraint_slots_per_s": slots / total,
                  "msm_s": sum(v for k, v in sections.items() if "encode" in k or "msm" in k),
                  "poly_s": sum(v for k, v in sections.items() if "encode" not in k and "msm" not in k),
                  "sections_ms": {k: round(v * 1e3, 3) for k, v in sections.items()},
                  "reference": {"cpu_total_s": 45.70, "cuda_total_s": 21.08, "cpu_msm_s": 24.33, "cuda_msm_s": 1.27,
                                "source": "packages/backend/prove/optimization/timing.*.md (whole prove incl. host glue; other hardware)"}}))
