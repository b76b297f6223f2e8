"""prove2's p_comb-shaped pointwise expression at the production evaluation domain (16384 x 512 = 2^23 elements, 7 leaves,
15 arithmetic nodes): node-by-node passes vs the one-pass kernel (tkmk_poly_expr_eval).  Leaf NTTs excluded (cached)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tokamak-zk-evm_amd"))
import numpy as np  # noqa: E402
import tkmk  # noqa: E402
from tkmk.poly import DensePolynomialExt as P, PolyExpr as E  # noqa: E402

tkmk.set_device(0)
xs, ys = 16384, 512
tkmk.init_ntt_domain_for_size(xs * ys)
L = [P.from_coeffs(tkmk.fr_random_device(10 + k, 4096 * 256), 4096, 256) for k in range(7)]
k = [np.frombuffer(tkmk.fr_random_device(30 + i, 1).to_host(), np.uint8).copy() for i in range(4)]
p = [E.poly(x) for x in L]
# kappa-weighted copy-constraint style combination: products of shifted / plain witnesses, (X - 1) factors, Lagrange terms
expr = E.sub(
    E.add(E.mul(E.mul(p[0], p[1]), E.scale(k[0], p[2])),
          E.scale(k[1], E.mul_x_minus_one(E.sub(E.mul(p[1], p[3]), E.mul(p[2], p[4]))))),
    E.weighted_sum([(k[2], E.mul(p[5], E.sub(p[1], E.scalar(k[3])))), (k[3], E.mul(p[6], p[0])), (k[0], p[4])]))
cache = {}
expr._one_pass(xs, ys, cache)          # fills the leaf-NTT cache
tkmk.synchronize()


def timed(fn, reps=5):
    fn()
    tkmk.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    tkmk.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


a = expr._one_pass(xs, ys, cache)
b, _ = expr._on_domain(xs, ys, cache)
assert (a.to_host(1 << 20) == b.to_host(1 << 20)).all()
print(json.dumps({"domain": [xs, ys], "leaves": 7, "one_pass_ms": round(timed(lambda: expr._one_pass(xs, ys, cache)), 3),
                  "node_by_node_ms": round(timed(lambda: expr._on_domain(xs, ys, cache)), 3)}))
