"""CRS handle + encode_poly: the work-alike of Sigma1::encode_poly / encode_poly_from_xy_powers
(packages/backend/libs/src/iotools/mod.rs:2033-2113; macro twin libs/src/group_structures/mod.rs:59-119).

The reference decodes and re-uploads the needed CRS sub-grid (up to 2^22 x 96 B) on EVERY commit; here the
xy_powers table is uploaded once and stays resident in HBM (17 of the 19 commits of a
proof read sub-grids of the same table), and a commit is: degree scan -> gather of the coefficient box and the
matching CRS rows (two strided device copies) -> one MSM with device operands."""
import ctypes

import numpy as np

import tkmk
from tkmk.poly import DensePolynomialExt


class Sigma1:
    dist = None            # torch.distributed module with an initialised process group: the commits of a round are spread over the ranks
    comm_device = "cuda"   # where the gathered results travel ("cpu" for a gloo rehearsal)

    def __init__(self, xy_powers, rs_x_size, rs_y_size):
        """xy_powers[i*rs_y_size + j] = [tau_x^i tau_y^j]G as 96-byte affine records (host array or DeviceBuffer);
        rs_x_size = max(2n, 2(l_D - l)), rs_y_size = 2 s_max (iotools/mod.rs:2050-2051)"""
        if tkmk._len(xy_powers) * 32 != rs_x_size * rs_y_size * 96:
            raise ValueError("xy_powers has the wrong length")
        self.xy_powers = xy_powers if isinstance(xy_powers, tkmk.DeviceBuffer) else tkmk.DeviceBuffer.from_host(xy_powers)
        self.rs_x_size, self.rs_y_size = rs_x_size, rs_y_size

    def _gather(self, poly: DensePolynomialExt):
        """(scalars, bases, n) of the commit MSM for `poly`, or None for the zero polynomial"""
        poly.optimize_size()
        tx, ty = poly.x_degree + 1, poly.y_degree + 1
        if tx > self.rs_x_size or ty > self.rs_y_size:
            raise ValueError("Insufficient length of sigma.sigma_1.xy_powers")
        if tx * ty == 0:
            return None
        # compact (tx x ty) scalar box and the matching CRS sub-grid, both gathered on the device
        scalars = tkmk.DeviceBuffer(32 * tx * ty)
        bases = tkmk.DeviceBuffer(96 * tx * ty)
        lib = tkmk.lib()
        tkmk._check(lib.tkmk_memcpy_2d_d2d(tkmk._p(scalars), ctypes.c_size_t(32 * ty), tkmk._p(poly.poly), ctypes.c_size_t(32 * poly.y_size),
                                           ctypes.c_size_t(32 * ty), ctypes.c_size_t(tx)), "tkmk_memcpy_2d_d2d")
        tkmk._check(lib.tkmk_memcpy_2d_d2d(tkmk._p(bases), ctypes.c_size_t(96 * ty), tkmk._p(self.xy_powers), ctypes.c_size_t(96 * self.rs_y_size),
                                           ctypes.c_size_t(96 * ty), ctypes.c_size_t(tx)), "tkmk_memcpy_2d_d2d")
        return scalars, bases, tx * ty

    def encode_poly(self, poly: DensePolynomialExt):
        """-> 96-byte affine commitment (G1serde); all-zero = G1serde::zero()"""
        job = self._gather(poly)
        if job is None:
            return np.zeros(96, np.uint8)
        return tkmk.projective_to_affine_bytes(tkmk.msm(job[0], job[1]))

    def encode_polys(self, polys):
        """commitments of several independent polynomials (e.g. prove0's U, V, W, Q_AX, Q_AY, B:
        prove/src/lib.rs prove0) in one pipelined tkmk_msm_multi call -> [96-byte affine, ...]"""
        jobs = [self._gather(p) for p in polys]
        live = [j for j in jobs if j is not None]
        if self.dist is not None and self.dist.get_world_size() > 1 and live:
            # every rank holds the same polynomials (the rounds are replicated, Fiat-Shamir keeps them in lock step); each runs
            # the MSMs it owns through its own pipelined multi-MSM and ONE all_gather returns all commitments
            from tkmk import sharding
            proj = sharding.commits_balanced(tkmk.msm_multi, self.dist, live, [j[2] for j in live], device=self.comm_device)
            res = tkmk.projective_to_affine_bytes(np.ascontiguousarray(proj).reshape(-1))
        else:
            res = tkmk.projective_to_affine_bytes(tkmk.msm_multi(live)) if live else np.zeros(0, np.uint8)
        out, k = [], 0
        for j in jobs:
            if j is None:
                out.append(np.zeros(96, np.uint8))
            else:
                out.append(res[96 * k:96 * (k + 1)].copy())
                k += 1
        return out
