"""`preprocess` round over the device path: the work-alike of Preprocess::gen
(packages/backend/preprocess/src/lib.rs:32-82): permutation polynomials s0, s1 (Permutation::to_poly, two inverse
bivariate NTTs), their commitments (two encode_poly MSMs against the resident xy_powers table, issued as one pipelined
call) and the fixed public-input binding commitment O_pub_fix (a gathered MSM over the tail of gamma_inv_o_inst,
libs/src/group_structures/mod.rs:145-182), then the Solidity-verifier formatting.

The CRS arrives as device-resident tables (tkmk.sigma.Sigma1 for xy_powers; gamma_inv_o_inst as a 96-byte-record buffer):
reading sigma_preprocess.rkyv is a storage concern outside the hot path."""
import numpy as np

import tkmk
from tkmk import proofio, witness
from tkmk.r1cs import hex_to_fr


def encode_O_pub_fix(gamma_inv_o_inst, a_pub_function, setup_params):
    """group_structures/mod.rs:145-182: scalars = a_pub_function (hex), bases = the last m_function entries of
    gamma_inv_o_inst; zero when m_function == 0; length mismatches raise (the reference panics)"""
    m_function = setup_params["l"] - setup_params["l_free"]
    if m_function == 0:
        return np.zeros(96, np.uint8)
    if len(a_pub_function) != m_function:
        raise ValueError("a_pub_function length mismatch: expected m_function=%d, got %d" % (m_function, len(a_pub_function)))
    n_bases = tkmk._len(gamma_inv_o_inst) * 32 // 96
    if n_bases < m_function:
        raise ValueError("gamma_inv_o_inst length is smaller than m_function")
    scalars = np.frombuffer(b"".join(hex_to_fr(v).to_bytes(32, "little") for v in a_pub_function), np.uint8).copy()
    start = n_bases - m_function
    if isinstance(gamma_inv_o_inst, tkmk.DeviceBuffer):
        bases = tkmk.DeviceBuffer(96 * m_function)
        tkmk._check(tkmk.lib().tkmk_memcpy_d2d(tkmk._p(bases), tkmk.ctypes.c_void_p(gamma_inv_o_inst.ptr + 96 * start),
                                               tkmk.ctypes.c_size_t(96 * m_function)), "tkmk_memcpy_d2d")
        scalars = tkmk.DeviceBuffer.from_host(scalars)
    else:
        bases = np.ascontiguousarray(gamma_inv_o_inst[96 * start:])
    return tkmk.projective_to_affine_bytes(tkmk.msm(scalars, bases))


class Preprocess:
    def __init__(self, s0, s1, O_pub_fix):
        self.s0, self.s1, self.O_pub_fix = s0, s1, O_pub_fix

    @classmethod
    def gen(cls, sigma1, gamma_inv_o_inst, permutation_raw, instance, setup_params):
        m_i = setup_params["l_D"] - setup_params["l"]
        s_max = setup_params["s_max"]
        tkmk.init_ntt_domain_for_size(4 * max(m_i, setup_params["n"]) * 2 * s_max)      # libs/src/utils/mod.rs:51-58
        s0XY, s1XY = witness.permutation_to_poly(permutation_raw, m_i, s_max)
        s0, s1 = sigma1.encode_polys([s0XY, s1XY])
        return cls(s0, s1, encode_O_pub_fix(gamma_inv_o_inst, instance["a_pub_function"], setup_params))

    def convert_format_for_solidity_verifier(self):
        return proofio.format_preprocess({"s0": self.s0, "s1": self.s1, "O_pub_fix": self.O_pub_fix})

    @classmethod
    def recover_from_format(cls, fmt):
        p = proofio.recover_preprocess(fmt)
        return cls(p["s0"], p["s1"], p["O_pub_fix"])
