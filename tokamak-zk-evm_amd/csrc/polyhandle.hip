// polyhandle.hip — the opaque univariate polynomial object behind bls12_381_polynomial_* (include/tkmk.h): work-alike of
// icicle_bls12_381::polynomials::DensePolynomial as the reference uses it for the storage of DensePolynomialExt
// (packages/backend/libs/src/bivariate_polynomial/mod.rs:112-127: from_coeffs, clone, copy_coeffs, get_coeff, coeffs_mut_slice,
// eval :1725-1737, divide :2070 in the legacy row-by-row div_by_vanishing).  A handle owns one device buffer of plain Fr
// coefficients, low degree first; clone = device copy.  Every operation is a few calls into this library's own entries
// (tkmk_poly_lincomb, tkmk_poly_eval, tkmk_bintt, ...) plus one kernel of its own, the long division.
#include <vector>

#include "common.h"

struct tkmk_polynomial {
    fr_t *c = nullptr;   // device
    size_t n = 0;        // coefficients held (>= 1)
};

static tkmk_error ph_new(size_t n, tkmk_polynomial **out) {
    tkmk_polynomial *p = new tkmk_polynomial();
    p->n = n ? n : 1;
    tkmk_error e = tkmk_malloc((void **)&p->c, p->n * sizeof(fr_t));
    if (e != TKMK_SUCCESS) {
        delete p;
        return e;
    }
    if (!n) e = tkmk_memset(p->c, 0, sizeof(fr_t));
    if (e != TKMK_SUCCESS) {
        (void)tkmk_free(p->c);
        delete p;
        return e;
    }
    *out = p;
    return TKMK_SUCCESS;
}
TK_API tkmk_error bls12_381_polynomial_delete(tkmk_polynomial *p) {
    if (!p) return TKMK_SUCCESS;
    tkmk_error e = tkmk_free(p->c);
    delete p;
    return e;
}
TK_API tkmk_error bls12_381_polynomial_create_from_coefficients(const tkmk_fr *coeffs, size_t n, bool on_device, tkmk_polynomial **out) {
    if (!out || (!coeffs && n)) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    tkmk_polynomial *p = nullptr;
    TK_TRY(ph_new(n, &p));
    tkmk_error e = n ? (on_device ? tkmk_memcpy_d2d(p->c, coeffs, n * sizeof(fr_t)) : tkmk_memcpy_h2d(p->c, coeffs, n * sizeof(fr_t))) : TKMK_SUCCESS;
    if (e != TKMK_SUCCESS) {
        (void)bls12_381_polynomial_delete(p);
        return e;
    }
    *out = p;
    return TKMK_SUCCESS;
}
// from_rou_evals: n (a power of two inside the NTT domain) evaluations on the n-th roots of unity -> coefficients (inverse NTT)
TK_API tkmk_error bls12_381_polynomial_create_from_rou_evaluations(const tkmk_fr *evals, size_t n, bool on_device, tkmk_polynomial **out) {
    if (!out || !evals) return TKMK_ERR_INVALID_POINTER;
    if (n == 0 || (n & (n - 1))) return TKMK_ERR_INVALID_ARGUMENT;
    tkmk_polynomial *p = nullptr;
    TK_TRY(bls12_381_polynomial_create_from_coefficients(evals, n, on_device, &p));
    tkmk_error e = tkmk_bintt((const tkmk_fr *)p->c, n, 1, TKMK_NTT_INVERSE, nullptr, nullptr, true, nullptr, (tkmk_fr *)p->c);
    if (e != TKMK_SUCCESS) {
        (void)bls12_381_polynomial_delete(p);
        return e;
    }
    *out = p;
    return TKMK_SUCCESS;
}
TK_API tkmk_error bls12_381_polynomial_clone(const tkmk_polynomial *p, tkmk_polynomial **out) {
    if (!p || !out) return TKMK_ERR_INVALID_POINTER;
    return bls12_381_polynomial_create_from_coefficients((const tkmk_fr *)p->c, p->n, true, out);
}
TK_API tkmk_error bls12_381_polynomial_nof_coeffs(const tkmk_polynomial *p, size_t *n) {
    if (!p || !n) return TKMK_ERR_INVALID_POINTER;
    *n = p->n;
    return TKMK_SUCCESS;
}
// highest index holding a non-zero coefficient, -1 for the zero polynomial
TK_API tkmk_error bls12_381_polynomial_degree(const tkmk_polynomial *p, int64_t *degree) {
    if (!p || !degree) return TKMK_ERR_INVALID_POINTER;
    if (p->n > 0xffffffffull) return TKMK_ERR_INVALID_ARGUMENT;
    int64_t yd;
    return tkmk_poly_find_degree((const tkmk_fr *)p->c, (uint32_t)p->n, 1, degree, &yd, nullptr);
}
// copy_coeffs(start_idx, slice): coefficients [start, start + count) into a host or device buffer
TK_API tkmk_error bls12_381_polynomial_copy_coeffs(const tkmk_polynomial *p, size_t start, size_t count, tkmk_fr *out, bool out_on_device) {
    if (!p || (!out && count)) return TKMK_ERR_INVALID_POINTER;
    if (start > p->n || count > p->n - start) return TKMK_ERR_INVALID_ARGUMENT;
    if (!count) return TKMK_SUCCESS;
    return out_on_device ? tkmk_memcpy_d2d(out, p->c + start, count * sizeof(fr_t)) : tkmk_memcpy_d2h(out, p->c + start, count * sizeof(fr_t));
}
TK_API tkmk_error bls12_381_polynomial_get_coeff(const tkmk_polynomial *p, size_t idx, tkmk_fr *out_host) {
    return bls12_381_polynomial_copy_coeffs(p, idx, 1, out_host, false);
}
// coeffs_mut_slice: the device buffer itself (valid until the handle is deleted)
TK_API tkmk_error bls12_381_polynomial_coeffs_device_ptr(tkmk_polynomial *p, tkmk_fr **ptr, size_t *n) {
    if (!p || !ptr || !n) return TKMK_ERR_INVALID_POINTER;
    *ptr = (tkmk_fr *)p->c;
    *n = p->n;
    return TKMK_SUCCESS;
}
TK_API tkmk_error bls12_381_polynomial_evaluate(const tkmk_polynomial *p, const tkmk_fr *x_host, tkmk_fr *out_host) {
    if (!p || !x_host || !out_host) return TKMK_ERR_INVALID_POINTER;
    if (p->n > 0xffffffffull) return TKMK_ERR_INVALID_ARGUMENT;
    tkmk_fr one{};
    one.limbs[0] = 1;
    return tkmk_poly_eval((const tkmk_fr *)p->c, (uint32_t)p->n, 1, x_host, &one, out_host, nullptr);
}
static tkmk_error ph_lincomb2(const tkmk_polynomial *a, const tkmk_polynomial *b, bool subtract, tkmk_polynomial **out) {
    if (!a || !b || !out) return TKMK_ERR_INVALID_POINTER;
    const size_t n = a->n > b->n ? a->n : b->n;
    if (n > 0xffffffffull) return TKMK_ERR_INVALID_ARGUMENT;
    tkmk_polynomial *r = nullptr;
    TK_TRY(ph_new(n, &r));
    tkmk_fr c[2] = {};
    c[0].limbs[0] = 1;
    if (subtract) {
        fr_t one = Fr::zero();
        one.l[0] = 1;
        fr_t m1 = Fr::from_mont(Fr::neg(Fr::to_mont(one)));
        for (int i = 0; i < 8; i++) c[1].limbs[i] = m1.l[i];
    } else {
        c[1].limbs[0] = 1;
    }
    const tkmk_fr *ptrs[2] = {(const tkmk_fr *)a->c, (const tkmk_fr *)b->c};
    uint32_t xs[2] = {(uint32_t)a->n, (uint32_t)b->n}, ys[2] = {1, 1};
    tkmk_error e = tkmk_poly_lincomb(2, c, ptrs, xs, ys, nullptr, nullptr, (tkmk_fr *)r->c, (uint32_t)n, 1, nullptr);
    if (e == TKMK_SUCCESS) e = tkmk_device_synchronize();
    if (e != TKMK_SUCCESS) {
        (void)bls12_381_polynomial_delete(r);
        return e;
    }
    *out = r;
    return TKMK_SUCCESS;
}
TK_API tkmk_error bls12_381_polynomial_add(const tkmk_polynomial *a, const tkmk_polynomial *b, tkmk_polynomial **out) { return ph_lincomb2(a, b, false, out); }
TK_API tkmk_error bls12_381_polynomial_subtract(const tkmk_polynomial *a, const tkmk_polynomial *b, tkmk_polynomial **out) { return ph_lincomb2(a, b, true, out); }
TK_API tkmk_error bls12_381_polynomial_multiply_by_scalar(const tkmk_polynomial *a, const tkmk_fr *s_host, tkmk_polynomial **out) {
    if (!a || !s_host || !out) return TKMK_ERR_INVALID_POINTER;
    if (a->n > 0xffffffffull) return TKMK_ERR_INVALID_ARGUMENT;
    tkmk_polynomial *r = nullptr;
    TK_TRY(ph_new(a->n, &r));
    const tkmk_fr *ptr = (const tkmk_fr *)a->c;
    uint32_t xs = (uint32_t)a->n, ys = 1;
    tkmk_error e = tkmk_poly_lincomb(1, s_host, &ptr, &xs, &ys, nullptr, nullptr, (tkmk_fr *)r->c, xs, 1, nullptr);
    if (e == TKMK_SUCCESS) e = tkmk_device_synchronize();
    if (e != TKMK_SUCCESS) {
        (void)bls12_381_polynomial_delete(r);
        return e;
    }
    *out = r;
    return TKMK_SUCCESS;
}
// product through the NTT on the next power of two past deg a + deg b (the domain must cover it: bls12_381_ntt_init_domain)
TK_API tkmk_error bls12_381_polynomial_multiply(const tkmk_polynomial *a, const tkmk_polynomial *b, tkmk_polynomial **out) {
    if (!a || !b || !out) return TKMK_ERR_INVALID_POINTER;
    int64_t da, db;
    TK_TRY(bls12_381_polynomial_degree(a, &da));
    TK_TRY(bls12_381_polynomial_degree(b, &db));
    if (da < 0 || db < 0) return ph_new(0, out);
    size_t N = 1;
    while (N < (size_t)(da + db + 1)) N <<= 1;
    tkmk_polynomial *ea = nullptr, *eb = nullptr;
    TK_TRY(ph_new(N, &ea));
    tkmk_error e = ph_new(N, &eb);
    if (e != TKMK_SUCCESS) {
        (void)bls12_381_polynomial_delete(ea);
        return e;
    }
    auto load = [&](tkmk_polynomial *dst, const tkmk_polynomial *src, int64_t deg) -> tkmk_error {
        TK_TRY(tkmk_memset(dst->c, 0, N * sizeof(fr_t)));
        TK_TRY(tkmk_memcpy_d2d(dst->c, src->c, (size_t)(deg + 1) * sizeof(fr_t)));
        return tkmk_bintt((const tkmk_fr *)dst->c, N, 1, TKMK_NTT_FORWARD, nullptr, nullptr, true, nullptr, (tkmk_fr *)dst->c);
    };
    e = load(ea, a, da);
    if (e == TKMK_SUCCESS) e = load(eb, b, db);
    if (e == TKMK_SUCCESS) {
        tkmk_vecops_config c = tkmk_vecops_default_config();
        c.is_a_on_device = c.is_b_on_device = c.is_result_on_device = true;
        e = bls12_381_vector_mul((const tkmk_fr *)ea->c, (const tkmk_fr *)eb->c, N, &c, (tkmk_fr *)ea->c);
    }
    if (e == TKMK_SUCCESS) e = tkmk_bintt((const tkmk_fr *)ea->c, N, 1, TKMK_NTT_INVERSE, nullptr, nullptr, true, nullptr, (tkmk_fr *)ea->c);
    (void)bls12_381_polynomial_delete(eb);
    if (e != TKMK_SUCCESS) {
        (void)bls12_381_polynomial_delete(ea);
        return e;
    }
    *out = ea;
    return TKMK_SUCCESS;
}
// slice(offset, stride, size): out[i] = p[offset + i * stride] (even / odd parts and the row / column extraction of
// get_univariate_polynomial_x / _y, mod.rs:1760-1782)
TK_API tkmk_error bls12_381_polynomial_slice(const tkmk_polynomial *p, size_t offset, size_t stride, size_t size, tkmk_polynomial **out) {
    if (!p || !out) return TKMK_ERR_INVALID_POINTER;
    if (stride == 0 || size == 0 || offset >= p->n || (size - 1) > (p->n - 1 - offset) / stride) return TKMK_ERR_INVALID_ARGUMENT;
    tkmk_polynomial *r = nullptr;
    TK_TRY(ph_new(size, &r));
    tkmk_error e = tkmk_memcpy_2d_d2d(r->c, sizeof(fr_t), p->c + offset, stride * sizeof(fr_t), sizeof(fr_t), size);
    if (e != TKMK_SUCCESS) {
        (void)bls12_381_polynomial_delete(r);
        return e;
    }
    *out = r;
    return TKMK_SUCCESS;
}

// ---- long division: num = quot * den + rem, deg rem < deg den.  One workgroup: the quotient coefficients are a serial chain
// (q_k depends on the remainder left by q_{k+1}), each step an axpy over the dd low coefficients of the denominator. ----
__global__ __launch_bounds__(1024) void k_poly_long_div(fr_t *__restrict__ r, const fr_t *__restrict__ den, fr_t inv_lead_mont, uint32_t dn,
                                                       uint32_t dd, fr_t *__restrict__ q) {
    __shared__ fr_t qk_mont;
    for (int64_t k = (int64_t)dn - dd; k >= 0; k--) {
        __syncthreads();
        if (threadIdx.x == 0) {
            fr_t top = Fr::canon(tk_load(r + k + dd));
            fr_t qk = Fr::mul(top, inv_lead_mont);          // plain * Montgomery -> plain
            tk_store(q + k, qk);
            qk_mont = Fr::to_mont(qk);
            tk_store(r + k + dd, Fr::zero());
        }
        __syncthreads();
        const fr_t m = qk_mont;
        for (uint32_t j = threadIdx.x; j < dd; j += blockDim.x) {
            fr_t t = Fr::mul(Fr::canon(tk_load(den + j)), m);   // plain
            tk_store(r + k + j, Fr::sub(Fr::canon(tk_load(r + k + j)), t));
        }
    }
}
TK_API tkmk_error bls12_381_polynomial_divide(const tkmk_polynomial *num, const tkmk_polynomial *den, tkmk_polynomial **quot, tkmk_polynomial **rem) {
    if (!num || !den || !quot || !rem) return TKMK_ERR_INVALID_POINTER;
    int64_t dn, dd;
    TK_TRY(bls12_381_polynomial_degree(num, &dn));
    TK_TRY(bls12_381_polynomial_degree(den, &dd));
    if (dd < 0) return TKMK_ERR_INVALID_ARGUMENT;   // division by the zero polynomial
    tkmk_polynomial *q = nullptr, *r = nullptr;
    if (dn < dd) {                                   // quotient 0, remainder = numerator
        TK_TRY(ph_new(0, &q));
        tkmk_error e = bls12_381_polynomial_clone(num, &r);
        if (e != TKMK_SUCCESS) {
            (void)bls12_381_polynomial_delete(q);
            return e;
        }
        *quot = q, *rem = r;
        return TKMK_SUCCESS;
    }
    TK_TRY(ph_new((size_t)(dn - dd + 1), &q));
    tkmk_error e = bls12_381_polynomial_create_from_coefficients((const tkmk_fr *)num->c, (size_t)dn + 1, true, &r);   // working copy of the numerator
    fr_t lead;
    if (e == TKMK_SUCCESS) e = tkmk_memcpy_d2h(&lead, den->c + dd, sizeof lead);
    if (e == TKMK_SUCCESS) {
        fr_t inv = Fr::inv(Fr::to_mont(Fr::canon(lead)));
        hipLaunchKernelGGL(k_poly_long_div, 1, 1024, 0, 0, r->c, (const fr_t *)den->c, inv, (uint32_t)dn, (uint32_t)dd, q->c);
        if (hipGetLastError() != hipSuccess) e = TKMK_ERR_UNKNOWN;
        if (e == TKMK_SUCCESS) e = tkmk_device_synchronize();
    }
    if (e != TKMK_SUCCESS) {
        (void)bls12_381_polynomial_delete(q);
        if (r) (void)bls12_381_polynomial_delete(r);
        return e;
    }
    r->n = dd > 0 ? (size_t)dd : 1;   // the remainder: the dd low coefficients of the working copy (dd = 0: the zero polynomial — r[0] was cleared)
    *quot = q, *rem = r;
    return TKMK_SUCCESS;
}
