#include "hostcheck_unsat.h"

// BN254 instantiation (8 saturated / 10 x 28-bit unsaturated limbs)
extern "C" {
void hc_bn254_g1u_full(int mode, const uint8_t *pts, uint32_t k, uint8_t *out) { g1u_full<bn254_fq_params>(mode, pts, k, out); }
void hc_bn254_fr_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *o, size_t n) { binop<ff<bn254_fr_params>>(op, a, b, o, n); }
void hc_bn254_fq_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *o, size_t n) { binop<ff<bn254_fq_params>>(op, a, b, o, n); }
void hc_bn254_fqu_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *o, size_t n) { fqu_op<bn254_fq_params>(op, a, b, o, n); }
void hc_bn254_g1u_accumulate(const uint8_t *pts, const uint8_t *negate, size_t n, uint8_t *out) {
    g1u_accumulate<bn254_fq_params>(pts, negate, n, out);
}
}
