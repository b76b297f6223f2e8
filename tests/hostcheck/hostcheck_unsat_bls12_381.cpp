#include "hostcheck_unsat.h"

extern "C" {
void hc_g1u_full(int mode, const uint8_t *pts, uint32_t k, uint8_t *out) { g1u_full<bls12_381_fq_params>(mode, pts, k, out); }
void hc_fqu_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *o, size_t n) { fqu_op<bls12_381_fq_params>(op, a, b, o, n); }
void hc_g1u_accumulate(const uint8_t *pts, const uint8_t *negate, size_t n, uint8_t *out) {
    g1u_accumulate<bls12_381_fq_params>(pts, negate, n, out);
}
}
