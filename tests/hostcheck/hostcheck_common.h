// hostcheck_common.h — TEST-ONLY helpers shared by the host-compiled checks of the product's field / curve templates
#pragma once
#include <string.h>

#include "ec.h"

template <class F>
static void load(typename F::E &e, const uint8_t *p) { memcpy(e.l, p, 4 * F::N); }
template <class F>
static void store(uint8_t *p, const typename F::E &e) { memcpy(p, e.l, 4 * F::N); }

template <class F>
static void binop(int op, const uint8_t *a, const uint8_t *b, uint8_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) {
        typename F::E x, y, z;
        load<F>(x, a + 4 * F::N * i);
        load<F>(y, b + 4 * F::N * i);
        switch (op) {
            case 0: z = F::add(x, y); break;
            case 1: z = F::sub(x, y); break;
            case 2: z = F::from_mont(F::mul(F::to_mont(x), F::to_mont(y))); break;
            case 3: z = F::mul(x, F::to_mont(y)); break;  // plain * mont -> plain (NTT butterfly form)
            case 4: z = F::from_mont(F::inv(F::to_mont(x))); break;
            case 5: z = F::neg(x); break;
            default: z = F::zero();
        }
        store<F>(o + 4 * F::N * i, z);
    }
}
