// tkmk_base.hpp — the few types and helpers every host header of this directory starts from: the error type (Rust panics of the reference
// become tkmk::Error), the ABI's scalar / point records under the reference's names, the host trace.  Split out of tkmk_host.hpp so that
// the single-element field arithmetic (tkmk_fr.hpp) can sit below the polynomial layer that uses it.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/tkmk.h"

namespace tkmk {

struct Error : std::runtime_error {
    tkmk_error code;
    Error(tkmk_error c, const std::string &where) : std::runtime_error(where + ": " + tkmk_error_string(c)), code(c) {}
    explicit Error(const std::string &msg) : std::runtime_error(msg), code(TKMK_ERR_INVALID_ARGUMENT) {}
};
inline void check(tkmk_error e, const char *where) {
    if (e != TKMK_SUCCESS) throw Error(e, where);
}
// TKMK_HOST_TRACE=1: one stderr line per transform / commit batch / division issued by the host side (sizes and a host clock),
// to read a proof's operation sequence next to a kernel trace
inline bool host_trace_on() {
    static const bool on = getenv("TKMK_HOST_TRACE") != nullptr;
    return on;
}
inline void host_trace(const char *fmt, ...) {
    if (!host_trace_on()) return;
    static const auto t0 = std::chrono::steady_clock::now();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "[tkmk host %9.3f ms] ", ms);
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}

using ScalarField = tkmk_fr;
using G1Affine = tkmk_g1_affine;

inline ScalarField fr_from_u32(uint32_t v) {
    ScalarField f{};
    f.limbs[0] = v;
    return f;
}
inline bool fr_is_zero(const ScalarField &a) {
    uint32_t x = 0;
    for (uint32_t l : a.limbs) x |= l;
    return x == 0;
}
inline bool fr_eq(const ScalarField &a, const ScalarField &b) { return std::memcmp(&a, &b, sizeof a) == 0; }

}  // namespace tkmk
