// tkmk_host.hpp — C++17 host-side mirror of the reference's Rust `libs` types over the C ABI (include/tkmk.h).
//
// The reference's host language (Rust) is not available in this build environment, so the interface a maintainer
// programs against is restated here in C++ with the reference's names, argument meaning and error behaviour
// (Rust panics become tkmk::Error exceptions):
//   DeviceVec<T>            icicle_runtime::memory::DeviceVec            (RAII device buffer)
//   init_ntt_domain_for_size  libs/src/bivariate_polynomial/mod.rs:33-55 (global, grow-only)
//   DensePolynomialExt      libs/src/bivariate_polynomial/mod.rs:112-118 + trait BivariatePolynomial :1283-1416
//   PolyExpr                libs/src/bivariate_polynomial/mod.rs:141-436  (fused evaluation-domain evaluator)
//   Sigma1::encode_poly     libs/src/iotools/mod.rs:2033-2113, libs/src/group_structures/mod.rs:59-119
// Every method is a few calls into libtkmk_hip.so; the coefficient matrix stays in HBM (the reference copies it to the
// host for find_degree / resize / mul_monomial / scaling / divisions).  Header-only; link with -ltkmk_hip.
#pragma once
#include "tkmk_base.hpp"
#include "tkmk_fr.hpp"

namespace tkmk {

template <class T>
class DeviceVec {
    T *p_ = nullptr;
    size_t n_ = 0;

  public:
    DeviceVec() = default;
    explicit DeviceVec(size_t n) : n_(n) { check(tkmk_malloc((void **)&p_, n * sizeof(T)), "DeviceVec::device_malloc"); }
    DeviceVec(const DeviceVec &) = delete;
    DeviceVec &operator=(const DeviceVec &) = delete;
    DeviceVec(DeviceVec &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr, o.n_ = 0; }
    DeviceVec &operator=(DeviceVec &&o) noexcept {
        if (this != &o) {
            if (p_) tkmk_free(p_);
            p_ = o.p_, n_ = o.n_, o.p_ = nullptr, o.n_ = 0;
        }
        return *this;
    }
    ~DeviceVec() {
        if (p_) tkmk_free(p_);
    }
    static DeviceVec from_host(const T *src, size_t n) {
        DeviceVec d(n);
        d.copy_from_host(src, n);
        return d;
    }
    static DeviceVec from_host(const std::vector<T> &v) { return from_host(v.data(), v.size()); }
    void copy_from_host(const T *src, size_t n) { check(tkmk_memcpy_h2d(p_, src, n * sizeof(T)), "DeviceVec::copy_from_host"); }
    void copy_to_host(T *dst, size_t n, size_t first = 0) const { check(tkmk_memcpy_d2h(dst, p_ + first, n * sizeof(T)), "DeviceVec::copy_to_host"); }
    std::vector<T> to_host() const {
        std::vector<T> v(n_);
        if (n_) copy_to_host(v.data(), n_);
        return v;
    }
    DeviceVec clone() const {
        DeviceVec d(n_);
        check(tkmk_memcpy_d2d(d.p_, p_, n_ * sizeof(T)), "DeviceVec::copy");
        return d;
    }
    T *ptr() { return p_; }
    const T *ptr() const { return p_; }
    size_t len() const { return n_; }
};

// ---- NTT domain: global and grow-only, like init_ntt_domain_for_size (mod.rs:33-55) ----
inline void init_ntt_domain_for_size(size_t size) {
    if (size == 0) throw Error("NTT domain size must be non-zero.");
    if (size & (size - 1)) throw Error("NTT domain size must be a power of two.");
    uint64_t cur = 0;   // the library's own record (the domain is process-global; another host side may have grown it)
    check(bls12_381_ntt_domain_size(&cur), "ntt::domain_size");
    if (cur >= size) return;
    if (cur) check(bls12_381_ntt_release_domain(), "ntt::release_domain");
    ScalarField root;
    check(bls12_381_get_root_of_unity(size, &root), "ntt::get_root_of_unity");
    tkmk_ntt_init_domain_config cfg{nullptr, false, nullptr};
    check(bls12_381_ntt_init_domain(&root, &cfg), "ntt::initialize_domain");
}

inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
inline size_t next_pow2(size_t n) {
    size_t p = 1;
    while (p < n) p <<= 1;
    return p;
}
// _find_size_as_twopower (mod.rs:72-86)
inline std::pair<size_t, size_t> find_size_as_twopower(size_t tx, size_t ty) {
    if (tx == 0 || ty == 0) throw Error("Invalid target sizes for resize");
    return {next_pow2(tx), next_pow2(ty)};
}

inline tkmk_vecops_config dev_cfg() {
    tkmk_vecops_config c = tkmk_vecops_default_config();
    c.is_a_on_device = c.is_b_on_device = c.is_result_on_device = true;
    return c;
}

// ---- One proof over G GPUs: the calling thread's place in a sharded prover (no reference counterpart) ----
// world == 1 (the default): everything below is the single-GPU code path, unchanged.  world > 1 (installed for the span of a call by
// the sharded ProverContext, host/tkmk_service.hpp ShardSpan): matrices are DISTRIBUTED in the two layouts of include/tkmk_dist.h —
//   COLS  coefficient matrices: rank r holds all rows of the columns iy = r mod G; local column k = global column r + G k
//   ROWS  evaluations on a domain: rank r holds the rows [r h, (r + 1) h), h = x_size / G
// and every method of DensePolynomialExt / PolyExpr / Sigma1 works on the local part, with the few collectives the operation needs
// (a transform: one all-to-all; find_degree, eval, div_by_ruffini's remainder row: an all-gather of a few values).  G is a power of two.
struct Shard {
    uint32_t world = 1, rank = 0;
    size_t cols_of(size_t total) const { return total > rank ? (total - rank + world - 1) / world : 0; }   // |{iy < total : iy = rank mod world}|
    size_t rows_of(size_t total) const { return cols_of(total); }                                         // the same count for an interleaved row set
};
struct DistCtx {
    void *comm = nullptr;
    Shard shard;
    tkmk_error (*all_gather_host)(void *, const void *, size_t, void *) = nullptr;
    tkmk_error (*fwd_cols_to_rows)(void *, const tkmk_fr *, size_t, size_t, size_t, size_t, int, tkmk_fr *) = nullptr;
    tkmk_error (*inv_rows_to_cols)(void *, tkmk_fr *, size_t, size_t, int, tkmk_fr *) = nullptr;
    tkmk_error (*rows_rotate)(void *, const tkmk_fr *, size_t, size_t, size_t, tkmk_fr *) = nullptr;
    tkmk_error (*ring_shift)(void *, const void *, size_t, int, void *) = nullptr;
    bool on() const { return shard.world > 1; }
    uint32_t G() const { return shard.world; }
    uint32_t r() const { return shard.rank; }
    // host values of every rank, rank order
    template <class T>
    std::vector<T> gather(const T &mine) const {
        std::vector<T> all(shard.world);
        check(all_gather_host(comm, &mine, sizeof(T), all.data()), "tkmk_comm_all_gather_host");
        return all;
    }
};
inline DistCtx &dist_ctx() {
    static thread_local DistCtx c;
    return c;
}

class DensePolynomialExt {
  public:
    // x_size * ly() coefficients, element (ix, local column k) at ix * ly() + k.  Single GPU: ly() = y_size, k = iy.
    // Sharded prover: a DISTRIBUTED matrix holds the columns iy = rank mod G (ly() = |those|); a REPLICATED one (rep: small polynomials
    // built from host values — vanishing polynomials, blinding terms, the Lagrange polynomials K, L, a_free) holds all columns on every rank.
    DeviceVec<ScalarField> poly;
    int64_t x_degree = -1, y_degree = -1;
    size_t x_size = 1, y_size = 1;
    bool rep = true;
    DeviceVec<ScalarField> commit_slice_;   // sharded prover: this rank's columns of a REPLICATED polynomial while its commit is in flight (Sigma1::job)

    DensePolynomialExt() = default;
    DensePolynomialExt(DeviceVec<ScalarField> &&c, size_t xs, size_t ys, int64_t xd, int64_t yd, bool replicated = !dist_ctx().on())
        : poly(std::move(c)), x_degree(xd), y_degree(yd), x_size(xs), y_size(ys), rep(replicated || !dist_ctx().on()) {}

    bool distributed() const { return !rep && dist_ctx().on(); }
    static size_t local_cols(size_t ys, bool replicated) { return (replicated || !dist_ctx().on()) ? ys : dist_ctx().shard.cols_of(ys); }
    size_t ly() const { return local_cols(y_size, rep); }

    static DensePolynomialExt zero() {
        ScalarField z{};
        return DensePolynomialExt(DeviceVec<ScalarField>::from_host(&z, 1), 1, 1, -1, -1, true);
    }
    // from_coeffs (mod.rs:1527-1551): takes ownership of a device vector (sharded prover: the local columns of a distributed matrix
    // unless `replicated`)
    static DensePolynomialExt from_coeffs(DeviceVec<ScalarField> &&coeffs, size_t x_size, size_t y_size, bool replicated = !dist_ctx().on()) {
        if (x_size * local_cols(y_size, replicated) != coeffs.len()) throw Error("Mismatch between the coefficient vector and the polynomial size");
        if (!is_pow2(x_size) || !is_pow2(y_size)) throw Error("The input sizes for from_coeffs must be powers of two.");
        return DensePolynomialExt(std::move(coeffs), x_size, y_size, (int64_t)x_size - 1, (int64_t)y_size - 1, replicated);
    }
    // host values: the same on every rank, hence replicated
    static DensePolynomialExt from_coeffs(const std::vector<ScalarField> &coeffs, size_t x_size, size_t y_size) {
        DensePolynomialExt p = from_coeffs(DeviceVec<ScalarField>::from_host(coeffs), x_size, y_size, true);
        int64_t xd = -1, yd = -1;   // the values are here: their exact degree costs nothing
        for (size_t i = 0; i < x_size; i++)
            for (size_t j = 0; j < y_size; j++)
                if (!fr_is_zero(coeffs[i * y_size + j])) xd = std::max<int64_t>(xd, (int64_t)i), yd = std::max<int64_t>(yd, (int64_t)j);
        p.x_degree = xd, p.y_degree = yd;
        return p;
    }
    // from_rou_evals (mod.rs:1615-1644): inverse _biNTT of device-resident evaluations.  Sharded prover: `evals` is this rank's ROWS slab
    // (h x y_size), the result a distributed COLS matrix (one all-to-all).
    static DensePolynomialExt from_rou_evals(const DeviceVec<ScalarField> &evals, size_t x_size, size_t y_size,
                                             const ScalarField *coset_x = nullptr, const ScalarField *coset_y = nullptr) {
        if (!is_pow2(x_size) || !is_pow2(y_size)) throw Error("The input sizes for from_rou_evals must be powers of two.");
        const DistCtx &dc = dist_ctx();
        if (dc.on()) {
            if (coset_x || coset_y) throw Error("from_rou_evals: cosets are not used by the sharded prover");
            if (x_size % dc.G() || y_size % dc.G()) throw Error("from_rou_evals: the domain is smaller than the number of ranks");
            if (evals.len() < x_size / dc.G() * y_size) throw Error("Insufficient buffer length for from_rou_evals");
            DeviceVec<ScalarField> slab = evals.clone();   // the exchange consumes its input
            return from_rou_evals_consume(std::move(slab), x_size, y_size);
        }
        if (evals.len() < x_size * y_size) throw Error("Insufficient buffer length for from_rou_evals");
        DeviceVec<ScalarField> coeffs(x_size * y_size);
        host_trace("from_rou_evals %zu x %zu%s", x_size, y_size, coset_x || coset_y ? " coset" : "");
        check(tkmk_bintt(evals.ptr(), x_size, y_size, TKMK_NTT_INVERSE, coset_x, coset_y, true, nullptr, coeffs.ptr()), "_biNTT");
        return from_coeffs(std::move(coeffs), x_size, y_size);
    }
    // the same for evaluations the caller no longer needs (sharded prover: saves the copy; single GPU: identical)
    static DensePolynomialExt from_rou_evals_consume(DeviceVec<ScalarField> &&evals, size_t x_size, size_t y_size) {
        const DistCtx &dc = dist_ctx();
        if (!dc.on()) return from_rou_evals(evals, x_size, y_size);
        DeviceVec<ScalarField> coeffs(x_size * (y_size / dc.G()));
        host_trace("from_rou_evals %zu x %zu (rank %u of %u)", x_size, y_size, dc.r(), dc.G());
        check(dc.inv_rows_to_cols(dc.comm, evals.ptr(), x_size, y_size, 0, coeffs.ptr()), "tkmk_dist_inv_rows_to_cols");
        return from_coeffs(std::move(coeffs), x_size, y_size, false);
    }
    // evaluations in the COLS layout — what the witness side produces (one column per placement: u, v, w, b, s0, s1) and prove1's walk
    // works on — to coefficients: a change of layout (one all-to-all) in front of the inverse transform.  Single GPU: from_rou_evals.
    static DensePolynomialExt from_rou_evals_cols(const DeviceVec<ScalarField> &evals_cols, size_t x_size, size_t y_size) {
        const DistCtx &dc = dist_ctx();
        if (!dc.on()) return from_rou_evals(evals_cols, x_size, y_size);
        if (x_size % dc.G() || y_size % dc.G()) throw Error("from_rou_evals: the domain is smaller than the number of ranks");
        if (evals_cols.len() < x_size * (y_size / dc.G())) throw Error("Insufficient buffer length for from_rou_evals");
        DeviceVec<ScalarField> slab(x_size / dc.G() * y_size);
        check(dc.fwd_cols_to_rows(dc.comm, evals_cols.ptr(), x_size, y_size, x_size, y_size, 3, slab.ptr()), "tkmk_dist_fwd_cols_to_rows");
        return from_rou_evals_consume(std::move(slab), x_size, y_size);
    }
    // small evaluation vectors made from host values (unit evaluations, the public inputs): the same on every rank -> replicated result
    static DensePolynomialExt from_rou_evals_rep(const DeviceVec<ScalarField> &evals, size_t x_size, size_t y_size) {
        if (!is_pow2(x_size) || !is_pow2(y_size)) throw Error("The input sizes for from_rou_evals must be powers of two.");
        if (evals.len() < x_size * y_size) throw Error("Insufficient buffer length for from_rou_evals");
        DeviceVec<ScalarField> coeffs(x_size * y_size);
        check(tkmk_bintt(evals.ptr(), x_size, y_size, TKMK_NTT_INVERSE, nullptr, nullptr, true, nullptr, coeffs.ptr()), "_biNTT");
        return from_coeffs(std::move(coeffs), x_size, y_size, true);
    }
    // to_rou_evals (mod.rs:1646-1674) without the reference's D->H->D round trip (single GPU)
    void to_rou_evals(const ScalarField *coset_x, const ScalarField *coset_y, DeviceVec<ScalarField> &evals) const {
        if (dist_ctx().on()) throw Error("to_rou_evals: the sharded prover evaluates through evals_on (ROWS layout)");
        if (evals.len() < x_size * y_size) throw Error("Insufficient buffer length for to_rou_evals");
        host_trace("to_rou_evals %zu x %zu%s", x_size, y_size, coset_x || coset_y ? " coset" : "");
        check(tkmk_bintt(poly.ptr(), x_size, y_size, TKMK_NTT_FORWARD, coset_x, coset_y, true, nullptr, evals.ptr()), "_biNTT");
    }
    // this polynomial as a distributed matrix (a replicated one gives up the other ranks' columns; y_size grows to G if smaller)
    DensePolynomialExt to_distributed() const {
        const DistCtx &dc = dist_ctx();
        if (!dc.on() || !rep) return clone();
        DensePolynomialExt wide = clone();
        if (wide.y_size < dc.G()) wide.resize(wide.x_size, dc.G());
        const size_t lc = wide.y_size / dc.G();
        DeviceVec<ScalarField> loc(wide.x_size * lc);
        // local element (i, k) = replicated element (i, r + G k) = flat index r + G (i lc + k): one strided copy
        check(tkmk_memcpy_2d_d2d(loc.ptr(), sizeof(ScalarField), wide.poly.ptr() + dc.r(), (size_t)dc.G() * sizeof(ScalarField), sizeof(ScalarField), wide.x_size * lc),
              "to_distributed");
        DensePolynomialExt out(std::move(loc), wide.x_size, wide.y_size, x_degree, y_degree, false);
        return out;
    }
    // evaluations of this polynomial on the xs x ys domain: `resize` + forward _biNTT of the reference (mod.rs:1646-1674, 1920-1960),
    // without materialising the zero padding when the matrix is smaller than the domain (tkmk_bintt_padded).
    // Sharded prover: -> this rank's ROWS slab (xs / G x ys); X pass on the local columns, one all-to-all, Y pass on the local rows.
    DeviceVec<ScalarField> evals_on(size_t xs, size_t ys) const {
        const DistCtx &dc = dist_ctx();
        if (dc.on()) {
            if (rep) return to_distributed().evals_on(xs, ys);
            if (xs % dc.G() || ys % dc.G()) throw Error("evals_on: the domain is smaller than the number of ranks");
            if (x_size > xs || y_size > ys) {
                DensePolynomialExt r = clone();
                r.resize(std::min(x_size, xs), std::min(y_size, ys));
                if (r.x_size > xs || r.y_size > ys) throw Error("evals_on: the polynomial does not fit the domain");
                return r.evals_on(xs, ys);
            }
            DeviceVec<ScalarField> out(xs / dc.G() * ys);
            host_trace("evals_on %zu x %zu -> %zu x %zu (rank %u of %u)", x_size, y_size, xs, ys, dc.r(), dc.G());
            check(dc.fwd_cols_to_rows(dc.comm, poly.ptr(), x_size, y_size, xs, ys, 0, out.ptr()), "tkmk_dist_fwd_cols_to_rows");
            return out;
        }
        if (x_size <= xs && y_size <= ys) {
            DeviceVec<ScalarField> out(xs * ys);
            host_trace("evals_on %zu x %zu -> %zu x %zu (padded)", x_size, y_size, xs, ys);
            check(tkmk_bintt_padded(poly.ptr(), x_size, y_size, xs, ys, nullptr, nullptr, nullptr, out.ptr()), "_biNTT");
            return out;
        }
        DensePolynomialExt r = clone();   // larger than the domain in one direction: resize drops the (zero) excess first
        r.resize(xs, ys);
        host_trace("evals_on %zu x %zu -> %zu x %zu (resize)", x_size, y_size, xs, ys);
        check(tkmk_bintt(r.poly.ptr(), xs, ys, TKMK_NTT_FORWARD, nullptr, nullptr, true, nullptr, r.poly.ptr()), "_biNTT");
        return std::move(r.poly);
    }
    // a replicated polynomial's evaluations on a domain, computed whole on every rank (the X-only / Y-only leaves of the fused evaluator)
    DeviceVec<ScalarField> evals_on_rep(size_t xs, size_t ys) const {
        if (!rep) throw Error("evals_on_rep: the polynomial is distributed");
        if (x_size > xs || y_size > ys) throw Error("evals_on_rep: the polynomial does not fit the domain");
        DeviceVec<ScalarField> out(xs * ys);
        check(tkmk_bintt_padded(poly.ptr(), x_size, y_size, xs, ys, nullptr, nullptr, nullptr, out.ptr()), "_biNTT");
        return out;
    }
    DensePolynomialExt clone() const { return DensePolynomialExt(poly.clone(), x_size, y_size, x_degree, y_degree, rep); }
    std::vector<ScalarField> copy_coeffs() const { return poly.to_host(); }   // the LOCAL coefficients
    ScalarField get_coeff(uint64_t ix, uint64_t iy) const {
        if (!(ix <= x_size && iy <= y_size)) throw Error("The index at which to get a coefficient exceeds the coefficient size.");
        ScalarField v{};
        if (!distributed()) {
            poly.copy_to_host(&v, 1, ix * y_size + iy);
            return v;
        }
        const DistCtx &dc = dist_ctx();   // the owner's value reaches every rank
        if (iy % dc.G() == dc.r()) poly.copy_to_host(&v, 1, ix * ly() + iy / dc.G());
        return dc.gather(v)[iy % dc.G()];
    }
    // coefficient (0, 0) += s (`&poly + &scalar`, mod.rs:1042-1116): column 0 belongs to rank 0
    void add_to_constant_term(const ScalarField &s) {
        if (distributed() && dist_ctx().r() != 0) return;
        if (poly.len() == 0) return;
        ScalarField c0;
        poly.copy_to_host(&c0, 1, 0);
        c0 = fr_add(c0, s);
        check(tkmk_memcpy_h2d(poly.ptr(), &c0, sizeof c0), "memcpy_h2d");
    }
    std::pair<int64_t, int64_t> degree() const { return {x_degree, y_degree}; }

    // ---- degrees: kept as UPPER BOUNDS through every operation instead of measured (r4) ----
    // The reference measures a matrix's degree (find_degree: a pass over the matrix and a host round trip, mod.rs:1480-1515) wherever it
    // needs its shape: before every commit, product, vanishing division and K0 product — 26 times per proof.  Here x_degree / y_degree
    // carry a bound that every operation maintains exactly as the algebra gives it (a sum's degree is at most the largest operand's plus
    // its shift, a product's the sum, a quotient's the difference; polynomials built from host values get their exact degree from those
    // values), and the shape decisions read the bound.  For generic inputs the bounds ARE the degrees (the 19 boxes of
    // tests/golden/encode_dims.json are reproduced: tests/test_gpu_fullsize.py); for a degenerate input (a leading coefficient that happens
    // to vanish) a matrix may be one power of two larger than the reference's and a commit run over a few zero coefficients more — the
    // polynomials, hence the proof bytes, are the same.  TKMK_PROVER_EXACT_DEGREES=1 restores the measuring form; the testing-mode builds
    // measure as well and refuse a bound that lies below a measured degree.
    static bool measure_degrees() {
        static const bool on = [] {
            const char *e = getenv("TKMK_PROVER_EXACT_DEGREES");
            return e && atoi(e) != 0;
        }();
        return on;
    }
    // the degree the shape decisions use
    std::pair<int64_t, int64_t> shape_degree() const {
        if (measure_degrees()) return find_degree();
#ifdef TKMK_TESTING_MODE
        auto [xd, yd] = find_degree();
        if (xd > x_degree || yd > y_degree)
            throw Error("degree bound violated: measured (" + std::to_string(xd) + ", " + std::to_string(yd) + ") above the bound (" + std::to_string(x_degree) + ", " +
                        std::to_string(y_degree) + ")");
#endif
        return {std::min<int64_t>(x_degree, (int64_t)x_size - 1), std::min<int64_t>(y_degree, (int64_t)y_size - 1)};
    }
    DensePolynomialExt &&with_degree(int64_t xd, int64_t yd) && {
        x_degree = std::min<int64_t>(xd, (int64_t)x_size - 1), y_degree = std::min<int64_t>(yd, (int64_t)y_size - 1);
        return std::move(*this);
    }

    // find_degree (mod.rs:1480-1515)
    std::pair<int64_t, int64_t> find_degree() const {
        int64_t xd = -1, yd = -1;
        const size_t lc = ly();
        if (lc) check(tkmk_poly_find_degree(poly.ptr(), (uint32_t)x_size, (uint32_t)lc, &xd, &yd, nullptr), "find_degree");
        if (!distributed()) return {xd, yd};
        const DistCtx &dc = dist_ctx();
        struct { int64_t x, y; } mine{xd, yd < 0 ? -1 : (int64_t)dc.r() + (int64_t)dc.G() * yd};   // local column k is global column r + G k
        int64_t gx = -1, gy = -1;
        for (auto &q : dc.gather(mine)) gx = std::max(gx, q.x), gy = std::max(gy, q.y);
        return {gx, gy};
    }
    // resize (mod.rs:1784-1806)
    void resize(size_t target_x_size, size_t target_y_size) {
        auto [nx, ny] = find_size_as_twopower(target_x_size, target_y_size);
        if (x_size == nx && y_size == ny) return;
        const size_t sl = ly(), dl = local_cols(ny, rep);
        DeviceVec<ScalarField> dst(nx * dl);
        if (dl) {
            if (sl) check(tkmk_poly_place(poly.ptr(), (uint32_t)x_size, (uint32_t)sl, dst.ptr(), (uint32_t)nx, (uint32_t)dl, 0, 0, nullptr), "resize");
            else check(tkmk_memset(dst.ptr(), 0, nx * dl * sizeof(ScalarField)), "resize");
        }
        poly = std::move(dst);
        x_size = nx, y_size = ny;
        x_degree = std::min<int64_t>(x_degree, (int64_t)nx - 1), y_degree = std::min<int64_t>(y_degree, (int64_t)ny - 1);
    }
    // optimize_size (mod.rs:1808-1818)
    void optimize_size() {
        auto [xd, yd] = shape_degree();
        x_degree = xd, y_degree = yd;
        if (xd + 1 == 0 || yd + 1 == 0) return;
        resize((size_t)(xd + 1), (size_t)(yd + 1));
    }
    // mul_monomial (mod.rs:1820-1844)
    DensePolynomialExt mul_monomial(size_t x_exponent, size_t y_exponent) const {
        if (x_exponent == 0 && y_exponent == 0) return clone();
        if (distributed() && y_exponent) return lincomb({Term(fr_from_u32(1), this, (uint32_t)x_exponent, (uint32_t)y_exponent)});
        auto [nx, ny] = find_size_as_twopower((size_t)(x_degree + 1) + x_exponent, (size_t)(y_degree + 1) + y_exponent);
        if (x_size + x_exponent > nx || y_size + y_exponent > ny) throw Error("mul_monomial: coefficient block does not fit the target");
        const size_t sl = ly(), dl = local_cols(ny, rep);
        DeviceVec<ScalarField> dst(nx * dl);
        if (dl)
            check(tkmk_poly_place(poly.ptr(), (uint32_t)x_size, (uint32_t)sl, dst.ptr(), (uint32_t)nx, (uint32_t)dl, (uint32_t)x_exponent,
                                  (uint32_t)y_exponent, nullptr),
                  "mul_monomial");
        return from_coeffs(std::move(dst), nx, ny, rep).with_degree(x_degree < 0 ? -1 : x_degree + (int64_t)x_exponent, y_degree < 0 ? -1 : y_degree + (int64_t)y_exponent);
    }
    // scale_coeffs_x / scale_coeffs_y (mod.rs:1553-1613).  Distributed: local column k is global column r + G k, so the Y factor of
    // element (i, k) is fy^r (fy^G)^k.
    DensePolynomialExt scale_coeffs(const ScalarField *fx, const ScalarField *fy) const {
        const size_t lc = ly();
        DeviceVec<ScalarField> dst(x_size * lc);
        if (!lc) return from_coeffs(std::move(dst), x_size, y_size, rep).with_degree(x_degree, y_degree);
        if (!distributed() || !fy) {
            check(tkmk_poly_scale_coeffs(poly.ptr(), (uint32_t)x_size, (uint32_t)lc, fx, fy, dst.ptr(), nullptr), "_scale_coeffs");
            return from_coeffs(std::move(dst), x_size, y_size, rep).with_degree(x_degree, y_degree);
        }
        const DistCtx &dc = dist_ctx();
        const ScalarField fy_g = fr_pow(*fy, dc.G()), fy_r = fr_pow(*fy, dc.r());
        check(tkmk_poly_scale_coeffs(poly.ptr(), (uint32_t)x_size, (uint32_t)lc, fx, &fy_g, dst.ptr(), nullptr), "_scale_coeffs");
        if (dc.r()) {
            tkmk_vecops_config c = dev_cfg();
            c.is_a_on_device = false;
            check(bls12_381_scalar_mul_vec(&fy_r, dst.ptr(), dst.len(), &c, dst.ptr()), "_scale_coeffs");
        }
        return from_coeffs(std::move(dst), x_size, y_size, false).with_degree(x_degree, y_degree);
    }
    DensePolynomialExt scale_coeffs_x(const ScalarField &f) const { return scale_coeffs(&f, nullptr); }
    DensePolynomialExt scale_coeffs_y(const ScalarField &f) const { return scale_coeffs(nullptr, &f); }
    // eval_x / eval_y / eval (mod.rs:1719-1750)
    DensePolynomialExt eval_x(const ScalarField &x) const {   // -> a Y-only polynomial with this one's distribution
        const size_t lc = ly();
        DeviceVec<ScalarField> out(lc);
        if (lc) check(tkmk_poly_eval_x(poly.ptr(), (uint32_t)x_size, (uint32_t)lc, &x, out.ptr(), nullptr), "eval_x");
        return from_coeffs(std::move(out), 1, y_size, rep).with_degree(x_degree < 0 ? -1 : 0, y_degree);
    }
    DensePolynomialExt eval_y(const ScalarField &y) const {
        if (distributed()) throw Error("eval_y: not used by the sharded prover");
        DeviceVec<ScalarField> out(x_size);
        check(tkmk_poly_eval_y(poly.ptr(), (uint32_t)x_size, (uint32_t)y_size, &y, out.ptr(), nullptr), "eval_y");
        return from_coeffs(std::move(out), x_size, 1, rep);
    }
    ScalarField eval(const ScalarField &x, const ScalarField &y) const {
        ScalarField r{};
        if (!distributed()) {
            check(tkmk_poly_eval(poly.ptr(), (uint32_t)x_size, (uint32_t)y_size, &x, &y, &r, nullptr), "eval");
            return r;
        }
        // sum_k sum_i c[i][k] x^i y^(r + G k) = y^r * (local matrix evaluated at (x, y^G)); the G partial values are added on the host
        const DistCtx &dc = dist_ctx();
        const size_t lc = ly();
        if (lc) {
            const ScalarField y_g = fr_pow(y, dc.G());
            check(tkmk_poly_eval(poly.ptr(), (uint32_t)x_size, (uint32_t)lc, &x, &y_g, &r, nullptr), "eval");
            r = fr_mul(r, fr_pow(y, dc.r()));
        }
        ScalarField sum{};
        for (const ScalarField &q : dc.gather(r)) sum = fr_add(sum, q);
        return sum;
    }

    // ---- arithmetic (mod.rs:532-1281): operands are brought to the common (max) shape first ----
    static std::tuple<DensePolynomialExt, DensePolynomialExt> same_shape(const DensePolynomialExt &a, const DensePolynomialExt &b) {
        size_t nx = std::max(a.x_size, b.x_size), ny = std::max(a.y_size, b.y_size);
        DensePolynomialExt l = a.clone(), r = b.clone();
        l.resize(nx, ny);
        r.resize(nx, ny);
        return {std::move(l), std::move(r)};
    }
    friend DensePolynomialExt operator+(const DensePolynomialExt &a, const DensePolynomialExt &b) {
        if (dist_ctx().on()) return lincomb({Term(fr_from_u32(1), &a), Term(fr_from_u32(1), &b)});
        auto [l, r] = same_shape(a, b);
        tkmk_vecops_config c = dev_cfg();
        check(bls12_381_vector_add(l.poly.ptr(), r.poly.ptr(), l.poly.len(), &c, l.poly.ptr()), "add");
        return from_coeffs(std::move(l.poly), l.x_size, l.y_size).with_degree(std::max(a.x_degree, b.x_degree), std::max(a.y_degree, b.y_degree));
    }
    friend DensePolynomialExt operator-(const DensePolynomialExt &a, const DensePolynomialExt &b) {
        if (dist_ctx().on()) {
            ScalarField zero{}, one = fr_from_u32(1), minus_one;
            tkmk_vecops_config hc = tkmk_vecops_default_config();   // host in, host out
            check(bls12_381_vector_sub(&zero, &one, 1, &hc, &minus_one), "sub");
            return lincomb({Term(one, &a), Term(minus_one, &b)});
        }
        auto [l, r] = same_shape(a, b);
        tkmk_vecops_config c = dev_cfg();
        check(bls12_381_vector_sub(l.poly.ptr(), r.poly.ptr(), l.poly.len(), &c, l.poly.ptr()), "sub");
        return from_coeffs(std::move(l.poly), l.x_size, l.y_size).with_degree(std::max(a.x_degree, b.x_degree), std::max(a.y_degree, b.y_degree));
    }
    friend DensePolynomialExt operator*(const DensePolynomialExt &a, const ScalarField &s) {
        DensePolynomialExt out = a.clone();
        tkmk_vecops_config c = dev_cfg();
        c.is_a_on_device = false;
        if (out.poly.len()) check(bls12_381_scalar_mul_vec(&s, out.poly.ptr(), out.poly.len(), &c, out.poly.ptr()), "scalar_mul");
        return from_coeffs(std::move(out.poly), out.x_size, out.y_size, out.rep).with_degree(a.x_degree, a.y_degree);
    }
    // sum_t c_t * X^ox_t Y^oy_t * p_t in ONE pass (tkmk_poly_lincomb): poly_comb! and the `&a * &s`, `&a + &b`, mul_monomial chains
    // around it (prove/src/lib.rs:30-124), which cost one pass and one temporary per operator when evaluated step by step
    struct Term {
        ScalarField c;
        const DensePolynomialExt *p;
        uint32_t ox = 0, oy = 0;
        Term(const ScalarField &c_, const DensePolynomialExt *p_, uint32_t ox_ = 0, uint32_t oy_ = 0) : c(c_), p(p_), ox(ox_), oy(oy_) {}
    };
    static DensePolynomialExt lincomb(const std::vector<Term> &terms) {
        if (terms.empty()) return zero();
        size_t nx = 1, ny = 1;
        bool all_rep = true;
        int64_t bx = -1, by = -1;   // the sum's degree is at most the largest (operand degree + shift)
        for (const Term &t : terms) {
            nx = std::max(nx, next_pow2(t.p->x_size + t.ox)), ny = std::max(ny, next_pow2(t.p->y_size + t.oy));
            all_rep = all_rep && !t.p->distributed();
            if (t.p->x_degree >= 0 && t.p->y_degree >= 0 && !fr_is_zero(t.c)) bx = std::max<int64_t>(bx, t.p->x_degree + t.ox), by = std::max<int64_t>(by, t.p->y_degree + t.oy);
#ifdef TKMK_TESTING_MODE
            (void)t.p->shape_degree();   // the row truncation below trusts the operands' bounds: the testing builds measure them
#endif
        }
        // The sum has no more rows than its degree bound says, whatever the operands' matrices are allocated to (a vanishing quotient sits
        // in its numerator's 4 m_I rows although it has fewer than 2 m_I): the output gets next_pow2(bound + 1) rows and every operand is
        // read up to there only — the rows beyond hold zeros by the same bounds.  (Columns keep their allocated width: the kernel takes one
        // number for an operand's width and stride.)
        if (!measure_degrees()) nx = std::min(nx, next_pow2((size_t)std::max<int64_t>(bx, 0) + 1));
        const DistCtx &dc = dist_ctx();
        // sharded prover: operands that are all replicated and small give a replicated result (every rank computes the same few values);
        // anything else is a distributed matrix, each rank combining its columns of every operand
        const bool out_rep = !dc.on() || (all_rep && nx * ny <= (1u << 16));
        if (!out_rep) ny = std::max<size_t>(ny, dc.G()), nx = std::max<size_t>(nx, 1);   // at least one column per rank
        std::vector<ScalarField> c;
        std::vector<const tkmk_fr *> ptr;
        std::vector<uint32_t> xs, ys, ox, oy;
        std::vector<DeviceVec<ScalarField>> temps;   // local column sets of replicated operands, columns received from the neighbour for Y shifts
        std::vector<std::tuple<const DensePolynomialExt *, uint32_t, const tkmk_fr *>> fetched;
        for (const Term &t : terms) {
            const DensePolynomialExt &p = *t.p;
            if (t.ox >= nx) continue;                                        // wholly beyond the bound: a zero polynomial
            const uint32_t rows_in = (uint32_t)std::min<size_t>(p.x_size, nx - t.ox);   // rows of this operand that land inside the output
            if (out_rep) {
                c.push_back(t.c), ptr.push_back(p.poly.ptr()), xs.push_back(rows_in), ys.push_back((uint32_t)p.y_size), ox.push_back(t.ox), oy.push_back(t.oy);
                continue;
            }
            const uint32_t G = dc.G(), r = dc.r();
            if (p.rep) {
                // a replicated operand contributes its columns r, r + G, ...: a strided copy of that set (a Y shift that is not a multiple
                // of G is applied to the whole small matrix first)
                DensePolynomialExt d = (t.oy % G) ? p.mul_monomial(0, t.oy).to_distributed() : p.to_distributed();
                const size_t lc = d.ly(), lx = std::min<size_t>(d.x_size, nx - t.ox);
                temps.push_back(std::move(d.poly));
                if (!lc) continue;
                c.push_back(t.c), ptr.push_back(temps.back().ptr()), xs.push_back((uint32_t)lx), ys.push_back((uint32_t)lc), ox.push_back(t.ox);
                oy.push_back((t.oy % G) ? 0u : t.oy / G);
                continue;
            }
            const size_t lc = p.ly();
            if (t.oy % G == 0) {
                if (!lc) continue;
                c.push_back(t.c), ptr.push_back(p.poly.ptr()), xs.push_back(rows_in), ys.push_back((uint32_t)lc), ox.push_back(t.ox), oy.push_back(t.oy / G);
                continue;
            }
            // Y^oy p: global column j of p becomes column j + oy.  With s = oy mod G, my local column k (global r + G k) receives the
            // column r + G k - oy of p = local column k - q' of rank (r - s) mod G, q' = oy / G + (r < s): the whole local matrix of that
            // rank, fetched once (ring shift), then an ordinary local column offset
            if (p.y_size % G) throw Error("lincomb: a Y shift needs at least one column per rank");
            const uint32_t sft = t.oy % G;
            const tkmk_fr *shifted = nullptr;   // the same operand under the same rotation again (X^a Y p and X^b Y p): one exchange serves both
            for (const auto &f : fetched)
                if (std::get<0>(f) == &p && std::get<1>(f) == sft) shifted = std::get<2>(f);
            if (!shifted) {
                temps.emplace_back(p.x_size * lc);
                check(dc.ring_shift(dc.comm, p.poly.ptr(), p.x_size * lc * sizeof(ScalarField), (int)sft, temps.back().ptr()), "tkmk_comm_ring_shift");
                shifted = temps.back().ptr();
                fetched.emplace_back(&p, sft, shifted);
            }
            c.push_back(t.c), ptr.push_back(shifted), xs.push_back(rows_in), ys.push_back((uint32_t)lc), ox.push_back(t.ox);
            oy.push_back(t.oy / G + (r < sft ? 1u : 0u));
        }
        const size_t out_l = local_cols(ny, out_rep);
        DeviceVec<ScalarField> out(nx * out_l);
        if (out_l) {
            if (host_trace_on()) {
                size_t in_elems = 0;
                for (size_t t = 0; t < ptr.size(); t++) in_elems += (size_t)xs[t] * ys[t];
                host_trace("lincomb %zu terms -> %zu x %zu (operands %.1f MB, output %.1f MB)", ptr.size(), nx, out_l, in_elems * 32e-6, nx * out_l * 32e-6);
            }
            if (ptr.empty()) check(tkmk_memset(out.ptr(), 0, nx * out_l * sizeof(ScalarField)), "lincomb");
            else
                check(tkmk_poly_lincomb((uint32_t)ptr.size(), c.data(), ptr.data(), xs.data(), ys.data(), ox.data(), oy.data(), out.ptr(), (uint32_t)nx, (uint32_t)out_l,
                                        nullptr),
                      "tkmk_poly_lincomb");
            if (!temps.empty()) check(tkmk_device_synchronize(), "synchronize");   // the temporaries go out of scope here
        }
        return from_coeffs(std::move(out), nx, ny, out_rep).with_degree(bx, by);
    }
    // _mul (mod.rs:1846-1996)
    friend DensePolynomialExt operator*(const DensePolynomialExt &a, const DensePolynomialExt &b) {
        auto [lx, ly] = a.shape_degree();
        auto [rx, ry] = b.shape_degree();
        if (lx + ly == 0 && rx + ry > 0) return b * a.get_coeff(0, 0);
        if (rx + ry == 0 && lx + ly > 0) return a * b.get_coeff(0, 0);
        if (rx + ry == 0 && lx + ly == 0) {
            DensePolynomialExt one = a.clone();
            one.resize(1, 1);
            return one * b.get_coeff(0, 0);
        }
        size_t tx = (size_t)(lx + rx + 1), ty = (size_t)(ly + ry + 1);
        auto [xs, ys] = find_size_as_twopower(tx, ty);
        const DistCtx &dc = dist_ctx();
        if (dc.on()) xs = std::max<size_t>(xs, dc.G()), ys = std::max<size_t>(ys, dc.G());   // at least one row and one column per rank
        DeviceVec<ScalarField> le = a.evals_on(xs, ys), re = b.evals_on(xs, ys);
        tkmk_vecops_config c = dev_cfg();
        check(bls12_381_vector_mul(le.ptr(), re.ptr(), le.len(), &c, le.ptr()), "mul");
        return from_rou_evals_consume(std::move(le), xs, ys).with_degree(lx + rx, ly + ry);
    }

    // this * scale * (1 + X + ... + X^(m-1)); with scale = 1/m the factor is K_0 = unit evaluations at index 0 of the m-th roots
    // (the reference multiplies by it with `&K0 * &poly`, three NTTs: lib.rs:2238-2246, 3012-3040)
    DensePolynomialExt mul_ones_x(size_t m, const ScalarField &scale) const {
        auto [xd, yd] = shape_degree();
        if (xd < 0) return zero();
        size_t ox = next_pow2((size_t)xd + m);
        const size_t lc = ly();
        DeviceVec<ScalarField> out(ox * lc);
        host_trace("mul_ones_x %zu x %zu by m = %zu", x_size, y_size, m);
        if (lc) check(tkmk_poly_mul_ones_x(poly.ptr(), (uint32_t)x_size, (uint32_t)lc, (uint32_t)m, &scale, (uint32_t)ox, out.ptr(), nullptr), "mul_ones_x");
        return from_coeffs(std::move(out), ox, y_size, rep).with_degree(xd + (int64_t)m - 1, yd);
    }
    // div_by_vanishing_opt (mod.rs:2284-2410).  Distributed: every row is local, and the Y recurrence's stride d is a multiple of G, so a
    // column's predecessor j - d lives on the same rank d / G local columns earlier: both passes are local.
    std::pair<DensePolynomialExt, DensePolynomialExt> div_by_vanishing_opt(int64_t denom_x_degree, int64_t denom_y_degree) {
        if (denom_x_degree <= 0 || denom_y_degree <= 0 || !is_pow2((size_t)denom_x_degree) || !is_pow2((size_t)denom_y_degree))
            throw Error("The denominators must have degress as powers of two.");
        optimize_size();
        if (x_degree < denom_x_degree || y_degree < denom_y_degree) throw Error("The numerator must have grater degrees than denominators.");
        size_t c = (size_t)denom_x_degree, d = (size_t)denom_y_degree;
        size_t xs = (x_size / c) * c, ys = (y_size / d) * d;
        size_t ysl = ys, dl = d;
        if (distributed()) {
            const size_t G = dist_ctx().G();
            if (d % G || ys % G) throw Error("div_by_vanishing_opt: the Y divisor must be a multiple of the number of ranks");
            ysl = ys / G, dl = d / G;
        }
        DeviceVec<ScalarField> qx(xs * ysl), qy(c * ysl);
        host_trace("div_by_vanishing_opt %zu x %zu by (%zu, %zu)", xs, ys, c, d);
        check(tkmk_poly_div_by_vanishing_opt(poly.ptr(), (uint32_t)xs, (uint32_t)ysl, (uint32_t)c, (uint32_t)dl, qx.ptr(), qy.ptr(), nullptr),
              "div_by_vanishing_opt");
        DensePolynomialExt quo_x = from_coeffs(std::move(qx), xs, ys, rep), quo_y = from_coeffs(std::move(qy), c, ys, rep);
        // degree bounds of the quotients of an EXACT division P = Q_X (X^c - 1) + Q_Y (Y^d - 1) with deg_x Q_Y < c: deg_x Q_X <= deg_x P - c
        // (rows above that are zero: the top c rows of the recurrence vanish by exactness and the zero propagates down through the rows
        // where P has none), deg_y Q_X <= deg_y P, deg_y Q_Y <= deg_y P - d.  (The reference writes x_size - c - 1 etc. into the fields —
        // the matrix shape — and measures again before it commits.)
        if (xs > c) quo_x.x_degree = std::min<int64_t>(x_degree - (int64_t)c, (int64_t)(xs - c) - 1), quo_x.y_degree = std::min<int64_t>(y_degree, (int64_t)ys - 1);
        else quo_x.x_degree = quo_x.y_degree = -1;
        if (ys > d) quo_y.x_degree = (int64_t)c - 1, quo_y.y_degree = std::min<int64_t>(y_degree - (int64_t)d, (int64_t)(ys - d) - 1);
        else quo_y.x_degree = quo_y.y_degree = -1;
        return {std::move(quo_x), std::move(quo_y)};
    }
    // div_by_ruffini (mod.rs:2412-2458).  Distributed: Q_X is a Horner scan along X, local in every column; the remainder row P(x, Y) —
    // one value per column — is gathered (y_size values), and its division by (Y - y) is done whole on every rank (Q_Y: replicated).
    std::tuple<DensePolynomialExt, DensePolynomialExt, ScalarField> div_by_ruffini(const ScalarField &x, const ScalarField &y) const {
        ScalarField r;
        if (!distributed()) {
            DeviceVec<ScalarField> qx(x_size * y_size), qy(y_size);
            check(tkmk_poly_div_by_ruffini(poly.ptr(), (uint32_t)x_size, (uint32_t)y_size, &x, &y, qx.ptr(), qy.ptr(), &r, nullptr), "div_by_ruffini");
            return {from_coeffs(std::move(qx), x_size, y_size, rep).with_degree(std::max<int64_t>(x_degree - 1, 0), y_degree),
                    from_coeffs(std::move(qy), 1, y_size, rep).with_degree(0, std::max<int64_t>(y_degree - 1, 0)), r};
        }
        const DistCtx &dc = dist_ctx();
        const size_t lc = ly(), G = dc.G();
        if (y_size % G) throw Error("div_by_ruffini: needs at least one column per rank");
        DeviceVec<ScalarField> qx(x_size * lc), scratch_qy(lc), rem(lc);
        ScalarField ignored;
        check(tkmk_poly_div_by_ruffini(poly.ptr(), (uint32_t)x_size, (uint32_t)lc, &x, &y, qx.ptr(), scratch_qy.ptr(), &ignored, nullptr), "div_by_ruffini");
        check(tkmk_poly_eval_x(poly.ptr(), (uint32_t)x_size, (uint32_t)lc, &x, rem.ptr(), nullptr), "div_by_ruffini: remainder row");
        std::vector<ScalarField> mine = rem.to_host(), all(y_size), whole(y_size);
        check(dc.all_gather_host(dc.comm, mine.data(), lc * sizeof(ScalarField), all.data()), "tkmk_comm_all_gather_host");
        for (size_t q = 0; q < G; q++)
            for (size_t k = 0; k < lc; k++) whole[q + G * k] = all[q * lc + k];
        DeviceVec<ScalarField> row = DeviceVec<ScalarField>::from_host(whole), qrow(y_size), qy(y_size);
        // P(x, Y) as a 1 x y_size matrix: its "X division" is trivial (one row), its Y division is the one wanted
        check(tkmk_poly_div_by_ruffini(row.ptr(), 1, (uint32_t)y_size, &x, &y, qrow.ptr(), qy.ptr(), &r, nullptr), "div_by_ruffini");
        return {from_coeffs(std::move(qx), x_size, y_size, false).with_degree(std::max<int64_t>(x_degree - 1, 0), y_degree),
                from_coeffs(std::move(qy), 1, y_size, true).with_degree(0, std::max<int64_t>(y_degree - 1, 0)), r};
    }
};

// ---- PolyExpr (mod.rs:141-436) ----
class PolyExpr {
  public:
    enum Kind { Poly, Scalar, Add, Sub, Mul, Scale, MulXMinusOne, Sum };
    Kind kind;
    const DensePolynomialExt *leaf = nullptr;
    size_t shift_ord_x = 0, shift_ord_y = 0;   // a Poly node may stand for leaf(w_ordx^-1 X, w_ordy^-1 Y); 0 = that axis unshifted
    ScalarField scalar_{};
    std::vector<PolyExpr> kids;

    static PolyExpr poly(const DensePolynomialExt &p) {
        PolyExpr e{Poly};
        e.leaf = &p;
        return e;
    }
    // p(w_{ord_x}^-1 X, w_{ord_y}^-1 Y) — what the reference builds with scale_coeffs_x / _y before evaluating (prove2's r_omegaX,
    // r_omegaX_omegaY, lib.rs:1969-1975).  On an evaluation domain whose sizes are multiples of the orders these are p's
    // evaluations rotated by size/order, so the node shares p's transform.
    static PolyExpr poly_root_shifted(const DensePolynomialExt &p, size_t ord_x, size_t ord_y) {
        if ((ord_x && !is_pow2(ord_x)) || (ord_y && !is_pow2(ord_y))) throw Error("root shifts must have power-of-two orders");
        PolyExpr e{Poly};
        e.leaf = &p;
        e.shift_ord_x = ord_x > 1 ? ord_x : 0, e.shift_ord_y = ord_y > 1 ? ord_y : 0;
        return e;
    }
    static PolyExpr scalar(const ScalarField &s) {
        PolyExpr e{Scalar};
        e.scalar_ = s;
        return e;
    }
    static PolyExpr add(PolyExpr l, PolyExpr r) { return binary(Add, std::move(l), std::move(r)); }
    static PolyExpr sub(PolyExpr l, PolyExpr r) { return binary(Sub, std::move(l), std::move(r)); }
    static PolyExpr mul(PolyExpr l, PolyExpr r) { return binary(Mul, std::move(l), std::move(r)); }
    static PolyExpr scale(const ScalarField &s, PolyExpr e) {
        PolyExpr out{Scale};
        out.scalar_ = s;
        out.kids.push_back(std::move(e));
        return out;
    }
    static PolyExpr mul_x_minus_one(PolyExpr e) {
        PolyExpr out{MulXMinusOne};
        out.kids.push_back(std::move(e));
        return out;
    }
    static PolyExpr weighted_sum(std::vector<std::pair<ScalarField, PolyExpr>> terms) {
        PolyExpr out{Sum};
        for (auto &t : terms) out.kids.push_back(scale(t.first, std::move(t.second)));
        return out;
    }

    using DegreeMemo = std::map<const DensePolynomialExt *, std::pair<int64_t, int64_t>>;
    std::pair<int64_t, int64_t> degree_bound() const {
        DegreeMemo memo;   // a leaf that occurs several times is measured once (find_degree is a device round trip)
        return degree_bound(memo);
    }
    std::pair<int64_t, int64_t> degree_bound(DegreeMemo &memo) const {
        switch (kind) {
            case Poly: {
                auto it = memo.find(leaf);
                if (it == memo.end()) it = memo.emplace(leaf, leaf->shape_degree()).first;
                return it->second;
            }
            case Scalar: return fr_is_zero(scalar_) ? std::make_pair<int64_t, int64_t>(-1, -1) : std::make_pair<int64_t, int64_t>(0, 0);
            case Add:
            case Sub: {
                auto l = kids[0].degree_bound(memo), r = kids[1].degree_bound(memo);
                return {std::max(l.first, r.first), std::max(l.second, r.second)};
            }
            case Mul: {
                auto l = kids[0].degree_bound(memo), r = kids[1].degree_bound(memo);
                if (l.first < 0 || l.second < 0 || r.first < 0 || r.second < 0) return {-1, -1};
                return {l.first + r.first, l.second + r.second};
            }
            case Scale: return fr_is_zero(scalar_) ? std::make_pair<int64_t, int64_t>(-1, -1) : kids[0].degree_bound(memo);
            case MulXMinusOne: {
                auto d = kids[0].degree_bound(memo);
                if (d.first < 0 || d.second < 0) return {-1, -1};
                return {d.first + 1, d.second};
            }
            default: {
                std::pair<int64_t, int64_t> acc{-1, -1};
                for (auto &k : kids) {
                    auto d = k.degree_bound(memo);
                    acc = {std::max(acc.first, d.first), std::max(acc.second, d.second)};
                }
                return acc;
            }
        }
    }
    static size_t domain_size_for_degree(int64_t degree) { return degree < 0 ? 1 : next_pow2((size_t)degree + 1); }

    DensePolynomialExt evaluate_fused() const {
        auto d = degree_bound();
        return evaluate_fused_with_domain(domain_size_for_degree(d.first), domain_size_for_degree(d.second));
    }
    // leaf evaluations by (polynomial, domain): pass the same cache to several evaluations on one domain and every leaf is transformed once
    using LeafCache = std::map<std::tuple<const DensePolynomialExt *, size_t, size_t>, std::shared_ptr<DeviceVec<ScalarField>>>;
    DensePolynomialExt evaluate_fused_with_domain(size_t target_x_size, size_t target_y_size, LeafCache *shared_cache = nullptr) const {
        if (!is_pow2(target_x_size) || !is_pow2(target_y_size)) throw Error("Fused polynomial expression domains must be powers of two.");
        auto d = degree_bound();
        if (domain_size_for_degree(d.first) > target_x_size || domain_size_for_degree(d.second) > target_y_size)
            throw Error("Fused polynomial expression domain is too small for the expression degree.");
        LeafCache own_cache;
        LeafCache &cache = shared_cache ? *shared_cache : own_cache;
        // one kernel pass over the leaf evaluations when the tree fits the device evaluator (tkmk_poly_expr_eval);
        // otherwise node by node
        Program pg;
        int need = compile(pg, 0);
        const DistCtx &dc = dist_ctx();
        const bool fits = need > 0 && need <= 6 && !pg.leaves.empty() && pg.leaves.size() <= 16 && pg.consts.size() <= 16 && pg.code.size() <= 100;
        if (dc.on()) {
            // sharded prover: every rank evaluates its ROWS slab [r h, (r + 1) h) of the domain.  Distributed leaves are transformed with one
            // all-to-all each (evals_on); replicated X-only / Y-only leaves are transformed whole (1-D, small) and read through their
            // slab piece / broadcast; a root shift along X is a rotation that crosses slabs and is made beforehand (rows_rotate).
            if (!fits) throw Error("Fused polynomial expression: too large for the one-pass evaluator (sharded prover)");
            const size_t G = dc.G(), h = target_x_size / G;
            if (target_x_size % G || target_y_size % G) throw Error("Fused polynomial expression: the domain is smaller than the number of ranks");
            std::vector<tkmk_expr_leaf> views;
            std::vector<std::shared_ptr<DeviceVec<ScalarField>>> rotated;
            for (const Program::Leaf &l : pg.leaves) {
                if ((l.ord_x && target_x_size % l.ord_x) || (l.ord_y && target_y_size % l.ord_y))
                    throw Error("Fused polynomial expression: a root shift's order does not divide the domain.");
                tkmk_expr_leaf v{};
                const bool x_only = l.p->rep && l.p->y_size == 1, y_only = l.p->rep && l.p->x_size == 1;
                if (x_only || y_only) {
                    const size_t lx = y_only ? 1 : target_x_size, ly = x_only ? 1 : target_y_size;
                    auto key = std::make_tuple(l.p, lx, ly);
                    auto it = cache.find(key);
                    if (it == cache.end()) it = cache.emplace(key, std::make_shared<DeviceVec<ScalarField>>(l.p->evals_on_rep(lx, ly))).first;
                    if (x_only && l.ord_x) throw Error("Fused polynomial expression: root shift of a replicated X-only leaf (sharded prover)");
                    v.data = it->second->ptr() + (x_only && lx > 1 ? dc.r() * h : 0);
                    v.x_len = (uint32_t)(lx > 1 ? h : 1), v.y_len = (uint32_t)ly;
                    v.rot_y = (l.ord_y && ly > 1) ? (uint32_t)(target_y_size / l.ord_y) : 0;
                    views.push_back(v);
                    continue;
                }
                std::shared_ptr<DeviceVec<ScalarField>> slab = leaf_evals(l.p, target_x_size, target_y_size, cache);
                const size_t rot_x = l.ord_x ? target_x_size / l.ord_x : 0;
                if (rot_x % target_x_size) {
                    if (rot_x > h) throw Error("Fused polynomial expression: the root shift spans more than one rank's rows");
                    auto rt = std::make_shared<DeviceVec<ScalarField>>(h * target_y_size);
                    check(dc.rows_rotate(dc.comm, slab->ptr(), h, target_y_size, rot_x, rt->ptr()), "tkmk_dist_rows_rotate");
                    rotated.push_back(rt);
                    slab = rt;
                }
                v.data = slab->ptr();
                v.x_len = (uint32_t)h, v.y_len = (uint32_t)target_y_size;
                v.rot_y = l.ord_y ? (uint32_t)((target_y_size / l.ord_y) % target_y_size) : 0;
                views.push_back(v);
            }
            DeviceVec<ScalarField> out(h * target_y_size);
            check(tkmk_poly_expr_eval_views_slab(pg.code.data(), (uint32_t)pg.code.size(), views.data(), (uint32_t)views.size(), pg.consts.data(),
                                                 (uint32_t)pg.consts.size(), (uint32_t)target_x_size, (uint32_t)(dc.r() * h), (uint32_t)h, (uint32_t)target_y_size,
                                                 out.ptr(), nullptr),
                  "tkmk_poly_expr_eval_views_slab");
            check(tkmk_device_synchronize(), "synchronize");   // the rotated copies go out of scope
            return DensePolynomialExt::from_rou_evals_consume(std::move(out), target_x_size, target_y_size).with_degree(d.first, d.second);
        }
        if (fits) {
            // leaves as VIEWS: an X-only (Y-only) polynomial is transformed in one dimension and broadcast; a root-shifted leaf
            // reads the unshifted polynomial's evaluations rotated by size / order
            std::vector<tkmk_expr_leaf> views;
            for (const Program::Leaf &l : pg.leaves) {
                const bool col = l.p->y_size == 1 && target_y_size > 1, row = l.p->x_size == 1 && target_x_size > 1;
                const size_t lx = row ? 1 : target_x_size, ly = col ? 1 : target_y_size;
                if ((l.ord_x && target_x_size % l.ord_x) || (l.ord_y && target_y_size % l.ord_y))
                    throw Error("Fused polynomial expression: a root shift's order does not divide the domain.");
                tkmk_expr_leaf v{};
                v.data = leaf_evals(l.p, lx, ly, cache)->ptr();
                v.x_len = (uint32_t)lx, v.y_len = (uint32_t)ly;
                v.rot_x = (l.ord_x && lx > 1) ? (uint32_t)(target_x_size / l.ord_x) : 0;
                v.rot_y = (l.ord_y && ly > 1) ? (uint32_t)(target_y_size / l.ord_y) : 0;
                views.push_back(v);
            }
            DeviceVec<ScalarField> out(target_x_size * target_y_size);
            check(tkmk_poly_expr_eval_views(pg.code.data(), (uint32_t)pg.code.size(), views.data(), (uint32_t)views.size(), pg.consts.data(),
                                            (uint32_t)pg.consts.size(), (uint32_t)target_x_size, (uint32_t)target_y_size, out.ptr(), nullptr),
                  "tkmk_poly_expr_eval_views");
            return DensePolynomialExt::from_rou_evals(out, target_x_size, target_y_size).with_degree(d.first, d.second);
        }
        DeviceVec<ScalarField> evals = on_domain(target_x_size, target_y_size, cache);
        return DensePolynomialExt::from_rou_evals(evals, target_x_size, target_y_size).with_degree(d.first, d.second);
    }

  private:
    explicit PolyExpr(Kind k) : kind(k) {}
    static PolyExpr binary(Kind k, PolyExpr l, PolyExpr r) {
        PolyExpr e{k};
        e.kids.push_back(std::move(l));
        e.kids.push_back(std::move(r));
        return e;
    }
    static std::shared_ptr<DeviceVec<ScalarField>> leaf_evals(const DensePolynomialExt *leaf, size_t xs, size_t ys, LeafCache &cache) {
        auto key = std::make_tuple(leaf, xs, ys);
        auto it = cache.find(key);
        if (it == cache.end()) {
            it = cache.emplace(key, std::make_shared<DeviceVec<ScalarField>>(leaf->evals_on(xs, ys))).first;
        }
        return it->second;
    }
    // postfix program for tkmk_poly_expr_eval; compile() returns the stack depth the node needs, or -1
    struct Program {
        std::vector<tkmk_expr_instr> code;
        struct Leaf {
            const DensePolynomialExt *p;
            size_t ord_x, ord_y;
        };
        std::vector<Leaf> leaves;
        std::vector<ScalarField> consts;
        uint8_t leaf_index(const DensePolynomialExt *p, size_t ord_x, size_t ord_y) {
            for (size_t i = 0; i < leaves.size(); i++)
                if (leaves[i].p == p && leaves[i].ord_x == ord_x && leaves[i].ord_y == ord_y) return (uint8_t)i;
            leaves.push_back({p, ord_x, ord_y});
            return (uint8_t)(leaves.size() - 1);
        }
        uint8_t const_index(const ScalarField &c) {
            for (size_t i = 0; i < consts.size(); i++)
                if (fr_eq(consts[i], c)) return (uint8_t)i;
            consts.push_back(c);
            return (uint8_t)(consts.size() - 1);
        }
    };
    int compile(Program &pg, int depth) const {
        if (pg.leaves.size() > 200 || pg.consts.size() > 200 || pg.code.size() > 1000) return -1;
        switch (kind) {
            case Poly: pg.code.push_back({TKMK_EXPR_LEAF, pg.leaf_index(leaf, shift_ord_x, shift_ord_y)}); return 1;
            case Scalar: pg.code.push_back({TKMK_EXPR_CONST, pg.const_index(scalar_)}); return 1;
            case Add:
            case Sub:
            case Mul: {
                int l = kids[0].compile(pg, depth);
                if (l < 0) return -1;
                int r = kids[1].compile(pg, depth + 1);
                if (r < 0) return -1;
                pg.code.push_back({(uint8_t)(kind == Add ? TKMK_EXPR_ADD : kind == Sub ? TKMK_EXPR_SUB : TKMK_EXPR_MUL), 0});
                return std::max(l, 1 + r);
            }
            case Scale: {
                int d = kids[0].compile(pg, depth);
                if (d < 0) return -1;
                pg.code.push_back({TKMK_EXPR_SCALE, pg.const_index(scalar_)});
                return d;
            }
            case MulXMinusOne: {
                int d = kids[0].compile(pg, depth);
                if (d < 0) return -1;
                pg.code.push_back({TKMK_EXPR_MUL_X_MINUS_ONE, 0});
                return d;
            }
            default: {
                if (kids.empty()) {
                    pg.code.push_back({TKMK_EXPR_CONST, pg.const_index(ScalarField{})});
                    return 1;
                }
                int need = 0;
                for (size_t i = 0; i < kids.size(); i++) {
                    int d = kids[i].compile(pg, depth + (i ? 1 : 0));
                    if (d < 0) return -1;
                    need = std::max(need, d + (i ? 1 : 0));
                    if (i) pg.code.push_back({TKMK_EXPR_ADD, 0});
                }
                return need;
            }
        }
    }
    // every node returns a buffer it owns (leaf evaluations are copied out of the cache like the reference does)
    DeviceVec<ScalarField> on_domain(size_t xs, size_t ys, LeafCache &cache) const {
        size_t n = xs * ys;
        tkmk_vecops_config c = dev_cfg();
        switch (kind) {
            case Poly: {
                if (!shift_ord_x && !shift_ord_y) return leaf_evals(leaf, xs, ys, cache)->clone();
                auto inv_root = [](size_t ord) {
                    ScalarField w;
                    check(bls12_381_get_root_of_unity(ord, &w), "get_root_of_unity");
                    ScalarField wi;
                    tkmk_vecops_config hc = tkmk_vecops_default_config();   // host in, host out (host Fr arithmetic lives above this header)
                    check(bls12_381_vector_inv(&w, 1, &hc, &wi), "vector_inv");
                    return wi;
                };
                ScalarField wx = shift_ord_x ? inv_root(shift_ord_x) : fr_from_u32(1), wy = shift_ord_y ? inv_root(shift_ord_y) : fr_from_u32(1);
                return leaf->scale_coeffs(shift_ord_x ? &wx : nullptr, shift_ord_y ? &wy : nullptr).evals_on(xs, ys);   // the reference's own route
            }
            case Scalar: {
                std::vector<ScalarField> v(n, scalar_);
                return DeviceVec<ScalarField>::from_host(v);
            }
            case Add:
            case Sub:
            case Mul: {
                DeviceVec<ScalarField> l = kids[0].on_domain(xs, ys, cache), r = kids[1].on_domain(xs, ys, cache);
                auto fn = kind == Add ? bls12_381_vector_add : kind == Sub ? bls12_381_vector_sub : bls12_381_vector_mul;
                check(fn(l.ptr(), r.ptr(), n, &c, l.ptr()), "fused pointwise");
                return l;
            }
            case Scale: {
                DeviceVec<ScalarField> e = kids[0].on_domain(xs, ys, cache);
                if (fr_eq(scalar_, fr_from_u32(1))) return e;
                c.is_a_on_device = false;
                check(bls12_381_scalar_mul_vec(&scalar_, e.ptr(), n, &c, e.ptr()), "fused scale");
                return e;
            }
            case MulXMinusOne: {
                DeviceVec<ScalarField> e = kids[0].on_domain(xs, ys, cache);
                check(tkmk_poly_mul_x_minus_one_evals(e.ptr(), (uint32_t)xs, (uint32_t)ys, e.ptr(), nullptr), "fused x-1");
                return e;
            }
            default: {
                std::vector<ScalarField> z(n);
                DeviceVec<ScalarField> acc = DeviceVec<ScalarField>::from_host(z);
                for (auto &k : kids) {
                    DeviceVec<ScalarField> t = k.on_domain(xs, ys, cache);
                    check(bls12_381_vector_add(acc.ptr(), t.ptr(), n, &c, acc.ptr()), "fused sum");
                }
                return acc;
            }
        }
    }
};

// ---- Sigma1 with a device-resident xy_powers table: encode_poly (iotools/mod.rs:2033-2113) ----
// The table is kept in the MSM's resident ("converted") form from construction on: the CRS is fixed for its lifetime, so the
// per-call base conversion of bls12_381_msm is paid once here; and a commit reads the coefficient box and the matching CRS
// sub-grid through strided VIEWS (tkmk_msm_multi_ex), where the reference copies both, point by point, before every MSM
// (iotools/mod.rs:2061-2088).
// What a commit actually ran over: the (x_degree + 1) x (y_degree + 1) coefficient box of encode_poly (iotools/mod.rs:2055-2060; the
// `msm=AxB` dims of the reference's timing reports, prove/optimization/timing.local.cpu.current.md "Encode Details"), or the whole
// evaluation grid of a Lagrange-basis commit.  Off unless a sink is installed (tkmk_prover_prove_ex's commit_boxes_json_out).
struct CommitBox {
    std::string name;
    size_t x, y;
    const char *basis;   // "coeff" | "evals"
};
inline std::vector<CommitBox> *&commit_box_sink() {
    static thread_local std::vector<CommitBox> *sink = nullptr;
    return sink;
}

// One proof over G GPUs: the commit table xy_powers is sharded like the coefficient matrices it is
// multiplied with — rank r of G holds the grid columns iy = r mod G (COLS layout, struct Shard above) — so that any coefficient box
// [0, tx) x [0, ty) splits evenly whatever its width, and a rank's share of a commit is a view of its own columns on both sides.
// The commit batches of a sharded prover go through its communicator (tkmk_msm_multi_ex_sharded of libtkmk_dist.so: every rank runs
// its share, ONE all-gather of 144 bytes per commit, the partials summed on the device).  Installed per host thread by the sharded
// context for the span of a call; absent, batches run on this GPU alone.
struct CommitComm {
    void *comm = nullptr;
    tkmk_error (*multi_ex_sharded)(void *comm, const tkmk_msm_job_ex *jobs, int n_jobs, const tkmk_msm_config *cfg, int bases_form, tkmk_g1_projective *results) = nullptr;
};
inline CommitComm &commit_comm() {
    static thread_local CommitComm c;
    return c;
}

class Sigma1 {
    DeviceVec<G1Affine> xy_powers_;   // level 0: the table in resident form; with table_c_: levels 1 .. table_factor_ - 1 behind it
    size_t rs_x_, rs_y_;              // the GRID (all ranks' columns); this rank holds local_cols_ of its rs_y_ columns, all rs_x_ rows
    uint32_t table_c_ = 0, table_factor_ = 0;
    Shard shard_;
    size_t local_cols_;

  public:
    // xy_powers[i*rs_y + j] = [tau_x^i tau_y^j]G (plain affine records);  rs_x = max(2n, 2(l_D - l)), rs_y = 2 s_max.
    // table_c > 0 (a resident prover that amortises it over many proofs): the table is expanded ONCE into its 2^(table_c j)
    // multiples (ICICLE's msm_precompute_bases, MSMConfig::precompute_factor — left at 1 by the reference), so that every large
    // commit runs as ONE bucket set with table_c-bit windows: 13 instead of 16 bucket additions per point at table_c = 20.
    // HBM: windows x the table (2^24 points at table_c = 20: 21 GB of the 288).
    // shard.world > 1: `xy_powers` holds THIS RANK'S columns only (cols_of_grid(...) below cuts them out of a whole grid), row-major
    // rs_x x local_cols; rs_y_size stays the grid's width.
    Sigma1(DeviceVec<G1Affine> &&xy_powers, size_t rs_x_size, size_t rs_y_size, uint32_t table_c = 0, Shard shard = Shard{})
        : xy_powers_(std::move(xy_powers)), rs_x_(rs_x_size), rs_y_(rs_y_size), shard_(shard), local_cols_(shard.cols_of(rs_y_size)) {
        if (shard_.world < 1 || shard_.rank >= shard_.world) throw Error("Sigma1: invalid shard");
        if (xy_powers_.len() != rs_x_ * local_cols_) throw Error("xy_powers has the wrong length");
        tkmk_msm_config cfg = tkmk_msm_default_config();
        cfg.are_points_on_device = cfg.are_results_on_device = true;
        if (table_c >= 2 && xy_powers_.len() >= 2) {
            const uint32_t windows = 255 / table_c + 1;
            if ((uint64_t)xy_powers_.len() * windows >= (1ull << 31)) throw Error("xy_powers is too large for a precomputed table");
            DeviceVec<G1Affine> table(xy_powers_.len() * windows);
            cfg.c = (int)table_c;
            cfg.precompute_factor = (int)windows;
            check(bls12_381_msm_precompute_bases(xy_powers_.ptr(), (int)xy_powers_.len(), &cfg, table.ptr()), "msm::precompute_bases");
            xy_powers_ = std::move(table);
            table_c_ = table_c, table_factor_ = windows;
        } else {
            check(bls12_381_msm_convert_bases(xy_powers_.ptr(), xy_powers_.len(), &cfg, xy_powers_.ptr()), "msm::convert_bases");
        }
    }
    size_t rs_x() const { return rs_x_; }
    size_t rs_y() const { return rs_y_; }
    size_t table_len() const { return rs_x_ * local_cols_; }   // rows of one table level on THIS rank
    size_t local_cols() const { return local_cols_; }
    const G1Affine *level0() const { return xy_powers_.ptr(); }   // this rank's rs_x x local_cols monomial points, resident (converted) form
    const Shard &shard() const { return shard_; }
    // columns r, r + G, ... of a row-major rs_x x rs_y grid of points: what rank r of G keeps (one strided copy on the device: local
    // element (i, k) = grid element (i, r + G k) = flat index r + G (i lc + k) when G divides rs_y)
    static DeviceVec<G1Affine> cols_of_grid(const DeviceVec<G1Affine> &grid, size_t rs_x, size_t rs_y, Shard shard) {
        if (grid.len() != rs_x * rs_y) throw Error("cols_of_grid: the grid has the wrong length");
        if (rs_y % shard.world) throw Error("cols_of_grid: the grid's width is not a multiple of the number of ranks");
        const size_t lc = rs_y / shard.world;
        DeviceVec<G1Affine> out(rs_x * lc);
        if (out.len())
            check(tkmk_memcpy_2d_d2d(out.ptr(), sizeof(G1Affine), grid.ptr() + shard.rank, (size_t)shard.world * sizeof(G1Affine), sizeof(G1Affine), rs_x * lc), "cols_of_grid");
        return out;
    }
    // rows r, r + G, ... of a row-major grid (the walk-ordered prefix table of prove1: one row per grid column)
    static DeviceVec<G1Affine> rows_of_grid(const DeviceVec<G1Affine> &grid, size_t rs_x, size_t rs_y, Shard shard) {
        if (grid.len() != rs_x * rs_y) throw Error("rows_of_grid: the grid has the wrong length");
        const size_t rows = shard.rows_of(rs_x);
        DeviceVec<G1Affine> out(rows * rs_y);
        if (rows)
            check(tkmk_memcpy_2d_d2d(out.ptr(), rs_y * sizeof(G1Affine), grid.ptr() + (size_t)shard.rank * rs_y, (size_t)shard.world * rs_y * sizeof(G1Affine),
                                     rs_y * sizeof(G1Affine), rows),
                  "rows_of_grid");
        return out;
    }
    uint32_t table_c() const { return table_c_; }
    // the MSM job of one commit: coefficient box x CRS sub-grid, both as views (msm_size 0 for the zero polynomial)
    tkmk_msm_job_ex job(DensePolynomialExt &poly, const char *name = nullptr) const {
        // the box the MSM runs over: the polynomial's degree bound (DensePolynomialExt::shape_degree) — or its MEASURED degree when the
        // caller records the boxes (the reference's own encode_poly measures: iotools/mod.rs:2055-2060; tests/golden/encode_dims.json).
        // A view addresses the box inside the matrix as it is: nothing is resized or copied.
        const auto deg = commit_box_sink() ? poly.find_degree() : poly.shape_degree();
        size_t tx = (size_t)(deg.first + 1), ty = (size_t)(deg.second + 1);
        if (deg.first < 0 || deg.second < 0) tx = ty = 0;
        if (tx > rs_x_ || ty > rs_y_) throw Error("Insufficient length of sigma.sigma_1.xy_powers");
        if (auto *sink = commit_box_sink()) sink->push_back({name ? name : "?", tx, ty, "coeff"});
        // this rank's columns of the box: grid columns r, r + G, ... < ty = columns 0 .. mine - 1 of the local table, and the same columns
        // of the coefficient matrix (distributed: its own local columns; replicated: that column set is cut out first).  world = 1: the
        // whole box, strides = the matrices' own.
        const size_t mine = shard_.cols_of(ty);
        const ScalarField *scalars = poly.poly.ptr();
        size_t scalar_stride = poly.ly();
        if (shard_.world > 1 && poly.rep) {
            DensePolynomialExt d = poly.to_distributed();
            scalar_stride = d.ly();
            poly.commit_slice_ = std::move(d.poly);
            scalars = poly.commit_slice_.ptr();
        }
        tkmk_msm_job_ex j{};
        j.scalars = scalars;
        j.bases = xy_powers_.ptr();
        j.msm_size = (int)(tx * mine);
        j.scalar_cols = (uint32_t)mine, j.scalar_stride = (uint32_t)scalar_stride;
        j.base_cols = (uint32_t)mine, j.base_stride = (uint32_t)local_cols_;
        j.base_index = nullptr;
        j.base_table_len = table_len();
        // large commits through the expanded table; small ones (the wide windows' two-pass sort needs 2^18 entries, and a 2^19-bucket
        // reduction is not worth paying for a few thousand points) through level 0 with the ordinary multi-window path
        if (table_c_ && (uint64_t)tx * mine * table_factor_ >= (1ull << 20)) j.table_c = table_c_, j.table_factor = table_factor_;
        return j;
    }
    static G1Affine to_affine(const tkmk_g1_projective &res) {
        G1Affine out{};
        bool inf = true;
        for (uint32_t l : res.z.limbs) inf &= l == 0;
        if (!inf) out.x = res.x, out.y = res.y;  // canonical (x, y, 1): dropping z is G1Affine::from(projective)
        return out;
    }
    static tkmk_msm_config device_cfg() {
        tkmk_msm_config cfg = tkmk_msm_default_config();
        cfg.are_scalars_on_device = cfg.are_points_on_device = true;
        return cfg;
    }
    // independent MSM jobs over converted tables in one pipelined call -> affine results; (0,0) = G1serde::zero()
    // stream: the caller's stream for the batch's scratch frame (a helper thread names its own, so that its frame does not share the
    // default stream's arena with the main thread: tkmk.h "re-entrant per stream")
    static std::vector<G1Affine> run_jobs(const std::vector<tkmk_msm_job_ex> &jobs, tkmk_stream stream = nullptr) {
        tkmk_msm_config cfg = device_cfg();
        cfg.stream_handle = stream;
        std::vector<tkmk_g1_projective> res(jobs.size());
        if (host_trace_on()) {
            std::string d;
            for (auto &j : jobs) d += " " + std::to_string(j.msm_size) + (j.table_c ? "t" : "") + (j.base_index ? "i" : "");
            host_trace("commit batch of %zu:%s", jobs.size(), d.c_str());
        }
        const CommitComm &cc = commit_comm();
        if (cc.comm) check(cc.multi_ex_sharded(cc.comm, jobs.data(), (int)jobs.size(), &cfg, TKMK_BASES_CONVERTED, res.data()), "tkmk_msm_multi_ex_sharded");
        else check(tkmk_msm_multi_ex(jobs.data(), (int)jobs.size(), &cfg, TKMK_BASES_CONVERTED, res.data()), "tkmk_msm_multi_ex");
        std::vector<G1Affine> out;
        for (auto &r : res) out.push_back(to_affine(r));
        return out;
    }
    // -> affine commitment
    G1Affine encode_poly(DensePolynomialExt &poly, const char *name = nullptr) const { return run_jobs({job(poly, name)})[0]; }
    // the MSM job of a polynomial given by its EVALUATIONS on the grid of this table, which holds the Lagrange-basis points of that grid
    // (lagrange_of below): every row of the table against the evaluation vector, no view; zeros and small values cost the MSM next to
    // nothing.  Sharded prover: the table object is built over THIS RANK'S part of the grid (its columns, or its stretch of the walk) and
    // `evals` is the same part of the evaluations — COLS layout on both sides.
    tkmk_msm_job_ex job_evals(const DeviceVec<ScalarField> &evals, const char *name = nullptr, size_t whole_x = 0, size_t whole_y = 0) const {
        if (shard_.world != 1) throw Error("job_evals: a Lagrange-basis table is built over the rank's own part of the grid");
        if (evals.len() < rs_x_ * rs_y_) throw Error("evaluation vector shorter than the Lagrange table");
        if (auto *sink = commit_box_sink()) sink->push_back({name ? name : "?", whole_x ? whole_x : rs_x_, whole_y ? whole_y : rs_y_, "evals"});
        tkmk_msm_job_ex j{};
        j.scalars = evals.ptr();
        j.bases = xy_powers_.ptr();
        j.msm_size = (int)(rs_x_ * rs_y_);
        j.base_table_len = table_len();
        if (table_c_ && (uint64_t)rs_x_ * rs_y_ * table_factor_ >= (1ull << 20)) j.table_c = table_c_, j.table_factor = table_factor_;
        return j;
    }
    // The Lagrange-basis twin of the grid [0, xs) x [0, ys) of this table: [L_i(tau_x) L_j(tau_y)] G = (1 / N) x the inverse NTT over G1
    // points of the monomial sub-grid (tkmk_g1_ntt is unscaled; tkmk_g1_scale folds the 1 / N in, once per circuit), with the same commit
    // table treatment.  commit(P) = MSM(evaluations of P, this).
    static ScalarField inverse_of_grid_size(size_t xs, size_t ys) { return fr_inv(fr_mul(fr_from_u32((uint32_t)xs), fr_from_u32((uint32_t)ys))); }
    DeviceVec<G1Affine> lagrange_points(size_t xs, size_t ys) const {   // plain affine records, row-major xs x ys
        if (shard_.world != 1) throw Error("Lagrange table: needs the whole grid (build it before the rows are sharded)");
        if (xs > rs_x_ || ys > rs_y_ || !is_pow2(xs) || !is_pow2(ys)) throw Error("Lagrange table: the grid must be a power-of-two corner of xy_powers");
        DeviceVec<G1Affine> lam(xs * ys);
        host_trace("lagrange_points %zu x %zu", xs, ys);
        check(tkmk_g1_ntt(xy_powers_.ptr(), TKMK_BASES_CONVERTED, (uint32_t)rs_y_, (uint32_t)xs, (uint32_t)ys, TKMK_NTT_INVERSE, lam.ptr(), nullptr), "tkmk_g1_ntt");
        const ScalarField inv_n = inverse_of_grid_size(xs, ys);
        check(tkmk_g1_scale(lam.ptr(), xs * ys, &inv_n, lam.ptr(), nullptr), "tkmk_g1_scale");
        return lam;
    }
    Sigma1 lagrange_of(size_t xs, size_t ys) const { return Sigma1(lagrange_points(xs, ys), xs, ys, table_c_); }
    // The table over which a PIECEWISE-CONSTANT evaluation vector (constant along the column-by-column walk of the xs x ys grid except at
    // a few jumps) commits as an MSM of its jumps: S_j = sum_{j' <= j} Lambda_{walk(j')}, and sum_j r_j Lambda_walk(j) = sum_j (r_j - r_{j+1}) S_j.
    // `lagrange` = lagrange_points(xs, ys).  The returned table has xs * ys rows in walk order.
    static DeviceVec<G1Affine> lagrange_prefix_points(const DeviceVec<G1Affine> &lagrange, size_t xs, size_t ys) {   // plain affine, xs * ys rows in walk order
        DeviceVec<G1Affine> pre(xs * ys);
        host_trace("lagrange_prefix_of %zu x %zu", xs, ys);
        check(tkmk_g1_prefix_sums(lagrange.ptr(), TKMK_BASES_PLAIN, (uint32_t)xs, (uint32_t)ys, 1, pre.ptr(), nullptr), "tkmk_g1_prefix_sums");
        return pre;
    }
    Sigma1 lagrange_prefix_of(const DeviceVec<G1Affine> &lagrange, size_t xs, size_t ys) const {
        return Sigma1(lagrange_prefix_points(lagrange, xs, ys), xs * ys, 1, table_c_);
    }
    // commitments of independent polynomials in one pipelined call
    std::vector<G1Affine> encode_polys(const std::vector<DensePolynomialExt *> &polys, const std::vector<const char *> &names = {}) const {
        std::vector<tkmk_msm_job_ex> jobs;
        for (size_t k = 0; k < polys.size(); k++) jobs.push_back(job(*polys[k], k < names.size() ? names[k] : nullptr));
        return run_jobs(jobs);
    }
};

}  // namespace tkmk
