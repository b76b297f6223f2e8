// host_driver.cpp — exercises the C++ host mirror (tokamak-zk-evm_amd/host/tkmk_host.hpp) on inputs written by
// tests/test_gpu_host_cpp.py and dumps every result for comparison with the oracle.  Mirrors the flow of the
// reference's own polynomial tests (packages/backend/libs/src/tests.rs): build, resize, scale, evaluate, multiply,
// divide, commit.
//   in : u32 xs, ys, bxs, bys, c, d, rs_x, rs_y | a[xs*ys] | b[bxs*bys] | fx fy x y s (5 Fr) | crs[rs_x*rs_y] (96 B each)
//   out: records { u32 tag, u64 nbytes, payload }
#include <cstdio>
#include <cstdlib>

#include "tkmk_host.hpp"

using namespace tkmk;

static FILE *g_out;
static void emit(uint32_t tag, const void *p, uint64_t n) {
    fwrite(&tag, 4, 1, g_out);
    fwrite(&n, 8, 1, g_out);
    if (n) fwrite(p, 1, n, g_out);
}
static void emit_poly(uint32_t tag, const DensePolynomialExt &p) {
    int64_t hdr[4] = {(int64_t)p.x_size, (int64_t)p.y_size, p.x_degree, p.y_degree};
    emit(tag, hdr, sizeof hdr);
    auto c = p.copy_coeffs();
    emit(tag + 1, c.data(), c.size() * sizeof(ScalarField));
}
template <class T>
static std::vector<T> rd(FILE *f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) {
        fprintf(stderr, "short input\n");
        exit(2);
    }
    return v;
}

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *in = fopen(argv[1], "rb");
    g_out = fopen(argv[2], "wb");
    if (!in || !g_out) return 2;
    try {
        check(tkmk_set_device(0), "set_device");
        auto h = rd<uint32_t>(in, 8);
        size_t xs = h[0], ys = h[1], bxs = h[2], bys = h[3], c = h[4], d = h[5], rs_x = h[6], rs_y = h[7];
        auto a = rd<ScalarField>(in, xs * ys);
        auto b = rd<ScalarField>(in, bxs * bys);
        auto sc = rd<ScalarField>(in, 5);
        auto crs = rd<G1Affine>(in, rs_x * rs_y);
        const ScalarField &fx = sc[0], &fy = sc[1], &x = sc[2], &y = sc[3], &s = sc[4];
        init_ntt_domain_for_size(1 << 16);
        init_ntt_domain_for_size(1 << 12);  // grow-only: a smaller request is a no-op

        DensePolynomialExt A = DensePolynomialExt::from_coeffs(a, xs, ys), B = DensePolynomialExt::from_coeffs(b, bxs, bys);
        auto deg = A.find_degree();
        int64_t dg[2] = {deg.first, deg.second};
        emit(1, dg, sizeof dg);
        DensePolynomialExt Ao = A.clone();
        Ao.optimize_size();
        emit_poly(10, Ao);
        emit_poly(20, A.scale_coeffs_x(fx).scale_coeffs_y(fy));
        ScalarField ev = A.eval(x, y);
        emit(30, &ev, sizeof ev);
        emit_poly(32, A.eval_x(x));
        emit_poly(34, A.eval_y(y));
        emit_poly(40, A * B);
        emit_poly(42, (A + B) - (B * s));
        // coset evaluations and back (tests.rs:134-180)
        DeviceVec<ScalarField> evs(xs * ys);
        A.to_rou_evals(&fx, &fy, evs);
        auto evh = evs.to_host();
        emit(50, evh.data(), evh.size() * sizeof(ScalarField));
        emit_poly(52, DensePolynomialExt::from_rou_evals(evs, xs, ys, &fx, &fy));
        // divisions
        DensePolynomialExt Ad = A.clone();
        auto q = Ad.div_by_vanishing_opt((int64_t)c, (int64_t)d);
        emit_poly(60, q.first);
        emit_poly(62, q.second);
        auto r = A.div_by_ruffini(x, y);
        emit_poly(70, std::get<0>(r));
        emit_poly(72, std::get<1>(r));
        emit(74, &std::get<2>(r), sizeof(ScalarField));
        // fused expression: A*B + s*(X-1)*B - (fx*A + fy*B)
        PolyExpr e = PolyExpr::sub(PolyExpr::add(PolyExpr::mul(PolyExpr::poly(A), PolyExpr::poly(B)),
                                                 PolyExpr::scale(s, PolyExpr::mul_x_minus_one(PolyExpr::poly(B)))),
                                   PolyExpr::weighted_sum({{fx, PolyExpr::poly(A)}, {fy, PolyExpr::poly(B)}}));
        emit_poly(80, e.evaluate_fused());
        // commitment
        Sigma1 sigma(DeviceVec<G1Affine>::from_host(crs), rs_x, rs_y);
        DensePolynomialExt Ac = A.clone();
        G1Affine cm = sigma.encode_poly(Ac);
        emit(90, &cm, sizeof cm);
        DensePolynomialExt Z = DensePolynomialExt::from_coeffs(std::vector<ScalarField>(4), 2, 2);
        G1Affine cz = sigma.encode_poly(Z);
        emit(92, &cz, sizeof cz);
        // the same two commits plus B's through the pipelined multi-commit call
        DensePolynomialExt A2 = A.clone(), Z2 = DensePolynomialExt::from_coeffs(std::vector<ScalarField>(4), 2, 2), B2 = B.clone();
        std::vector<G1Affine> many = sigma.encode_polys({&A2, &Z2, &B2});
        emit(94, many.data(), many.size() * sizeof(G1Affine));
        // error behaviour: the reference panics, the mirror throws
        uint32_t threw = 0;
        try {
            DensePolynomialExt::from_coeffs(a, xs + 1, ys);
        } catch (const Error &) {
            threw |= 1;
        }
        try {
            DensePolynomialExt t = A.clone();
            t.div_by_vanishing_opt(3, 4);
        } catch (const Error &) {
            threw |= 2;
        }
        try {
            DensePolynomialExt big = DensePolynomialExt::from_coeffs(std::vector<ScalarField>(4 * rs_x * rs_y, fr_from_u32(1)), 2 * rs_x, 2 * rs_y);
            sigma.encode_poly(big);
        } catch (const Error &) {
            threw |= 4;
        }
        emit(99, &threw, sizeof threw);
    } catch (const std::exception &ex) {
        fprintf(stderr, "host_driver: %s\n", ex.what());
        return 1;
    }
    fclose(g_out);
    return 0;
}
