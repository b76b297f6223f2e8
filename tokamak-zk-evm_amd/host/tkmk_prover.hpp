// tkmk_prover.hpp — the prover over the C++ host side: work-alike of Prover::init / prove0 .. prove4 and of the round loop of main
// (packages/backend/prove/src/lib.rs:675-1206, 1446-3206; packages/backend/prove/src/main.rs:27-97), with the reference's
// structure names (Mixer, Binding, Proof0..Proof4, Proof4Test).  Same design as tkmk/prove.py, which it matches byte for byte
// (tests/test_gpu_prove.py::test_native_prove_binary): every polynomial stays in HBM from init to the last commitment, the
// independent commitments of one round are ONE tkmk_msm_multi call (6 / 2 / 9), prove1's running product is a device scan,
// prove2's p_comb is one pass of the fused expression evaluator.
#pragma once
#include <array>
#include <chrono>
#include <functional>
#include <future>
#include <initializer_list>
#include <map>
#include <memory>
#include <sys/random.h>

#include <cerrno>

#include "tkmk_fq_host.hpp"
#include "tkmk_fr.hpp"
#include "tkmk_json.hpp"
#include "tkmk_protocol.hpp"
#include "tkmk_witness.hpp"

namespace tkmk {

struct Mixer {   // lib.rs:251-263
    ScalarField rU_X, rU_Y, rV_X, rV_Y;
    std::array<ScalarField, 4> rW_X, rW_Y;   // 3 random values resized to 4 with a zero (lib.rs:1045-1060)
    std::array<ScalarField, 2> rB_X, rB_Y;
    ScalarField rR_X, rR_Y, rO_mid;
    // ScalarCfg::generate_random (lib.rs:1040-1080): uniform below r — 255 bits from the kernel's CSPRNG (getrandom), redrawn
    // until the value is below r (rejection sampling: no modular bias; r is 0.906 of 2^255, so 1.1 draws on average)
    static ScalarField random_scalar() {
        for (;;) {
            uint8_t b[32];
            size_t got = 0;
            while (got < sizeof b) {
                ssize_t k = ::getrandom(b + got, sizeof b - got, 0);
                if (k < 0) {
                    if (errno == EINTR) continue;
                    throw Error("getrandom failed: no entropy source for the blinding scalars");
                }
                got += (size_t)k;
            }
            b[31] &= 0x7f;
            frh::U256 v;
            std::memcpy(v.l, b, 32);
            if (!frh::geq(v, frh::MOD)) return frh::store(v);
        }
    }
    static Mixer random() {
        auto draw = [] { return random_scalar(); };
        Mixer m;
        m.rU_X = draw(), m.rU_Y = draw(), m.rV_X = draw(), m.rV_Y = draw();
        m.rW_X = {draw(), draw(), draw(), ScalarField{}};
        m.rW_Y = {draw(), draw(), draw(), ScalarField{}};
        m.rB_X = {draw(), draw()};
        m.rB_Y = {draw(), draw()};
        m.rO_mid = draw(), m.rR_X = draw(), m.rR_Y = draw();
        return m;
    }
};
// testing hook (not in the reference): blinding scalars from a JSON document {"rU_X": "0x..", ..., "rW_X": [4 values], "rB_X": [2 values], ...}
// so that two implementations can be compared byte for byte.  Never used unless the caller asks for it explicitly
// (bin/prove --testing-mixer FILE, tkmk_prover_prove's testing_mixer_json argument).
inline Mixer mixer_from_json(const json::Value &j) {
    auto one = [&](const char *k) { return fr_from_hex(j.at(k).as_string()); };
    Mixer m;
    m.rU_X = one("rU_X"), m.rU_Y = one("rU_Y"), m.rV_X = one("rV_X"), m.rV_Y = one("rV_Y");
    m.rO_mid = one("rO_mid"), m.rR_X = one("rR_X"), m.rR_Y = one("rR_Y");
    auto fill = [&](const char *k, ScalarField *dst, size_t n) {
        const auto &items = j.at(k).items();
        if (items.size() != n) throw Error(std::string("mixer field ") + k + " has the wrong length");
        for (size_t i = 0; i < n; i++) dst[i] = fr_from_hex(items[i].as_string());
    };
    fill("rW_X", m.rW_X.data(), 4), fill("rW_Y", m.rW_Y.data(), 4), fill("rB_X", m.rB_X.data(), 2), fill("rB_Y", m.rB_Y.data(), 2);
    return m;
}
struct Binding {
    G1Affine A_free, O_pub_free, O_mid, O_prv;
};
struct Proof0 {
    G1Affine U, V, W, Q_AX, Q_AY, B;
};
struct Proof1 {
    G1Affine R;
};
struct Proof2 {
    G1Affine Q_CX, Q_CY;
};
struct Proof3 {
    ScalarField V_eval, R_eval, R_omegaX_eval, R_omegaX_omegaY_eval;
};
struct Proof4 {
    G1Affine Pi_X, Pi_Y, M_X, M_Y, N_X, N_Y;
};
struct Proof4Test {
    G1Affine Pi_CX, Pi_CY, Pi_AX, Pi_AY, Pi_B, M_X, M_Y, N_X, N_Y;
};
struct Proof {
    Binding binding;
    Proof0 proof0;
    Proof1 proof1;
    Proof2 proof2;
    Proof3 proof3;
    Proof4 proof4;
    // convert_format_for_solidity_verifier (lib.rs:452-513)
    FormattedEntries convert_format_for_solidity_verifier() const {
        FormattedEntries f;
        for (const G1Affine *p : {&proof0.U, &proof0.V, &proof0.W, &binding.O_mid, &binding.O_prv, &proof0.Q_AX, &proof0.Q_AY, &proof2.Q_CX,
                                  &proof2.Q_CY, &proof4.Pi_X, &proof4.Pi_Y, &proof0.B, &proof1.R, &proof4.M_Y, &proof4.M_X, &proof4.N_Y,
                                  &proof4.N_X, &binding.O_pub_free, &binding.A_free})
            split_push(f, *p);
        for (const ScalarField *s : {&proof3.R_eval, &proof3.R_omegaX_eval, &proof3.R_omegaX_omegaY_eval, &proof3.V_eval})
            f.part2.push_back(scalar_to_hex(*s));
        return f;
    }
    std::string to_json() const { return entries_json("proof_entries_part1", "proof_entries_part2", convert_format_for_solidity_verifier()); }
};

// the CRS parts `prove` uses (SigmaHolder::load of combined_sigma: libs/src/group_structures/mod.rs:313-551 for the names)
struct ProverSigma {
    Sigma1 sigma1;
    DeviceVec<G1Affine> gamma_inv_o_inst, eta_inv_li_o_inter_alpha4_kj, delta_inv_li_o_prv;
    std::vector<G1Affine> delta_inv_alphak_xh_tx, delta_inv_alpha4_xj_tx, delta_inv_alphak_yi_ty;   // 3x3, 2, 4x3
    G1Affine delta, eta;
    bool binding_tables_converted = false;   // ProverContext keeps the three binding tables in the MSM's resident form
    // host copies of the few xy_powers entries the blinding terms of U, V, W, B touch: (a, 0) and (0, b) for small a, b and around n, m_I, s_max
    std::map<std::pair<size_t, size_t>, G1Affine> xy_edge;
    const G1Affine &xy_at(size_t a, size_t b) const {
        auto it = xy_edge.find({a, b});
        if (it == xy_edge.end()) throw Error("xy_powers entry not kept on the host");
        return it->second;
    }
    // shard.world > 1 (one proof over several GPUs): sigma1 keeps this rank's grid columns only; *whole_grid (when asked for) receives the
    // whole xy_powers grid in the MSM's resident form without a table — what the Lagrange-basis tables are derived from before they
    // are sharded the same way; the caller drops it afterwards.  The binding tables are replicated (every rank commits the index lists
    // of its own placements).
    static ProverSigma from_payload(const CrsPayload &crs, const SetupParams &sp, uint32_t table_c = 0, Shard shard = Shard{},
                                    std::unique_ptr<Sigma1> *whole_grid = nullptr) {
        size_t m_i = sp.l_D - sp.l, rs_x = std::max(2 * sp.n, 2 * m_i), rs_y = 2 * sp.s_max;
        auto want = [&](CrsPayload::Section s, size_t pts, const char *name) {
            if (crs.points(s) != pts) throw Error(std::string("CRS section ") + name + " does not match setupParams.json");
        };
        want(CrsPayload::XyPowers, rs_x * rs_y, "xy_powers");
        want(CrsPayload::GammaInvOInst, sp.l, "gamma_inv_o_inst");
        want(CrsPayload::EtaInvLiOInterAlpha4Kj, m_i * sp.s_max, "eta_inv_li_o_inter_alpha4_kj");
        want(CrsPayload::DeltaInvLiOPrv, (sp.m_D - sp.l_D) * sp.s_max, "delta_inv_li_o_prv");
        want(CrsPayload::DeltaInvAlphakXhTx, 9, "delta_inv_alphak_xh_tx");
        want(CrsPayload::DeltaInvAlpha4XjTx, 2, "delta_inv_alpha4_xj_tx");
        want(CrsPayload::DeltaInvAlphakYiTy, 12, "delta_inv_alphak_yi_ty");
        auto host = [&](CrsPayload::Section s) { return std::vector<G1Affine>(crs.g1(s), crs.g1(s) + crs.points(s)); };
        const G1Affine *singles = crs.g1(CrsPayload::G1Singles);   // G, x, y, delta, eta, lagrange_KL
        std::map<std::pair<size_t, size_t>, G1Affine> edge;
        {
            const G1Affine *xy = crs.g1(CrsPayload::XyPowers);
            for (size_t base : {(size_t)0, (size_t)sp.n, m_i})
                for (size_t k = 0; k < 4; k++)
                    if (base + k < rs_x) edge[{base + k, 0}] = xy[(base + k) * rs_y];
            for (size_t base : {(size_t)0, (size_t)sp.s_max})
                for (size_t k = 0; k < 4; k++)
                    if (base + k < rs_y) edge[{0, base + k}] = xy[base + k];
        }
        DeviceVec<G1Affine> grid = crs.upload(CrsPayload::XyPowers);
        DeviceVec<G1Affine> mine = shard.world > 1 ? Sigma1::cols_of_grid(grid, rs_x, rs_y, shard) : DeviceVec<G1Affine>();
        if (shard.world > 1 && whole_grid) whole_grid->reset(new Sigma1(std::move(grid), rs_x, rs_y, 0));
        ProverSigma out{Sigma1(shard.world > 1 ? std::move(mine) : std::move(grid), rs_x, rs_y, table_c, shard),
                           crs.upload(CrsPayload::GammaInvOInst),
                           crs.upload(CrsPayload::EtaInvLiOInterAlpha4Kj),
                           crs.upload(CrsPayload::DeltaInvLiOPrv),
                           host(CrsPayload::DeltaInvAlphakXhTx),
                           host(CrsPayload::DeltaInvAlpha4XjTx),
                           host(CrsPayload::DeltaInvAlphakYiTy),
                           singles[3],
                           singles[4],
                           false,
                           {}};
        out.xy_edge = std::move(edge);
        return out;
    }
};

struct ProverInputs {   // the documents Prover::init reads (lib.rs:679-835)
    SetupParams sp;
    std::vector<SubcircuitInfo> infos;
    std::vector<size_t> n_consts;   // subcircuitInfo.json "Nconsts", by position in infos
    std::vector<PlacementVariables> pv;
    std::vector<Permutation> perm;
    std::vector<ScalarField> a_pub_user, a_pub_block;
    std::string qap_path;
};

namespace prover_detail {

using Poly = DensePolynomialExt;
using Term = Poly::Term;   // {coefficient, polynomial, X shift, Y shift}

// poly_comb! (lib.rs:30-38): sum of c_i * p_i — one fused pass over the operands (tkmk_poly_lincomb)
inline Poly poly_comb(std::initializer_list<Term> terms) { return Poly::lincomb(std::vector<Term>(terms)); }
// A linear combination kept symbolic until its matrix is needed.  Sums, scalar multiples and monomial shifts of combinations are
// combinations of the same operands: a chain of poly_comb! / `&a * &s` / `&a + &b` / mul_monomial steps, each of which the reference
// (and a step-by-step transcription) materialises as a matrix of the output's size, becomes ONE pass over the base operands with the
// coefficients multiplied out on the host.  Field arithmetic is exact, so the coefficients of the result are the same numbers.
struct Lin {
    std::vector<Term> t;
    Lin() {}
    Lin &add(const ScalarField &c, const Poly &p, uint32_t ox = 0, uint32_t oy = 0) {
        t.emplace_back(c, &p, ox, oy);
        return *this;
    }
    Lin &add(const ScalarField &c, const Lin &o, uint32_t ox = 0, uint32_t oy = 0) {
        for (const Term &x : o.t) t.emplace_back(fr_mul(c, x.c), x.p, x.ox + ox, x.oy + oy);
        return *this;
    }
    Lin &add(const std::vector<Term> &terms) {
        t.insert(t.end(), terms.begin(), terms.end());
        return *this;
    }
    // equal (operand, shift) pairs merged, zero coefficients dropped; large operands first (a combination of more terms than one launch
    // takes is folded in groups: the small ones then share the last group)
    std::vector<Term> merged() const {
        std::vector<Term> m;
        for (const Term &x : t) {
            bool found = false;
            for (Term &y : m)
                if (y.p == x.p && y.ox == x.ox && y.oy == x.oy) {
                    y.c = fr_add(y.c, x.c), found = true;
                    break;
                }
            if (!found) m.push_back(x);
        }
        m.erase(std::remove_if(m.begin(), m.end(), [](const Term &x) { return fr_is_zero(x.c); }), m.end());
        std::stable_sort(m.begin(), m.end(), [](const Term &a, const Term &b) { return a.p->x_size * a.p->y_size > b.p->x_size * b.p->y_size; });
        return m;
    }
    Poly materialize() const { return Poly::lincomb(merged()); }
};
inline Poly constant_poly(const ScalarField &c) { return Poly::from_coeffs(std::vector<ScalarField>{c}, 1, 1); }
inline Poly sparse(const std::vector<std::pair<size_t, ScalarField>> &entries, size_t xs, size_t ys) {
    std::vector<ScalarField> c(xs * ys);
    for (auto &e : entries) c.at(e.first) = e.second;
    return Poly::from_coeffs(c, xs, ys);
}
// lib.rs:48-68: (sum c_i T^i) * (T^exponent - 1) along one axis
template <size_t K>
inline Poly low_degree_times_vanishing(const std::array<ScalarField, K> &coeffs, size_t exponent, bool x_axis) {
    if (exponent == 0) throw Error("low_degree_times_vanishing: exponent must be positive");   // assert!(exponent > 0)
    size_t size = next_pow2(exponent + K);
    std::vector<ScalarField> c(size);
    for (size_t i = 0; i < K; i++) {   // accumulated: the two copies overlap when exponent < K
        c[i] = fr_sub(c[i], coeffs[i]);
        c[i + exponent] = fr_add(c[i + exponent], coeffs[i]);
    }
    return x_axis ? Poly::from_coeffs(c, size, 1) : Poly::from_coeffs(c, 1, size);
}
inline Poly vanishing(size_t size, bool x_axis) {   // lib.rs:849-894
    std::vector<std::pair<size_t, ScalarField>> e = {{0, fr_neg(fr_one())}, {size, fr_one()}};
    return x_axis ? sparse(e, 2 * size, 1) : sparse(e, 1, 2 * size);
}
inline Poly unit_evals(size_t size, size_t index, bool x_axis) {   // the Lagrange polynomials K, L, K0 (lib.rs:2018-2100)
    std::vector<ScalarField> e(size);
    e.at(index) = fr_one();
    DeviceVec<ScalarField> d = DeviceVec<ScalarField>::from_host(e);
    return x_axis ? Poly::from_rou_evals_rep(d, size, 1) : Poly::from_rou_evals_rep(d, 1, size);   // host values: the same on every rank
}
// &poly + &scalar / &poly - &scalar (bivariate_polynomial/mod.rs:1042-1116, 1189-1262): only coefficient (0,0) changes
inline void add_const_in_place(Poly &p, const ScalarField &s) { p.add_to_constant_term(s); }
inline Poly add_const(const Poly &p, const ScalarField &s) {
    Poly out = p.clone();
    add_const_in_place(out, s);
    return out;
}
inline Poly sub_const(const Poly &p, const ScalarField &s) { return add_const(p, fr_neg(s)); }
// the small products of lib.rs:70-124 as shifted terms of one fused pass (X * p = p shifted by one row, Y * p = by one column)
inline Poly mul_by_x_minus_one(const Poly &p) { return Poly::lincomb({Term(fr_one(), &p, 1, 0), Term(fr_neg(fr_one()), &p)}); }
inline Poly mul_by_one_minus_x(const Poly &p) { return Poly::lincomb({Term(fr_one(), &p), Term(fr_neg(fr_one()), &p, 1, 0)}); }
inline Poly mul_by_linear(const Poly &p, const std::array<ScalarField, 2> &c, bool x_axis) {   // lib.rs:80-94
    return Poly::lincomb({Term(c[0], &p), x_axis ? Term(c[1], &p, 1, 0) : Term(c[1], &p, 0, 1)});
}
// terms of mul_by_term9 (lib.rs:96-124): (constant + cx X + cy Y) * p
inline std::vector<Term> term9(const Poly &p, const std::array<ScalarField, 2> &rB_X, const std::array<ScalarField, 2> &rB_Y,
                               const ScalarField &t_mi_eval, const ScalarField &t_smax_eval) {
    ScalarField constant = fr_add(fr_mul(t_mi_eval, rB_X[0]), fr_mul(t_smax_eval, rB_Y[0]));
    return {Term(constant, &p), Term(fr_mul(t_mi_eval, rB_X[1]), &p, 1, 0), Term(fr_mul(t_smax_eval, rB_Y[1]), &p, 0, 1)};
}
// several equally long linear combinations of G1 points in one batched MSM call (G1serde `+`, `-`, `* scalar`)
inline std::vector<G1Affine> g1_lincombs(const std::vector<std::vector<std::pair<ScalarField, G1Affine>>> &rows, tkmk_stream stream = nullptr) {
    size_t k = rows.at(0).size();
    std::vector<ScalarField> sc;
    std::vector<G1Affine> pts;
    for (auto &r : rows) {
        if (r.size() != k) throw Error("g1_lincombs: ragged rows");
        for (auto &t : r) sc.push_back(t.first), pts.push_back(t.second);
    }
    tkmk_msm_config cfg = tkmk_msm_default_config();
    cfg.batch_size = (int)rows.size();
    cfg.are_points_shared_in_batch = false;
    cfg.stream_handle = stream;   // a helper thread names its own stream (include/tkmk.h: one host thread per stream)
    std::vector<tkmk_g1_projective> res(rows.size());
    check(bls12_381_msm(sc.data(), pts.data(), (int)k, &cfg, res.data()), "msm::msm");
    std::vector<G1Affine> out;
    for (auto &r : res) out.push_back(projective_to_affine(r));
    return out;
}

}  // namespace prover_detail

class Prover {
    using Poly = DensePolynomialExt;

  public:
    SetupParams sp;
    size_t m_i = 0;
    const ProverSigma *sigma = nullptr;
    Mixer mixer;
    Poly bXY, uXY, vXY, wXY, rXY, a_free_X, t_n, t_mi, t_smax, s0XY, s1XY;
    Poly q0XY, q1XY, q2XY, q3XY;
    std::unique_ptr<Poly> w_zk, term_b_zk;   // ProverCache (lib.rs:297-301)
    // the same two polynomials as their X-only and Y-only summands (a column and a row): what prove4's fused combinations read instead
    // of the 2 m_I x 2 s_max matrix of zeros their sum is stored as
    std::unique_ptr<Poly> w_zk_x, w_zk_y, b_zk_x, b_zk_y;
    // the Lagrange polynomials of prove2 / prove4 (lib.rs:2018-2100) depend on the setup parameters only: a resident context
    // builds them once and shares them; a stand-alone Prover builds them on first use
    struct LagrangePolys {
        Poly K_last, L_last, K0, KL;   // K_{m_I - 1}(X), L_{s_max - 1}(Y), K_0(X), K_last * L_last
        static std::shared_ptr<const LagrangePolys> make(size_t m_i, size_t s_max);
    };
    std::shared_ptr<const LagrangePolys> lagrange;
    // Evaluation-basis commitments (a resident context with Lagrange-basis tables): u, v, w and b exist as evaluations on the roots of
    // unity before they exist as coefficients (read_R1CS_gen_uvwXY, gen_bXY), and those are mostly zeros and small numbers.  With the
    // tables set, prove0 commits U, V, W, B as MSM(evaluations, Lagrange table) + the blinding terms; the points are the same.
    DeviceVec<ScalarField> u_ev, v_ev, w_ev, b_ev;
    // the permutation polynomials' evaluations on the m_I x s_max grid and their identity part (w_x^row, w_y^col) — kept by a resident
    // context: prove1's f and g are then formed ON THE GRID (the forward transforms of lib.rs:1813-1830 are linear and exact, so the
    // values are the same) instead of from coefficients through two transforms.  All in the COLS layout of a sharded prover.
    DeviceVec<ScalarField> s0_ev, s1_ev;
    const DeviceVec<ScalarField> *s0_identity = nullptr, *s1_identity = nullptr;
    const Sigma1 *lagrange_n = nullptr, *lagrange_mi = nullptr;   // grids n x s_max (u, v, w) and m_I x s_max (b)
    // prove1's r is a running product of g / f along the column-by-column walk of the m_I x s_max grid, and g / f = 1 wherever the copy
    // permutation is the identity: r is constant between the few cells the permutation touches.  Over the prefix sums of the Lagrange
    // points in walk order it commits as an MSM of its jumps (zero scalars elsewhere, which the MSM skips).
    const Sigma1 *lagrange_mi_prefix = nullptr;
    const LagrangePolys &lagrange_polys() {
        if (!lagrange) lagrange = LagrangePolys::make(m_i, sp.s_max);
        return *lagrange;
    }
    std::map<std::string, double> timing;
    // EARLY COMMITS (a resident single-GPU context sets commit_stream): the commitments of a round that do not wait for the round's
    // polynomial work — prove0's U, V, W, B from the evaluations init left behind, prove4's M_X, M_Y, N_Y from R and the round-3 openings —
    // are issued from a helper thread on their own stream BEFORE that work starts and collected after the round's last commit batch.
    // They are the sparse or small MSMs whose time goes to sorts and latency-bound tails; those now run under the transforms and
    // streaming passes of the same round instead of behind them.  Same jobs, same points: not a byte of the proof changes
    // (TKMK_PROVER_EARLY_COMMITS=0 restores the in-line order).
    tkmk_stream commit_stream = nullptr;
    bool can_commit_early() const {
        static const bool enabled = [] {
            const char *e = getenv("TKMK_PROVER_EARLY_COMMITS");
            return !(e && atoi(e) == 0);
        }();
        return enabled && commit_stream && !dist_ctx().on() && !commit_comm().comm && !commit_box_sink() && tkmk_msm_get_pipeline_streams() > 1;
    }
    // background: the batch's accumulate kernels take one workgroup per CU (tkmk_stream_set_background) — for a dense commit nothing waits
    // for until the end of the round (prove4's M_X under the round's polynomial work: 54.4 -> 52.2 ms); prove0's evaluation commits are what
    // the round ends on and stay foreground (measured: background costs them 1 ms)
    std::future<std::vector<G1Affine>> commit_early(std::vector<tkmk_msm_job_ex> jobs, bool background = false) const {
        tkmk_stream st = commit_stream;
        check(tkmk_stream_set_background(st, background ? 1 : 0), "tkmk_stream_set_background");
        return std::async(std::launch::async, [jobs, st] { return Sigma1::run_jobs(jobs, st); });
    }
    // The blinding points of the evaluation-basis commitments: commit(p + sum_k c_k T^k (T^e - 1)) = MSM(evaluations of p, Lagrange table)
    // + sum_k c_k ([tau^(e+k)]G - [tau^k]G), and the second summand depends on the mixer and the CRS only — one small batched MSM per
    // proof, made by the helper thread of `init`'s binding commitments before its own batch (tkmk_service.hpp), i.e. under prove0's work (the
    // reference has no such step: it commits the blinded coefficients, lib.rs:1744-1782, 1940-1956); U, V, W, B and R are then ONE
    // affine addition each on the host when their MSM returns.  blinds_ready (valid when a helper makes them) is waited for by blinds().
    struct Blinds {
        G1Affine U, V, W, B, R;
        G1Affine O_mid, O_prv;   // the blinding terms of the two binding commitments (lib.rs:1100-1160), same treatment
    };
    mutable std::unique_ptr<Blinds> blinds_;
    std::shared_future<void> blinds_ready;
    Blinds compute_blinds(tkmk_stream st) const {
        using namespace prover_detail;
        const Mixer &mx = mixer;
        const size_t n = sp.n, s_max = sp.s_max;
        using Row = std::vector<std::pair<ScalarField, G1Affine>>;
        auto vanishing_terms = [&](Row &row, const ScalarField *coef, size_t k_count, size_t exponent, bool x_axis) {
            for (size_t k = 0; k < k_count; k++) {
                row.push_back({coef[k], x_axis ? sigma->xy_at(exponent + k, 0) : sigma->xy_at(0, exponent + k)});
                row.push_back({fr_neg(coef[k]), x_axis ? sigma->xy_at(k, 0) : sigma->xy_at(0, k)});
            }
        };
        Row ru, rv, rw, rb, rr;
        vanishing_terms(ru, &mx.rU_X, 1, n, true), vanishing_terms(ru, &mx.rU_Y, 1, s_max, false);
        vanishing_terms(rv, &mx.rV_X, 1, n, true), vanishing_terms(rv, &mx.rV_Y, 1, s_max, false);
        vanishing_terms(rw, mx.rW_X.data(), mx.rW_X.size(), n, true), vanishing_terms(rw, mx.rW_Y.data(), mx.rW_Y.size(), s_max, false);
        vanishing_terms(rb, mx.rB_X.data(), mx.rB_X.size(), m_i, true), vanishing_terms(rb, mx.rB_Y.data(), mx.rB_Y.size(), s_max, false);
        vanishing_terms(rr, &mx.rR_X, 1, m_i, true), vanishing_terms(rr, &mx.rR_Y, 1, s_max, false);
        const auto &xh = sigma->delta_inv_alphak_xh_tx, &xj = sigma->delta_inv_alpha4_xj_tx, &yi = sigma->delta_inv_alphak_yi_ty;
        Row mid = {{mx.rO_mid, sigma->delta}};
        Row prv = {   // lib.rs:1146-1160 without the core commitment
            {fr_neg(mx.rO_mid), sigma->eta},
            {mx.rU_X, xh[0]}, {mx.rV_X, xh[3]}, {mx.rW_X[0], xh[6]}, {mx.rW_X[1], xh[7]}, {mx.rW_X[2], xh[8]},
            {mx.rB_X[0], xj[0]}, {mx.rB_X[1], xj[1]},
            {mx.rU_Y, yi[0]}, {mx.rV_Y, yi[3]}, {mx.rW_Y[0], yi[6]}, {mx.rW_Y[1], yi[7]}, {mx.rW_Y[2], yi[8]},
            {mx.rB_Y[0], yi[9]}, {mx.rB_Y[1], yi[10]}};
        size_t width = 0;
        for (Row *r : {&ru, &rv, &rw, &rb, &rr, &mid, &prv}) width = std::max(width, r->size());
        ScalarField zero{};
        for (Row *r : {&ru, &rv, &rw, &rb, &rr, &mid, &prv})
            while (r->size() < width) r->push_back({zero, r->front().second});
        auto b = g1_lincombs({ru, rv, rw, rb, rr, mid, prv}, st);
        return Blinds{b[0], b[1], b[2], b[3], b[4], b[5], b[6]};
    }
    const Blinds &blinds() const {
        if (blinds_ready.valid()) blinds_ready.get();   // rethrows what the helper met
        if (!blinds_) blinds_.reset(new Blinds(compute_blinds(nullptr)));
        return *blinds_;
    }

    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // Prover::init (lib.rs:675-1206) after the JSON documents are parsed
    static std::pair<std::unique_ptr<Prover>, Binding> init(const ProverInputs &in, const ProverSigma &sigma, const Mixer &mixer) {
        using namespace prover_detail;
        double t0 = now();
        std::unique_ptr<Prover> p(new Prover());
        p->sp = in.sp;
        const SetupParams &sp = in.sp;
        size_t m_i = sp.l_D - sp.l, n = sp.n, s_max = sp.s_max;
        if (sp.l_D < sp.l) throw Error("Invalid setup params: l_D must be >= l.");   // setup_shape / validate_setup_shape (libs/src/utils/mod.rs:21-46)
        if (!is_pow2(n)) throw Error("n is not a power of two.");
        if (!is_pow2(s_max)) throw Error("s_max is not a power of two.");
        if (!is_pow2(m_i)) throw Error("m_I is not a power of two.");
        p->m_i = m_i;
        if (sigma.binding_tables_converted) throw Error("Prover::init takes plain binding tables; this reference string belongs to a ProverContext");
        p->sigma = &sigma;
        p->mixer = mixer;
        init_ntt_domain_for_size(4 * std::max(m_i, n) * 2 * s_max);   // prover_verifier_ntt_domain_size (libs/src/utils/mod.rs:51-58)
        p->bXY = gen_bXY(in.pv, in.infos, sp);
        std::map<size_t, SubcircuitR1CS> r1cs_cache;
        auto r1cs_of = [&](size_t id) -> const SubcircuitR1CS & {
            auto it = r1cs_cache.find(id);
            if (it == r1cs_cache.end()) {
                R1csBinary b = R1csBinary::read(in.qap_path + "/r1cs/subcircuit" + std::to_string(in.infos.at(id).id) + ".r1cs");
                it = r1cs_cache.emplace(id, SubcircuitR1CS::from_r1cs_sparse_only(b, sp, in.infos.at(id), in.n_consts.at(id))).first;
            }
            return it->second;
        };
        auto uvw = read_R1CS_gen_uvwXY(r1cs_of, in.pv, in.infos, sp);
        p->uXY = std::move(uvw[0]), p->vXY = std::move(uvw[1]), p->wXY = std::move(uvw[2]);
        p->rXY = Poly::zero();
        p->a_free_X = gen_a_free_X(in.a_pub_user, in.a_pub_block, sp);
        p->t_n = vanishing(n, true), p->t_mi = vanishing(m_i, true), p->t_smax = vanishing(s_max, false);
        auto s01 = permutation_to_poly(in.perm, m_i, s_max);
        p->s0XY = std::move(s01.first), p->s1XY = std::move(s01.second);
        p->timing["init.build"] = now() - t0;

        double t1 = now();
        const Mixer &mx = mixer;
        Binding b;
        b.A_free = sigma.sigma1.encode_poly(p->a_free_X, "A_free");
        b.O_pub_free = encode_O_pub_free(sigma.gamma_inv_o_inst, in.pv, in.infos);
        G1Affine O_mid_core = encode_O_mid_no_zk(sigma.eta_inv_li_o_inter_alpha4_kj, in.pv, in.infos, sp);
        G1Affine O_prv_core = encode_O_prv_no_zk(sigma.delta_inv_li_o_prv, in.pv, in.infos, sp);
        const auto &xh = sigma.delta_inv_alphak_xh_tx, &xj = sigma.delta_inv_alpha4_xj_tx, &yi = sigma.delta_inv_alphak_yi_ty;
        ScalarField zero{};
        std::vector<std::pair<ScalarField, G1Affine>> mid = {{fr_one(), O_mid_core}, {mx.rO_mid, sigma.delta}};
        while (mid.size() < 16) mid.push_back({zero, O_mid_core});
        std::vector<std::pair<ScalarField, G1Affine>> prv = {   // lib.rs:1146-1160
            {fr_one(), O_prv_core}, {fr_neg(mx.rO_mid), sigma.eta},
            {mx.rU_X, xh[0]}, {mx.rV_X, xh[3]}, {mx.rW_X[0], xh[6]}, {mx.rW_X[1], xh[7]}, {mx.rW_X[2], xh[8]},
            {mx.rB_X[0], xj[0]}, {mx.rB_X[1], xj[1]},
            {mx.rU_Y, yi[0]}, {mx.rV_Y, yi[3]}, {mx.rW_Y[0], yi[6]}, {mx.rW_Y[1], yi[7]}, {mx.rW_Y[2], yi[8]},
            {mx.rB_Y[0], yi[9]}, {mx.rB_Y[1], yi[10]}};
        auto both = g1_lincombs({mid, prv});
        b.O_mid = both[0], b.O_prv = both[1];
        p->timing["init.binding"] = now() - t1;
        p->timing["init.total"] = now() - t0;
        return {std::move(p), b};
    }

    // f = b + th0 s0 + th1 s1 + th2,  g = b + th0 X + th1 Y + th2 (lib.rs:1807-1811)
    std::pair<Poly, Poly> fg(const std::vector<ScalarField> &th) const {
        using namespace prover_detail;
        ScalarField zero{};
        Poly X_mono = Poly::from_coeffs(std::vector<ScalarField>{zero, fr_one()}, 2, 1);
        Poly Y_mono = Poly::from_coeffs(std::vector<ScalarField>{zero, fr_one()}, 1, 2);
        Poly f = Poly::lincomb({Term(fr_one(), &bXY), Term(th[0], &s0XY), Term(th[1], &s1XY)});
        Poly g = Poly::lincomb({Term(fr_one(), &bXY), Term(th[0], &X_mono), Term(th[1], &Y_mono)});
        add_const_in_place(f, th[2]);
        add_const_in_place(g, th[2]);
        return {std::move(f), std::move(g)};
    }
    // what prove2 and prove4 both need of r, f and g (lib.rs:1975-1990 and 2703-2725 recompute them): f, g, r(w_x^-1 X, Y),
    // r(w_x^-1 X, w_y^-1 Y) and the two differences r_D1, r_D2 — six m_I x s_max matrices, built once per (r, thetas)
    struct CopyOperands {
        std::vector<ScalarField> thetas;
        const void *r_data = nullptr;
        Poly f, g, r_omegaX, r_omegaX_omegaY, r_D1, r_D2;
    };
    std::unique_ptr<CopyOperands> copy_operands_;
    const CopyOperands &copy_operands(const std::vector<ScalarField> &thetas) {
        using namespace prover_detail;
        if (copy_operands_ && copy_operands_->r_data == (const void *)rXY.poly.ptr() && copy_operands_->thetas.size() == thetas.size() &&
            std::memcmp(copy_operands_->thetas.data(), thetas.data(), thetas.size() * sizeof(ScalarField)) == 0)
            return *copy_operands_;
        std::unique_ptr<CopyOperands> c(new CopyOperands());
        c->thetas = thetas, c->r_data = (const void *)rXY.poly.ptr();
        const ScalarField one = fr_one(), minus_one = fr_neg(one);
        const ScalarField w_inv_x = fr_inv(root_of_unity(m_i)), w_inv_y = fr_inv(root_of_unity(sp.s_max));
        auto f_g = fg(thetas);
        c->f = std::move(f_g.first), c->g = std::move(f_g.second);
        c->r_omegaX = rXY.scale_coeffs(&w_inv_x, nullptr);
        c->r_omegaX_omegaY = c->r_omegaX.scale_coeffs(nullptr, &w_inv_y);
        c->r_D1 = poly_comb({{one, &rXY}, {minus_one, &c->r_omegaX}});
        c->r_D2 = poly_comb({{one, &rXY}, {minus_one, &c->r_omegaX_omegaY}});
        copy_operands_ = std::move(c);
        return *copy_operands_;
    }
    Poly blinded_R() const {
        using namespace prover_detail;
        return poly_comb({{fr_one(), &rXY}, {mixer.rR_X, &t_mi}, {mixer.rR_Y, &t_smax}});
    }
    Poly blinded_V() const {
        using namespace prover_detail;
        return poly_comb({{fr_one(), &vXY}, {mixer.rV_X, &t_n}, {mixer.rV_Y, &t_smax}});
    }

    // prove0 (lib.rs:1446-1782)
    Proof0 prove0() {
        using namespace prover_detail;
        const Mixer &mx = mixer;
        size_t n = sp.n, s_max = sp.s_max;
        const bool from_evaluations = lagrange_n && lagrange_mi && u_ev.len() && v_ev.len() && w_ev.len() && b_ev.len();
        // the evaluation-basis commits need nothing this round computes: they start now, under the transforms of p0 and its division
        std::future<std::vector<G1Affine>> early;
        if (from_evaluations && can_commit_early())
            early = commit_early({lagrange_n->job_evals(u_ev, "U", n, s_max), lagrange_n->job_evals(v_ev, "V", n, s_max), lagrange_n->job_evals(w_ev, "W", n, s_max),
                                  lagrange_mi->job_evals(b_ev, "B", m_i, s_max)});
        Poly p0XY = uXY * vXY - wXY;
        auto q01 = p0XY.div_by_vanishing_opt((int64_t)n, (int64_t)s_max);
        q0XY = std::move(q01.first), q1XY = std::move(q01.second);
        Poly rW_X = Poly::from_coeffs(std::vector<ScalarField>(mx.rW_X.begin(), mx.rW_X.end()), 4, 1);
        Poly rW_Y = Poly::from_coeffs(std::vector<ScalarField>(mx.rW_Y.begin(), mx.rW_Y.end()), 1, 4);
        ScalarField one = fr_one(), minus_one = fr_neg(one);
        Poly Q_AX_XY = poly_comb({{one, &q0XY}, {mx.rU_X, &vXY}, {mx.rV_X, &uXY}, {minus_one, &rW_X}, {fr_mul(mx.rU_X, mx.rV_X), &t_n},
                                  {fr_mul(mx.rU_Y, mx.rV_X), &t_smax}});
        Poly Q_AY_XY = poly_comb({{one, &q1XY}, {mx.rU_Y, &vXY}, {mx.rV_Y, &uXY}, {minus_one, &rW_Y}, {fr_mul(mx.rU_X, mx.rV_Y), &t_n},
                                  {fr_mul(mx.rU_Y, mx.rV_Y), &t_smax}});
        if (from_evaluations) return prove0_from_evaluations(Q_AX_XY, Q_AY_XY, early);
        // the coefficient route commits the blinded polynomials themselves: (sum c_k X^k)(X^n - 1) + (sum c_k Y^k)(Y^s_max - 1) as matrices
        w_zk.reset(new Poly(low_degree_times_vanishing(mx.rW_X, n, true) + low_degree_times_vanishing(mx.rW_Y, s_max, false)));
        term_b_zk.reset(new Poly(low_degree_times_vanishing(mx.rB_X, m_i, true) + low_degree_times_vanishing(mx.rB_Y, s_max, false)));
        Poly UXY = poly_comb({{one, &uXY}, {mx.rU_X, &t_n}, {mx.rU_Y, &t_smax}});
        Poly VXY = blinded_V();
        Poly WXY = poly_comb({{one, &wXY}, {one, w_zk.get()}});
        Poly BXY = poly_comb({{one, &bXY}, {one, term_b_zk.get()}});
        auto c = sigma->sigma1.encode_polys({&UXY, &VXY, &WXY, &Q_AX_XY, &Q_AY_XY, &BXY}, {"U", "V", "W", "Q_AX", "Q_AY", "B"});
        return Proof0{c[0], c[1], c[2], c[3], c[4], c[5]};
    }
    // U, V, W, B from the evaluations (same points as the coefficient route above):
    //   commit(p + sum_k c_k T^k (T^e - 1)) = MSM(evaluations of p, Lagrange table) + sum_k c_k ([tau^(e+k)]G - [tau^k]G)
    // early (valid): the four evaluation commits were issued before the round's polynomial work and are collected here
    Proof0 prove0_from_evaluations(Poly &Q_AX_XY, Poly &Q_AY_XY, std::future<std::vector<G1Affine>> &early) {
        using namespace prover_detail;
        const size_t n = sp.n, s_max = sp.s_max;
        std::vector<G1Affine> c(6);
        if (early.valid()) {
            std::vector<G1Affine> q = Sigma1::run_jobs({sigma->sigma1.job(Q_AX_XY, "Q_AX"), sigma->sigma1.job(Q_AY_XY, "Q_AY")});
            std::vector<G1Affine> e = early.get();
            c = {e[0], e[1], e[2], q[0], q[1], e[3]};
        } else {
            c = Sigma1::run_jobs({lagrange_n->job_evals(u_ev, "U", n, s_max), lagrange_n->job_evals(v_ev, "V", n, s_max), lagrange_n->job_evals(w_ev, "W", n, s_max),
                                  sigma->sigma1.job(Q_AX_XY, "Q_AX"), sigma->sigma1.job(Q_AY_XY, "Q_AY"), lagrange_mi->job_evals(b_ev, "B", m_i, s_max)});
        }
        // the Lagrange tables carry the transform's 1 / N (tkmk_g1_scale at open): the MSM is the commitment of the unblinded polynomial
        const Blinds &bl = blinds();
        return Proof0{fqh::g1_affine_add(c[0], bl.U), fqh::g1_affine_add(c[1], bl.V), fqh::g1_affine_add(c[2], bl.W), c[3], c[4], fqh::g1_affine_add(c[5], bl.B)};
    }

    // prove1 (lib.rs:1784-1956)
    Proof1 prove1(const std::vector<ScalarField> &thetas) {
        const size_t s_max = sp.s_max;
        const DistCtx &dc = dist_ctx();
        const size_t lc = Poly::local_cols(s_max, false);   // columns of the m_I x s_max grid on this rank (all of them on one GPU)
        const size_t cells = m_i * lc;
        tkmk_vecops_config c = dev_cfg();
        DeviceVec<ScalarField> f_ev(cells), g_ev(cells), tr(cells + 1), sfx(cells + 1);
        const bool on_grid = b_ev.len() >= cells && s0_ev.len() >= cells && s1_ev.len() >= cells && s0_identity && s1_identity && m_i >= 2 && s_max >= 2;
        if (on_grid) {
            // f = b + th0 s0 + th1 s1 + th2 and g = b + th0 X + th1 Y + th2 evaluated where their operands already are evaluations
            auto combine = [&](const DeviceVec<ScalarField> &a0, const DeviceVec<ScalarField> &a1, DeviceVec<ScalarField> &out) {
                const ScalarField coef[3] = {fr_one(), thetas[0], thetas[1]};
                const tkmk_fr *ptr[3] = {b_ev.ptr(), a0.ptr(), a1.ptr()};
                const uint32_t xs[3] = {(uint32_t)m_i, (uint32_t)m_i, (uint32_t)m_i}, ys[3] = {(uint32_t)lc, (uint32_t)lc, (uint32_t)lc};
                check(tkmk_poly_lincomb(3, coef, ptr, xs, ys, nullptr, nullptr, out.ptr(), (uint32_t)m_i, (uint32_t)lc, nullptr), "tkmk_poly_lincomb");
                tkmk_vecops_config sc = dev_cfg();
                sc.is_a_on_device = false;
                check(bls12_381_scalar_add_vec(&thetas[2], out.ptr(), cells, &sc, out.ptr()), "scalar_add");
            };
            combine(s0_ev, s1_ev, f_ev);
            combine(*s0_identity, *s1_identity, g_ev);
        } else {
            if (dc.on()) throw Error("prove1: a sharded prover needs the evaluation tables of a resident context");
            auto f_g = fg(thetas);
            f_g.first.resize(m_i, s_max);
            f_g.second.resize(m_i, s_max);
            f_g.first.to_rou_evals(nullptr, nullptr, f_ev);
            f_g.second.to_rou_evals(nullptr, nullptr, g_ev);
        }
        // r[last] = 1, r[idx] = r[idx + 1] * (g / f)[idx + 1] over the TRANSPOSED (s_max x m_i) order (lib.rs:1858-1866): the walk goes down
        // column 0, then column 1, ...
        check(bls12_381_vector_div(g_ev.ptr(), f_ev.ptr(), cells, &c, g_ev.ptr()), "vector_div");
        check(bls12_381_matrix_transpose(g_ev.ptr(), (uint32_t)m_i, (uint32_t)lc, &c, tr.ptr()), "transpose");
        std::vector<ScalarField> next_top;   // sharded: r at the top of the column that FOLLOWS each local column in the walk
        if (!dc.on()) {
            check(tkmk_vec_suffix_product(tr.ptr(), cells, sfx.ptr(), nullptr), "tkmk_vec_suffix_product");
        } else {
            // My columns are c(k) = rank + G k.  The running product at (k, row) is the product of the rest of column k and of ALL later
            // columns, mine and the other ranks'.  The column totals of every rank are gathered (s_max values); the other ranks' columns
            // between c(k) and c(k + 1) enter as ONE factor O_k multiplied onto the first element of my next column, those after my last
            // column as one element appended to the vector — and the ordinary scan over my columns gives the global running product.
            const uint32_t G = dc.G(), r = dc.r();
            DeviceVec<ScalarField> totals(lc);
            tkmk_vecops_config pc = dev_cfg();
            pc.batch_size = (int)lc;
            check(bls12_381_vector_product(tr.ptr(), m_i, &pc, totals.ptr()), "vector_product");
            std::vector<ScalarField> mine = totals.to_host(), all((size_t)G * lc);
            check(dc.all_gather_host(dc.comm, mine.data(), lc * sizeof(ScalarField), all.data()), "tkmk_comm_all_gather_host");
            auto total_of = [&](size_t col) { return all[(col % G) * lc + col / G]; };
            std::vector<ScalarField> others(lc, fr_one());   // O_k
            for (size_t k = 0; k < lc; k++)
                for (size_t col = r + G * k + 1; col < std::min<size_t>(r + G * (k + 1), s_max); col++) others[k] = fr_mul(others[k], total_of(col));
            if (lc > 1) {
                std::vector<ScalarField> firsts(lc - 1);
                DeviceVec<ScalarField> compact(lc - 1);
                check(tkmk_memcpy_2d_d2d(compact.ptr(), sizeof(ScalarField), tr.ptr() + m_i, m_i * sizeof(ScalarField), sizeof(ScalarField), lc - 1), "gather column heads");
                compact.copy_to_host(firsts.data(), lc - 1);
                for (size_t k = 1; k < lc; k++) firsts[k - 1] = fr_mul(firsts[k - 1], others[k - 1]);
                compact.copy_from_host(firsts.data(), lc - 1);
                check(tkmk_memcpy_2d_d2d(tr.ptr() + m_i, m_i * sizeof(ScalarField), compact.ptr(), sizeof(ScalarField), sizeof(ScalarField), lc - 1), "scatter column heads");
            }
            check(tkmk_memcpy_h2d(tr.ptr() + cells, &others[lc - 1], sizeof(ScalarField)), "append");
            check(tkmk_vec_suffix_product(tr.ptr(), cells + 1, sfx.ptr(), nullptr), "tkmk_vec_suffix_product");
            // the value at the top of every column, for the jumps across column ends below
            DeviceVec<ScalarField> tops(lc);
            check(tkmk_memcpy_2d_d2d(tops.ptr(), sizeof(ScalarField), sfx.ptr(), m_i * sizeof(ScalarField), sizeof(ScalarField), lc), "gather column tops");
            std::vector<ScalarField> my_tops = tops.to_host(), all_tops((size_t)G * lc);
            check(dc.all_gather_host(dc.comm, my_tops.data(), lc * sizeof(ScalarField), all_tops.data()), "tkmk_comm_all_gather_host");
            next_top.assign(lc, ScalarField{});
            for (size_t k = 0; k < lc; k++) {
                const size_t col = r + G * k + 1;
                if (col < s_max) next_top[k] = all_tops[(col % G) * lc + col / G];
            }
        }
        {
            DeviceVec<ScalarField> r_ev(cells);
            check(bls12_381_matrix_transpose(sfx.ptr(), (uint32_t)lc, (uint32_t)m_i, &c, r_ev.ptr()), "transpose");
            rXY = Poly::from_rou_evals_cols(r_ev, m_i, s_max);
        }
        if (lagrange_mi_prefix && m_i * s_max >= 2) {
            // jumps of r along the walk: d_j = r_j - r_{j+1} (d_last = r_last), zero wherever g / f = 1
            DeviceVec<ScalarField> d(cells);
            if (cells > 1) check(bls12_381_vector_sub(sfx.ptr(), sfx.ptr() + 1, cells - 1, &c, d.ptr()), "vector_sub");
            check(tkmk_memcpy_d2d(d.ptr() + (cells - 1), sfx.ptr() + (cells - 1), sizeof(ScalarField)), "memcpy");
            if (dc.on()) {   // the successor of a column's last cell is the top of the NEXT GLOBAL column, another rank's
                DeviceVec<ScalarField> compact(lc);
                check(tkmk_memcpy_2d_d2d(compact.ptr(), sizeof(ScalarField), sfx.ptr() + (m_i - 1), m_i * sizeof(ScalarField), sizeof(ScalarField), lc), "gather column ends");
                std::vector<ScalarField> ends = compact.to_host();
                for (size_t k = 0; k < lc; k++) ends[k] = fr_sub(ends[k], next_top[k]);
                compact.copy_from_host(ends.data(), lc);
                check(tkmk_memcpy_2d_d2d(d.ptr() + (m_i - 1), m_i * sizeof(ScalarField), compact.ptr(), sizeof(ScalarField), sizeof(ScalarField), lc), "scatter column ends");
            }
            G1Affine core = Sigma1::run_jobs({lagrange_mi_prefix->job_evals(d, "R", m_i * s_max, 1)})[0];
            return Proof1{fqh::g1_affine_add(core, blinds().R)};
        }
        Poly RXY = blinded_R();
        return Proof1{sigma->sigma1.encode_poly(RXY, "R")};
    }

    // prove2 (lib.rs:1958-2270)
    Proof2 prove2(const std::vector<ScalarField> &thetas, const ScalarField &kappa0) {
        using namespace prover_detail;
        const Mixer &mx = mixer;
        size_t s_max = sp.s_max;
        ScalarField kappa0_sq = fr_mul(kappa0, kappa0), one = fr_one(), inv_m_i = fr_inv(fr_from_u32((uint32_t)m_i));
        const CopyOperands &co = copy_operands(thetas);   // kept for prove4
        const Poly &fXY = co.f, &gXY = co.g;
        const LagrangePolys &lg = lagrange_polys();
        const Poly &K_last = lg.K_last, &L_last = lg.L_last, &KL = lg.KL, &K0 = lg.K0;

        // the reference's expression tree (lib.rs:2107-2171) with three operands in cheaper but equal forms: r(w^-1 X, Y) and
        // r(w^-1 X, w^-1 Y) as root-shifted leaves of r (rotations of r's evaluations instead of two more 2^25-point transforms),
        // KL as the product of its X-only and Y-only factors, K0 as an X-only leaf (1-D transforms, broadcast in the evaluator)
        using E = PolyExpr;
        auto r_g = [&]() { return E::mul(E::poly(rXY), E::poly(gXY)); };
        E p1 = E::mul(E::sub(E::poly(rXY), E::scalar(one)), E::mul(E::poly(K_last), E::poly(L_last)));
        E p2 = E::mul_x_minus_one(E::sub(r_g(), E::mul(E::poly_root_shifted(rXY, m_i, 0), E::poly(fXY))));
        // p3 = K0 * h, h = r g - r(w^-1 X, w^-1 Y) f: only K0 pushes the X-degree past 2 m_I.  The reference evaluates everything on the
        // 4 m_I x 2 s_max domain (lib.rs:2160-2171); here p1 + kappa0 p2 and h are evaluated on 2 m_I x 2 s_max (half the transform
        // and pointwise work), and K0 * h is the sliding-window sum of h's coefficients.  Same polynomial, coefficient for coefficient.
        E h_expr = E::sub(r_g(), E::mul(E::poly_root_shifted(rXY, m_i, s_max), E::poly(fXY)));
        std::vector<std::pair<ScalarField, E>> terms;
        terms.emplace_back(one, std::move(p1));
        terms.emplace_back(kappa0, std::move(p2));
        E::LeafCache leaves;
        Poly p12 = E::weighted_sum(std::move(terms)).evaluate_fused_with_domain(2 * m_i, 2 * s_max, &leaves);
        Poly hXY = h_expr.evaluate_fused_with_domain(2 * m_i, 2 * s_max, &leaves);
        leaves.clear();
        Poly k0_h = hXY.mul_ones_x(m_i, inv_m_i);
        Poly p_comb = Poly::lincomb({Term(one, &p12), Term(kappa0_sq, &k0_h)});
        auto q23 = p_comb.div_by_vanishing_opt((int64_t)m_i, (int64_t)s_max);
        q2XY = std::move(q23.first), q3XY = std::move(q23.second);
        ScalarField minus_one = fr_neg(one);
        const Poly &r_D1 = co.r_D1, &r_D2 = co.r_D2;
        Poly g_D = poly_comb({{one, &gXY}, {minus_one, &fXY}});

        auto q_c = [&](const Poly &quot, const std::array<ScalarField, 2> &rB, const ScalarField &rR, bool x_axis) {
            // d_comb = (rB[0] + rB[1] T) * r_D + rR * g_D with T = X or Y (mul_by_linear, lib.rs:80-94), in one pass each
            auto d_comb = [&](const Poly &r_D) {
                return Poly::lincomb({Term(rB[0], &r_D), x_axis ? Term(rB[1], &r_D, 1, 0) : Term(rB[1], &r_D, 0, 1), Term(rR, &g_D)});
            };
            Poly d1_comb = d_comb(r_D1), d2_comb = d_comb(r_D2);
            Poly k0_d2 = d2_comb.mul_ones_x(m_i, inv_m_i);   // K0 * d2_comb: K0 = (1/m_I)(1 + X + ... + X^(m_I - 1))
            // kappa0 * (X - 1) * d1_comb enters as two shifted terms
            return Poly::lincomb({Term(one, &quot), Term(rR, &KL), Term(kappa0, &d1_comb, 1, 0), Term(fr_neg(kappa0), &d1_comb), Term(kappa0_sq, &k0_d2)});
        };
        Poly Q_CX_XY = q_c(q2XY, mx.rB_X, mx.rR_X, true);
        Poly Q_CY_XY = q_c(q3XY, mx.rB_Y, mx.rR_Y, false);
        auto c = sigma->sigma1.encode_polys({&Q_CX_XY, &Q_CY_XY}, {"Q_CX", "Q_CY"});
        return Proof2{c[0], c[1]};
    }

    // prove3 (lib.rs:2272-2354): V(chi, zeta), R(chi, zeta), R(chi / w_x, zeta), R(chi / w_x, zeta / w_y) with V = v + rV_X t_n + rV_Y t_smax
    // and R = r + rR_X t_mI + rR_Y t_smax.  The reference forms V, R, R(w_x^-1 X, Y) and R(w_x^-1 X, w_y^-1 Y) as coefficient matrices
    // (2 m_I x 2 s_max each) and evaluates those; a coefficient scaling followed by an evaluation is the evaluation at the scaled point, and
    // the blinding terms are two-coefficient polynomials — so four evaluations of the m_I x s_max matrices v and r plus closed forms give
    // the same four field elements.
    Proof3 prove3(const ScalarField &chi, const ScalarField &zeta) const {
        const ScalarField one = fr_one();
        const ScalarField w_inv_x = fr_inv(root_of_unity(m_i)), w_inv_y = fr_inv(root_of_unity(sp.s_max));
        const ScalarField chi_w = fr_mul(chi, w_inv_x), zeta_w = fr_mul(zeta, w_inv_y);
        auto vanish = [&](const ScalarField &at, size_t size) { return fr_sub(fr_pow(at, size), one); };
        auto R_at = [&](const ScalarField &x, const ScalarField &y) {
            return fr_add(rXY.eval(x, y), fr_add(fr_mul(mixer.rR_X, vanish(x, m_i)), fr_mul(mixer.rR_Y, vanish(y, sp.s_max))));
        };
        Proof3 out;
        out.V_eval = fr_add(vXY.eval(chi, zeta), fr_add(fr_mul(mixer.rV_X, vanish(chi, sp.n)), fr_mul(mixer.rV_Y, vanish(zeta, sp.s_max))));
        out.R_eval = R_at(chi, zeta);
        out.R_omegaX_eval = R_at(chi_w, zeta);
        out.R_omegaX_omegaY_eval = R_at(chi_w, zeta_w);
        return out;
    }

    // prove4 (lib.rs:2356-3206)
    // test_parts = true also commits Pi_A, Pi_C, Pi_B one by one (Proof4Test: what the reference's testing-mode verifier looks at);
    // the proof itself carries only their sums
    std::pair<Proof4, Proof4Test> prove4(const Proof3 &proof3, const std::vector<ScalarField> &thetas, const ScalarField &kappa0,
                                        const ScalarField &chi, const ScalarField &zeta, const ScalarField &kappa1, bool test_parts = false) {
        using namespace prover_detail;
        const Mixer &mx = mixer;
        size_t n = sp.n, s_max = sp.s_max;
        ScalarField one = fr_one(), minus_one = fr_neg(one);
        auto ev = [&](const Poly &p) { return p.eval(chi, zeta); };

        // M, N: openings of R at (chi / w_x, zeta) and (chi / w_x, zeta / w_y) (lib.rs:2534-2701).  First in this round (the reference has
        // them after Pi_A; they depend on R and the round-3 openings only): their commitments can then run under everything that follows.
        ScalarField w_inv_x = fr_inv(root_of_unity(m_i)), w_inv_y = fr_inv(root_of_unity(s_max));
        const ScalarField chi_w = fr_mul(w_inv_x, chi), zeta_w = fr_mul(w_inv_y, zeta);
        // R = r + rR_X t_mI + rR_Y t_smax and V = v + rV_X t_n + rV_Y t_smax enter this round only inside sums: they stay symbolic (Lin) and
        // are never stored as the 2 m_I x 2 s_max matrices their blinding terms would make of them
        Lin R_lin, V_lin;
        R_lin.add(one, rXY).add(mx.rR_X, t_mi).add(mx.rR_Y, t_smax);
        V_lin.add(one, vXY).add(mx.rV_X, t_n).add(mx.rV_Y, t_smax);
        auto minus_constant = [&](const Lin &p, const Poly &minus_c) {   // p - c as one pass (minus_c: the 1 x 1 polynomial -c)
            Lin d = p;
            return d.add(one, minus_c).materialize();
        };
        const Poly m_R_wx = constant_poly(fr_neg(proof3.R_omegaX_eval)), m_R_wxy = constant_poly(fr_neg(proof3.R_omegaX_omegaY_eval));
        const Poly m_R = constant_poly(fr_neg(proof3.R_eval)), m_V = constant_poly(fr_neg(proof3.V_eval));
        auto M = minus_constant(R_lin, m_R_wx).div_by_ruffini(chi_w, zeta);
        auto N = minus_constant(R_lin, m_R_wxy).div_by_ruffini(chi_w, zeta_w);
        std::future<std::vector<G1Affine>> early;   // declared after M and N: it ends (and is waited for) before the polynomials its jobs read
        if (!test_parts && can_commit_early())
            early = commit_early({sigma->sigma1.job(std::get<0>(M), "M_X"), sigma->sigma1.job(std::get<1>(M), "M_Y"), sigma->sigma1.job(std::get<1>(N), "N_Y")}, true);

        // Pi_A: arithmetic constraints + the opening of V (lib.rs:2383-2532); t_n(chi) = chi^n - 1, t_smax(zeta) = zeta^s_max - 1
        const ScalarField t_n_eval = fr_sub(fr_pow(chi, n), one), t_smax_eval = fr_sub(fr_pow(zeta, s_max), one), small_v_eval = ev(vXY);
        Poly rW_X = Poly::from_coeffs(std::vector<ScalarField>(mx.rW_X.begin(), mx.rW_X.end()), 4, 1);
        Poly rW_Y = Poly::from_coeffs(std::vector<ScalarField>(mx.rW_Y.begin(), mx.rW_Y.end()), 1, 4);
        if (!w_zk_x) {
            w_zk_x.reset(new Poly(low_degree_times_vanishing(mx.rW_X, n, true)));
            w_zk_y.reset(new Poly(low_degree_times_vanishing(mx.rW_Y, s_max, false)));
        }
        Lin pA;   // lib.rs:2440-2500, one pass over u, v, w, q0, q1 and a handful of rows and columns
        pA.add(kappa1, V_lin).add(kappa1, m_V)
            .add(small_v_eval, uXY).add(minus_one, wXY).add(fr_neg(t_n_eval), q0XY).add(fr_neg(t_smax_eval), q1XY)
            .add(fr_mul(small_v_eval, mx.rU_X), t_n).add(fr_mul(small_v_eval, mx.rU_Y), t_smax)
            .add(fr_neg(fr_add(fr_mul(mx.rU_X, t_n_eval), fr_mul(mx.rU_Y, t_smax_eval))), vXY)
            .add(t_n_eval, rW_X).add(t_smax_eval, rW_Y).add(minus_one, *w_zk_x).add(minus_one, *w_zk_y);
        auto piA = pA.materialize().div_by_ruffini(chi, zeta);

        // Pi_C: copy constraints (lib.rs:2703-3130).  The reference builds pC, term5, term6, term10, the two mul_by_term9 products, LHS_zk1,
        // LHS_zk2, R - R(chi, zeta) and their weighted sum as ten matrices; only K0 * (...) needs its operand as a matrix (a product with a
        // polynomial, not a shift), everything else is one combination of g, f, q2, q3, KL, r_D1, r and a few rows and columns.
        const CopyOperands &co = copy_operands(thetas);   // prove2 left them
        const Poly &fXY = co.f, &gXY = co.g;
        ScalarField t_mi_eval = fr_sub(fr_pow(chi, m_i), one), t_s_max_eval = t_smax_eval;
        const LagrangePolys &lg = lagrange_polys();
        const Poly &K0 = lg.K0;
        // r(w_x^-1 X, Y) at (chi, zeta) is r at (chi / w_x, zeta): the evaluations need no scaled copy
        const ScalarField K0_eval = ev(K0), small_r = ev(rXY), small_r_wx = rXY.eval(chi_w, zeta), small_r_wxy = rXY.eval(chi_w, zeta_w);
        const ScalarField chi_m1 = fr_sub(chi, one), kappa0_sq = fr_mul(kappa0, kappa0);
        // r_D1 = r - r(w_x^-1 X, Y), r_D2 = r - r(w_x^-1 X, w_y^-1 Y) stay matrices (m_I x s_max): each is read under three shifts, and a
        // Y shift of a matrix is one ring exchange of it in the sharded prover — of one operand rather than of two
        const Poly &r_D1 = co.r_D1, &r_D2 = co.r_D2;
        const ScalarField r_D1_eval = fr_sub(small_r, small_r_wx), r_D2_eval = fr_sub(small_r, small_r_wxy);   // evaluation is linear
        if (!b_zk_x) {
            b_zk_x.reset(new Poly(low_degree_times_vanishing(mx.rB_X, m_i, true)));
            b_zk_y.reset(new Poly(low_degree_times_vanishing(mx.rB_Y, s_max, false)));
        }
        const ScalarField c10 = fr_add(fr_mul(mx.rR_X, t_mi_eval), fr_mul(mx.rR_Y, t_s_max_eval));
        Lin term5, term6, term10, b_zk, pC, r_d1_t, r_d2_lin, LHS_zk1, LHS_zk2, LHS_for_copy;
        term5.add(small_r, gXY).add(fr_neg(small_r_wx), fXY);
        term6.add(small_r, gXY).add(fr_neg(small_r_wxy), fXY);
        term10.add(c10, gXY).add(fr_neg(c10), fXY);   // (rR_X t_mI(chi) + rR_Y t_smax(zeta)) * (g - f)
        b_zk.add(one, *b_zk_x).add(one, *b_zk_y);
        pC.add(fr_sub(small_r, one), lg.KL).add(fr_mul(kappa0, chi_m1), term5).add(fr_mul(kappa0_sq, K0_eval), term6)
            .add(fr_neg(t_mi_eval), q2XY).add(fr_neg(t_s_max_eval), q3XY);
        r_d1_t.add(term9(r_D1, mx.rB_X, mx.rB_Y, t_mi_eval, t_s_max_eval)).add(one, term10);   // mul_by_term9(r_D1) + term10 (lib.rs:96-124)
        LHS_zk1.add(fr_mul(chi_m1, r_D1_eval), b_zk).add(one, r_d1_t).add(minus_one, r_d1_t, 1, 0).add(chi_m1, term10);   // (1 - X) r_d1_t: two shifts
        r_d2_lin.add(term9(r_D2, mx.rB_X, mx.rB_Y, t_mi_eval, t_s_max_eval)).add(one, term10);
        Poly r_d2_t = r_d2_lin.materialize();
        Poly k0_r_d2_t = r_d2_t.mul_ones_x(m_i, fr_inv(fr_from_u32((uint32_t)m_i)));   // K0 * r_d2_t
        LHS_zk2.add(fr_mul(K0_eval, r_D2_eval), b_zk).add(K0_eval, term10).add(minus_one, k0_r_d2_t);
        const ScalarField k1_2 = fr_mul(kappa1, kappa1);
        LHS_for_copy.add(k1_2, pC).add(fr_mul(k1_2, kappa0), LHS_zk1).add(fr_mul(fr_mul(k1_2, kappa0), kappa0), LHS_zk2)
            .add(fr_mul(k1_2, kappa1), R_lin).add(fr_mul(k1_2, kappa1), m_R);
        auto piC = LHS_for_copy.materialize().div_by_ruffini(chi, zeta);

        // Pi_B: opening of a_free (lib.rs:3137-3181)
        ScalarField A_eval = ev(a_free_X);
        auto piB = sub_const(a_free_X, A_eval).div_by_ruffini(chi, zeta);

        // N_X is M_X: both numerators are R minus a constant and both are divided by the same X - chi / w_x first, and a constant
        // only moves the remainder of that division, never its quotient (the reference commits the same polynomial twice,
        // lib.rs:2572-2700).  One 2^22-point commitment instead of two; the Y-quotients differ and are both committed.
        ScalarField k1_4 = fr_mul(k1_2, k1_2), zero{};
        if (!test_parts) {
            // Pi_X = Pi_AX + Pi_CX + kappa1^4 Pi_B and Pi_Y = Pi_AY + Pi_CY are the proof's entries (lib.rs:3180-3184): the commitment is
            // linear, so the three quotient polynomials are added first (one pass) and committed ONCE — the reference commits Pi_AX
            // (2^23 coefficients), Pi_CX (2^24) and Pi_B apart and adds the points
            Poly pi_x = Poly::lincomb({Term(one, &std::get<0>(piA)), Term(one, &std::get<0>(piC)), Term(k1_4, &std::get<0>(piB))});
            Poly pi_y = Poly::lincomb({Term(one, &std::get<1>(piA)), Term(one, &std::get<1>(piC))});
            if (early.valid()) {
                auto cc = sigma->sigma1.encode_polys({&pi_x, &pi_y}, {"Pi_X", "Pi_Y"});
                std::vector<G1Affine> e = early.get();
                return {Proof4{cc[0], cc[1], e[0], e[1], e[0], e[2]}, Proof4Test{}};
            }
            auto cc = sigma->sigma1.encode_polys({&pi_x, &pi_y, &std::get<0>(M), &std::get<1>(M), &std::get<1>(N)}, {"Pi_X", "Pi_Y", "M_X", "M_Y", "N_Y"});
            return {Proof4{cc[0], cc[1], cc[2], cc[3], cc[2], cc[4]}, Proof4Test{}};
        }
        // the reference's own commit list (lib.rs:2572-3184); N_X is committed too here so that its box is on record next to M_X's
        auto c = sigma->sigma1.encode_polys({&std::get<0>(piA), &std::get<1>(piA), &std::get<0>(M), &std::get<1>(M), &std::get<1>(N),
                                             &std::get<0>(piC), &std::get<1>(piC), &std::get<0>(piB), &std::get<0>(N)},
                                            {"Pi_AX", "Pi_AY", "M_X", "M_Y", "N_Y", "Pi_CX", "Pi_CY", "Pi_B", "N_X"});
        if (std::memcmp(&c[8], &c[2], sizeof(G1Affine)) != 0) throw Error("prove4: N_X differs from M_X");   // the identity the fast path relies on
        const G1Affine &Pi_AX = c[0], &Pi_AY = c[1], &M_X = c[2], &M_Y = c[3], &N_X = c[2], &N_Y = c[4], &Pi_CX = c[5], &Pi_CY = c[6], &Pi_B0 = c[7];
        auto sums = g1_lincombs({{{k1_4, Pi_B0}, {zero, Pi_B0}, {zero, Pi_B0}},   // encode(pi_B) * kappa1^4 (lib.rs:3180)
                                 {{one, Pi_AX}, {one, Pi_CX}, {k1_4, Pi_B0}},      // lib.rs:3183-3184
                                 {{one, Pi_AY}, {one, Pi_CY}, {zero, Pi_AY}}});
        Proof4 p4{sums[1], sums[2], M_X, M_Y, N_X, N_Y};
        Proof4Test t{Pi_CX, Pi_CY, Pi_AX, Pi_AY, sums[0], M_X, M_Y, N_X, N_Y};
        return {p4, t};
    }
};

inline std::shared_ptr<const Prover::LagrangePolys> Prover::LagrangePolys::make(size_t m_i, size_t s_max) {
    using namespace prover_detail;
    Poly K_last = unit_evals(m_i, m_i - 1, true), L_last = unit_evals(s_max, s_max - 1, false), K0 = unit_evals(m_i, 0, true);
    Poly KL = K_last * L_last;
    return std::make_shared<const LagrangePolys>(LagrangePolys{std::move(K_last), std::move(L_last), std::move(K0), std::move(KL)});
}

// the round loop of prove/src/main.rs:47-76
inline Proof run_rounds(Prover &prover, const Binding &binding, std::map<std::string, double> *times = nullptr, bool test_parts = false) {
    TranscriptManager manager;
    auto timed = [&](const char *name, const std::function<void()> &fn) {
        double t = Prover::now();
        host_trace("== %s", name);
        fn();
        check(tkmk_device_synchronize(), "synchronize");
        if (times) (*times)[name] = Prover::now() - t;
    };
    Proof proof;
    proof.binding = binding;
    timed("prove0", [&] { proof.proof0 = prover.prove0(); });
    const Proof0 &p0 = proof.proof0;
    manager.add_proof0(p0.U, p0.V, p0.W, p0.Q_AX, p0.Q_AY, p0.B);
    std::vector<ScalarField> thetas = manager.get_thetas();
    timed("prove1", [&] { proof.proof1 = prover.prove1(thetas); });
    manager.add_proof1(proof.proof1.R);
    ScalarField kappa0 = manager.get_kappa0();
    timed("prove2", [&] { proof.proof2 = prover.prove2(thetas, kappa0); });
    manager.add_proof2(proof.proof2.Q_CX, proof.proof2.Q_CY);
    auto cz = manager.get_chi_zeta();
    timed("prove3", [&] { proof.proof3 = prover.prove3(cz.first, cz.second); });
    const Proof3 &p3 = proof.proof3;
    manager.add_proof3(p3.V_eval, p3.R_eval, p3.R_omegaX_eval, p3.R_omegaX_omegaY_eval);
    ScalarField kappa1 = manager.get_kappa1();
    timed("prove4", [&] { proof.proof4 = prover.prove4(p3, thetas, kappa0, cz.first, cz.second, kappa1, test_parts).first; });
    return proof;
}

}  // namespace tkmk
