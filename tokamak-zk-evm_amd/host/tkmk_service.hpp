// tkmk_service.hpp — the prover as a resident service: everything that does not change between proofs of one circuit is loaded
// ONCE (ProverContext::open), and a proof is one call (ProverContext::prove) from the synthesizer's files to proof.json.
//
// The reference runs one process per proof (prove/src/main.rs:27-97) and pays for the circuit-static part every time inside
// Prover::init (prove/src/lib.rs:675-1206): setupParams.json / subcircuitInfo.json, the .r1cs of every used subcircuit
// (libs/src/iotools/mod.rs:1287-1420), the mmap + validation of combined_sigma.rkyv (prove/src/sigma_source.rs:22-32).  Here these
// live in the context: the constraint library as device-resident CSR (tkmk_r1cs_library), the CRS in HBM in the MSM's resident
// form, the static wire lists that route witness values to b(X,Y) and to the binding commitments, the identity part of s0 / s1,
// and the NTT domain.  Per proof the host only parses the three synthesizer documents (placementVariables.json on all cores,
// tkmk_fastparse.hpp) and uploads the witness once; every loop over placements x wires of the reference's init
// (gen_bXY, read_R1CS_gen_uvwXY, encode_O_pub_free / O_mid / O_prv, Permutation::to_poly) is a device launch over that copy.
// Results are those of Prover::init + prove0..prove4 of tkmk_prover.hpp on the same inputs and blinding scalars, bit for bit
// (tests/test_gpu_service.py).  `bin/prove` is open + prove + exit on the same code.
#pragma once
#include <sys/stat.h>

#include <fstream>
#include <future>

#include "tkmk_fastparse.hpp"
#include "tkmk_inputs.hpp"
#include "tkmk_prover.hpp"

namespace tkmk {

// One proof over the G GPUs of a node (include/tkmk_prover.h tkmk_prover_open_sharded; no reference counterpart: the
// reference is single-device).  Every rank holds 1 / G of every commit table AND of every polynomial: coefficient matrices live in the
// COLS layout (rank r: the columns iy = r mod G), evaluations on the large domains in the ROWS layout (host/tkmk_host.hpp DistCtx,
// include/tkmk_dist.h), a transform crosses between them with one all-to-all.  The witness side divides by placement (a placement IS a
// column of u, v, w, b, s0, s1); a rank commits its columns of every polynomial against its columns of the table, and the shares of a
// round's commitments meet in ONE all-gather.  Replicated: the host's parse of the three documents, the Fiat-Shamir transcript (every rank
// derives the same challenges from the same commitments, so the rounds need no message) and the few small polynomials built from host
// values.  The entries of libtkmk_dist.so arrive as function pointers (the prover library does not link RCCL).
struct ShardLink {
    void *comm = nullptr;
    Shard shard;
    tkmk_error (*multi_ex_sharded)(void *, const tkmk_msm_job_ex *, int, const tkmk_msm_config *, int, tkmk_g1_projective *) = nullptr;
    tkmk_error (*broadcast_host)(void *, void *, size_t, int) = nullptr;
    tkmk_error (*device_turn)(void *, int) = nullptr;
    tkmk_error (*all_gather_host)(void *, const void *, size_t, void *) = nullptr;
    tkmk_error (*agree)(void *, tkmk_error) = nullptr;
    tkmk_error (*abort)(void *) = nullptr;
    tkmk_error (*fwd_cols_to_rows)(void *, const tkmk_fr *, size_t, size_t, size_t, size_t, int, tkmk_fr *) = nullptr;
    tkmk_error (*inv_rows_to_cols)(void *, tkmk_fr *, size_t, size_t, int, tkmk_fr *) = nullptr;
    tkmk_error (*rows_rotate)(void *, const tkmk_fr *, size_t, size_t, size_t, tkmk_fr *) = nullptr;
    tkmk_error (*ring_shift)(void *, const void *, size_t, int, void *) = nullptr;
    tkmk_error (*relayout_cols_to_rows)(void *, const void *, size_t, size_t, size_t, void *) = nullptr;
    tkmk_error (*relayout_rows_to_cols)(void *, const void *, size_t, size_t, size_t, void *) = nullptr;
    explicit operator bool() const { return comm != nullptr; }
};
// for the span of one call on a sharded context: this thread's polynomials are distributed (dist_ctx()), its commit batches go through the
// communicator, and (loopback transport only) this virtual rank holds the device turn except while it waits for its peers
struct ShardSpan {
    const ShardLink &link;
    explicit ShardSpan(const ShardLink &l) : link(l) {
        if (!link) return;
        check(link.device_turn(link.comm, 1), "tkmk_comm_device_turn");
        commit_comm() = CommitComm{link.comm, link.multi_ex_sharded};
        DistCtx &dc = dist_ctx();
        dc.comm = link.comm, dc.shard = link.shard;
        dc.all_gather_host = link.all_gather_host, dc.fwd_cols_to_rows = link.fwd_cols_to_rows, dc.inv_rows_to_cols = link.inv_rows_to_cols;
        dc.rows_rotate = link.rows_rotate, dc.ring_shift = link.ring_shift;
    }
    ~ShardSpan() {
        if (!link) return;
        dist_ctx() = DistCtx{};
        commit_comm() = CommitComm{};
        (void)link.device_turn(link.comm, 0);
    }
};

// The four binding commitments of Prover::init (A_free, O_pub_free, O_mid, O_prv: lib.rs:1086-1160) enter the proof document but not the
// transcript: no round waits for them.  On a single-GPU context their MSM batch is issued from a helper thread on its own stream as
// soon as the routed witness lists exist, and collected after prove4: its sort and tail kernels (latency-bound, a few ms) then run under
// prove0's transforms and streaming passes instead of in front of them.  Members are destroyed in reverse order: the future first
// (it waits for the batch), the operand buffers after it.
struct PendingBinding {
    DeviceVec<ScalarField> mid_sc, prv_sc, pub_sc;
    DeviceVec<uint32_t> mid_ix, prv_ix, pub_ix;
    std::future<std::vector<G1Affine>> cores;   // A_free, O_pub_free, O_mid_core, O_prv_core
};

struct ProveTiming {   // seconds
    double parse = 0, upload = 0, build = 0, binding = 0, init = 0;
    double prove[5] = {0, 0, 0, 0, 0};
    double write = 0, total = 0;
};

class ProverContext {
  public:
    SetupParams sp{};
    size_t m_i = 0;
    std::vector<SubcircuitInfo> infos;
    std::vector<size_t> n_consts;
    std::string lib_dir;
    unsigned threads = 1;
    std::unique_ptr<ProverSigma> sigma;   // tables in the MSM's resident form (see open)
    std::string crs_source;               // which container the CRS came from
    ShardLink link;                       // set: this context is rank link.shard.rank of a sharded prover

  private:
    tkmk_r1cs_library *lib_ = nullptr;
    // static (local wire, row) lists per subcircuit kind, concatenated; kind k owns [first[k], first[k + 1])
    struct WireLists {
        std::vector<uint32_t> first;
        DeviceVec<uint32_t> wire, row;
        uint32_t count(size_t k) const { return first[k + 1] - first[k]; }
    };
    WireLists iface_, prv_, pub_;
    DeviceVec<ScalarField> s0_identity_, s1_identity_, xp_, yp_;   // w_x^row / w_y^col matrices and the two power tables
    std::shared_ptr<const Prover::LagrangePolys> lagrange_;           // K_last, L_last, K0, KL of prove2 / prove4
    std::unique_ptr<Sigma1> lagrange_n_, lagrange_mi_own_;            // Lagrange-basis commit tables of the n x s_max and m_I x s_max grids
    std::unique_ptr<Sigma1> lagrange_mi_prefix_;                      // prefix sums of the m_I x s_max one in prove1's walk order
    const Sigma1 *lagrange_mi_ = nullptr;                             // (= lagrange_n_ when n == m_I)

    tkmk_stream binding_stream_ = nullptr;                          // the helper thread's stream (PendingBinding)
    tkmk_stream commit_stream_ = nullptr;                           // the stream of a round's early commits (Prover::commit_early)
    ScalarField *pinned_ = nullptr;                                 // witness staging
    uint64_t pinned_cap_ = 0;
    std::vector<uint32_t> n_wires_;

    static WireLists make_lists(const std::vector<std::vector<std::pair<uint32_t, uint32_t>>> &per_kind) {
        WireLists L;
        std::vector<uint32_t> w, r;
        L.first.push_back(0);
        for (auto &v : per_kind) {
            for (auto &e : v) w.push_back(e.first), r.push_back(e.second);
            L.first.push_back((uint32_t)w.size());
        }
        if (w.empty()) w.push_back(0), r.push_back(0);
        L.wire = DeviceVec<uint32_t>::from_host(w);
        L.row = DeviceVec<uint32_t>::from_host(r);
        return L;
    }
    ScalarField *staging(uint64_t elements) {
        if (elements > pinned_cap_) {
            if (pinned_) check(tkmk_host_free(pinned_), "host_free");
            pinned_ = nullptr;
            pinned_cap_ = elements + elements / 8 + 1024;
            check(tkmk_host_malloc((void **)&pinned_, pinned_cap_ * sizeof(ScalarField)), "host_malloc");
        }
        return pinned_;
    }

  public:
    ProverContext() = default;
    ProverContext(const ProverContext &) = delete;
    ProverContext &operator=(const ProverContext &) = delete;
    ~ProverContext() {
        if (lib_) tkmk_r1cs_library_destroy(lib_);
        if (pinned_) tkmk_host_free(pinned_);
        if (binding_stream_) tkmk_stream_destroy(binding_stream_);
        if (commit_stream_) tkmk_stream_destroy(commit_stream_);
    }

    static SetupParams read_setup_params(const std::string &dir) {
        json::Value jp = json::read_file(dir + "/setupParams.json");
        return SetupParams{jp.at("l").as_size(),   jp.at("l_user_out").as_size(), jp.at("l_user").as_size(), jp.at("l_free").as_size(),
                           jp.at("l_D").as_size(), jp.at("m_D").as_size(),        jp.at("n").as_size(),      jp.at("s_D").as_size(),
                           jp.at("s_max").as_size()};
    }

    // circuit-static state: subcircuit library from <lib_dir>, reference string from <crs_dir> (combined_sigma.rkyv, or the flat
    // combined_sigma.tkcrs payload when only that is present)
    // link set: load_sigma returns this rank's columns of xy_powers; everything derived from them (table expansion, the Lagrange-basis
    // tables and their prefix sums) is made from those columns, 1 / G of the work per rank
    static std::unique_ptr<ProverContext> open(const std::string &lib_dir, const std::string &crs_dir,
                                               const std::function<std::unique_ptr<ProverSigma>(const SetupParams &, std::string &)> &load_sigma,
                                               const ShardLink &link = ShardLink{}) {
        std::unique_ptr<ProverContext> c(new ProverContext());
        c->link = link;
        ShardSpan span(c->link);
        c->lib_dir = lib_dir;
        c->threads = host_threads();
        c->sp = read_setup_params(lib_dir);
        const SetupParams &sp = c->sp;
        if (sp.l_D < sp.l) throw Error("Invalid setup params: l_D must be >= l.");   // setup_shape / validate_setup_shape (libs/src/utils/mod.rs:21-46)
        c->m_i = sp.l_D - sp.l;
        if (!is_pow2(sp.n)) throw Error("n is not a power of two.");
        if (!is_pow2(sp.s_max)) throw Error("s_max is not a power of two.");
        if (!is_pow2(c->m_i)) throw Error("m_I is not a power of two.");
        if (c->link && c->link.shard.world > 1) {
            const size_t G = c->link.shard.world;
            if (!is_pow2(G)) throw Error("sharded prover: the number of ranks must be a power of two");
            if (G > sp.s_max || G > std::min(sp.n, c->m_i)) throw Error("sharded prover: more ranks than the circuit has columns (s_max) or rows (n, m_I)");
        }
        {
            const json::Value jinfo = json::read_file(lib_dir + "/subcircuitInfo.json");
            for (const json::Value &e : jinfo.items()) {
                SubcircuitInfo si;
                si.id = e.at("id").as_size();
                si.name = e.at("name").as_string();
                si.Nwires = e.at("Nwires").as_size();
                si.Out_idx = {e.at("Out_idx").items().at(0).as_size(), e.at("Out_idx").items().at(1).as_size()};
                si.In_idx = {e.at("In_idx").items().at(0).as_size(), e.at("In_idx").items().at(1).as_size()};
                for (const json::Value &g : e.at("flattenMap").items()) si.flattenMap.push_back(g.as_size());
                c->infos.push_back(std::move(si));
                c->n_consts.push_back(e.at("Nconsts").as_size());
            }
        }
        init_ntt_domain_for_size(4 * std::max(c->m_i, sp.n) * 2 * sp.s_max);   // prover_verifier_ntt_domain_size (libs/src/utils/mod.rs:51-58)

        // constraint library -> device CSR, and the static wire lists
        const size_t K = c->infos.size();
        std::vector<SubcircuitR1CS> r1cs(K);
        std::vector<uint32_t> n_rows(K), n_wires(K);
        std::vector<const uint32_t *> rp(3 * K), wr(3 * K);
        std::vector<const tkmk_fr *> cf(3 * K);
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> iface(K), prv(K), pub(K);
        for (size_t k = 0; k < K; k++) {
            const SubcircuitInfo &info = c->infos[k];
            if (info.flattenMap.size() != info.Nwires) throw Error("subcircuitInfo.json: flattenMap length differs from Nwires for subcircuit " + std::to_string(info.id));
            R1csBinary b = R1csBinary::read(lib_dir + "/r1cs/subcircuit" + std::to_string(info.id) + ".r1cs");
            r1cs[k] = SubcircuitR1CS::from_r1cs_sparse_only(b, sp, info, c->n_consts[k]);
            n_rows[k] = r1cs[k].n_constraints, n_wires[k] = r1cs[k].n_wires;
            for (int m = 0; m < 3; m++) rp[3 * k + m] = r1cs[k].row_ptr[m].data(), wr[3 * k + m] = r1cs[k].wire[m].data(), cf[3 * k + m] = r1cs[k].coeff[m].data();
            for (size_t j = 0; j < info.Nwires; j++) {
                size_t g = info.flattenMap[j];
                if (g >= sp.m_D) throw Error("subcircuitInfo.json: flattenMap entry outside the global wire range");
                if (g >= sp.l && g < sp.l_D) iface[k].push_back({(uint32_t)j, (uint32_t)(g - sp.l)});
                else if (g >= sp.l_D) prv[k].push_back({(uint32_t)j, (uint32_t)(g - sp.l_D)});
            }
            const std::array<size_t, 2> *rng = nullptr;   // encode_O_pub_free (libs/src/group_structures/mod.rs:184-229)
            if (info.name == "bufferPubOut") rng = &info.Out_idx;
            else if (info.name == "bufferPubIn" || info.name == "bufferBlockIn") rng = &info.In_idx;
            if (rng)
                for (size_t j = (*rng)[0]; j < (*rng)[0] + (*rng)[1]; j++) {
                    if (j >= info.Nwires) throw Error("subcircuitInfo.json: public wire range outside the subcircuit");
                    if (info.flattenMap[j] >= sp.l) throw Error("subcircuitInfo.json: public wire mapped outside [0, l)");
                    pub[k].push_back({(uint32_t)j, (uint32_t)info.flattenMap[j]});
                }
        }
        check(tkmk_r1cs_library_create((uint32_t)K, n_rows.data(), n_wires.data(), rp.data(), wr.data(), cf.data(), &c->lib_), "tkmk_r1cs_library_create");
        c->n_wires_ = n_wires;
        c->iface_ = make_lists(iface), c->prv_ = make_lists(prv), c->pub_ = make_lists(pub);

        // Permutation::to_poly's identity part (libs/src/iotools/mod.rs:419-437): s0[row][col] = w_x^row, s1[row][col] = w_y^col — this rank's
        // columns of them (COLS layout: local column k is column rank + G k; all columns on one GPU)
        {
            const size_t m_i = c->m_i, s_max = sp.s_max;
            const Shard sh = c->link ? c->link.shard : Shard{};
            const size_t lc = sh.cols_of(s_max);
            std::vector<ScalarField> ones(m_i * lc, fr_from_u32(1));
            DeviceVec<ScalarField> base = DeviceVec<ScalarField>::from_host(ones);
            ScalarField wx = root_of_unity(m_i), wy = root_of_unity(s_max), one = fr_from_u32(1);
            const ScalarField wy_g = fr_pow(wy, sh.world), wy_r = fr_pow(wy, sh.rank);
            c->s0_identity_ = DeviceVec<ScalarField>(m_i * lc), c->s1_identity_ = DeviceVec<ScalarField>(m_i * lc);
            check(tkmk_poly_scale_coeffs(base.ptr(), (uint32_t)m_i, (uint32_t)lc, &wx, &one, c->s0_identity_.ptr(), nullptr), "s0 powers");
            check(tkmk_poly_scale_coeffs(base.ptr(), (uint32_t)m_i, (uint32_t)lc, &one, &wy_g, c->s1_identity_.ptr(), nullptr), "s1 powers");
            if (sh.rank) {
                tkmk_vecops_config vc = dev_cfg();
                vc.is_a_on_device = false;
                check(bls12_381_scalar_mul_vec(&wy_r, c->s1_identity_.ptr(), m_i * lc, &vc, c->s1_identity_.ptr()), "s1 powers");
            }
            // the two power tables the redirected cells read (indexed by the GLOBAL X / Y of a permutation entry): whole on every rank
            std::vector<ScalarField> xp(m_i), yp(s_max);
            ScalarField acc = one;
            for (size_t i = 0; i < m_i; i++) xp[i] = acc, acc = fr_mul(acc, wx);
            acc = one;
            for (size_t j = 0; j < s_max; j++) yp[j] = acc, acc = fr_mul(acc, wy);
            c->xp_ = DeviceVec<ScalarField>::from_host(xp), c->yp_ = DeviceVec<ScalarField>::from_host(yp);
        }

        host_trace("open: loading the CRS");
        c->sigma = load_sigma(sp, c->crs_source);
        host_trace("open: CRS resident");
        // binding tables -> resident form, in place (Sigma1 converts xy_powers itself)
        tkmk_msm_config cfg = tkmk_msm_default_config();
        cfg.are_points_on_device = cfg.are_results_on_device = true;
        // (expanding delta_inv_li_o_prv into a commit table like xy_powers was measured: 1.2 ms per proof for 50 GB and 6 s at open
        // — the table has (m_D - l_D) * s_max = 4 * 10^7 rows of which a proof touches 3 * 10^6: not kept)
        for (DeviceVec<G1Affine> *t : {&c->sigma->gamma_inv_o_inst, &c->sigma->eta_inv_li_o_inter_alpha4_kj, &c->sigma->delta_inv_li_o_prv})
            check(bls12_381_msm_convert_bases(t->ptr(), t->len(), &cfg, t->ptr()), "msm::convert_bases");
        c->sigma->binding_tables_converted = true;
        // Lagrange-basis twins of the commit table for the grids u, v, w (n x s_max) and b (m_I x s_max) live on: prove0 then commits
        // them from their evaluations.  Only with a commit table (a context that amortises seconds of one-time work); TKMK_PROVER_LAGRANGE=0
        // keeps the coefficient route.
        {
            const char *e = getenv("TKMK_PROVER_LAGRANGE");
            const size_t n = c->sp.n, m_i = c->m_i, s_max = c->sp.s_max;
            if (c->sigma->sigma1.table_c() && !(e && atoi(e) == 0) && is_pow2(n) && is_pow2(m_i) && is_pow2(s_max)) {
                const Sigma1 &s1 = c->sigma->sigma1;
                if (!c->link || c->link.shard.world == 1) {   // (a one-rank communicator holds whole tables: nothing to cut)
                    DeviceVec<G1Affine> lam_mi = s1.lagrange_points(m_i, s_max);
                    c->lagrange_mi_prefix_.reset(new Sigma1(s1.lagrange_prefix_of(lam_mi, m_i, s_max)));
                    if (m_i == n) {
                        c->lagrange_n_.reset(new Sigma1(std::move(lam_mi), m_i, s_max, s1.table_c()));
                        c->lagrange_mi_ = c->lagrange_n_.get();
                    } else {
                        c->lagrange_mi_own_.reset(new Sigma1(std::move(lam_mi), m_i, s_max, s1.table_c()));
                        c->lagrange_mi_ = c->lagrange_mi_own_.get();
                        c->lagrange_n_.reset(new Sigma1(s1.lagrange_of(n, s_max)));
                    }
                } else {
                    // sharded: the group transforms are divided like the field transforms of a proof (r4).  The inverse NTT over G1 points runs
                    // its X pass over this rank's columns of the monomial grid (the level-0 part of its own commit table), changes layout
                    // (one all-to-all of 96-byte points), runs its Y pass over this rank's rows, and changes back: 1 / G of the
                    // butterflies per rank, and the result IS this rank's columns of the Lagrange-basis grid — a plain table over its own
                    // m_I x (s_max / G) part, committed against its own columns of the evaluations (COLS layout on both sides).
                    const Shard sh = c->link.shard;
                    const ShardLink &lk = c->link;
                    const size_t G = sh.world, lc = sh.cols_of(s_max);
                    auto lagrange_cols = [&](size_t xs, size_t ys) {   // -> plain affine, xs x (ys / G)
                        const size_t lcs = ys / G, h = xs / G;
                        DeviceVec<G1Affine> a(xs * lcs), rows(h * ys), b(h * ys);
                        host_trace("lagrange_points %zu x %zu (rank %u of %zu)", xs, ys, sh.rank, G);
                        check(tkmk_g1_ntt_axes(s1.level0(), TKMK_BASES_CONVERTED, (uint32_t)s1.local_cols(), (uint32_t)xs, (uint32_t)lcs, TKMK_NTT_INVERSE, TKMK_G1_NTT_AXIS_X,
                                               a.ptr(), nullptr),
                              "tkmk_g1_ntt_axes");
                        check(lk.relayout_cols_to_rows(lk.comm, a.ptr(), xs, ys, sizeof(G1Affine), rows.ptr()), "tkmk_dist_relayout_cols_to_rows");
                        check(tkmk_g1_ntt_axes(rows.ptr(), TKMK_BASES_PLAIN, (uint32_t)ys, (uint32_t)h, (uint32_t)ys, TKMK_NTT_INVERSE, TKMK_G1_NTT_AXIS_Y, b.ptr(), nullptr),
                              "tkmk_g1_ntt_axes");
                        check(lk.relayout_rows_to_cols(lk.comm, b.ptr(), xs, ys, sizeof(G1Affine), a.ptr()), "tkmk_dist_relayout_rows_to_cols");
                        const ScalarField inv_n = Sigma1::inverse_of_grid_size(xs, ys);   // the transform's 1 / N, on this rank's columns
                        check(tkmk_g1_scale(a.ptr(), xs * lcs, &inv_n, a.ptr(), nullptr), "tkmk_g1_scale");
                        return a;
                    };
                    DeviceVec<G1Affine> lam_mi = lagrange_cols(m_i, s_max);
                    {
                        // the walk-ordered prefix sums S_j = sum_{j' <= j} Lambda_walk(j') of prove1, this rank's stretches of them (the walk
                        // goes down column 0, then column 1, ...).  My columns in walk order, the column totals of ALL ranks gathered
                        // (s_max points), and in front of every one of my columns ONE extra point = the sum of the other ranks' columns the
                        // walk passes between my previous column and this one: the ordinary running sum over that sequence is the global one.
                        DeviceVec<G1Affine> walk(lc * m_i);
                        for (size_t k = 0; k < lc; k++)
                            check(tkmk_memcpy_2d_d2d(walk.ptr() + k * m_i, sizeof(G1Affine), lam_mi.ptr() + k, lc * sizeof(G1Affine), sizeof(G1Affine), m_i), "walk order");
                        std::vector<ScalarField> ones_h(std::max<size_t>(m_i, G), fr_from_u32(1));
                        DeviceVec<ScalarField> ones = DeviceVec<ScalarField>::from_host(ones_h);
                        std::vector<tkmk_msm_job_ex> jobs(lc);
                        for (size_t k = 0; k < lc; k++) {
                            tkmk_msm_job_ex j{};
                            j.scalars = ones.ptr(), j.bases = walk.ptr() + k * m_i, j.msm_size = (int)m_i, j.base_table_len = m_i;
                            jobs[k] = j;
                        }
                        tkmk_msm_config mc = Sigma1::device_cfg();
                        std::vector<tkmk_g1_projective> tot(lc);
                        check(tkmk_msm_multi_ex(jobs.data(), (int)lc, &mc, TKMK_BASES_PLAIN, tot.data()), "column totals");
                        std::vector<G1Affine> mine(lc), all(lc * G);
                        for (size_t k = 0; k < lc; k++) mine[k] = Sigma1::to_affine(tot[k]);
                        check(lk.all_gather_host(lk.comm, mine.data(), lc * sizeof(G1Affine), all.data()), "tkmk_comm_all_gather_host");
                        auto total_of = [&](size_t col) { return all[(col % G) * lc + col / G]; };
                        std::vector<G1Affine> between(lc, G1Affine{});   // (0, 0) = the point at infinity
                        if (G > 1) {
                            std::vector<ScalarField> sc(lc * (G - 1), fr_from_u32(1));
                            std::vector<G1Affine> pts(lc * (G - 1), G1Affine{});
                            for (size_t k = 0; k < lc; k++) {
                                const size_t lo = k ? sh.rank + G * (k - 1) + 1 : 0, hi = sh.rank + G * k;   // the other ranks' columns in [lo, hi)
                                for (size_t col = lo, at = 0; col < hi; col++, at++) pts[k * (G - 1) + at] = total_of(col);
                            }
                            tkmk_msm_config hc = tkmk_msm_default_config();
                            hc.batch_size = (int)lc, hc.are_points_shared_in_batch = false;
                            std::vector<tkmk_g1_projective> res(lc);
                            check(bls12_381_msm(sc.data(), pts.data(), (int)(G - 1), &hc, res.data()), "sums between my columns");
                            for (size_t k = 0; k < lc; k++) between[k] = Sigma1::to_affine(res[k]);
                        }
                        DeviceVec<G1Affine> seq(lc * (m_i + 1)), sums(lc * (m_i + 1)), pre(lc * m_i);
                        for (size_t k = 0; k < lc; k++) {
                            check(tkmk_memcpy_h2d(seq.ptr() + k * (m_i + 1), &between[k], sizeof(G1Affine)), "offset point");
                            check(tkmk_memcpy_d2d(seq.ptr() + k * (m_i + 1) + 1, walk.ptr() + k * m_i, m_i * sizeof(G1Affine)), "walk order");
                        }
                        check(tkmk_g1_prefix_sums(seq.ptr(), TKMK_BASES_PLAIN, (uint32_t)(lc * (m_i + 1)), 1, 0, sums.ptr(), nullptr), "tkmk_g1_prefix_sums");
                        for (size_t k = 0; k < lc; k++)
                            check(tkmk_memcpy_d2d(pre.ptr() + k * m_i, sums.ptr() + k * (m_i + 1) + 1, m_i * sizeof(G1Affine)), "strip offsets");
                        c->lagrange_mi_prefix_.reset(new Sigma1(std::move(pre), lc * m_i, 1, s1.table_c()));
                    }
                    std::unique_ptr<Sigma1> local_mi(new Sigma1(std::move(lam_mi), m_i, lc, s1.table_c()));
                    if (m_i == n) {
                        c->lagrange_n_ = std::move(local_mi);
                        c->lagrange_mi_ = c->lagrange_n_.get();
                    } else {
                        c->lagrange_mi_own_ = std::move(local_mi);
                        c->lagrange_mi_ = c->lagrange_mi_own_.get();
                        c->lagrange_n_.reset(new Sigma1(lagrange_cols(n, s_max), n, lc, s1.table_c()));
                    }
                }
            }
        }
        host_trace("open: Lagrange polynomials");
        c->lagrange_ = Prover::LagrangePolys::make(c->m_i, c->sp.s_max);
        host_trace("open: done");
        check(tkmk_device_synchronize(), "synchronize");
        check(tkmk_release_scratch(), "release_scratch");   // the one-time transforms' scratch (1.6 GB for the group NTT) is not a proof's
        check(tkmk_device_synchronize(), "synchronize");
        return c;
    }

    // Prover::init (prove/src/lib.rs:675-1206) from the synthesizer's directory
    // pending (optional): when given and this context is not sharded, the binding batch is left running and *pending holds it;
    // finish_binding() completes the returned Binding later.  Otherwise the Binding is complete on return.
    std::pair<std::unique_ptr<Prover>, Binding> init(const std::string &synth_dir, const Mixer &mixer, ProveTiming &tm, std::unique_ptr<PendingBinding> *pending = nullptr) {
        using namespace prover_detail;
        const double t0 = Prover::now();
        const size_t n = sp.n, s_max = sp.s_max, K = infos.size();
        host_trace("init: begin");

        // ---- the three per-proof documents.  permutation.json first (every host thread, ~1 ms); then the (fourteen times larger) witness
        // document is read on a helper — with every host thread again — while this one, mostly waiting on the device, builds the
        // permutation's two polynomials; the two meet before the witness upload.
        // Sharded: every rank reads the whole document's structure and every placement's kind, but converts (and thereby checks) only the
        // values of its own placements — so an input error may show on one rank only.  The ranks therefore AGREE on the outcome of this
        // phase before any of them goes on: a malformed document is then the same error on every rank, at the same point.
        const Shard sh = link ? link.shard : Shard{};
        WitnessLayout W;
        std::vector<ScalarField> a_pub_user, a_pub_block;
        std::unique_ptr<Prover> p(new Prover());
        p->sp = sp, p->m_i = m_i, p->sigma = sigma.get(), p->mixer = mixer, p->lagrange = lagrange_;
        if (!link) {
            if (!commit_stream_) check(tkmk_stream_create(&commit_stream_), "stream_create");
            p->commit_stream = commit_stream_;
        }
        MappedFile perm_file(synth_dir + "/permutation.json");
        PermutationColumns perm = parse_permutation_fast(perm_file.data(), perm_file.size(), threads);
        host_trace("init: permutation.json parsed (%zu entries)", perm.size());
        // (the helper signals as soon as the values are staged; unmapping the 88 MB document — milliseconds of page-table work — happens
        // after the signal, beside the upload)
        std::promise<void> witness_staged;
        std::future<void> witness_read = witness_staged.get_future();
        struct Joined {
            std::thread t;
            ~Joined() { if (t.joinable()) t.join(); }
        } witness_reader{std::thread([&] {
            std::unique_ptr<MappedFile> pv_file;
            try {
                pv_file.reset(new MappedFile(synth_dir + "/placementVariables.json"));
                W = parse_placement_variables_fast(pv_file->data(), pv_file->size(), n_wires_, [&](uint64_t total) { return staging(total); }, threads, sh.world, sh.rank);
                host_trace("init: placementVariables.json parsed (%zu placements, %llu values kept)", W.id.size(), (unsigned long long)W.total);
                if (W.id.size() > s_max) throw Error("placement_variables length exceeds s_max.");
                const json::Value jinst = json::read_file(synth_dir + "/instance.json");
                a_pub_user = hex_list(jinst.at("a_pub_user"));
                a_pub_block = hex_list(jinst.at("a_pub_block"));
                witness_staged.set_value();
            } catch (...) {
                witness_staged.set_exception(std::current_exception());
            }
        })};
        DeviceVec<uint32_t> d_dst, d_x, d_y;   // alive until the kernels reading them have run (synchronised before init returns)
        {
            // Permutation::to_poly's redirects: distinct destinations (the reference's serial loop lets the last entry win)
            std::vector<uint32_t> dst, srcx, srcy;
            const size_t cells = m_i * s_max;
            std::vector<uint64_t> bitmap((cells + 63) / 64, 0);
            bool dup = false;
            dst.reserve(perm.size());
            for (size_t e = 0; e < perm.size(); e++) {
                if (perm.row[e] >= m_i || perm.col[e] >= s_max || perm.X[e] >= m_i || perm.Y[e] >= s_max) throw Error("permutation entry out of range");
                uint32_t d = perm.row[e] * (uint32_t)s_max + perm.col[e];
                if (bitmap[d >> 6] >> (d & 63) & 1) dup = true;
                bitmap[d >> 6] |= 1ull << (d & 63);
                dst.push_back(d);
            }
            srcx = perm.X, srcy = perm.Y;
            if (dup) {   // keep the last writer of every cell
                std::fill(bitmap.begin(), bitmap.end(), 0);
                std::vector<uint32_t> d2, x2, y2;
                for (size_t e = perm.size(); e-- > 0;) {
                    uint32_t d = dst[e];
                    if (bitmap[d >> 6] >> (d & 63) & 1) continue;
                    bitmap[d >> 6] |= 1ull << (d & 63);
                    d2.push_back(d), x2.push_back(srcx[e]), y2.push_back(srcy[e]);
                }
                dst.swap(d2), srcx.swap(x2), srcy.swap(y2);
            }
            // Permutation::to_poly (libs/src/iotools/mod.rs:419-455).  Sharded: this rank redirects the cells of its own columns
            // (col = rank mod G; local cell row * lc + col / G); the sources X, Y stay global indices into the two power tables
            const size_t lc = sh.cols_of(s_max);
            if (sh.world > 1) {
                std::vector<uint32_t> d2, x2, y2;
                for (size_t e = 0; e < dst.size(); e++) {
                    const uint32_t row = dst[e] / (uint32_t)s_max, col = dst[e] % (uint32_t)s_max;
                    if (col % sh.world != sh.rank) continue;
                    d2.push_back(row * (uint32_t)lc + col / sh.world), x2.push_back(srcx[e]), y2.push_back(srcy[e]);
                }
                dst.swap(d2), srcx.swap(x2), srcy.swap(y2);
            }
            DeviceVec<ScalarField> e0 = s0_identity_.clone(), e1 = s1_identity_.clone();
            if (!dst.empty()) {
                d_dst = DeviceVec<uint32_t>::from_host(dst), d_x = DeviceVec<uint32_t>::from_host(srcx), d_y = DeviceVec<uint32_t>::from_host(srcy);
                check(tkmk_fr_scatter_table(xp_.ptr(), xp_.len(), d_x.ptr(), d_dst.ptr(), dst.size(), e0.ptr(), e0.len(), nullptr), "tkmk_fr_scatter_table");
                check(tkmk_fr_scatter_table(yp_.ptr(), yp_.len(), d_y.ptr(), d_dst.ptr(), dst.size(), e1.ptr(), e1.len(), nullptr), "tkmk_fr_scatter_table");
            }
            p->s0XY = Poly::from_rou_evals_cols(e0, m_i, s_max), p->s1XY = Poly::from_rou_evals_cols(e1, m_i, s_max);
            p->s0_ev = std::move(e0), p->s1_ev = std::move(e1);   // prove1 forms f and g on the grid from these
            p->s0_identity = &s0_identity_, p->s1_identity = &s1_identity_;
        }
        host_trace("init: s0 / s1 issued");
        std::string input_error;
        try {
            witness_read.get();
            host_trace("init: witness documents joined");
        } catch (const std::exception &e) {
            if (sh.world == 1) throw;
            input_error = e.what();
            if (input_error.empty()) input_error = "input error";
        }
        if (sh.world > 1) {
            const tkmk_error verdict = link.agree(link.comm, input_error.empty() ? TKMK_SUCCESS : TKMK_ERR_INVALID_ARGUMENT);
            if (!input_error.empty()) throw Error(input_error);
            if (verdict != TKMK_SUCCESS) throw Error("another rank of the sharded prover could not read its share of the synthesizer's documents");
        }
        const size_t P = W.id.size();
        // placements grouped by kind: slot list, variable offsets, position of each kind's run.  Sharded: a placement is a column of
        // u, v, w, b — this rank takes the placements q = rank mod G; `slots` are their local columns q / G, `gslots` the global ones
        // (the binding tables are indexed by the global slot)
        const size_t lc = sh.cols_of(s_max);
        std::vector<uint32_t> mine_q;
        for (size_t q = sh.rank; q < P; q += sh.world) mine_q.push_back((uint32_t)q);
        const size_t PL = mine_q.size();
        std::vector<uint32_t> kind_first(K + 1, 0), slots(PL ? PL : 1), gslots(PL ? PL : 1), ids_local(PL ? PL : 1);
        std::vector<uint64_t> offs(PL ? PL : 1), offs_local(PL ? PL : 1);
        for (size_t k = 0; k < PL; k++) kind_first[W.id[mine_q[k]] + 1]++, ids_local[k] = W.id[mine_q[k]], offs_local[k] = W.off[mine_q[k]];
        for (size_t k = 0; k < K; k++) kind_first[k + 1] += kind_first[k];
        {
            std::vector<uint32_t> cur(kind_first.begin(), kind_first.end() - 1);
            for (size_t k = 0; k < PL; k++) {
                const uint32_t q = mine_q[k];
                uint32_t at = cur[W.id[q]]++;
                slots[at] = (uint32_t)k, gslots[at] = q, offs[at] = W.off[q];
            }
        }
        tm.parse = Prover::now() - t0;

        // ---- one upload of the witness and of the small index arrays
        const double t1 = Prover::now();
        DeviceVec<ScalarField> d_vars((size_t)W.total + 1);
        if (W.total) check(tkmk_memcpy_h2d(d_vars.ptr(), pinned_, (size_t)W.total * sizeof(ScalarField)), "witness upload");
        DeviceVec<uint32_t> d_id = DeviceVec<uint32_t>::from_host(ids_local);
        DeviceVec<uint64_t> d_off = DeviceVec<uint64_t>::from_host(offs_local);
        DeviceVec<uint32_t> d_slots = DeviceVec<uint32_t>::from_host(slots), d_gslots = DeviceVec<uint32_t>::from_host(gslots);
        DeviceVec<uint64_t> d_offs = DeviceVec<uint64_t>::from_host(offs);
        tm.upload = Prover::now() - t1;

        // ---- polynomials
        const double t2 = Prover::now();
        {   // read_R1CS_gen_uvwXY (libs/src/iotools/mod.rs:1287-1420): one column per placement — this rank's placements, its columns
            DeviceVec<ScalarField> u(n * lc), v(n * lc), w(n * lc);
            check(tkmk_r1cs_library_eval(lib_, d_vars.ptr(), d_id.ptr(), d_off.ptr(), (uint32_t)PL, (uint32_t)n, (uint32_t)lc, u.ptr(), v.ptr(), w.ptr(), nullptr),
                  "tkmk_r1cs_library_eval");
            p->uXY = Poly::from_rou_evals_cols(u, n, s_max), p->vXY = Poly::from_rou_evals_cols(v, n, s_max), p->wXY = Poly::from_rou_evals_cols(w, n, s_max);
            if (lagrange_n_) p->u_ev = std::move(u), p->v_ev = std::move(v), p->w_ev = std::move(w);   // prove0 commits from these
        }
        // gen_bXY + the (scalar, CRS row) lists of O_mid / O_prv / O_pub_free, kind by kind
        uint64_t n_mid = 0, n_prv = 0, n_pub = 0;
        std::vector<uint64_t> mid_at(K), prv_at(K), pub_at(K);
        for (size_t k = 0; k < K; k++) {
            uint64_t cnt = kind_first[k + 1] - kind_first[k];
            mid_at[k] = n_mid, prv_at[k] = n_prv, pub_at[k] = n_pub;
            n_mid += cnt * iface_.count(k), n_prv += cnt * prv_.count(k), n_pub += cnt * pub_.count(k);
        }
        // nVar checks of encode_statement_common (libs/src/group_structures/mod.rs:231-264, 294-296), over ALL placements
        {
            std::vector<PlacementVariables> shape(P);
            uint64_t all_mid = 0, all_prv = 0;
            for (size_t q = 0; q < P; q++) shape[q].subcircuitId = W.id[q], all_mid += iface_.count(W.id[q]), all_prv += prv_.count(W.id[q]);
            if (all_mid != count_o_mid_nvar(shape, infos) || all_prv != count_o_prv_nvar(shape, infos)) throw Error("nVar mismatch while encoding statement");
        }
        if (n_mid >= (1ull << 31) || n_prv >= (1ull << 31)) throw Error("binding commitment too large");
        DeviceVec<ScalarField> b_ev(m_i * lc), mid_sc(n_mid + 1), prv_sc(n_prv + 1), pub_sc(n_pub + 1);
        DeviceVec<uint32_t> mid_ix(n_mid + 1), prv_ix(n_prv + 1), pub_ix(n_pub + 1);
        if (m_i * lc) check(tkmk_memset(b_ev.ptr(), 0, m_i * lc * sizeof(ScalarField)), "memset");
        for (size_t k = 0; k < K; k++) {
            uint32_t cnt = kind_first[k + 1] - kind_first[k];
            if (!cnt) continue;
            const uint64_t *vo = d_offs.ptr() + kind_first[k];
            const uint32_t *sl = d_slots.ptr() + kind_first[k], *gsl = d_gslots.ptr() + kind_first[k];
            if (sh.world == 1) {
                check(tkmk_witness_route(d_vars.ptr(), vo, sl, cnt, iface_.wire.ptr() + iface_.first[k], iface_.row.ptr() + iface_.first[k], iface_.count(k), b_ev.ptr(),
                                         (uint32_t)s_max, mid_sc.ptr() + mid_at[k], mid_ix.ptr() + mid_at[k], (uint32_t)s_max, 1, nullptr),
                      "tkmk_witness_route");
            } else {
                // the interface matrix takes the LOCAL column of a placement, the binding table's row index its GLOBAL slot: two calls
                check(tkmk_witness_route(d_vars.ptr(), vo, sl, cnt, iface_.wire.ptr() + iface_.first[k], iface_.row.ptr() + iface_.first[k], iface_.count(k), b_ev.ptr(),
                                         (uint32_t)lc, nullptr, nullptr, 0, 0, nullptr),
                      "tkmk_witness_route");
                check(tkmk_witness_route(d_vars.ptr(), vo, gsl, cnt, iface_.wire.ptr() + iface_.first[k], iface_.row.ptr() + iface_.first[k], iface_.count(k), nullptr, 0,
                                         mid_sc.ptr() + mid_at[k], mid_ix.ptr() + mid_at[k], (uint32_t)s_max, 1, nullptr),
                      "tkmk_witness_route");
            }
            check(tkmk_witness_route(d_vars.ptr(), vo, gsl, cnt, prv_.wire.ptr() + prv_.first[k], prv_.row.ptr() + prv_.first[k], prv_.count(k), nullptr, 0,
                                     prv_sc.ptr() + prv_at[k], prv_ix.ptr() + prv_at[k], (uint32_t)s_max, 1, nullptr),
                  "tkmk_witness_route");
            check(tkmk_witness_route(d_vars.ptr(), vo, gsl, cnt, pub_.wire.ptr() + pub_.first[k], pub_.row.ptr() + pub_.first[k], pub_.count(k), nullptr, 0,
                                     pub_sc.ptr() + pub_at[k], pub_ix.ptr() + pub_at[k], 1, 0, nullptr),
                  "tkmk_witness_route");
        }
        p->bXY = Poly::from_rou_evals_cols(b_ev, m_i, s_max);
        p->b_ev = std::move(b_ev);   // prove1 forms f and g on the grid from it; with the Lagrange tables prove0 commits B from it too
        if (lagrange_mi_) p->lagrange_n = lagrange_n_.get(), p->lagrange_mi = lagrange_mi_, p->lagrange_mi_prefix = lagrange_mi_prefix_.get();
        p->rXY = Poly::zero();
        p->a_free_X = gen_a_free_X(a_pub_user, a_pub_block, sp);
        p->t_n = vanishing(n, true), p->t_mi = vanishing(m_i, true), p->t_smax = vanishing(s_max, false);
        check(tkmk_device_synchronize(), "synchronize");
        tm.build = Prover::now() - t2;

        // ---- binding commitments (lib.rs:1086-1160): four MSMs over resident tables in one pipelined call
        const double t3 = Prover::now();
        const Mixer &mx = mixer;
        Binding b;
        // sharded: the binding tables are replicated and a rank commits the (scalar, row) lists of its own placements
        auto indexed = [](const DeviceVec<ScalarField> &sc, const DeviceVec<uint32_t> &ix, uint64_t cnt, const DeviceVec<G1Affine> &table) {
            tkmk_msm_job_ex j{};
            j.scalars = sc.ptr(), j.bases = table.ptr(), j.msm_size = (int)cnt, j.base_index = ix.ptr(), j.base_table_len = table.len();
            return j;
        };
        std::vector<tkmk_msm_job_ex> binding_jobs = {sigma->sigma1.job(p->a_free_X, "A_free"), indexed(pub_sc, pub_ix, n_pub, sigma->gamma_inv_o_inst),
                                                     indexed(mid_sc, mid_ix, n_mid, sigma->eta_inv_li_o_inter_alpha4_kj),
                                                     indexed(prv_sc, prv_ix, n_prv, sigma->delta_inv_li_o_prv)};
        static const bool async_binding = [] {
            const char *e = getenv("TKMK_PROVER_ASYNC_BINDING");
            return !(e && atoi(e) == 0);
        }();
        // (at one pipeline stream — bench.py's serialised profiling pass: every kernel alone on the device — the batch runs in line)
        if (pending && !link && async_binding && tkmk_msm_get_pipeline_streams() > 1) {
            if (!binding_stream_) check(tkmk_stream_create(&binding_stream_), "stream_create");
            pending->reset(new PendingBinding());
            PendingBinding &pb = **pending;
            pb.mid_sc = std::move(mid_sc), pb.prv_sc = std::move(prv_sc), pb.pub_sc = std::move(pub_sc);   // the jobs point into these blocks
            pb.mid_ix = std::move(mid_ix), pb.prv_ix = std::move(prv_ix), pb.pub_ix = std::move(pub_ix);
            tkmk_stream bs = binding_stream_;
            // the helper first makes the proof's blinding points (one small MSM + a host Horner, 2 ms: prove0 needs them at its END, the
            // binding commitments are needed after prove4), then runs the binding batch
            std::shared_ptr<std::promise<void>> blinds_done = std::make_shared<std::promise<void>>();
            p->blinds_ready = blinds_done->get_future().share();
            Prover *pp = p.get();   // alive until this future has been collected (ProverContext::prove: `pending` goes before the prover)
            pb.cores = std::async(std::launch::async, [binding_jobs, bs, pp, blinds_done] {
                if (blinds_done) {
                    try {
                        pp->blinds_.reset(new Prover::Blinds(pp->compute_blinds(bs)));
                        blinds_done->set_value();
                    } catch (...) {
                        blinds_done->set_exception(std::current_exception());
                    }
                }
                return Sigma1::run_jobs(binding_jobs, bs);
            });
            tm.binding = Prover::now() - t3;
            tm.init = Prover::now() - t0;
            p->timing["init.parse"] = tm.parse, p->timing["init.upload"] = tm.upload, p->timing["init.build"] = tm.build;
            p->timing["init.binding"] = tm.binding, p->timing["init.total"] = tm.init;
            return {std::move(p), b};
        }
        b = finish_binding(Sigma1::run_jobs(binding_jobs), p->blinds());
        tm.binding = Prover::now() - t3;
        tm.init = Prover::now() - t0;
        p->timing["init.parse"] = tm.parse, p->timing["init.upload"] = tm.upload, p->timing["init.build"] = tm.build;
        p->timing["init.binding"] = tm.binding, p->timing["init.total"] = tm.init;
        return {std::move(p), b};
    }

    // the blinding terms on top of the four core commitments (lib.rs:1100-1160): O_mid = core + rO_mid delta, O_prv = core - rO_mid eta + the
    // fourteen mixer-weighted CRS points — both sums come with the proof's other blinding points (Prover::Blinds), one affine addition here
    Binding finish_binding(const std::vector<G1Affine> &cm, const Prover::Blinds &bl) const {
        Binding b;
        b.A_free = cm[0], b.O_pub_free = cm[1];
        b.O_mid = fqh::g1_affine_add(cm[2], bl.O_mid), b.O_prv = fqh::g1_affine_add(cm[3], bl.O_prv);
        return b;
    }

    // main() of prove/src/main.rs:27-97 after check_device: init, five rounds, <out_dir>/proof.json
    // flags (include/tkmk_prover.h): TKMK_PROVE_TEST_PARTS = the reference's commit list in prove4 (Pi_AX, Pi_CX, Pi_B, M_X, N_X one by
    // one), TKMK_PROVE_COEFFICIENT_BASIS = U, V, W, B, R committed from coefficients although the Lagrange tables are resident.  Neither
    // changes a byte of the proof.
    Proof prove(const std::string &synth_dir, const std::string &out_dir, const Mixer &mixer_in, ProveTiming *timing = nullptr, int flags = 0) {
        ProveTiming tm;
        const double t0 = Prover::now();
        ShardSpan span(link);
        Mixer mixer = mixer_in;
        // the ranks of a sharded prover commit shares of ONE polynomial: they must blind it with the same scalars — rank 0's
        static_assert(std::is_trivially_copyable<Mixer>::value, "Mixer travels as bytes");
        if (link) check(link.broadcast_host(link.comm, &mixer, sizeof mixer, 0), "tkmk_comm_broadcast_host");
        std::pair<std::unique_ptr<Prover>, Binding> pb;
        std::unique_ptr<PendingBinding> pending;   // declared after pb: destroyed (and thereby waited for) BEFORE the prover whose a_free_X it reads
        std::map<std::string, double> times;
        Proof proof;
        try {
            pb = init(synth_dir, mixer, tm, &pending);
            if (flags & 2) pb.first->lagrange_n = pb.first->lagrange_mi = pb.first->lagrange_mi_prefix = nullptr;
            proof = run_rounds(*pb.first, pb.second, &times, (flags & 1) != 0);
        } catch (const Error &e) {
            if (pending && pending->cores.valid()) pending->cores.wait();   // the batch reads a_free_X of the prover about to go
            // A rank that leaves between two collectives must not leave its peers waiting in the next one.  What the replicated inputs
            // decide — a malformed document, a shape that does not fit (invalid argument / pointer) — fails every rank at the same point
            // and the communicator lives on; a failure only this rank may have had (memory, a device or transport error) aborts it.
            const bool replicated_cause = e.code == TKMK_ERR_INVALID_ARGUMENT || e.code == TKMK_ERR_INVALID_POINTER;
            if (link && link.shard.world > 1 && !replicated_cause) (void)link.abort(link.comm);
            throw;
        } catch (...) {
            if (pending && pending->cores.valid()) pending->cores.wait();
            if (link && link.shard.world > 1) (void)link.abort(link.comm);
            throw;
        }
        if (pending) {   // collect the binding commitments issued during init
            const double tb = Prover::now();
            proof.binding = finish_binding(pending->cores.get(), pb.first->blinds());
            pending.reset();
            tm.binding += Prover::now() - tb;
        }
        int k = 0;
        for (const char *name : {"prove0", "prove1", "prove2", "prove3", "prove4"}) tm.prove[k++] = times[name];
        const double tw = Prover::now();
        // sharded: the ranks hold one and the same proof; they agree that all of them got there, and rank 0 alone writes the file
        // (G writers truncating one path at once is how a reader sees half a document)
        if (link && link.shard.world > 1) check(link.agree(link.comm, TKMK_SUCCESS), "tkmk_comm_agree");
        if (!out_dir.empty() && (!link || link.shard.rank == 0)) {
            ::mkdir(out_dir.c_str(), 0777);
            const std::string path = out_dir + "/proof.json", tmp_path = path + ".tmp";
            {
                std::ofstream f(tmp_path);
                if (!f) throw Error("cannot write " + tmp_path);
                f << proof.to_json();
                f.close();
                if (!f) throw Error("cannot write " + tmp_path);
            }
            if (::rename(tmp_path.c_str(), path.c_str()) != 0) throw Error("cannot write " + path);
        }
        tm.write = Prover::now() - tw;
        tm.total = Prover::now() - t0;
        if (timing) *timing = tm;
        return proof;
    }
};

}  // namespace tkmk
