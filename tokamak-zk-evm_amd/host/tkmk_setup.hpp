// tkmk_setup.hpp — CRS generation over the C++ host side: work-alike of the trusted setup's evaluation phase and Sigma::gen / Sigma1::gen /
// Sigma2::gen (packages/backend/setup/trusted-setup/src/main.rs:99-196, packages/backend/libs/src/group_structures/mod.rs:313-551,752-777,
// packages/backend/libs/src/field_structures/mod.rs:67-165).  The C++ twin of tkmk/setup.py, whose payload it must reproduce byte for byte
// (tests/test_gpu_prove.py::test_native_setup_binary): Lagrange values by one inverse NTT of the power vector, the QAP mixture through
// tkmk_r1cs_eval_rows on the transposed sparse matrices, outer products with gathers + a vector multiplication, every G1 point through the
// fixed-base batched scalar multiplication; Sigma2's ten G2 points through bls12_381_g2_msm (the generator is validated on the
// host, tkmk_g2.hpp).
#pragma once
#include <algorithm>
#include <fstream>
#include <numeric>

#include "tkmk_g2.hpp"
#include "tkmk_protocol.hpp"
#include "tkmk_rkyv.hpp"
#include "tkmk_witness.hpp"

namespace tkmk {

struct Tau {   // libs/src/field_structures/mod.rs:44-64
    ScalarField x, y, alpha, gamma, delta, eta;
};

namespace setup_detail {

// gen_evaled_lagrange_bases (libs/src/vector_operations/mod.rs:19-28): L_i(val), i < size -> device vector
inline DeviceVec<ScalarField> lagrange_bases(const ScalarField &val, size_t size) {
    std::vector<ScalarField> ones(size, fr_one());
    DeviceVec<ScalarField> d_ones = DeviceVec<ScalarField>::from_host(ones), pows(size);
    check(tkmk_poly_scale_coeffs(d_ones.ptr(), (uint32_t)size, 1, &val, nullptr, pows.ptr(), nullptr), "tkmk_poly_scale_coeffs");
    DeviceVec<ScalarField> out(size);
    check(tkmk_bintt(pows.ptr(), size, 1, TKMK_NTT_INVERSE, nullptr, nullptr, true, nullptr, out.ptr()), "_biNTT");
    return out;
}
// scale * col[j] * row[i] at [j * n_row + i] (type_scaled_outer_product_2d!)
inline DeviceVec<ScalarField> outer_scaled(const std::vector<ScalarField> &col, const DeviceVec<ScalarField> &row, size_t n_row, const ScalarField &scale) {
    size_t k = col.size();
    if (k == 0) return DeviceVec<ScalarField>(1);
    std::vector<ScalarField> scaled(k);
    for (size_t j = 0; j < k; j++) scaled[j] = fr_mul(col[j], scale);
    DeviceVec<ScalarField> d_col = DeviceVec<ScalarField>::from_host(scaled);
    std::vector<uint32_t> ia(k * n_row), ib(k * n_row);
    for (size_t j = 0; j < k; j++)
        for (size_t i = 0; i < n_row; i++) ia[j * n_row + i] = (uint32_t)j, ib[j * n_row + i] = (uint32_t)i;
    DeviceVec<uint32_t> d_ia = DeviceVec<uint32_t>::from_host(ia), d_ib = DeviceVec<uint32_t>::from_host(ib);
    DeviceVec<ScalarField> a(k * n_row), b(k * n_row);
    check(tkmk_gather_rows_device(d_col.ptr(), 32, d_ia.ptr(), k * n_row, a.ptr(), nullptr), "gather");
    check(tkmk_gather_rows_device(row.ptr(), 32, d_ib.ptr(), k * n_row, b.ptr(), nullptr), "gather");
    tkmk_vecops_config c = dev_cfg();
    check(bls12_381_vector_mul(a.ptr(), b.ptr(), k * n_row, &c, a.ptr()), "vector_mul");
    return a;
}
inline DeviceVec<G1Affine> points(const DeviceVec<ScalarField> &scalars, size_t n, const G1Affine &g) {
    DeviceVec<G1Affine> out(n);
    check(tkmk_g1_batch_scalar_mul_device(scalars.ptr(), &g, n, out.ptr(), nullptr), "tkmk_g1_batch_scalar_mul_device");
    return out;
}
inline DeviceVec<G1Affine> points(const std::vector<ScalarField> &scalars, const G1Affine &g) {
    return points(DeviceVec<ScalarField>::from_host(scalars), scalars.size(), g);
}

}  // namespace setup_detail

// o_evaled_vec of the setup (main.rs:129-160, from_r1cs_to_evaled_qap_mixture): o_j = alpha u_j + alpha^2 v_j + alpha^3 w_j at tau_x
inline std::vector<ScalarField> evaled_qap_mixture(const std::string &qap_path, const std::vector<SubcircuitInfo> &infos, const std::vector<size_t> &n_consts,
                                                    const SetupParams &sp, const Tau &tau) {
    DeviceVec<ScalarField> x_lag = setup_detail::lagrange_bases(tau.x, sp.n);
    ScalarField alpha[3] = {tau.alpha, fr_mul(tau.alpha, tau.alpha), fr_mul(fr_mul(tau.alpha, tau.alpha), tau.alpha)};
    std::vector<uint32_t> slot0 = {0};
    DeviceVec<uint32_t> d_slot = DeviceVec<uint32_t>::from_host(slot0);
    std::vector<ScalarField> o_vec(sp.m_D);
    for (size_t s = 0; s < infos.size(); s++) {
        const SubcircuitInfo &info = infos[s];
        R1csBinary b = R1csBinary::read(qap_path + "/r1cs/subcircuit" + std::to_string(info.id) + ".r1cs");
        SubcircuitR1CS r = SubcircuitR1CS::from_r1cs_sparse_only(b, sp, info, n_consts.at(s));
        std::vector<ScalarField> o(r.n_wires);
        for (int m = 0; m < 3; m++) {
            size_t nnz = r.wire[m].size();
            if (nnz == 0) continue;
            // transpose: CSR over constraints -> CSR over wires (stable by constraint row)
            std::vector<uint32_t> t_ptr(r.n_wires + 1, 0), t_rows(nnz);
            std::vector<ScalarField> t_coeff(nnz);
            for (uint32_t w : r.wire[m]) t_ptr[w + 1]++;
            std::partial_sum(t_ptr.begin(), t_ptr.end(), t_ptr.begin());
            std::vector<uint32_t> cur(t_ptr.begin(), t_ptr.end() - 1);
            for (uint32_t row = 0; row < r.n_constraints; row++)
                for (uint32_t k = r.row_ptr[m][row]; k < r.row_ptr[m][row + 1]; k++) {
                    uint32_t dst = cur[r.wire[m][k]]++;
                    t_rows[dst] = row;
                    t_coeff[dst] = r.coeff[m][k];
                }
            std::vector<ScalarField> zeros(r.n_wires);
            DeviceVec<ScalarField> out = DeviceVec<ScalarField>::from_host(zeros);
            DeviceVec<uint32_t> d_ptr = DeviceVec<uint32_t>::from_host(t_ptr), d_rows = DeviceVec<uint32_t>::from_host(t_rows);
            DeviceVec<ScalarField> d_coeff = DeviceVec<ScalarField>::from_host(t_coeff);
            // "constraints" = wires, "wires" = constraint rows, "variables" = the Lagrange vector, one slot
            check(tkmk_r1cs_eval_rows(d_ptr.ptr(), d_rows.ptr(), d_coeff.ptr(), r.n_wires, (uint32_t)nnz, x_lag.ptr(), (uint32_t)sp.n, 1, d_slot.ptr(),
                                      r.n_wires, out.ptr(), nullptr),
                  "tkmk_r1cs_eval_rows");
            std::vector<ScalarField> h = out.to_host();
            for (uint32_t j = 0; j < r.n_wires; j++)
                if (!fr_is_zero(h[j])) o[j] = fr_add(o[j], fr_mul(alpha[m], h[j]));
        }
        for (uint32_t j = 0; j < r.n_wires; j++)
            if (!fr_is_zero(o[j])) o_vec.at(info.flattenMap.at(j)) = o[j];
    }
    return o_vec;
}

struct Sigma {
    // payload sections in TKCRS001 order (tkmk_protocol.hpp CrsPayload::Section)
    std::vector<G1Affine> singles;   // G, x, y, delta, eta, lagrange_KL
    DeviceVec<G1Affine> xy_powers, gamma_inv_o_inst, eta_inv_li_o_inter_alpha4_kj, delta_inv_li_o_prv;
    std::vector<G1Affine> delta_inv_alphak_xh_tx, delta_inv_alpha4_xj_tx, delta_inv_alphak_yi_ty;
    std::vector<std::array<uint8_t, 192>> g2;   // H, alpha, alpha2, alpha3, alpha4, gamma, delta, eta, x, y (all zero without a G2 generator)

    static Sigma gen(const SetupParams &sp, const Tau &tau, const std::string &qap_path, const std::vector<SubcircuitInfo> &infos,
                     const std::vector<size_t> &n_consts, const G1Affine &g1, const g2h::Affine *g2_gen) {
        using namespace setup_detail;
        size_t n = sp.n, s_max = sp.s_max, l = sp.l, l_free = sp.l_free, m_i = sp.l_D - sp.l;
        if (sp.l_D < sp.l) throw Error("Invalid setup params: l_D must be >= l.");   // setup_shape / validate_setup_shape (libs/src/utils/mod.rs:21-46)
        if (!is_pow2(n)) throw Error("n is not a power of two.");
        if (!is_pow2(s_max)) throw Error("s_max is not a power of two.");
        if (!is_pow2(m_i)) throw Error("m_I is not a power of two.");
        if (l_free != 0 && !is_pow2(l_free)) throw Error("l is not a power of two.");      // validate_public_wire_size (:48-52)
        init_ntt_domain_for_size(std::max(std::max(n, l_free), std::max(m_i, s_max)));   // trusted_setup_ntt_domain_size (libs/src/utils/mod.rs:60-66)
        ScalarField gi = fr_inv(tau.gamma), di = fr_inv(tau.delta), ei = fr_inv(tau.eta);
        DeviceVec<ScalarField> k_dev = lagrange_bases(tau.x, m_i), l_dev = lagrange_bases(tau.y, s_max), m_dev = lagrange_bases(tau.x, l_free);
        std::vector<ScalarField> k_vec = k_dev.to_host(), l_vec = l_dev.to_host(), m_vec = m_dev.to_host();
        std::vector<ScalarField> o_vec = evaled_qap_mixture(qap_path, infos, n_consts, sp, tau);
        Sigma out;

        // xy_powers[h * 2 s_max + i] = [x^h y^i]G
        size_t h_max = std::max(2 * n, 2 * m_i), rs_y = 2 * s_max;
        {
            std::vector<ScalarField> ones(h_max * rs_y, fr_one());
            DeviceVec<ScalarField> d_ones = DeviceVec<ScalarField>::from_host(ones), mon(h_max * rs_y);
            check(tkmk_poly_scale_coeffs(d_ones.ptr(), (uint32_t)h_max, (uint32_t)rs_y, &tau.x, &tau.y, mon.ptr(), nullptr), "tkmk_poly_scale_coeffs");
            out.xy_powers = points(mon, h_max * rs_y, g1);
        }
        // gamma_inv_o_inst (:405-440)
        std::vector<ScalarField> gamma_scal(l);
        for (size_t j = 0; j < l; j++) {
            const ScalarField &lag = j < sp.l_user_out ? l_vec.at(0) : j < sp.l_user ? l_vec.at(1) : j < l_free ? l_vec.at(2) : l_vec.at(3);
            ScalarField v = fr_mul(lag, o_vec[j]);
            if (j < l_free) v = fr_add(v, m_vec[j]);
            gamma_scal[j] = fr_mul(gi, v);
        }
        out.gamma_inv_o_inst = points(gamma_scal, g1);
        ScalarField a2 = fr_mul(tau.alpha, tau.alpha), a4 = fr_mul(a2, a2);
        std::vector<ScalarField> inter(m_i);
        for (size_t j = 0; j < m_i; j++) inter[j] = fr_add(o_vec[l + j], fr_mul(a4, k_vec[j]));
        out.eta_inv_li_o_inter_alpha4_kj = points(outer_scaled(inter, l_dev, s_max, ei), m_i * s_max, g1);
        std::vector<ScalarField> prv(o_vec.begin() + sp.l_D, o_vec.begin() + sp.m_D);
        if (!prv.empty()) out.delta_inv_li_o_prv = points(outer_scaled(prv, l_dev, s_max, di), prv.size() * s_max, g1);

        ScalarField one = fr_one();
        ScalarField t_n = fr_sub(fr_pow(tau.x, n), one), t_mi = fr_sub(fr_pow(tau.x, m_i), one), t_s = fr_sub(fr_pow(tau.y, s_max), one);
        ScalarField apow[5] = {one, tau.alpha, a2, fr_mul(a2, tau.alpha), a4};
        std::vector<ScalarField> xh, xj, yi;
        for (int k = 1; k <= 3; k++)
            for (int h = 0; h < 3; h++) xh.push_back(fr_mul(fr_mul(fr_mul(di, apow[k]), fr_pow(tau.x, h)), t_n));
        for (int j = 0; j < 2; j++) xj.push_back(fr_mul(fr_mul(fr_mul(di, a4), fr_pow(tau.x, j)), t_mi));
        for (int k = 1; k <= 4; k++)
            for (int i = 0; i < 3; i++) yi.push_back(fr_mul(fr_mul(fr_mul(di, apow[k]), fr_pow(tau.y, i)), t_s));
        out.delta_inv_alphak_xh_tx = points(xh, g1).to_host();
        out.delta_inv_alpha4_xj_tx = points(xj, g1).to_host();
        out.delta_inv_alphak_yi_ty = points(yi, g1).to_host();
        std::vector<ScalarField> single_scalars = {one, tau.x, tau.y, tau.delta, tau.eta, fr_mul(l_vec.at(s_max - 1), k_vec.at(m_i - 1))};
        out.singles = points(single_scalars, g1).to_host();

        out.g2.assign(10, std::array<uint8_t, 192>{});
        if (g2_gen) {   // Sigma2::gen (:752-777) and H
            if (g2_gen->inf || !g2h::on_curve(*g2_gen)) throw Error("the G2 generator is not a point of the twist");
            ScalarField ks[10] = {one, tau.alpha, a2, apow[3], a4, tau.gamma, tau.delta, tau.eta, tau.x, tau.y};
            // ten multiples of one point = a batch of one-point G2 MSMs over shared points (bls12_381_g2_msm); results come back
            // as (x_affine, y_affine, 1) or (0, 1, 0)
            std::array<uint8_t, 192> h = g2h::encode(*g2_gen);
            tkmk_msm_config cfg = tkmk_msm_default_config();
            cfg.batch_size = 10;
            cfg.are_points_shared_in_batch = true;
            std::vector<tkmk_g2_projective> res(10);
            check(bls12_381_g2_msm(reinterpret_cast<const tkmk_fr *>(ks), reinterpret_cast<const tkmk_g2_affine *>(h.data()), 1, &cfg, res.data()), "g2 msm");
            for (int i = 0; i < 10; i++) {
                const uint8_t *r = reinterpret_cast<const uint8_t *>(&res[i]);
                bool inf = true;
                for (int b = 192; b < 288; b++) inf = inf && r[b] == 0;
                if (!inf) std::memcpy(out.g2[i].data(), r, 192);
            }
        }
        return out;
    }

    // <out_dir>/combined_sigma.tkcrs: "TKCRS001", u32 count, nine u32 lengths, the sections
    std::string write(const std::string &out_dir) const {
        std::vector<G1Affine> h_xy = xy_powers.to_host(), h_gamma = gamma_inv_o_inst.to_host(), h_eta = eta_inv_li_o_inter_alpha4_kj.to_host();
        std::vector<G1Affine> h_delta = delta_inv_li_o_prv.len() ? delta_inv_li_o_prv.to_host() : std::vector<G1Affine>();
        const std::vector<G1Affine> *g1s[8] = {&singles, &h_xy, &h_gamma, &h_eta, &h_delta, &delta_inv_alphak_xh_tx, &delta_inv_alpha4_xj_tx,
                                               &delta_inv_alphak_yi_ty};
        std::string path = out_dir + "/combined_sigma.tkcrs";
        std::ofstream f(path, std::ios::binary);
        if (!f) throw Error("cannot write " + path);
        f.write("TKCRS001", 8);
        uint32_t count = 9;
        f.write(reinterpret_cast<const char *>(&count), 4);
        for (int i = 0; i < 8; i++) {
            uint64_t bytes = (uint64_t)g1s[i]->size() * 96;
            if (bytes > 0xffffffffull) throw Error("CRS section exceeds the 4 GiB length field of the TKCRS001 payload");
            uint32_t n32 = (uint32_t)bytes;
            f.write(reinterpret_cast<const char *>(&n32), 4);
        }
        uint32_t g2_bytes = 10 * 192;
        f.write(reinterpret_cast<const char *>(&g2_bytes), 4);
        for (int i = 0; i < 8; i++) f.write(reinterpret_cast<const char *>(g1s[i]->data()), (std::streamsize)(g1s[i]->size() * 96));
        for (const auto &p : g2) f.write(reinterpret_cast<const char *>(p.data()), 192);
        if (!f) throw Error("short write on " + path);
        return path;
    }
    // bytes of the two archives the reference's setup writes (write_final_crs_artifacts, libs/src/iotools/mod.rs:271-300) with 32-bit
    // relative pointers: past 2 GiB rkyv itself cannot represent them
    uint64_t archive_bytes() const {
        return 96ull * (xy_powers.len() + gamma_inv_o_inst.len() + eta_inv_li_o_inter_alpha4_kj.len() + delta_inv_li_o_prv.len() + 29) + 10 * 192 + 65536;
    }
    // <out_dir>/sigma_verify.json: what the reference's `verify` reads (verify-rust/src/lib.rs:68-71) — SigmaVerify { G, H, sigma_1 { x, y },
    // sigma_2 { alpha, alpha2, alpha3, alpha4, gamma, delta, eta, x, y }, lagrange_KL } (libs/src/group_structures/mod.rs:849-860) through
    // serde_json's pretty writer (libs/src/iotools/mod.rs:210-218); a point is {"x": hex, "y": hex} with the coordinate as ONE big-endian
    // hex number over all its limbs (G1serde / G2serde: iotools/mod.rs:986-1059 — for G2 the 96-byte Fp2 element, imaginary part in the
    // high half, as the fixed generator of setup/trusted-setup/src/main.rs:75-78 is written).  The reader parses with from_hex, so any
    // "0x" + even-length hex is accepted; the digit count here (96 / 192) is ICICLE's own as far as known — parity of the exact text unpinned.
    std::string sigma_verify_json() const {
        auto hex_be = [](const uint8_t *le, size_t n) {
            static const char *d = "0123456789abcdef";
            std::string s = "0x";
            for (size_t i = n; i-- > 0;) s += d[le[i] >> 4], s += d[le[i] & 15];
            return s;
        };
        auto point = [&](const uint8_t *rec, size_t coord_bytes, const std::string &pad) {
            return "{\n" + pad + "  \"x\": \"" + hex_be(rec, coord_bytes) + "\",\n" + pad + "  \"y\": \"" + hex_be(rec + coord_bytes, coord_bytes) + "\"\n" + pad + "}";
        };
        auto g1 = [&](size_t i, const std::string &pad) { return point(reinterpret_cast<const uint8_t *>(&singles.at(i)), 48, pad); };
        auto g2p = [&](size_t i, const std::string &pad) { return point(g2.at(i).data(), 96, pad); };
        static const char *names[9] = {"alpha", "alpha2", "alpha3", "alpha4", "gamma", "delta", "eta", "x", "y"};
        std::string doc = "{\n  \"G\": " + g1(0, "  ") + ",\n  \"H\": " + g2p(0, "  ") + ",\n  \"sigma_1\": {\n    \"x\": " + g1(1, "    ") + ",\n    \"y\": " +
                          g1(2, "    ") + "\n  },\n  \"sigma_2\": {\n";
        for (int k = 0; k < 9; k++) doc += std::string("    \"") + names[k] + "\": " + g2p(1 + k, "    ") + (k < 8 ? ",\n" : "\n");
        doc += "  },\n  \"lagrange_KL\": " + g1(5, "  ") + "\n}";
        return doc;
    }
    bool has_sigma2() const {
        for (const auto &p : g2)
            for (uint8_t b : p)
                if (b) return true;
        return false;
    }
    std::string write_sigma_verify(const std::string &out_dir) const {
        std::string path = out_dir + "/sigma_verify.json";
        std::ofstream f(path);
        if (!f) throw Error("cannot write " + path);
        f << sigma_verify_json();
        f.close();
        if (!f) throw Error("short write on " + path);
        return path;
    }
    // <out_dir>/combined_sigma.rkyv and <out_dir>/sigma_preprocess.rkyv, the containers the reference's `prove` / `preprocess` open
    void write_rkyv(const std::string &out_dir, const SetupParams &sp) const {
        if (archive_bytes() >= (1ull << 31)) throw Error("this reference string does not fit an rkyv archive (32-bit relative pointers); use the .tkcrs payload");
        std::vector<G1Affine> h_xy = xy_powers.to_host(), h_gamma = gamma_inv_o_inst.to_host(), h_eta = eta_inv_li_o_inter_alpha4_kj.to_host();
        std::vector<G1Affine> h_delta = delta_inv_li_o_prv.len() ? delta_inv_li_o_prv.to_host() : std::vector<G1Affine>();
        std::vector<uint8_t> g2_bytes;
        for (const auto &p : g2) g2_bytes.insert(g2_bytes.end(), p.begin(), p.end());
        auto b = [](const std::vector<G1Affine> &v) { return reinterpret_cast<const uint8_t *>(v.data()); };
        const size_t m_i = sp.l_D - sp.l;
        rkyv::SigmaTables t{b(singles), b(h_xy), h_xy.size(), b(h_gamma), h_gamma.size(),
                            b(h_eta), std::vector<size_t>(m_i, sp.s_max),
                            b(h_delta), std::vector<size_t>(sp.m_D - sp.l_D, sp.s_max),
                            b(delta_inv_alphak_xh_tx), {3, 3, 3}, b(delta_inv_alpha4_xj_tx), delta_inv_alpha4_xj_tx.size(),
                            b(delta_inv_alphak_yi_ty), {3, 3, 3, 3}, g2_bytes.data()};
        auto put = [&](const std::string &name, const std::vector<uint8_t> &bytes) {
            std::string path = out_dir + "/" + name;
            std::ofstream f(path, std::ios::binary);
            f.write(reinterpret_cast<const char *>(bytes.data()), (std::streamsize)bytes.size());
            f.close();
            if (!f) throw Error("cannot write " + path);
        };
        put("combined_sigma.rkyv", rkyv::encode_combined_sigma(t));
        put("sigma_preprocess.rkyv", rkyv::encode_sigma_preprocess(b(h_xy), h_xy.size(), b(h_gamma), h_gamma.size()));
    }
};

}  // namespace tkmk
