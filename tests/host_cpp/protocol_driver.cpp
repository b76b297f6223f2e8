// protocol_driver.cpp — exercises tokamak-zk-evm_amd/host/tkmk_protocol.hpp (C++ protocol glue above the C ABI) on inputs
// written by tests/test_gpu_host_cpp.py and dumps every result as { u32 tag, u64 nbytes, payload } records.
//   in: u32 l, l_free, l_D, m_D, n, s_max, n_perm, n_fn, rs_x, rs_y | perm[n_perm] (4 x u32) | a_fn[n_fn] Fr | TKCRS001 payload
//       | u32 n_info | per info: u32 name_code, Nwires, Out[2], In[2], flattenMap[Nwires] | u32 n_pl | per placement: u32 id, n, vars[n] Fr
#include <cstdio>
#include <cstdlib>

#include "tkmk_protocol.hpp"

using namespace tkmk;

static FILE *g_out;
static void emit(uint32_t tag, const void *p, uint64_t n) {
    fwrite(&tag, 4, 1, g_out);
    fwrite(&n, 8, 1, g_out);
    if (n) fwrite(p, 1, n, g_out);
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
static const char *NAMES[] = {"bufferPubOut", "bufferPubIn", "bufferBlockIn", "bufferEVMIn", "ADD"};

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *in = fopen(argv[1], "rb");
    g_out = fopen(argv[2], "wb");
    if (!in || !g_out) return 2;
    try {
        check(tkmk_set_device(0), "set_device");
        // ---- host-only pieces: Keccak, transcript, formatting ----
        auto h0 = keccak256(nullptr, 0);
        emit(1, h0.data(), 32);
        const uint8_t abc[3] = {'a', 'b', 'c'};
        auto h1 = keccak256(abc, 3);
        emit(2, h1.data(), 32);
        std::vector<uint8_t> longmsg(300, 'a');
        auto h2 = keccak256(longmsg.data(), longmsg.size());
        emit(3, h2.data(), 32);

        auto hd = rd<uint32_t>(in, 10);
        SetupParams sp{hd[0], 1, 3, hd[1], hd[2], hd[3], hd[4], 2, hd[5]};
        size_t n_perm = hd[6], n_fn = hd[7], rs_x = hd[8], rs_y = hd[9];
        std::vector<Permutation> perm;
        for (size_t i = 0; i < n_perm; i++) {
            auto e = rd<uint32_t>(in, 4);
            perm.push_back({e[0], e[1], e[2], e[3]});
        }
        auto a_fn = rd<ScalarField>(in, n_fn);
        auto plen = rd<uint64_t>(in, 1);
        CrsPayload crs = CrsPayload::parse(rd<uint8_t>(in, plen[0]));
        uint64_t secs[CrsPayload::Count];
        for (int i = 0; i < CrsPayload::Count; i++) secs[i] = crs.length[i];
        emit(10, secs, sizeof secs);
        emit(11, crs.g1(CrsPayload::GammaInvOInst), crs.length[CrsPayload::GammaInvOInst]);

        auto n_info = rd<uint32_t>(in, 1)[0];
        std::vector<SubcircuitInfo> infos;
        for (uint32_t k = 0; k < n_info; k++) {
            auto e = rd<uint32_t>(in, 6);
            SubcircuitInfo si{k, NAMES[e[0]], e[1], {e[2], e[3]}, {e[4], e[5]}, {}};
            for (uint32_t w : rd<uint32_t>(in, e[1])) si.flattenMap.push_back(w);
            infos.push_back(si);
        }
        auto n_pl = rd<uint32_t>(in, 1)[0];
        std::vector<PlacementVariables> pv;
        for (uint32_t k = 0; k < n_pl; k++) {
            auto e = rd<uint32_t>(in, 2);
            pv.push_back({e[0], rd<ScalarField>(in, e[1])});
        }

        // ---- preprocess round over the device path ----
        Sigma1 sigma(crs.upload(CrsPayload::XyPowers), rs_x, rs_y);
        DeviceVec<G1Affine> gamma = crs.upload(CrsPayload::GammaInvOInst);
        Preprocess pre = Preprocess::gen(sigma, gamma, perm, a_fn, sp);
        emit(20, &pre.s0, 96);
        emit(21, &pre.s1, 96);
        emit(22, &pre.O_pub_fix, 96);
        std::string js = pre.to_json();
        emit(23, js.data(), js.size());
        Preprocess back = Preprocess::recover_from_format(pre.convert_format_for_solidity_verifier());
        uint32_t same = std::memcmp(&back, &pre, sizeof pre) == 0;
        emit(24, &same, 4);

        // ---- binding commitments ----
        G1Affine o_free = encode_O_pub_free(gamma, pv, infos);
        emit(30, &o_free, 96);
        DeviceVec<G1Affine> eta = crs.upload(CrsPayload::EtaInvLiOInterAlpha4Kj), delta = crs.upload(CrsPayload::DeltaInvLiOPrv);
        G1Affine o_mid = encode_O_mid_no_zk(eta, pv, infos, sp), o_prv = encode_O_prv_no_zk(delta, pv, infos, sp);
        emit(31, &o_mid, 96);
        emit(32, &o_prv, 96);

        // ---- transcript over the commitments just made ----
        TranscriptManager tm;
        tm.add_proof0(pre.s0, pre.s1, pre.O_pub_fix, o_free, o_mid, o_prv);
        auto th = tm.get_thetas();
        emit(40, th.data(), th.size() * sizeof(ScalarField));
        tm.add_proof1(o_mid);
        ScalarField k0 = tm.get_kappa0();
        emit(41, &k0, 32);
        tm.add_proof2(o_prv, pre.s0);
        auto cz = tm.get_chi_zeta();
        emit(42, &cz.first, 32);
        emit(43, &cz.second, 32);
        tm.add_proof3(th[0], th[1], th[2], k0);
        ScalarField k1 = tm.get_kappa1();
        emit(44, &k1, 32);

        uint32_t threw = 0;
        try {
            CrsPayload::parse(std::vector<uint8_t>{'n', 'o', 'p', 'e'});
        } catch (const Error &) {
            threw |= 1;
        }
        try {
            encode_O_pub_fix(gamma, std::vector<ScalarField>(a_fn.begin(), a_fn.end() - 1), sp);
        } catch (const Error &) {
            threw |= 2;
        }
        try {
            std::vector<SubcircuitInfo> bad = infos;
            bad.back().In_idx[1] += 1;
            encode_O_mid_no_zk(eta, pv, bad, sp);
        } catch (const Error &) {
            threw |= 4;
        }
        emit(99, &threw, 4);
    } catch (const std::exception &ex) {
        fprintf(stderr, "protocol_driver: %s\n", ex.what());
        return 1;
    }
    fclose(g_out);
    return 0;
}
