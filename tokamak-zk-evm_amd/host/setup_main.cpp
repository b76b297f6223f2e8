// setup_main.cpp — native trusted setup with the reference binary's argument surface (packages/backend/setup/trusted-setup/src/main.rs:27-46):
//   trusted-setup --output DIR [--fixed-tau] [--subcircuit-library DIR] [--format both|rkyv|tkcrs]
// tokamak-cli spawns it as `trusted-setup --output DIR --fixed-tau` (packages/cli/src/runtime.ts:1840-1848); without the flag the
// library is resolved the way host/tkmk_args.hpp describes.
// reads <lib>/setupParams.json, <lib>/subcircuitInfo.json, <lib>/r1cs/subcircuit{id}.r1cs; writes the reference's containers
// <out>/combined_sigma.rkyv and <out>/sigma_preprocess.rkyv (write_final_crs_artifacts, libs/src/iotools/mod.rs:271-300; host/tkmk_rkyv.hpp)
// <out>/sigma_verify.json (the verifier's part of the reference string, libs/src/iotools/mod.rs:186-218, 295-297)
// and <out>/combined_sigma.tkcrs (the flat TKCRS001 payload the reference derives from its archive: tkmk/crs.py — the fast path of
// `prove`).  --format picks one; a reference string past 2 GiB does not fit an rkyv archive (32-bit relative pointers) and is written
// as .tkcrs only.  --fixed-tau uses the hardcoded testing generators and tau of the
// reference (main.rs:68-80, libs/src/field_structures/mod.rs:43-64); otherwise tau is drawn from std::random_device and the generators are
// random multiples of the standard ones.  Needs an MI355X: no CPU fallback.
#include <chrono>
#include <cstdio>
#include <random>
#include <string>

#include "tkmk_args.hpp"
#include "tkmk_json.hpp"
#include "tkmk_setup.hpp"

using namespace tkmk;

static double Prover_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static ScalarField random_fr(std::random_device &rd) {
    uint8_t b[32];
    for (int i = 0; i < 32; i += 4) {
        uint32_t w = rd();
        std::memcpy(b + i, &w, 4);
    }
    ScalarField v = fr_from_le_bytes_mod_r(b, 32);
    return fr_is_zero(v) ? fr_one() : v;
}
static G1Affine g1_from_hex(const std::string &x, const std::string &y) {
    auto fq = [](const std::string &h) {
        tkmk_fq v{};
        size_t off = 2, nd = h.size() - off;
        uint8_t le[48] = {};
        for (size_t k = 0; k < nd && k < 96; k++) {
            char c = h[h.size() - 1 - k];
            int d = c >= '0' && c <= '9' ? c - '0' : c - 'a' + 10;
            le[k / 2] |= (uint8_t)(d << (4 * (k & 1)));
        }
        std::memcpy(v.limbs, le, 48);
        return v;
    };
    G1Affine p;
    p.x = fq(x), p.y = fq(y);
    return p;
}

static const char *USAGE =
    "Usage: trusted-setup --output <PATH> [--fixed-tau] [--subcircuit-library <PATH>] [--format both|rkyv|tkcrs]\n"
    "  --output            Output directory for combined_sigma.rkyv, sigma_preprocess.rkyv, combined_sigma.tkcrs\n"
    "  --fixed-tau         Use the hardcoded testing generators and tau\n"
    "  --subcircuit-library  Subcircuit library directory produced by the QAP compiler (default: see host/tkmk_args.hpp)\n";

int main(int argc, char **argv) {
    args::Spec spec{{"--output", "--subcircuit-library", "--format"}, {"--fixed-tau"}};
    args::Parsed a = args::parse(argc, argv, spec);
    if (a.help) {
        fputs(USAGE, stdout);
        return 0;
    }
    if (a.version) {
        printf("trusted-setup %s\n", TKMK_BACKEND_INTERFACE_VERSION);
        return 0;
    }
    if (a.error.empty() && !a.has("--output")) a.error = "the following required arguments were not provided: --output <PATH>";
    const std::string format = a.get("--format", "both"), out_dir = a.get("--output");
    if (a.error.empty() && format != "both" && format != "rkyv" && format != "tkcrs") a.error = "invalid value '" + format + "' for '--format'";
    if (!a.error.empty()) {
        fprintf(stderr, "error: %s\n\n%s", a.error.c_str(), USAGE);
        return 2;
    }
    const bool fixed_tau = a.flag("--fixed-tau");
    try {
        double t0 = Prover_now();
        const std::string lib_dir = args::resolve_subcircuit_library(a);
        printf("Subcircuit library: %s\n", lib_dir.c_str());   // which circuit this run is for (the reference names it by an embedded hash)
        fflush(stdout);
        int ndev = 0;
        if (tkmk_device_count(&ndev) != TKMK_SUCCESS || ndev < 1) throw Error("no HIP device: the MI355X backend has no CPU fallback");
        check(tkmk_set_device(0), "set_device");
        json::Value jp = json::read_file(lib_dir + "/setupParams.json");
        SetupParams sp{jp.at("l").as_size(),   jp.at("l_user_out").as_size(), jp.at("l_user").as_size(), jp.at("l_free").as_size(),
                       jp.at("l_D").as_size(), jp.at("m_D").as_size(),        jp.at("n").as_size(),      jp.at("s_D").as_size(),
                       jp.at("s_max").as_size()};
        std::vector<SubcircuitInfo> infos;
        std::vector<size_t> n_consts;
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
                infos.push_back(std::move(si));
                n_consts.push_back(e.at("Nconsts").as_size());
            }
        }
        Tau tau;
        G1Affine g1;
        g2h::Affine h2;
        if (fixed_tau) {
            printf("Using hardcoded G1, G2 generators and tau\n");
            tau = Tau{fr_from_hex("0x7234cd9b97845e0125e84ae3ae81354e004558d8c82a83425652bc7b9ed49f7d"),
                      fr_from_hex("0x6ed0eea55cbeeebdc7a41033ebd196ffecc1806fdbc13a8d41b8f1aa273a4037"),
                      fr_from_hex("0x7234cd9b97845e0125e84ae3ae81354e004558d8c82a83425652bc7b9ed49f7d"),
                      fr_from_hex("0x088dfe3d1b76775ec267d6d0e27b753ec904c76e0bc32ca8223dc2ae1a0ac6b4"),
                      fr_from_hex("0x04b8ce26374c547d8722ac51f5ed1e0f9cb891c332c69c865d96af150189a818"),
                      fr_from_hex("0x52eb2aeb35b72b94a19ea232e984850f2cda5542fdc10368955d8ac6274f8579")};
            g1 = g1_from_hex("0x0b001b4cc05fa01578be7d4e821d6ff58f2a05c584fba3cb31a37942dece65eadec9a878add2282f7c2513abb8d4ab05",
                             "0x15e237775397ed22eef43dd36cdca277c9cf6fa7e4ffff0a5bb4b20a82392caacf0f63fb6cdb02bccf2f5af14970d6b9");
            h2.x = g2h::f2_from_hex("0x1116094a7c01d4fd8abcfea69c658c92c037765bee00556b8d4063c33540b316ac68a2d913d3adc3b43c7d7cc7505cfc17206c8ae661f247979b3f1daa7fb6d5f7ce9c17b5ed1d7e8b421a2508b3f09a603e6a5fab3fcde7364fd178d656ac36");
            h2.y = g2h::f2_from_hex("0x15bf297a4b9842fb1a3a6f2dbf6b94de06997b11b2f72436c22efbb48d2f74b0de7239ea182a2ee50c23ae3d0be6fdee09459611409874fe4b04b1a7e42cb84eb4ae01728dc55dbd1343fda8d0fe94a299fc757acc1d2602a49a005b4ff90190");
        } else {
            std::random_device rd;
            tau = Tau{random_fr(rd), random_fr(rd), random_fr(rd), random_fr(rd), random_fr(rd), random_fr(rd)};
            G1Affine std_g1 = g1_from_hex("0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb",
                                          "0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1");
            std::vector<ScalarField> h = {random_fr(rd)};
            g1 = setup_detail::points(h, std_g1).to_host().at(0);
            g2h::Affine std_g2;
            std_g2.x = {g2h::f2_from_hex("0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")};
            std_g2.y = {g2h::f2_from_hex("0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801")};
            h2 = g2h::scalar_mul(random_fr(rd), std_g2);
        }
        printf("Setup parameters: n = %zu, s_max = %zu, l = %zu, l_free = %zu, m_I = %zu, m_D = %zu\n", sp.n, sp.s_max, sp.l, sp.l_free, sp.l_D - sp.l,
               sp.m_D);
        Sigma sigma = Sigma::gen(sp, tau, lib_dir, infos, n_consts, g1, &h2);
        check(tkmk_device_synchronize(), "synchronize");
        double t1 = Prover_now();
        printf("The sigma generation time: %.6f seconds\n", t1 - t0);
        bool fits = sigma.archive_bytes() < (1ull << 31);
        if (format == "rkyv" && !fits) throw Error("this reference string does not fit an rkyv archive (32-bit relative pointers); use --format tkcrs");
        if (format != "tkcrs" && fits) {
            sigma.write_rkyv(out_dir, sp);
            printf("combined_sigma.rkyv and sigma_preprocess.rkyv written to %s\n", out_dir.c_str());
        } else if (format == "both") {
            printf("reference string past 2 GiB: no rkyv archive (32-bit relative pointers), .tkcrs payload only\n");
        }
        if (format != "rkyv") {
            std::string path = sigma.write(out_dir);
            printf("combined_sigma.tkcrs written to %s\n", path.c_str());
        }
        if (sigma.has_sigma2()) printf("sigma_verify.json written to %s\n", sigma.write_sigma_verify(out_dir).c_str());
        printf("Total: %.3f s\n", Prover_now() - t0);
    } catch (const std::exception &ex) {
        fprintf(stderr, "trusted-setup: %s\n", ex.what());
        return 1;
    }
    return 0;
}
