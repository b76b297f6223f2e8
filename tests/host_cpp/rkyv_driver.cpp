// rkyv_driver.cpp — CPU-side test driver for host/tkmk_rkyv.hpp (no device is touched):
//   rkyv_driver decode ARCHIVE OUT.tkcrs [xy gamma eta delta rs_y | -] [order|auto] [nocheck]
//       -> prints the field order that validated; writes the nine sections as a TKCRS001 payload
//   rkyv_driver encode IN.tkcrs OUT.rkyv ORDER ETA_ROWS DELTA_ROWS   (rows of equal length; xh = 3x3, yi = 4x3)
//   rkyv_driver pre-decode ARCHIVE OUT.bin     -> xy_powers || gamma_inv_o_inst, sizes printed
//   rkyv_driver pre-encode IN.tkcrs OUT.rkyv
// tests/test_rkyv.py compares the results with tkmk/rkyv.py byte for byte.
#include <cstdio>
#include <fstream>
#include <iterator>

#include "tkmk_rkyv.hpp"

using namespace tkmk;

static std::vector<uint8_t> slurp(const std::string &p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) throw Error("cannot open " + p);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string &p, const std::vector<uint8_t> &v) {
    std::ofstream f(p, std::ios::binary);
    f.write((const char *)v.data(), (std::streamsize)v.size());
    if (!f) throw Error("cannot write " + p);
}
static rkyv::FieldOrder order_of(const std::string &s) {
    if (s == "rustc_size_groups") return rkyv::FieldOrder::RustcSizeGroups;
    if (s == "rustc_align_only") return rkyv::FieldOrder::RustcAlignOnly;
    if (s == "declared") return rkyv::FieldOrder::Declared;
    throw Error("unknown field order " + s);
}
static const char *order_key(rkyv::FieldOrder o) {
    return o == rkyv::FieldOrder::RustcSizeGroups ? "rustc_size_groups" : o == rkyv::FieldOrder::RustcAlignOnly ? "rustc_align_only" : "declared";
}

int main(int argc, char **argv) {
    try {
        std::string cmd = argc > 1 ? argv[1] : "";
        if (cmd == "decode" && argc >= 4) {
            auto bytes = std::make_shared<std::vector<uint8_t>>(slurp(argv[2]));
            rkyv::Expect ex;
            int a = 4;
            if (a < argc && std::string(argv[a]) == "-") a++;
            else if (a + 4 < argc) {
                ex.xy_powers = std::stoull(argv[a]), ex.gamma = std::stoull(argv[a + 1]), ex.eta = std::stoull(argv[a + 2]), ex.delta = std::stoull(argv[a + 3]);
                ex.rs_y = std::stoull(argv[a + 4]);
                a += 5;
            }
            std::vector<rkyv::FieldOrder> orders = {rkyv::FieldOrder::RustcSizeGroups, rkyv::FieldOrder::RustcAlignOnly, rkyv::FieldOrder::Declared};
            if (a < argc && std::string(argv[a]) != "auto") orders = {order_of(argv[a])};
            a++;
            if (a < argc && std::string(argv[a]) == "nocheck") ex.check_points = false;
            rkyv::FieldOrder got;
            CrsPayload c = rkyv::decode_combined_sigma(CrsPayload::View{bytes->data(), bytes->size()}, bytes, ex, &got, orders);
            std::vector<uint8_t> out = {'T', 'K', 'C', 'R', 'S', '0', '0', '1'};
            auto u32 = [&](uint32_t v) { out.insert(out.end(), (uint8_t *)&v, (uint8_t *)&v + 4); };
            u32((uint32_t)CrsPayload::Count);
            for (int i = 0; i < CrsPayload::Count; i++) u32((uint32_t)c.length[i]);
            for (int i = 0; i < CrsPayload::Count; i++) out.insert(out.end(), c.section[i], c.section[i] + c.length[i]);
            spit(argv[3], out);
            printf("%s\n", order_key(got));
            return 0;
        }
        if (cmd == "encode" && argc == 7) {
            CrsPayload c = CrsPayload::parse(slurp(argv[2]));
            size_t eta_rows = std::stoull(argv[5]), delta_rows = std::stoull(argv[6]);
            auto rows = [&](CrsPayload::Section s, size_t r) {
                if (r == 0) return std::vector<size_t>{};
                if (c.points(s) % r) throw Error("rows do not divide the section");
                return std::vector<size_t>(r, c.points(s) / r);
            };
            rkyv::SigmaTables t{c.bytes(CrsPayload::G1Singles),
                                c.bytes(CrsPayload::XyPowers), c.points(CrsPayload::XyPowers),
                                c.bytes(CrsPayload::GammaInvOInst), c.points(CrsPayload::GammaInvOInst),
                                c.bytes(CrsPayload::EtaInvLiOInterAlpha4Kj), rows(CrsPayload::EtaInvLiOInterAlpha4Kj, eta_rows),
                                c.bytes(CrsPayload::DeltaInvLiOPrv), rows(CrsPayload::DeltaInvLiOPrv, delta_rows),
                                c.bytes(CrsPayload::DeltaInvAlphakXhTx), rows(CrsPayload::DeltaInvAlphakXhTx, 3),
                                c.bytes(CrsPayload::DeltaInvAlpha4XjTx), c.points(CrsPayload::DeltaInvAlpha4XjTx),
                                c.bytes(CrsPayload::DeltaInvAlphakYiTy), rows(CrsPayload::DeltaInvAlphakYiTy, 4),
                                c.bytes(CrsPayload::G2Points)};
            spit(argv[3], rkyv::encode_combined_sigma(t, order_of(argv[4])));
            return 0;
        }
        if (cmd == "pre-decode" && argc == 4) {
            std::vector<uint8_t> b = slurp(argv[2]);
            rkyv::PreprocessSigma ps = rkyv::decode_sigma_preprocess(b.data(), b.size());
            std::vector<uint8_t> out(ps.xy_powers, ps.xy_powers + ps.xy_points * 96);
            out.insert(out.end(), ps.gamma_inv_o_inst, ps.gamma_inv_o_inst + ps.gamma_points * 96);
            spit(argv[3], out);
            printf("%zu %zu\n", ps.xy_points, ps.gamma_points);
            return 0;
        }
        if (cmd == "pre-encode" && argc == 4) {
            CrsPayload c = CrsPayload::parse(slurp(argv[2]));
            spit(argv[3], rkyv::encode_sigma_preprocess(c.bytes(CrsPayload::XyPowers), c.points(CrsPayload::XyPowers), c.bytes(CrsPayload::GammaInvOInst),
                                                        c.points(CrsPayload::GammaInvOInst)));
            return 0;
        }
        fprintf(stderr, "usage: rkyv_driver decode|encode|pre-decode|pre-encode ...\n");
        return 2;
    } catch (const std::exception &e) {
        fprintf(stderr, "rkyv_driver: %s\n", e.what());
        return 1;
    }
}
