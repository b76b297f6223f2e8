// preprocess_main.cpp — native `preprocess` with the reference binary's argument surface (packages/backend/preprocess/src/main.rs:12-63):
//   preprocess --crs DIR --synthesizer-stat DIR --output DIR --subcircuit-library DIR
// reads <lib>/setupParams.json, <synth>/permutation.json, <synth>/instance.json, <crs>/combined_sigma.tkcrs (the flat TKCRS001 payload
// the reference derives from its rkyv archive; the archive itself is not parsed), writes <out>/preprocess.json.  Exit code 0 on
// success; any failure prints the reason and exits non-zero (the reference panics).  Needs an MI355X: no CPU fallback.
#include <cstdio>
#include <cstring>
#include <string>

#include "tkmk_json.hpp"
#include "tkmk_protocol.hpp"

using namespace tkmk;

// ScalarField::from_hex on a HexString (libs/src/iotools/mod.rs:126-146): optional 0x, big-endian digits, reduced mod r
static ScalarField fr_from_hex(const std::string &h) {
    size_t off = h.rfind("0x", 0) == 0 || h.rfind("0X", 0) == 0 ? 2 : 0;
    std::string d = h.substr(off);
    if (d.size() > 64) throw Error("hex scalar longer than 32 bytes");
    if (d.size() % 2) d = "0" + d;
    uint8_t le[32] = {};
    size_t nb = d.size() / 2;
    for (size_t i = 0; i < nb; i++) le[nb - 1 - i] = (uint8_t)std::stoi(d.substr(2 * i, 2), nullptr, 16);
    static const uint32_t R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    ScalarField v{};
    std::memcpy(&v, le, 32);
    auto geq = [&]() {
        for (int i = 7; i >= 0; i--)
            if (v.limbs[i] != R[i]) return v.limbs[i] > R[i];
        return true;
    };
    while (geq()) {
        uint64_t br = 0;
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)v.limbs[i] - R[i] - br;
            v.limbs[i] = (uint32_t)t;
            br = (t >> 63) & 1;
        }
    }
    return v;
}

int main(int argc, char **argv) {
    std::string crs_dir, synth_dir, out_dir, lib_dir;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i], v = argv[i + 1];
        if (k == "--crs") crs_dir = v;
        else if (k == "--synthesizer-stat") synth_dir = v;
        else if (k == "--output") out_dir = v;
        else if (k == "--subcircuit-library") lib_dir = v;
        else {
            fprintf(stderr, "unknown argument %s\n", k.c_str());
            return 2;
        }
    }
    if (crs_dir.empty() || synth_dir.empty() || out_dir.empty() || lib_dir.empty()) {
        fprintf(stderr, "usage: preprocess --crs DIR --synthesizer-stat DIR --output DIR --subcircuit-library DIR\n");
        return 2;
    }
    try {
        int ndev = 0;
        if (tkmk_device_count(&ndev) != TKMK_SUCCESS || ndev < 1) throw Error("no HIP device: the MI355X backend has no CPU fallback");
        check(tkmk_set_device(0), "set_device");   // check_device(): device id 0 (libs/src/utils/mod.rs:88-110)
        json::Value jp = json::read_file(lib_dir + "/setupParams.json");
        SetupParams sp{jp.at("l").as_size(),   jp.at("l_user_out").as_size(), jp.at("l_user").as_size(), jp.at("l_free").as_size(),
                       jp.at("l_D").as_size(), jp.at("m_D").as_size(),        jp.at("n").as_size(),      jp.at("s_D").as_size(),
                       jp.at("s_max").as_size()};
        CrsPayload crs = CrsPayload::read(crs_dir + "/combined_sigma.tkcrs");
        size_t m_i = sp.l_D - sp.l, rs_x = std::max(2 * sp.n, 2 * m_i), rs_y = 2 * sp.s_max;
        if (crs.points(CrsPayload::XyPowers) != rs_x * rs_y || crs.points(CrsPayload::GammaInvOInst) != sp.l)
            throw Error("CRS sections do not match setupParams.json");
        std::vector<Permutation> perm;
        const json::Value jperm = json::read_file(synth_dir + "/permutation.json");   // named: items() refers into it
        for (const json::Value &e : jperm.items())
            perm.push_back({e.at("row").as_size(), e.at("col").as_size(), e.at("X").as_size(), e.at("Y").as_size()});
        std::vector<ScalarField> a_fn;
        const json::Value jinst = json::read_file(synth_dir + "/instance.json");
        for (const json::Value &e : jinst.at("a_pub_function").items()) a_fn.push_back(fr_from_hex(e.as_string()));
        Sigma1 sigma(crs.upload(CrsPayload::XyPowers), rs_x, rs_y);
        DeviceVec<G1Affine> gamma = crs.upload(CrsPayload::GammaInvOInst);
        Preprocess pre = Preprocess::gen(sigma, gamma, perm, a_fn, sp);
        std::string path = out_dir + "/preprocess.json";
        std::ofstream f(path);
        if (!f) throw Error("cannot write " + path);
        f << pre.to_json();
        printf("preprocess.json written to %s\n", out_dir.c_str());
    } catch (const std::exception &ex) {
        fprintf(stderr, "preprocess: %s\n", ex.what());
        return 1;
    }
    return 0;
}
