// preprocess_main.cpp — native `preprocess` with the reference binary's argument surface (packages/backend/preprocess/src/main.rs:12-63):
//   preprocess --crs DIR --synthesizer-stat DIR --output DIR [--subcircuit-library DIR]
// tokamak-cli spawns it with the first three flags only (packages/cli/src/cli.ts:537-546); the library is then resolved the way
// host/tkmk_args.hpp describes (the reference's release build embeds it: libs/src/subcircuit_library.rs:41-58).
// reads <lib>/setupParams.json, <synth>/permutation.json, <synth>/instance.json and the reference string <crs>/sigma_preprocess.rkyv
// (the reference's archive: preprocess/src/main.rs:47-53; host/tkmk_rkyv.hpp) — or <crs>/combined_sigma.tkcrs / combined_sigma.rkyv
// when that is what the directory holds; writes <out>/preprocess.json.  Exit code 0 on success; any failure prints the reason
// and exits non-zero (the reference panics).  Needs an MI355X: no CPU fallback.
#include <cstdio>
#include <cstring>
#include <string>

#include "tkmk_args.hpp"
#include "tkmk_crs_load.hpp"

using namespace tkmk;

static const char *USAGE =
    "Usage: preprocess --crs <PATH> --synthesizer-stat <PATH> --output <PATH> [--subcircuit-library <PATH>]\n"
    "  --crs               CRS output directory containing preprocess setup artifacts\n"
    "  --synthesizer-stat  Synthesizer output directory containing preprocess inputs\n"
    "  --output            Output directory for preprocess.json\n"
    "  --subcircuit-library  Subcircuit library directory produced by the QAP compiler (default: see host/tkmk_args.hpp)\n";

int main(int argc, char **argv) {
    args::Spec spec{{"--crs", "--synthesizer-stat", "--output", "--subcircuit-library"}, {}};
    args::Parsed a = args::parse(argc, argv, spec);
    if (a.help) {
        fputs(USAGE, stdout);
        return 0;
    }
    if (a.version) {
        printf("preprocess %s\n", TKMK_BACKEND_INTERFACE_VERSION);
        return 0;
    }
    for (const char *need : {"--crs", "--synthesizer-stat", "--output"})
        if (a.error.empty() && !a.has(need)) a.error = std::string("the following required arguments were not provided: ") + need + " <PATH>";
    if (!a.error.empty()) {
        fprintf(stderr, "error: %s\n\n%s", a.error.c_str(), USAGE);
        return 2;
    }
    const std::string crs_dir = a.get("--crs"), synth_dir = a.get("--synthesizer-stat"), out_dir = a.get("--output");
    try {
        const std::string lib_dir = args::resolve_subcircuit_library(a);
        printf("Subcircuit library: %s\n", lib_dir.c_str());   // which circuit this run is for (the reference names it by an embedded hash)
        fflush(stdout);
        int ndev = 0;
        if (tkmk_device_count(&ndev) != TKMK_SUCCESS || ndev < 1) throw Error("no HIP device: the MI355X backend has no CPU fallback");
        check(tkmk_set_device(0), "set_device");   // check_device(): device id 0 (libs/src/utils/mod.rs:88-110)
        json::Value jp = json::read_file(lib_dir + "/setupParams.json");
        SetupParams sp{jp.at("l").as_size(),   jp.at("l_user_out").as_size(), jp.at("l_user").as_size(), jp.at("l_free").as_size(),
                       jp.at("l_D").as_size(), jp.at("m_D").as_size(),        jp.at("n").as_size(),      jp.at("s_D").as_size(),
                       jp.at("s_max").as_size()};
        // the reference reads <crs>/sigma_preprocess.rkyv (preprocess/src/main.rs:47-53: xy_powers + gamma_inv_o_inst only); the flat
        // payload or the combined archive serve as well when that is what the directory holds
        size_t m_i = sp.l_D - sp.l, rs_x = std::max(2 * sp.n, 2 * m_i), rs_y = 2 * sp.s_max;
        const G1Affine *xy_host = nullptr, *gamma_host = nullptr;
        size_t xy_points = 0, gamma_points = 0;
        CrsPayload crs;
        std::shared_ptr<void> keep;
        const std::string pre_archive = crs_dir + "/sigma_preprocess.rkyv";
        if (!file_exists(crs_dir + "/combined_sigma.tkcrs") && file_exists(pre_archive)) {
            auto m = map_file(pre_archive);
            keep = m.second;
            rkyv::PreprocessSigma ps = rkyv::decode_sigma_preprocess(m.first.data(), m.first.size());
            xy_host = reinterpret_cast<const G1Affine *>(ps.xy_powers), xy_points = ps.xy_points;
            gamma_host = reinterpret_cast<const G1Affine *>(ps.gamma_inv_o_inst), gamma_points = ps.gamma_points;
        } else {
            crs = load_combined_sigma(crs_dir, sp);
            xy_host = crs.g1(CrsPayload::XyPowers), xy_points = crs.points(CrsPayload::XyPowers);
            gamma_host = crs.g1(CrsPayload::GammaInvOInst), gamma_points = crs.points(CrsPayload::GammaInvOInst);
        }
        if (xy_points != rs_x * rs_y || gamma_points != sp.l) throw Error("CRS sections do not match setupParams.json");
        std::vector<Permutation> perm;
        const json::Value jperm = json::read_file(synth_dir + "/permutation.json");   // named: items() refers into it
        for (const json::Value &e : jperm.items())
            perm.push_back({e.at("row").as_size(), e.at("col").as_size(), e.at("X").as_size(), e.at("Y").as_size()});
        std::vector<ScalarField> a_fn;
        const json::Value jinst = json::read_file(synth_dir + "/instance.json");
        for (const json::Value &e : jinst.at("a_pub_function").items()) a_fn.push_back(fr_from_hex(e.as_string()));
        Sigma1 sigma(DeviceVec<G1Affine>::from_host(xy_host, xy_points), rs_x, rs_y);
        DeviceVec<G1Affine> gamma = DeviceVec<G1Affine>::from_host(gamma_host, gamma_points);
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
