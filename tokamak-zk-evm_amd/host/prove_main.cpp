// prove_main.cpp — native `prove` with the reference binary's argument surface (packages/backend/prove/src/main.rs:8-97):
//   prove --crs DIR --synthesizer-stat DIR --output DIR --subcircuit-library DIR
// reads <lib>/setupParams.json, <lib>/subcircuitInfo.json, <lib>/r1cs/subcircuit{id}.r1cs, <synth>/placementVariables.json,
// <synth>/permutation.json, <synth>/instance.json and <crs>/combined_sigma.tkcrs (the flat TKCRS001 payload the reference derives from
// its rkyv archive; the archive itself is not parsed); writes <out>/proof.json in the Solidity-verifier format.  Exit code 0 on
// success; any failure prints the reason and exits non-zero (the reference panics).  Needs an MI355X: no CPU fallback.
// Test hook (not in the reference): TKMK_PROVE_MIXER=<file.json> fixes the blinding scalars so two implementations can be compared
// byte for byte; without it they come from std::random_device (ScalarCfg::generate_random in the reference, lib.rs:1040-1080).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "tkmk_inputs.hpp"
#include "tkmk_prover.hpp"

using namespace tkmk;

static Mixer mixer_from_json(const json::Value &j) {
    auto one = [&](const char *k) { return fr_from_hex(j.at(k).as_string()); };
    Mixer m;
    m.rU_X = one("rU_X"), m.rU_Y = one("rU_Y"), m.rV_X = one("rV_X"), m.rV_Y = one("rV_Y");
    m.rO_mid = one("rO_mid"), m.rR_X = one("rR_X"), m.rR_Y = one("rR_Y");
    auto fill = [&](const char *k, ScalarField *dst, size_t n) {
        std::vector<ScalarField> v = hex_list(j.at(k));
        if (v.size() != n) throw Error(std::string("mixer field ") + k + " has the wrong length");
        for (size_t i = 0; i < n; i++) dst[i] = v[i];
    };
    fill("rW_X", m.rW_X.data(), 4), fill("rW_Y", m.rW_Y.data(), 4), fill("rB_X", m.rB_X.data(), 2), fill("rB_Y", m.rB_Y.data(), 2);
    return m;
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
        fprintf(stderr, "usage: prove --crs DIR --synthesizer-stat DIR --output DIR --subcircuit-library DIR\n");
        return 2;
    }
    try {
        double t_start = Prover::now();
        int ndev = 0;
        if (tkmk_device_count(&ndev) != TKMK_SUCCESS || ndev < 1) throw Error("no HIP device: the MI355X backend has no CPU fallback");
        check(tkmk_set_device(0), "set_device");   // check_device(): device id 0 (libs/src/utils/mod.rs:88-110)
        void *warm = nullptr;                        // first allocation: the HIP runtime and the code objects come up here
        check(tkmk_malloc(&warm, 256), "malloc");
        check(tkmk_free(warm), "free");
        double t_dev = Prover::now();
        printf("Prover initialization...\n");
        ProverInputs in;
        in.qap_path = lib_dir;
        json::Value jp = json::read_file(lib_dir + "/setupParams.json");
        in.sp = SetupParams{jp.at("l").as_size(),   jp.at("l_user_out").as_size(), jp.at("l_user").as_size(), jp.at("l_free").as_size(),
                            jp.at("l_D").as_size(), jp.at("m_D").as_size(),        jp.at("n").as_size(),      jp.at("s_D").as_size(),
                            jp.at("s_max").as_size()};
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
                in.infos.push_back(std::move(si));
                in.n_consts.push_back(e.at("Nconsts").as_size());
            }
        }
        in.pv = read_placement_variables(synth_dir + "/placementVariables.json");
        {
            const json::Value jperm = json::read_file(synth_dir + "/permutation.json");
            for (const json::Value &e : jperm.items())
                in.perm.push_back({e.at("row").as_size(), e.at("col").as_size(), e.at("X").as_size(), e.at("Y").as_size()});
        }
        {
            const json::Value jinst = json::read_file(synth_dir + "/instance.json");
            in.a_pub_user = hex_list(jinst.at("a_pub_user"));
            in.a_pub_block = hex_list(jinst.at("a_pub_block"));
        }
        std::string crs_path = crs_dir + "/combined_sigma.tkcrs";
        if (!std::ifstream(crs_path)) throw Error("No reference string is found. Run the Setup first (expected " + crs_path + ").");
        double t_load = Prover::now();
        CrsPayload crs = CrsPayload::read(crs_path);
        ProverSigma sigma = ProverSigma::from_payload(crs, in.sp);
        double t_crs = Prover::now();
        const char *mixer_file = std::getenv("TKMK_PROVE_MIXER");
        Mixer mixer = mixer_file ? mixer_from_json(json::read_file(mixer_file)) : Mixer::random();

        auto pb = Prover::init(in, sigma, mixer);
        check(tkmk_device_synchronize(), "synchronize");
        std::map<std::string, double> times;
        Proof proof = run_rounds(*pb.first, pb.second, &times);

        printf("Writing the proof into JSON (formatted for Solidity verifier)...\n");
        std::string path = out_dir + "/proof.json";
        std::ofstream f(path);
        if (!f) throw Error("cannot write " + path);
        f << proof.to_json();
        f.close();
        printf("device.init %.3f s\nload.inputs %.3f s\nload.crs    %.3f s\n", t_dev - t_start, t_load - t_dev, t_crs - t_load);
        for (auto &kv : pb.first->timing) printf("%-11s %.3f s\n", kv.first.c_str(), kv.second);
        for (const char *k : {"prove0", "prove1", "prove2", "prove3", "prove4"}) printf("%-11s %.3f s\n", k, times[k]);
        double total = Prover::now() - t_start;
        printf("Prove completed. Total elapsed time: %.3fs (%.0f ms)\n", total, 1e3 * total);
    } catch (const std::exception &ex) {
        fprintf(stderr, "prove: %s\n", ex.what());
        return 1;
    }
    return 0;
}
