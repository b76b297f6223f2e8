// prove_main.cpp — native `prove` with the reference binary's argument surface (packages/backend/prove/src/main.rs:8-97):
//   prove --crs DIR --synthesizer-stat DIR --output DIR --subcircuit-library DIR
// reads <lib>/setupParams.json, <lib>/subcircuitInfo.json, <lib>/r1cs/subcircuit{id}.r1cs, <synth>/placementVariables.json,
// <synth>/permutation.json, <synth>/instance.json and the reference string <crs>/combined_sigma.rkyv (the reference's archive,
// host/tkmk_rkyv.hpp; <crs>/combined_sigma.tkcrs, the flat payload, is taken instead when present); writes <out>/proof.json in the
// Solidity-verifier format.  Exit code 0 on success; any failure prints the reason and exits non-zero (the reference panics).
// Needs an MI355X: no CPU fallback.  One process per proof = ProverContext::open + ::prove (host/tkmk_service.hpp); a host that
// proves repeatedly keeps the context (include/tkmk_prover.h).
// Test hook (not in the reference): --testing-mixer FILE fixes the blinding scalars so that two implementations can be compared
// byte for byte; a proof made that way is NOT zero-knowledge and the binary says so.  Without it they come from getrandom()
// (ScalarCfg::generate_random in the reference, lib.rs:1040-1080).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "tkmk_crs_load.hpp"
#include "tkmk_service.hpp"

using namespace tkmk;

int main(int argc, char **argv) {
    std::string crs_dir, synth_dir, out_dir, lib_dir, mixer_file;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i], v = argv[i + 1];
        if (k == "--crs") crs_dir = v;
        else if (k == "--synthesizer-stat") synth_dir = v;
        else if (k == "--output") out_dir = v;
        else if (k == "--subcircuit-library") lib_dir = v;
        else if (k == "--testing-mixer") mixer_file = v;
        else {
            fprintf(stderr, "unknown argument %s\n", k.c_str());
            return 2;
        }
    }
    if (crs_dir.empty() || synth_dir.empty() || out_dir.empty() || lib_dir.empty() || argc % 2 == 0) {
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
        auto ctx = ProverContext::open(lib_dir, crs_dir, [&](const SetupParams &sp, std::string &source) { return load_prover_sigma(crs_dir, sp, source); });
        double t_open = Prover::now();
        Mixer mixer;
        if (!mixer_file.empty()) {
            fprintf(stderr, "WARNING: --testing-mixer: blinding scalars are read from %s; this proof is NOT zero-knowledge. Testing only.\n", mixer_file.c_str());
            mixer = mixer_from_json(json::read_file(mixer_file));
        } else {
            mixer = Mixer::random();
        }
        ProveTiming tm;
        ctx->prove(synth_dir, out_dir, mixer, &tm);
        printf("Writing the proof into JSON (formatted for Solidity verifier)...\n");
        printf("device.init %.3f s\nopen.context %.3f s   (subcircuit library + %s -> HBM)\n", t_dev - t_start, t_open - t_dev, ctx->crs_source.c_str());
        printf("init.parse  %.3f s\ninit.upload %.3f s\ninit.build  %.3f s\ninit.binding %.3f s\ninit.total  %.3f s\n", tm.parse, tm.upload, tm.build, tm.binding, tm.init);
        for (int k = 0; k < 5; k++) printf("prove%d      %.3f s\n", k, tm.prove[k]);
        double total = Prover::now() - t_start;
        printf("Prove completed. Total elapsed time: %.3fs (%.0f ms)\n", total, 1e3 * total);
    } catch (const std::exception &ex) {
        fprintf(stderr, "prove: %s\n", ex.what());
        return 1;
    }
    return 0;
}
