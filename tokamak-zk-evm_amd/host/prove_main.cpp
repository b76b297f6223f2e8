// prove_main.cpp — native `prove` with the reference binary's argument surface (packages/backend/prove/src/main.rs:8-97):
//   prove --crs DIR --synthesizer-stat DIR --output DIR [--subcircuit-library DIR]
// tokamak-cli spawns it with the first three flags only (packages/cli/src/cli.ts:537-546); the library is then resolved the way
// host/tkmk_args.hpp describes (the reference's release build embeds it: libs/src/subcircuit_library.rs:41-58).
// reads <lib>/setupParams.json, <lib>/subcircuitInfo.json, <lib>/r1cs/subcircuit{id}.r1cs, <synth>/placementVariables.json,
// <synth>/permutation.json, <synth>/instance.json and the reference string <crs>/combined_sigma.rkyv (the reference's archive,
// host/tkmk_rkyv.hpp; <crs>/combined_sigma.tkcrs, the flat payload, is taken instead when present); writes <out>/proof.json in the
// Solidity-verifier format.  Exit code 0 on success; any failure prints the reason and exits non-zero (the reference panics).
// Needs an MI355X: no CPU fallback.  One process per proof = ProverContext::open + ::prove (host/tkmk_service.hpp); a host that
// proves repeatedly keeps the context (include/tkmk_prover.h).
// Blinding scalars come from getrandom() (ScalarCfg::generate_random in the reference, lib.rs:1040-1080).  The test hook
// --testing-mixer FILE (fixed blinding scalars, so that two implementations can be compared byte for byte; such a proof is NOT
// zero-knowledge) exists only in the -DTKMK_TESTING_MODE build of this file, bin/prove-testing — the reference gates its
// equivalent behind the compile-time feature `testing-mode` the same way; bin/prove refuses the flag.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "tkmk_args.hpp"
#include "tkmk_crs_load.hpp"
#include "tkmk_service.hpp"

using namespace tkmk;

static const char *USAGE =
    "Usage: prove --crs <PATH> --synthesizer-stat <PATH> --output <PATH> [--subcircuit-library <PATH>]\n"
    "  --crs               CRS output directory containing proof setup artifacts\n"
    "  --synthesizer-stat  Synthesizer output directory containing proving inputs\n"
    "  --output            Output directory for proof.json\n"
    "  --subcircuit-library  Subcircuit library directory produced by the QAP compiler (default: see host/tkmk_args.hpp)\n";

int main(int argc, char **argv) {
    args::Spec spec{{"--crs", "--synthesizer-stat", "--output", "--subcircuit-library", "--testing-mixer"}, {}};
    args::Parsed a = args::parse(argc, argv, spec);
    if (a.help) {
        fputs(USAGE, stdout);
        return 0;
    }
    if (a.version) {
        printf("prove %s\n", TKMK_BACKEND_INTERFACE_VERSION);
        return 0;
    }
    for (const char *need : {"--crs", "--synthesizer-stat", "--output"})
        if (a.error.empty() && !a.has(need)) a.error = std::string("the following required arguments were not provided: ") + need + " <PATH>";
    if (!a.error.empty()) {
        fprintf(stderr, "error: %s\n\n%s", a.error.c_str(), USAGE);
        return 2;
    }
    const std::string crs_dir = a.get("--crs"), synth_dir = a.get("--synthesizer-stat"), out_dir = a.get("--output"), mixer_file = a.get("--testing-mixer");
#ifndef TKMK_TESTING_MODE
    if (!mixer_file.empty()) {   // the reference gates fixed blinding scalars behind the compile-time feature `testing-mode`
        fprintf(stderr, "prove: --testing-mixer needs the testing-mode build (bin/prove-testing); this binary always draws its blinding scalars from getrandom()\n");
        return 2;
    }
#endif
    try {
        double t_start = Prover::now();
        const std::string lib_dir = args::resolve_subcircuit_library(a);
        printf("Subcircuit library: %s\n", lib_dir.c_str());   // which circuit this run is for (the reference names it by an embedded hash)
        fflush(stdout);
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
#ifdef TKMK_TESTING_MODE
        if (!mixer_file.empty()) {
            fprintf(stderr, "WARNING: --testing-mixer: blinding scalars are read from %s; this proof is NOT zero-knowledge. Testing only.\n", mixer_file.c_str());
            mixer = mixer_from_json(json::read_file(mixer_file));
        } else
#endif
            mixer = Mixer::random();
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
