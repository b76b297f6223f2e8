// args_driver.cpp — test driver over host/tkmk_args.hpp (host-only): parses its argv with the `prove` flag set and resolves the
// subcircuit library; prints "ok <crs>|<synth>|<out>|<library>" or "error: ..." (exit 2 usage, 1 resolution).  Built plain and with
// ASan + UBSan by the CPU test tier (tests/test_sanitizers.py).
#include <cstdio>

#include "tkmk_args.hpp"

using namespace tkmk;

int main(int argc, char **argv) {
    args::Spec spec{{"--crs", "--synthesizer-stat", "--output", "--subcircuit-library"}, {"--fixed-tau"}};
    args::Parsed a = args::parse(argc, argv, spec);
    if (!a.error.empty()) {
        printf("error: %s\n", a.error.c_str());
        return 2;
    }
    try {
        std::string lib = args::resolve_subcircuit_library(a);
        printf("ok %s|%s|%s|%s|%d\n", a.get("--crs").c_str(), a.get("--synthesizer-stat").c_str(), a.get("--output").c_str(), lib.c_str(), a.flag("--fixed-tau") ? 1 : 0);
    } catch (const std::exception &e) {
        printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
