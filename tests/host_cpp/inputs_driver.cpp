// test driver for host/tkmk_inputs.hpp and host/tkmk_fastparse.hpp (CPU only: no device is touched)
//   inputs_driver FILE                      serial scanner of placementVariables.json
//   inputs_driver fast FILE THREADS N0,N1,..  the multi-threaded reader; N_k = variables a placement of kind k must carry
//   inputs_driver perm FILE THREADS          the multi-threaded permutation.json reader
// placementVariables: prints "ok <placements> <variables>" and one line "<subcircuitId> <count> <hex of the xor of all 32-byte records>"
// per placement; permutation: "ok <entries>" and one line "row col X Y" per entry; or "error: <message>" (exit code 1)
#include <cstdio>
#include <iostream>
#include <sstream>

#include "tkmk_fastparse.hpp"
#include "tkmk_inputs.hpp"

using namespace tkmk;

int main(int argc, char **argv) {
    try {
        if (argc == 2) {
            auto pv = read_placement_variables(argv[1]);
            size_t total = 0;
            for (auto &p : pv) total += p.variables.size();
            std::cout << "ok " << pv.size() << " " << total << "\n";
            for (auto &p : pv) {
                ScalarField x{};
                for (auto &v : p.variables)
                    for (int i = 0; i < 8; i++) x.limbs[i] ^= v.limbs[i];
                std::cout << p.subcircuitId << " " << p.variables.size() << " " << scalar_to_hex(x) << "\n";
            }
            return 0;
        }
        std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "fast" && argc == 5) {
            MappedFile f(argv[2]);
            unsigned threads = (unsigned)std::stoul(argv[3]);
            std::vector<uint32_t> n_wires;
            std::stringstream ss(argv[4]);
            for (std::string tok; std::getline(ss, tok, ',');) n_wires.push_back((uint32_t)std::stoul(tok));
            std::vector<ScalarField> vars;
            WitnessLayout L = parse_placement_variables_fast(f.data(), f.size(), n_wires, [&](uint64_t total) {
                vars.assign(total + 1, ScalarField{});
                return vars.data(); }, threads);
            std::cout << "ok " << L.id.size() << " " << L.total << "\n";
            for (size_t q = 0; q < L.id.size(); q++) {
                ScalarField x{};
                for (uint32_t k = 0; k < n_wires[L.id[q]]; k++)
                    for (int i = 0; i < 8; i++) x.limbs[i] ^= vars[L.off[q] + k].limbs[i];
                std::cout << L.id[q] << " " << n_wires[L.id[q]] << " " << scalar_to_hex(x) << "\n";
            }
            return 0;
        }
        if (mode == "perm" && argc == 4) {
            MappedFile f(argv[2]);
            PermutationColumns p = parse_permutation_fast(f.data(), f.size(), (unsigned)std::stoul(argv[3]));
            std::cout << "ok " << p.size() << "\n";
            for (size_t e = 0; e < p.size(); e++) std::cout << p.row[e] << " " << p.col[e] << " " << p.X[e] << " " << p.Y[e] << "\n";
            return 0;
        }
        return 2;
    } catch (const std::exception &e) {
        std::cout << "error: " << e.what() << "\n";
        return 1;
    }
}
