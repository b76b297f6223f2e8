// test driver for host/tkmk_inputs.hpp: parses the placementVariables.json given as argv[1]; prints "ok <placements> <variables>" and one
// line "<subcircuitId> <count> <hex of the xor of all 32-byte records>" per placement, or "error: <message>" (exit code 1)
#include <cstdio>
#include <iostream>

#include "tkmk_inputs.hpp"

using namespace tkmk;

int main(int argc, char **argv) {
    if (argc != 2) return 2;
    try {
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
    } catch (const std::exception &e) {
        std::cout << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
