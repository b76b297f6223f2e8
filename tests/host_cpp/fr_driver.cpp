// test driver for host/tkmk_fr.hpp: reads "op a_hex b_hex" lines from stdin, prints the result as 0x-hex (one per line)
#include <iostream>
#include <sstream>

#include "tkmk_fr.hpp"
#include "tkmk_protocol.hpp"

using namespace tkmk;

int main() {
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream is(line);
        std::string op, a, b;
        is >> op >> a >> b;
        ScalarField x = fr_from_hex(a), y = fr_from_hex(b), r{};
        if (op == "add") r = fr_add(x, y);
        else if (op == "sub") r = fr_sub(x, y);
        else if (op == "mul") r = fr_mul(x, y);
        else if (op == "neg") r = fr_neg(x);
        else if (op == "inv") r = fr_inv(x);
        else if (op == "pow") r = fr_pow(x, std::stoull(b.substr(2), nullptr, 16));
        else if (op == "hex") r = x;
        else return 2;
        std::cout << scalar_to_hex(r) << "\n";
    }
    return 0;
}
