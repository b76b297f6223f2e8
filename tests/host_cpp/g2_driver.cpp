// test driver for host/tkmk_g2.hpp: argv[1], argv[2] = x, y of a G2 point as the reference's 96-byte hex constants; reads "op k_hex" lines
// (op = mul: [k]H; addmul: [k]H + H; oncurve) and prints the 192-byte encoding as hex (or 0/1); "g1add p_hex q_hex" (two 96-byte little-endian
// G1 affine records as hex) prints fqh::g1_affine_add(p, q) the same way (host/tkmk_fq_host.hpp: the prover's commitment + blinding point)
#include <iostream>
#include <sstream>

#include "tkmk_fq_host.hpp"
#include "tkmk_g2.hpp"

using namespace tkmk;

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    g2h::Affine h;
    h.x = g2h::f2_from_hex(argv[1]);
    h.y = g2h::f2_from_hex(argv[2]);
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream is(line);
        std::string op, k;
        is >> op >> k;
        if (op == "g1add") {
            std::string qh;
            is >> qh;
            auto rec = [](const std::string &h) {
                G1Affine a{};
                if (h.size() != 192) throw Error("g1add: a record is 96 bytes");
                uint8_t *b = reinterpret_cast<uint8_t *>(&a);
                for (size_t i = 0; i < 96; i++) b[i] = (uint8_t)std::stoul(h.substr(2 * i, 2), nullptr, 16);
                return a;
            };
            G1Affine r = fqh::g1_affine_add(rec(k), rec(qh));
            static const char *d = "0123456789abcdef";
            const uint8_t *b = reinterpret_cast<const uint8_t *>(&r);
            std::string out;
            for (size_t i = 0; i < 96; i++) out += d[b[i] >> 4], out += d[b[i] & 15];
            std::cout << out << "\n";
            continue;
        }
        if (op == "oncurve") {
            std::cout << (g2h::on_curve(h) ? 1 : 0) << "\n";
            continue;
        }
        g2h::Affine r = g2h::scalar_mul(fr_from_hex(k), h);
        if (op == "addmul") r = g2h::to_affine(g2h::add(g2h::to_jac(r), g2h::to_jac(h)));
        auto e = g2h::encode(r);
        auto back = g2h::encode(g2h::decode(e.data()));
        if (back != e) return 3;
        static const char *d = "0123456789abcdef";
        std::string out;
        for (uint8_t b : e) out += d[b >> 4], out += d[b & 15];
        std::cout << out << "\n";
    }
    return 0;
}
