// tkmk_witness.hpp — C++17 mirror of the witness side of the path, above libtkmk_hip.so's C ABI:
//   R1csBinary / SubcircuitR1CS::from_r1cs_sparse_only   libs/src/iotools/mod.rs:505-760   (iden3 .r1cs v1, same validations)
//   read_R1CS_gen_uvwXY                                  libs/src/iotools/mod.rs:1287-1420 (sparse rows x placement variables on the device)
//   gen_bXY, Instance::gen_a_free_X                      libs/src/polynomial_structures/mod.rs:104-162
// Include after tkmk_protocol.hpp (PlacementVariables, SubcircuitInfo, SetupParams).  JSON parsing stays with the caller.
#pragma once
#include <map>

#include "tkmk_protocol.hpp"

namespace tkmk {

struct R1csError : Error {
    explicit R1csError(const std::string &m) : Error(m) {}
};

// r = BLS12-381 scalar modulus, little-endian u32 limbs (the prime in every committed .r1cs header)
inline ScalarField fr_from_le_bytes_mod_r(const uint8_t *p, size_t n) {
    // ScalarField::from_bytes_le on a field-size buffer: values below 2^256 reduced by repeated subtraction (at most twice)
    static const uint32_t R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    ScalarField v{};
    std::memcpy(&v, p, n < 32 ? n : 32);
    for (size_t i = 32; i < n; i++)
        if (p[i]) throw R1csError("R1CS coefficient does not fit the scalar field");
    auto geq = [&]() {
        for (int i = 7; i >= 0; i--) {
            if (v.limbs[i] != R[i]) return v.limbs[i] > R[i];
        }
        return true;
    };
    while (geq()) {
        uint64_t br = 0;
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)v.limbs[i] - R[i] - br;
            v.limbs[i] = (uint32_t)d;
            br = (d >> 63) & 1;
        }
    }
    return v;
}

class R1csBinary {
  public:
    std::vector<uint8_t> data;
    size_t constraints_offset = 0, constraints_size = 0, field_size = 0;
    uint32_t n_wires = 0, n_constraints = 0;

    static R1csBinary read(const std::string &path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw R1csError("cannot open " + path);
        return parse(std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()));
    }
    static R1csBinary parse(std::vector<uint8_t> bytes) {
        R1csBinary b;
        b.data = std::move(bytes);
        const auto &d = b.data;
        auto need = [&](size_t off, size_t n) {
            if (off + n > d.size()) throw R1csError("unexpected end of R1CS file");
        };
        auto u32 = [&](size_t off) {
            uint32_t v;
            std::memcpy(&v, d.data() + off, 4);
            return v;
        };
        auto u64 = [&](size_t off) {
            uint64_t v;
            std::memcpy(&v, d.data() + off, 8);
            return v;
        };
        need(0, 4);
        if (std::memcmp(d.data(), "r1cs", 4) != 0) throw R1csError("invalid R1CS magic");
        need(4, 8);
        uint32_t version = u32(4), sections = u32(8);
        if (version != 1) throw R1csError("unsupported R1CS version");
        size_t off = 12, hoff = 0, hsize = 0;
        bool have_h = false, have_c = false;
        for (uint32_t s = 0; s < sections; s++) {
            need(off, 12);
            uint32_t stype = u32(off);
            uint64_t ssize = u64(off + 4);
            off += 12;
            if (off + ssize > d.size()) throw R1csError("R1CS section extends past end of file");
            if (stype == 1) hoff = off, hsize = ssize, have_h = true;
            else if (stype == 2) b.constraints_offset = off, b.constraints_size = ssize, have_c = true;
            off += ssize;
        }
        if (!have_h) throw R1csError("missing R1CS header section");
        if (!have_c) throw R1csError("missing R1CS constraints section");
        size_t cur = hoff;
        need(cur, 4);
        b.field_size = u32(cur);
        cur += 4;
        if (b.field_size == 0 || b.field_size % 8) throw R1csError("invalid R1CS field size");
        need(cur, b.field_size + 28);
        cur += b.field_size;
        b.n_wires = u32(cur);
        cur += 16 + 8;   // nWires, nPubOut, nPubIn, nPrvIn, nLabels(u64)
        b.n_constraints = u32(cur);
        cur += 4;
        if (cur > hoff + hsize) throw R1csError("R1CS header extends past section end");
        return b;
    }
    std::vector<uint8_t> prime() const {   // the header's modulus, little-endian
        size_t off = 12;
        for (;;) {
            uint32_t stype;
            uint64_t ssize;
            std::memcpy(&stype, data.data() + off, 4);
            std::memcpy(&ssize, data.data() + off + 4, 8);
            off += 12;
            if (stype == 1) return std::vector<uint8_t>(data.begin() + off + 4, data.begin() + off + 4 + field_size);
            off += ssize;
        }
    }
};

// CSR rows of A, B, C
struct SubcircuitR1CS {
    uint32_t n_wires = 0, n_constraints = 0;
    std::vector<uint32_t> row_ptr[3], wire[3];
    std::vector<ScalarField> coeff[3];

    static SubcircuitR1CS from_r1cs_sparse_only(const R1csBinary &b, const SetupParams &sp, const SubcircuitInfo &info, size_t n_consts) {
        if (b.n_wires != info.Nwires) throw R1csError("R1CS nWires mismatch for subcircuit " + std::to_string(info.id));
        if (b.n_constraints != n_consts) throw R1csError("R1CS nConstraints mismatch for subcircuit " + std::to_string(info.id));
        if (sp.n < n_consts) throw R1csError("n is smaller than the actual number of constraints.");
        SubcircuitR1CS r;
        r.n_wires = b.n_wires;
        r.n_constraints = b.n_constraints;
        const auto &d = b.data;
        size_t off = b.constraints_offset, end = b.constraints_offset + b.constraints_size, fs = b.field_size;
        for (int m = 0; m < 3; m++) r.row_ptr[m].push_back(0);
        for (uint32_t row = 0; row < b.n_constraints; row++)
            for (int m = 0; m < 3; m++) {
                if (off + 4 > d.size()) throw R1csError("unexpected end of R1CS file");
                uint32_t cnt;
                std::memcpy(&cnt, d.data() + off, 4);
                off += 4;
                for (uint32_t k = 0; k < cnt; k++) {
                    if (off + 4 + fs > d.size()) throw R1csError("unexpected end of R1CS file");
                    uint32_t w;
                    std::memcpy(&w, d.data() + off, 4);
                    off += 4;
                    if (w >= b.n_wires) throw R1csError("R1CS wire index exceeds nWires");
                    r.wire[m].push_back(w);
                    r.coeff[m].push_back(fr_from_le_bytes_mod_r(d.data() + off, fs));
                    off += fs;
                }
                r.row_ptr[m].push_back((uint32_t)r.wire[m].size());
            }
        if (off != end) throw R1csError("R1CS constraints section has trailing bytes");
        return r;
    }
};

// read_R1CS_gen_uvwXY: r1cs_of(subcircuit id) supplies the parsed constraint system of every used subcircuit
template <class R1csOf>
inline std::array<DensePolynomialExt, 3> read_R1CS_gen_uvwXY(R1csOf r1cs_of, const std::vector<PlacementVariables> &pv,
                                                              const std::vector<SubcircuitInfo> &infos, const SetupParams &sp) {
    const size_t n = sp.n, s_max = sp.s_max;
    if (pv.size() > s_max) throw Error("placement_variables length exceeds s_max.");
    std::map<size_t, std::vector<uint32_t>> by_id;
    for (size_t i = 0; i < pv.size(); i++) {
        if (pv[i].subcircuitId >= infos.size()) throw Error("Invalid subcircuit id in placement_variables.");
        by_id[pv[i].subcircuitId].push_back((uint32_t)i);
    }
    std::vector<ScalarField> zeros(s_max * n);
    DeviceVec<ScalarField> evals[3] = {DeviceVec<ScalarField>::from_host(zeros), DeviceVec<ScalarField>::from_host(zeros),
                                       DeviceVec<ScalarField>::from_host(zeros)};
    for (auto &kv : by_id) {
        const SubcircuitR1CS &r = r1cs_of(kv.first);
        std::vector<ScalarField> var;
        for (uint32_t slot : kv.second) {
            if (pv[slot].variables.size() != r.n_wires) throw Error("placement variable count does not match nWires");
            var.insert(var.end(), pv[slot].variables.begin(), pv[slot].variables.end());
        }
        DeviceVec<ScalarField> d_var = DeviceVec<ScalarField>::from_host(var);
        DeviceVec<uint32_t> d_slot = DeviceVec<uint32_t>::from_host(kv.second);
        for (int m = 0; m < 3; m++) {
            DeviceVec<uint32_t> d_ptr = DeviceVec<uint32_t>::from_host(r.row_ptr[m]);
            std::vector<uint32_t> w = r.wire[m];
            std::vector<ScalarField> c = r.coeff[m];
            if (w.empty()) w.push_back(0), c.push_back(ScalarField{});   // keep the device pointers valid; nnz stays 0
            DeviceVec<uint32_t> d_wire = DeviceVec<uint32_t>::from_host(w);
            DeviceVec<ScalarField> d_coeff = DeviceVec<ScalarField>::from_host(c);
            check(tkmk_r1cs_eval_rows(d_ptr.ptr(), d_wire.ptr(), d_coeff.ptr(), r.n_constraints, (uint32_t)r.wire[m].size(), d_var.ptr(), r.n_wires,
                                      (uint32_t)kv.second.size(), d_slot.ptr(), (uint32_t)n, evals[m].ptr(), nullptr),
                  "tkmk_r1cs_eval_rows");
        }
    }
    tkmk_vecops_config c = tkmk_vecops_default_config();
    c.is_a_on_device = c.is_result_on_device = true;
    std::array<DensePolynomialExt, 3> out = {DensePolynomialExt::zero(), DensePolynomialExt::zero(), DensePolynomialExt::zero()};
    for (int m = 0; m < 3; m++) {
        DeviceVec<ScalarField> t(n * s_max);
        check(bls12_381_matrix_transpose(evals[m].ptr(), (uint32_t)s_max, (uint32_t)n, &c, t.ptr()), "transpose");   // -> n x s_max
        out[m] = DensePolynomialExt::from_rou_evals(t, n, s_max);
    }
    return out;
}

// gen_bXY (polynomial_structures/mod.rs:132-162): interface wires [l, l_D) of every placement -> evaluations -> polynomial
inline DensePolynomialExt gen_bXY(const std::vector<PlacementVariables> &pv, const std::vector<SubcircuitInfo> &infos, const SetupParams &sp) {
    const size_t m_i = sp.l_D - sp.l, s_max = sp.s_max;
    std::vector<ScalarField> w(m_i * s_max);
    for (size_t i = 0; i < pv.size(); i++) {
        const SubcircuitInfo &info = infos.at(pv[i].subcircuitId);
        if (pv[i].variables.size() != info.flattenMap.size()) throw Error("Corrupted placement variables.");
        for (size_t j = 0; j < info.flattenMap.size(); j++) {
            size_t g = info.flattenMap[j];
            if (g >= sp.l && g < sp.l_D) w[(g - sp.l) * s_max + i] = pv[i].variables[j];
        }
    }
    DeviceVec<ScalarField> ev = DeviceVec<ScalarField>::from_host(w);
    return DensePolynomialExt::from_rou_evals(ev, m_i, s_max);
}
// Instance::gen_a_free_X (:104-130): user + block public inputs as evaluations over l_free points
inline DensePolynomialExt gen_a_free_X(const std::vector<ScalarField> &a_pub_user, const std::vector<ScalarField> &a_pub_block, const SetupParams &sp) {
    if (a_pub_user.size() < sp.l_user || a_pub_block.size() < sp.l_free - sp.l_user) throw Error("instance vectors are too short");
    std::vector<ScalarField> v(a_pub_user.begin(), a_pub_user.begin() + sp.l_user);
    v.insert(v.end(), a_pub_block.begin(), a_pub_block.begin() + (sp.l_free - sp.l_user));
    DeviceVec<ScalarField> ev = DeviceVec<ScalarField>::from_host(v);
    return DensePolynomialExt::from_rou_evals_rep(ev, sp.l_free, 1);   // the public inputs: the same on every rank
}

}  // namespace tkmk
