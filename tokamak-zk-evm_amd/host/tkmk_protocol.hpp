// tkmk_protocol.hpp — C++17 host-side protocol glue above libtkmk_hip.so's C ABI, next to tkmk_host.hpp: the compiled-language
// mirror of the reference's Rust for
//   RollingKeccakTranscript / TranscriptManager   prove/src/lib.rs:3211-3731 (byte layout fixed by the Solidity verifier)
//   split_g1 / scalar_to_hex / split_push! / pop_recover!, FormattedProof, FormattedPreprocess
//                                                 libs/src/iotools/mod.rs:1625-1700, prove/src/lib.rs:452-513, preprocess/src/lib.rs:84-146
//   the flat CRS payload "TKCRS001"               backend-wasm/tools/rkyv-decoder-wasm/src/lib.rs:118-204
//   msm_g1_bases, encode_O_pub_fix / _pub_free / O_mid / O_prv   libs/src/group_structures/mod.rs:127-300, 607-707
//   Preprocess::gen                                preprocess/src/lib.rs:32-82 (permutation polynomials passed in as evaluations)
// JSON parsing is left to the caller (the reference uses serde): inputs arrive as plain structs / vectors.
// Header-only; include after tkmk_host.hpp; link with -ltkmk_hip.
#pragma once
#include <array>
#include <cstdio>
#include <fstream>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <memory>

#include "tkmk_host.hpp"

namespace tkmk {

// ---------------------------------------------------------------------------------------------------------------------
// Keccak-256 (original padding 0x01) — the hash of the transcript
// ---------------------------------------------------------------------------------------------------------------------
inline std::array<uint8_t, 32> keccak256(const uint8_t *data, size_t len) {
    static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
                                    0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                                    0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                                    0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                                    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                                    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
    auto rol = [](uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; };
    const size_t rate = 136;
    std::vector<uint8_t> msg(data, data + len);
    msg.push_back(0x01);
    while (msg.size() % rate) msg.push_back(0);
    msg.back() |= 0x80;
    uint64_t a[5][5] = {};
    for (size_t off = 0; off < msg.size(); off += rate) {
        for (size_t i = 0; i < rate / 8; i++) {
            uint64_t w = 0;
            for (int b = 7; b >= 0; b--) w = (w << 8) | msg[off + 8 * i + b];
            a[i % 5][i / 5] ^= w;
        }
        for (int round = 0; round < 24; round++) {
            uint64_t c[5], d[5], b[5][5];
            for (int x = 0; x < 5; x++) c[x] = a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4];
            for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
            for (int x = 0; x < 5; x++)
                for (int y = 0; y < 5; y++) a[x][y] ^= d[x];
            for (int x = 0; x < 5; x++)
                for (int y = 0; y < 5; y++) b[y][(2 * x + 3 * y) % 5] = rol(a[x][y], ROT[x][y]);
            for (int x = 0; x < 5; x++)
                for (int y = 0; y < 5; y++) a[x][y] = b[x][y] ^ (~b[(x + 1) % 5][y] & b[(x + 2) % 5][y]);
            a[0][0] ^= RC[round];
        }
    }
    std::array<uint8_t, 32> out{};
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(a[i % 5][i / 5] >> (8 * b));
    return out;
}

// big-endian bytes of a little-endian limb struct (Fr: 32, Fq: 48)
template <class T>
inline std::vector<uint8_t> be_bytes(const T &v) {
    const uint8_t *p = reinterpret_cast<const uint8_t *>(&v);
    return std::vector<uint8_t>(std::reverse_iterator<const uint8_t *>(p + sizeof(T)), std::reverse_iterator<const uint8_t *>(p));
}

// prove/src/lib.rs:3211-3400
class RollingKeccakTranscript {
  public:
    std::array<uint8_t, 32> state_0{}, state_1{};
    uint32_t challenge_counter = 0;

    void update(const uint8_t *bytes, size_t len) {
        if (len > 32) throw Error("Input must be 32 bytes or less");
        uint8_t buf[100] = {};
        std::memcpy(buf + 4, state_0.data(), 32);
        std::memcpy(buf + 36, state_1.data(), 32);
        std::memcpy(buf + 100 - len, bytes, len);   // right-aligned in the last 32-byte slot
        buf[3] = 0;
        auto s0 = keccak256(buf, 100);
        buf[3] = 1;
        auto s1 = keccak256(buf, 100);               // both from the OLD state pair
        state_0 = s0;
        state_1 = s1;
    }
    std::array<uint8_t, 32> get_challenge_raw() {
        uint8_t buf[72] = {};
        buf[3] = 2;
        std::memcpy(buf + 4, state_0.data(), 32);
        std::memcpy(buf + 36, state_1.data(), 32);
        buf[68] = (uint8_t)(challenge_counter >> 24);
        buf[69] = (uint8_t)(challenge_counter >> 16);
        buf[70] = (uint8_t)(challenge_counter >> 8);
        buf[71] = (uint8_t)challenge_counter;
        challenge_counter++;
        return keccak256(buf, 72);
    }
    // FR_MASK: top 3 bits of the big-endian hash cleared; zero -> one (:3363-3394)
    ScalarField get_challenge() {
        auto raw = get_challenge_raw();
        raw[0] &= 0x1f;
        ScalarField s{};
        uint8_t *p = reinterpret_cast<uint8_t *>(&s);
        for (int i = 0; i < 32; i++) p[i] = raw[31 - i];
        if (fr_is_zero(s)) return fr_from_u32(1);
        return s;
    }
    std::vector<ScalarField> get_challenges(size_t count) {
        std::vector<ScalarField> out;
        for (size_t i = 0; i < count; i++) out.push_back(get_challenge());
        return out;
    }
    void commit_field_as_bytes(const ScalarField &s) {       // :3416-3426
        auto be = be_bytes(s);
        update(be.data(), 32);
    }
    void commit_bls12_381_field_element(const tkmk_fq &v) {  // :3429-3480
        auto be = be_bytes(v);
        uint8_t part1[32] = {};
        std::memcpy(part1 + 16, be.data(), 16);
        update(part1, 32);
        update(be.data() + 16, 32);
    }
    void commit_g1_point(const G1Affine &p) {                // :3482-3500
        commit_bls12_381_field_element(p.x);
        commit_bls12_381_field_element(p.y);
    }
};

// commit order of the rounds (prove/src/lib.rs:3528-3731)
class TranscriptManager {
  public:
    RollingKeccakTranscript transcript;
    void add_proof0(const G1Affine &U, const G1Affine &V, const G1Affine &W, const G1Affine &Q_AX, const G1Affine &Q_AY, const G1Affine &B) {
        for (const G1Affine *p : {&U, &V, &W, &Q_AX, &Q_AY, &B}) transcript.commit_g1_point(*p);
    }
    std::vector<ScalarField> get_thetas() { return transcript.get_challenges(3); }
    void add_proof1(const G1Affine &R) { transcript.commit_g1_point(R); }
    ScalarField get_kappa0() { return transcript.get_challenge(); }
    void add_proof2(const G1Affine &Q_CX, const G1Affine &Q_CY) {
        transcript.commit_g1_point(Q_CX);
        transcript.commit_g1_point(Q_CY);
    }
    std::pair<ScalarField, ScalarField> get_chi_zeta() {
        ScalarField chi = transcript.get_challenge();
        ScalarField zeta = transcript.get_challenge();
        return {chi, zeta};
    }
    void add_proof3(const ScalarField &V_eval, const ScalarField &R_eval, const ScalarField &R_omegaX_eval, const ScalarField &R_omegaX_omegaY_eval) {
        for (const ScalarField *s : {&V_eval, &R_eval, &R_omegaX_eval, &R_omegaX_omegaY_eval}) transcript.commit_field_as_bytes(*s);
    }
    ScalarField get_kappa1() { return transcript.get_challenge(); }
};

// ---------------------------------------------------------------------------------------------------------------------
// Solidity-verifier formatting (libs/src/iotools/mod.rs:1625-1700)
// ---------------------------------------------------------------------------------------------------------------------
inline std::string hex0x(const uint8_t *p, size_t n) {
    static const char *d = "0123456789abcdef";
    std::string s = "0x";
    for (size_t i = 0; i < n; i++) {
        s.push_back(d[p[i] >> 4]);
        s.push_back(d[p[i] & 15]);
    }
    return s;
}
// -> x_part1, x_part2, y_part1, y_part2
inline std::array<std::string, 4> split_g1(const G1Affine &p) {
    auto x = be_bytes(p.x), y = be_bytes(p.y);
    return {hex0x(x.data(), 16), hex0x(x.data() + 16, 32), hex0x(y.data(), 16), hex0x(y.data() + 16, 32)};
}
inline std::string scalar_to_hex(const ScalarField &s) {
    auto be = be_bytes(s);
    return hex0x(be.data(), 32);
}
inline std::vector<uint8_t> unhex(const std::string &h) {
    size_t off = h.rfind("0x", 0) == 0 ? 2 : 0;
    if ((h.size() - off) % 2) throw Error("Invalid format");
    std::vector<uint8_t> out;
    auto nib = [](char c) -> int { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; };
    for (size_t i = off; i < h.size(); i += 2) {
        int a = nib(h[i]), b = nib(h[i + 1]);
        if (a < 0 || b < 0) throw Error("Invalid format");
        out.push_back((uint8_t)(a * 16 + b));
    }
    return out;
}
inline tkmk_fq recover_basefield(const std::string &part1, const std::string &part2) {   // :1675-1685
    auto a = unhex(part1), b = unhex(part2);
    if (a.size() != 16 || b.size() != 32) throw Error("Invalid format");
    uint8_t be[48];
    std::memcpy(be, a.data(), 16);
    std::memcpy(be + 16, b.data(), 32);
    tkmk_fq v{};
    uint8_t *p = reinterpret_cast<uint8_t *>(&v);
    for (int i = 0; i < 48; i++) p[i] = be[47 - i];
    return v;
}
struct FormattedEntries {
    std::vector<std::string> part1, part2;
};
inline void split_push(FormattedEntries &f, const G1Affine &p) {   // split_push! (:1660-1673)
    auto s = split_g1(p);
    f.part1.push_back(s[0]);
    f.part2.push_back(s[1]);
    f.part1.push_back(s[2]);
    f.part2.push_back(s[3]);
}
inline G1Affine next_point(size_t idx, const FormattedEntries &f) {   // :1687-1693
    G1Affine p{};
    p.x = recover_basefield(f.part1.at(idx), f.part2.at(idx));
    p.y = recover_basefield(f.part1.at(idx + 1), f.part2.at(idx + 1));
    return p;
}
inline std::string entries_json(const char *k1, const char *k2, const FormattedEntries &f) {
    auto arr = [](const std::vector<std::string> &v) {
        std::string s = "[";
        for (size_t i = 0; i < v.size(); i++) s += (i ? ", \"" : "\"") + v[i] + "\"";
        return s + "]";
    };
    return std::string("{\n  \"") + k1 + "\": " + arr(f.part1) + ",\n  \"" + k2 + "\": " + arr(f.part2) + "\n}\n";
}

// ---------------------------------------------------------------------------------------------------------------------
// CRS payload "TKCRS001" (backend-wasm/tools/rkyv-decoder-wasm/src/lib.rs:118-204)
// ---------------------------------------------------------------------------------------------------------------------
struct CrsPayload {
    enum Section { G1Singles, XyPowers, GammaInvOInst, EtaInvLiOInterAlpha4Kj, DeltaInvLiOPrv, DeltaInvAlphakXhTx, DeltaInvAlpha4XjTx,
                   DeltaInvAlphakYiTy, G2Points, Count };
    // a borrowed view of the payload bytes: either an owned buffer (parse) or a read-only mapping of the file (read; xy_powers
    // alone is 384 MiB at the production shape, so sections go from the page cache to the device without a host copy)
    struct View {
        const uint8_t *p = nullptr;
        size_t n = 0;
        const uint8_t *data() const { return p; }
        size_t size() const { return n; }
    };
    View data;
    std::shared_ptr<void> keep;   // owns the buffer or the mapping
    std::vector<std::shared_ptr<void>> keep_more;   // sections that had to be assembled (rkyv archives: the single points)
    const uint8_t *section[Count]{};                // start of every section's records
    size_t length[Count]{};
    const char *container = "combined_sigma.tkcrs";

    static CrsPayload parse(std::vector<uint8_t> bytes) {
        auto owned = std::make_shared<std::vector<uint8_t>>(std::move(bytes));
        CrsPayload c = parse_view(View{owned->data(), owned->size()});
        c.keep = owned;
        return c;
    }
    static CrsPayload parse_view(View view) {
        CrsPayload c;
        c.data = view;
        const auto &d = c.data;
        if (d.size() < 12 || std::memcmp(d.data(), "TKCRS001", 8) != 0) throw Error("not a TKCRS001 payload");
        uint32_t count;
        std::memcpy(&count, d.data() + 8, 4);
        if (count != (uint32_t)Count) throw Error("unexpected section count");
        size_t head = 12 + 4 * (size_t)Count, off = head, total = head;
        if (d.size() < head) throw Error("truncated section table");
        for (int i = 0; i < Count; i++) {
            uint32_t n;
            std::memcpy(&n, d.data() + 12 + 4 * i, 4);
            c.length[i] = n;
            total += n;
        }
        if (total != d.size()) throw Error("section lengths do not add up to the payload size");
        for (int i = 0; i < Count; i++) {
            if (c.length[i] % (i == G2Points ? 192 : 96)) throw Error("section is not a whole number of points");
            c.section[i] = d.data() + off;
            off += c.length[i];
        }
        if (c.length[G1Singles] != 6 * 96 || c.length[G2Points] != 10 * 192) throw Error("unexpected size of the single-point sections");
        return c;
    }
    static CrsPayload read(const std::string &path) {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw Error("No reference string is found. Run the Setup first (expected " + path + ").");
        struct stat st;
        if (::fstat(fd, &st) != 0 || st.st_size <= 0) {
            ::close(fd);
            throw Error("cannot stat " + path);
        }
        size_t n = (size_t)st.st_size;
        void *m = ::mmap(nullptr, n, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) throw Error("cannot map " + path);
        std::shared_ptr<void> mapping(m, [n](void *q) { ::munmap(q, n); });
        CrsPayload c = parse_view(View{static_cast<const uint8_t *>(m), n});
        c.keep = mapping;
        return c;
    }
    size_t points(Section s) const { return length[s] / 96; }
    const G1Affine *g1(Section s) const { return reinterpret_cast<const G1Affine *>(section[s]); }
    const uint8_t *bytes(Section s) const { return section[s]; }
    DeviceVec<G1Affine> upload(Section s) const { return DeviceVec<G1Affine>::from_host(g1(s), points(s)); }
};

// ---------------------------------------------------------------------------------------------------------------------
// Binding commitments (libs/src/group_structures/mod.rs:127-300, 607-707): index / scalar lists on the host, bases gathered in HBM
// ---------------------------------------------------------------------------------------------------------------------
struct PlacementVariables {
    size_t subcircuitId;
    std::vector<ScalarField> variables;   // the reference keeps hex strings and parses them at use (ScalarField::from_hex)
};
struct SubcircuitInfo {
    size_t id;
    std::string name;
    size_t Nwires;
    std::array<size_t, 2> Out_idx, In_idx;   // [start, count]
    std::vector<size_t> flattenMap;
};
struct SetupParams {
    size_t l, l_user_out, l_user, l_free, l_D, m_D, n, s_D, s_max;
};

inline G1Affine projective_to_affine(const tkmk_g1_projective &r) {
    G1Affine out{};
    bool inf = true;
    for (uint32_t l : r.z.limbs) inf &= l == 0;
    if (!inf) out.x = r.x, out.y = r.y;
    return out;
}
// msm_g1_bases over rows `idx` of a device-resident table
inline G1Affine msm_gathered(const DeviceVec<G1Affine> &table, const std::vector<uint32_t> &idx, const std::vector<ScalarField> &scalars) {
    if (idx.size() != scalars.size()) throw Error("msm input length mismatch");
    if (idx.empty()) return G1Affine{};
    for (uint32_t i : idx)
        if (i >= table.len()) throw Error("CRS table index out of range");
    DeviceVec<uint32_t> di = DeviceVec<uint32_t>::from_host(idx);
    DeviceVec<G1Affine> bases(idx.size());
    check(tkmk_gather_rows_device(table.ptr(), 96, di.ptr(), idx.size(), bases.ptr(), nullptr), "tkmk_gather_rows_device");
    DeviceVec<ScalarField> ds = DeviceVec<ScalarField>::from_host(scalars);
    tkmk_msm_config cfg = tkmk_msm_default_config();
    cfg.are_scalars_on_device = cfg.are_points_on_device = true;
    tkmk_g1_projective res;
    check(bls12_381_msm(ds.ptr(), bases.ptr(), (int)idx.size(), &cfg, &res), "msm::msm");
    return projective_to_affine(res);
}
inline G1Affine encode_O_pub_fix(const DeviceVec<G1Affine> &gamma_inv_o_inst, const std::vector<ScalarField> &a_pub_function, const SetupParams &sp) {
    size_t m_function = sp.l - sp.l_free;   // :145-182
    if (m_function == 0) return G1Affine{};
    if (a_pub_function.size() != m_function) throw Error("a_pub_function length mismatch");
    if (gamma_inv_o_inst.len() < m_function) throw Error("gamma_inv_o_inst length is smaller than m_function");
    std::vector<uint32_t> idx;
    for (size_t i = 0; i < m_function; i++) idx.push_back((uint32_t)(gamma_inv_o_inst.len() - m_function + i));
    return msm_gathered(gamma_inv_o_inst, idx, a_pub_function);
}
inline G1Affine encode_O_pub_free(const DeviceVec<G1Affine> &gamma_inv_o_inst, const std::vector<PlacementVariables> &pv,
                                  const std::vector<SubcircuitInfo> &infos) {   // :184-229
    std::vector<uint32_t> idx;
    std::vector<ScalarField> wt;
    for (const auto &pl : pv) {
        const SubcircuitInfo &info = infos.at(pl.subcircuitId);
        const std::array<size_t, 2> *rng = nullptr;
        if (info.name == "bufferPubOut") rng = &info.Out_idx;
        else if (info.name == "bufferPubIn" || info.name == "bufferBlockIn") rng = &info.In_idx;
        if (!rng) continue;
        for (size_t j = (*rng)[0]; j < (*rng)[0] + (*rng)[1]; j++) {
            wt.push_back(pl.variables.at(j));
            idx.push_back((uint32_t)info.flattenMap.at(j));
        }
    }
    return msm_gathered(gamma_inv_o_inst, idx, wt);
}
inline size_t count_o_mid_nvar(const std::vector<PlacementVariables> &pv, const std::vector<SubcircuitInfo> &infos) {   // :231-251
    size_t n = 0;
    for (const auto &pl : pv) {
        const SubcircuitInfo &info = infos.at(pl.subcircuitId);
        if (info.name == "bufferPubOut") n += info.In_idx[1];
        else if (info.name == "bufferPubIn" || info.name == "bufferBlockIn" || info.name == "bufferEVMIn") n += info.Out_idx[1];
        else n += info.Out_idx[1] + info.In_idx[1];
        n += 1;
    }
    return n;
}
inline size_t count_o_prv_nvar(const std::vector<PlacementVariables> &pv, const std::vector<SubcircuitInfo> &infos) {   // :253-264
    size_t n = 0;
    for (const auto &pl : pv) {
        const SubcircuitInfo &info = infos.at(pl.subcircuitId);
        n += info.Nwires - info.In_idx[1] - info.Out_idx[1] - 1;
    }
    return n;
}
// encode_statement_common (:266-300); table flattened [global - offset][placement] with `inner` entries per global index
inline G1Affine encode_statement(size_t offset, size_t end, size_t n_var, const std::vector<PlacementVariables> &pv,
                                 const std::vector<SubcircuitInfo> &infos, const DeviceVec<G1Affine> &table, size_t inner) {
    std::vector<uint32_t> idx;
    std::vector<ScalarField> wt;
    for (size_t i = 0; i < pv.size(); i++) {
        const SubcircuitInfo &info = infos.at(pv[i].subcircuitId);
        for (size_t j = 0; j < info.Nwires; j++) {
            size_t g = info.flattenMap.at(j);
            if (g >= offset && g < end) {
                wt.push_back(pv[i].variables.at(j));
                idx.push_back((uint32_t)((g - offset) * inner + i));
            }
        }
    }
    if (idx.size() != n_var) throw Error("nVar mismatch while encoding statement");
    return msm_gathered(table, idx, wt);
}
inline G1Affine encode_O_mid_no_zk(const DeviceVec<G1Affine> &eta_table, const std::vector<PlacementVariables> &pv,
                                   const std::vector<SubcircuitInfo> &infos, const SetupParams &sp) {
    return encode_statement(sp.l, sp.l_D, count_o_mid_nvar(pv, infos), pv, infos, eta_table, sp.s_max);
}
inline G1Affine encode_O_prv_no_zk(const DeviceVec<G1Affine> &delta_table, const std::vector<PlacementVariables> &pv,
                                   const std::vector<SubcircuitInfo> &infos, const SetupParams &sp) {
    return encode_statement(sp.l_D, sp.m_D, count_o_prv_nvar(pv, infos), pv, infos, delta_table, sp.s_max);
}

// ---------------------------------------------------------------------------------------------------------------------
// preprocess round (preprocess/src/lib.rs:32-146)
// ---------------------------------------------------------------------------------------------------------------------
struct Permutation {
    size_t row, col, X, Y;
};
inline ScalarField root_of_unity(uint64_t size) {
    ScalarField w;
    check(bls12_381_get_root_of_unity(size, &w), "get_root_of_unity");
    return w;
}
// Permutation::to_poly (libs/src/iotools/mod.rs:419-455): evaluation matrices s0[row][col] = w_x^row, s1 = w_y^col with the listed
// cells redirected, then two inverse bivariate NTTs.  Powers are built on the device (scalar_mul_vec chain avoided: one
// geometric table per axis through tkmk_poly_scale_coeffs of an all-ones matrix).
inline std::pair<DensePolynomialExt, DensePolynomialExt> permutation_to_poly(const std::vector<Permutation> &perm, size_t m_i, size_t s_max) {
    std::vector<ScalarField> ones(m_i * s_max, fr_from_u32(1));
    DeviceVec<ScalarField> base = DeviceVec<ScalarField>::from_host(ones);
    ScalarField wx = root_of_unity(m_i), wy = root_of_unity(s_max), one = fr_from_u32(1);
    DeviceVec<ScalarField> s0(m_i * s_max), s1(m_i * s_max);
    check(tkmk_poly_scale_coeffs(base.ptr(), (uint32_t)m_i, (uint32_t)s_max, &wx, &one, s0.ptr(), nullptr), "s0 powers");   // w_x^row
    check(tkmk_poly_scale_coeffs(base.ptr(), (uint32_t)m_i, (uint32_t)s_max, &one, &wy, s1.ptr(), nullptr), "s1 powers");   // w_y^col
    std::vector<ScalarField> h0 = s0.to_host(), h1 = s1.to_host();
    std::vector<ScalarField> xp(m_i), yp(s_max);
    for (size_t i = 0; i < m_i; i++) xp[i] = h0[i * s_max];
    for (size_t j = 0; j < s_max; j++) yp[j] = h1[j];
    for (const Permutation &p : perm) {
        if (p.row >= m_i || p.col >= s_max || p.X >= m_i || p.Y >= s_max) throw Error("permutation entry out of range");
        h0[p.row * s_max + p.col] = xp[p.X];
        h1[p.row * s_max + p.col] = yp[p.Y];
    }
    DeviceVec<ScalarField> e0 = DeviceVec<ScalarField>::from_host(h0), e1 = DeviceVec<ScalarField>::from_host(h1);
    return {DensePolynomialExt::from_rou_evals(e0, m_i, s_max), DensePolynomialExt::from_rou_evals(e1, m_i, s_max)};
}
struct Preprocess {
    G1Affine s0, s1, O_pub_fix;
    static Preprocess gen(const Sigma1 &sigma1, const DeviceVec<G1Affine> &gamma_inv_o_inst, const std::vector<Permutation> &perm,
                          const std::vector<ScalarField> &a_pub_function, const SetupParams &sp) {
        size_t m_i = sp.l_D - sp.l;
        init_ntt_domain_for_size(4 * std::max(m_i, sp.n) * 2 * sp.s_max);   // libs/src/utils/mod.rs:51-58
        auto polys = permutation_to_poly(perm, m_i, sp.s_max);
        std::vector<G1Affine> cm = sigma1.encode_polys({&polys.first, &polys.second});
        return Preprocess{cm[0], cm[1], encode_O_pub_fix(gamma_inv_o_inst, a_pub_function, sp)};
    }
    FormattedEntries convert_format_for_solidity_verifier() const {
        FormattedEntries f;
        split_push(f, s0);
        split_push(f, s1);
        split_push(f, O_pub_fix);
        return f;
    }
    static Preprocess recover_from_format(const FormattedEntries &f) {
        if (f.part1.size() != 6 || f.part2.size() != 6) throw Error("unexpected preprocess entry count");
        return Preprocess{next_point(0, f), next_point(2, f), next_point(4, f)};
    }
    std::string to_json() const { return entries_json("preprocess_entries_part1", "preprocess_entries_part2", convert_format_for_solidity_verifier()); }
};

}  // namespace tkmk
