// tkmk_rkyv.hpp — reader (and fixture writer) for the reference's CRS archives, so that `prove --crs DIR` / `preprocess --crs DIR`
// open what the reference's trusted setup wrote:
//   <crs>/combined_sigma.rkyv    = rkyv::to_bytes::<_, 256>(&SigmaRkyv)            (libs/src/iotools/mod.rs:280-285; types :1701-1783)
//   <crs>/sigma_preprocess.rkyv  = rkyv::to_bytes::<_, 256>(&SigmaPreprocessRkyv)  (:287-294; types :1743-1754)
// read by SigmaZeroCopy::load (prove/src/sigma_source.rs:22-37) and preprocess/src/main.rs:47-53 through
// rkyv::check_archived_root / archived_root.  rkyv is a third-party crate (version 0.7, default features size_32 + std, plus
// alloc / bytecheck / validation: packages/backend/Cargo.toml:41) that is not vendored in the reference; its published format,
// restated here:
//   * the archive is written depth first: serialize() of a value first writes what the value points to, then the value's own
//     fixed-size "archived" form; the ROOT object is the last thing in the file, at len - size_of(Archived<Root>);
//   * a derived struct serializes its fields in declaration order (only out-of-line data is written at that point);
//   * Vec<T> -> the elements' own out-of-line data (element by element), padding to align_of(Archived<T>), then the archived
//     elements contiguously; its archived form is ArchivedVec = { RelPtr: i32 offset relative to the position of this field,
//     len: u32 } (little endian, 8 bytes, align 4);
//   * [u8; N] archives as itself (align 1): ArchivedG1SerdeRkyv = 96 bytes {x: 48 LE, y: 48 LE}, ArchivedG2SerdeRkyv = 192 bytes.
// What the format does NOT fix is the field order inside an archived struct: the derive emits plain (non-repr(C)) structs, so the
// compiler that built the reference chose it.  No archive ships with the reference ("parity unpinned" for the container: DESIGN.md
// §2); instead of trusting one guess, the reader tries the orders rustc has used — current rustc (group fields by
// log2(max(align, size)), larger first, stable), older rustc (by alignment, larger first, stable) and declaration order — and
// accepts the one under which the archive VALIDATES: every relative pointer lands inside the file on a block of the length the
// circuit's setupParams.json demands, the blocks appear in serialization order, and the single G1 points lie on the curve with
// xy_powers[0] == G, xy_powers[1] == [y]G's slot and xy_powers[rs_y] == [x]G's slot (libs/src/group_structures/mod.rs:313-551).
// The section order of the decoded view is the one of the reference's own decoder
// (backend-wasm/tools/rkyv-decoder-wasm/src/lib.rs:118-140), i.e. CrsPayload's.
#pragma once
#include <cstring>
#include <string>
#include <vector>

#include "tkmk_fq_host.hpp"
#include "tkmk_protocol.hpp"

namespace tkmk {
namespace rkyv {

enum class FieldOrder { RustcSizeGroups, RustcAlignOnly, Declared };
inline const char *field_order_name(FieldOrder o) {
    return o == FieldOrder::RustcSizeGroups ? "rustc (size groups)" : o == FieldOrder::RustcAlignOnly ? "rustc (alignment only)" : "declaration order";
}

struct Field {
    size_t size, align;
    size_t offset = 0;
};
// offsets of `fields` (declaration order in, offsets filled in place) -> {struct size, struct align}
inline std::pair<size_t, size_t> layout(std::vector<Field> &fields, FieldOrder order) {
    std::vector<size_t> idx(fields.size());
    for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
    auto tz = [](size_t v) {
        int k = 0;
        while (!(v & 1)) v >>= 1, k++;
        return k;
    };
    if (order == FieldOrder::RustcSizeGroups)
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
            return tz(std::max(fields[a].align, fields[a].size)) > tz(std::max(fields[b].align, fields[b].size));
        });
    else if (order == FieldOrder::RustcAlignOnly)
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return fields[a].align > fields[b].align; });
    size_t off = 0, align = 1;
    for (size_t i : idx) {
        Field &f = fields[i];
        off = (off + f.align - 1) / f.align * f.align;
        f.offset = off;
        off += f.size;
        align = std::max(align, f.align);
    }
    return {(off + align - 1) / align * align, align};
}

constexpr size_t G1 = 96, G2 = 192, VEC = 8;

// field positions of the three archived structs under one order
struct SigmaLayout {
    // ArchivedSigma1Rkyv, declaration order (libs/src/iotools/mod.rs:1727-1741)
    enum S1 { xy_powers, x, y, delta, eta, gamma_inv_o_inst, eta_inv_li_o_inter_alpha4_kj, delta_inv_li_o_prv, delta_inv_alphak_xh_tx, delta_inv_alpha4_xj_tx,
              delta_inv_alphak_yi_ty, S1_COUNT };
    // ArchivedSigmaRkyv (:1716-1724)
    enum R { G, H, sigma_1, sigma_2, lagrange_KL, R_COUNT };
    std::vector<Field> s1, root;
    size_t s1_size, root_size, root_align;
    explicit SigmaLayout(FieldOrder o) {
        s1 = {{VEC, 4}, {G1, 1}, {G1, 1}, {G1, 1}, {G1, 1}, {VEC, 4}, {VEC, 4}, {VEC, 4}, {VEC, 4}, {VEC, 4}, {VEC, 4}};
        auto a = layout(s1, o);
        s1_size = a.first;
        root = {{G1, 1}, {G2, 1}, {a.first, a.second}, {9 * G2, 1}, {G1, 1}};   // Sigma2Rkyv: nine G2 fields of one type keep their order
        auto b = layout(root, o);
        root_size = b.first, root_align = b.second;
    }
};

struct VecRef {
    size_t pos = 0, len = 0;   // absolute position of the elements, element count
};
inline VecRef read_vec(const uint8_t *d, size_t n, size_t at, size_t elem_size, const char *what) {
    if (at + 8 > n) throw Error(std::string("rkyv: vector header of ") + what + " outside the archive");
    int32_t rel;
    uint32_t len;
    std::memcpy(&rel, d + at, 4);
    std::memcpy(&len, d + at + 4, 4);
    int64_t pos = (int64_t)at + rel;
    if (pos < 0 || (uint64_t)pos + (uint64_t)len * elem_size > n) throw Error(std::string("rkyv: ") + what + " points outside the archive");
    return VecRef{(size_t)pos, len};
}

// a table of `rows` rows as ONE run of points: consecutive rows of an archive written by rkyv::to_bytes are adjacent (alignment 1)
struct Table {
    const uint8_t *p = nullptr;
    size_t points = 0;
    std::shared_ptr<std::vector<uint8_t>> owned;   // set when the rows were not adjacent and had to be gathered
};
inline Table read_nested(const uint8_t *d, size_t n, size_t at, const char *what) {
    VecRef outer = read_vec(d, n, at, VEC, what);
    if (outer.pos % 4) throw Error(std::string("rkyv: misaligned row headers of ") + what);
    Table t;
    bool adjacent = true;
    size_t expect = 0;
    std::vector<VecRef> rows(outer.len);
    for (size_t r = 0; r < outer.len; r++) {
        rows[r] = read_vec(d, n, outer.pos + r * VEC, G1, what);
        if (r && rows[r].pos != expect) adjacent = false;
        expect = rows[r].pos + rows[r].len * G1;
        t.points += rows[r].len;
    }
    if (outer.len == 0) return t;
    if (adjacent) {
        t.p = d + rows[0].pos;
    } else {
        t.owned = std::make_shared<std::vector<uint8_t>>();
        t.owned->reserve(t.points * G1);
        for (auto &r : rows) t.owned->insert(t.owned->end(), d + r.pos, d + r.pos + r.len * G1);
        t.p = t.owned->data();
    }
    return t;
}

// ---- combined_sigma.rkyv -> the decoder's nine sections (a CrsPayload whose sections point into the mapping) ----
struct Expect {   // section sizes the circuit demands (0 = do not check)
    size_t xy_powers = 0, gamma = 0, eta = 0, delta = 0;
    size_t rs_y = 0;   // for the xy_powers[rs_y] == x check
    bool check_points = true;   // false: structural checks only (the shape test of the reference's decoder uses byte patterns)
};
inline bool try_decode_sigma(const uint8_t *d, size_t n, FieldOrder order, const Expect &ex, CrsPayload &out, std::string &why) {
    try {
        SigmaLayout L(order);
        if (n < L.root_size) throw Error("rkyv: archive shorter than its root object");
        const size_t root = n - L.root_size;
        if (root % L.root_align) throw Error("rkyv: misaligned root object");
        const size_t s1 = root + L.root[SigmaLayout::sigma_1].offset, s2 = root + L.root[SigmaLayout::sigma_2].offset;
        auto f1 = [&](int f) { return s1 + L.s1[f].offset; };
        VecRef xy = read_vec(d, n, f1(SigmaLayout::xy_powers), G1, "xy_powers");
        VecRef gm = read_vec(d, n, f1(SigmaLayout::gamma_inv_o_inst), G1, "gamma_inv_o_inst");
        Table et = read_nested(d, n, f1(SigmaLayout::eta_inv_li_o_inter_alpha4_kj), "eta_inv_li_o_inter_alpha4_kj");
        Table dl = read_nested(d, n, f1(SigmaLayout::delta_inv_li_o_prv), "delta_inv_li_o_prv");
        Table xh = read_nested(d, n, f1(SigmaLayout::delta_inv_alphak_xh_tx), "delta_inv_alphak_xh_tx");
        VecRef xj = read_vec(d, n, f1(SigmaLayout::delta_inv_alpha4_xj_tx), G1, "delta_inv_alpha4_xj_tx");
        Table yi = read_nested(d, n, f1(SigmaLayout::delta_inv_alphak_yi_ty), "delta_inv_alphak_yi_ty");
        if (ex.xy_powers && xy.len != ex.xy_powers) throw Error("rkyv: xy_powers does not match setupParams.json");
        if (ex.gamma && gm.len != ex.gamma) throw Error("rkyv: gamma_inv_o_inst does not match setupParams.json");
        if (ex.eta && et.points != ex.eta) throw Error("rkyv: eta_inv_li_o_inter_alpha4_kj does not match setupParams.json");
        if (ex.delta && dl.points != ex.delta) throw Error("rkyv: delta_inv_li_o_prv does not match setupParams.json");
        // serialization order: every block lies before the root, xy_powers first
        if (xy.pos + xy.len * G1 > root || gm.pos < xy.pos + xy.len * G1) throw Error("rkyv: blocks out of serialization order");
        // the single points: on the curve, and consistent with the table (G = xy_powers[0], [y]G = xy_powers[1], [x]G = xy_powers[rs_y])
        auto single = [&](size_t at) {
            G1Affine p;
            std::memcpy(&p, d + at, G1);
            return p;
        };
        const G1Affine Gp = single(root + L.root[SigmaLayout::G].offset), X = single(f1(SigmaLayout::x)), Y = single(f1(SigmaLayout::y)),
                       De = single(f1(SigmaLayout::delta)), Et = single(f1(SigmaLayout::eta)), KL = single(root + L.root[SigmaLayout::lagrange_KL].offset);
        if (ex.check_points) {
            for (const G1Affine *p : {&Gp, &X, &Y, &De, &Et, &KL})
                if (!fqh::g1_on_curve_or_infinity(*p)) throw Error("rkyv: a single G1 point is not on the curve");
            if (xy.len >= 1 && std::memcmp(d + xy.pos, &Gp, G1) != 0) throw Error("rkyv: xy_powers[0] != G");
            if (xy.len >= 2 && std::memcmp(d + xy.pos + G1, &Y, G1) != 0) throw Error("rkyv: xy_powers[1] != sigma_1.y");
            if (ex.rs_y && xy.len > ex.rs_y && std::memcmp(d + xy.pos + ex.rs_y * G1, &X, G1) != 0) throw Error("rkyv: xy_powers[rs_y] != sigma_1.x");
        }

        CrsPayload c;
        c.container = "combined_sigma.rkyv";
        auto owned = [&](size_t bytes) {
            auto v = std::make_shared<std::vector<uint8_t>>(bytes);
            c.keep_more.push_back(v);
            return v->data();
        };
        uint8_t *g1s = owned(6 * G1);   // encode_sigma_g1: G, x, y, delta, eta, lagrange_KL
        int k = 0;
        for (const G1Affine *p : {&Gp, &X, &Y, &De, &Et, &KL}) std::memcpy(g1s + G1 * k++, p, G1);
        uint8_t *g2s = owned(10 * G2);  // encode_sigma_g2: H, then sigma_2's nine fields
        std::memcpy(g2s, d + root + L.root[SigmaLayout::H].offset, G2);
        std::memcpy(g2s + G2, d + s2, 9 * G2);
        auto set = [&](CrsPayload::Section s, const uint8_t *p, size_t pts, size_t rec) { c.section[s] = p, c.length[s] = pts * rec; };
        auto table = [&](CrsPayload::Section s, const Table &t) {
            if (t.owned) c.keep_more.push_back(t.owned);
            set(s, t.p, t.points, G1);
        };
        set(CrsPayload::G1Singles, g1s, 6, G1);
        set(CrsPayload::XyPowers, d + xy.pos, xy.len, G1);
        set(CrsPayload::GammaInvOInst, d + gm.pos, gm.len, G1);
        table(CrsPayload::EtaInvLiOInterAlpha4Kj, et);
        table(CrsPayload::DeltaInvLiOPrv, dl);
        table(CrsPayload::DeltaInvAlphakXhTx, xh);
        set(CrsPayload::DeltaInvAlpha4XjTx, d + xj.pos, xj.len, G1);
        table(CrsPayload::DeltaInvAlphakYiTy, yi);
        set(CrsPayload::G2Points, g2s, 10, G2);
        out = std::move(c);
        return true;
    } catch (const std::exception &e) {
        why = e.what();
        return false;
    }
}

// view = the archive's bytes (kept alive by `keep`); -> sections + the field order that validated
inline CrsPayload decode_combined_sigma(CrsPayload::View view, std::shared_ptr<void> keep, const Expect &ex, FieldOrder *order_out = nullptr,
                                        std::vector<FieldOrder> orders = {FieldOrder::RustcSizeGroups, FieldOrder::RustcAlignOnly, FieldOrder::Declared}) {
    std::string reasons;
    for (FieldOrder o : orders) {
        CrsPayload c;
        std::string why;
        if (try_decode_sigma(view.data(), view.size(), o, ex, c, why)) {
            c.data = view;
            c.keep = keep;
            if (order_out) *order_out = o;
            return c;
        }
        reasons += std::string("\n  ") + field_order_name(o) + ": " + why;
    }
    throw Error("Invalid sigma archive: combined_sigma.rkyv validates under none of the known field orders" + reasons);
}

// ---- sigma_preprocess.rkyv: root = { sigma_1: { xy_powers: Vec, gamma_inv_o_inst: Vec } } = two vector headers, 16 bytes ----
struct PreprocessSigma {
    const uint8_t *xy_powers = nullptr, *gamma_inv_o_inst = nullptr;
    size_t xy_points = 0, gamma_points = 0;
};
inline PreprocessSigma decode_sigma_preprocess(const uint8_t *d, size_t n) {
    if (n < 16 || (n - 16) % 4) throw Error("Invalid sigma_preprocess archive");
    VecRef xy = read_vec(d, n, n - 16, G1, "xy_powers"), gm = read_vec(d, n, n - 8, G1, "gamma_inv_o_inst");
    if (xy.pos + xy.len * G1 > n - 16 || gm.pos < xy.pos + xy.len * G1) throw Error("Invalid sigma_preprocess archive: blocks out of serialization order");
    return PreprocessSigma{d + xy.pos, d + gm.pos, xy.len, gm.len};
}

// ---- writer: what rkyv::to_bytes::<_, 256> produces for these types (trusted-setup output, test fixtures) ----
class Writer {
  public:
    std::vector<uint8_t> out;
    size_t pos() const { return out.size(); }
    void align(size_t a) {
        while (out.size() % a) out.push_back(0);
    }
    size_t block(const uint8_t *p, size_t bytes) {   // Vec<G1SerdeRkyv>: align 1
        size_t at = pos();
        out.insert(out.end(), p, p + bytes);
        return at;
    }
    // Vec<Vec<G1SerdeRkyv>>: rows first, then (aligned to 4) the row headers; -> position of the headers
    size_t nested(const uint8_t *p, const std::vector<size_t> &row_points) {
        std::vector<size_t> at;
        for (size_t r : row_points) {
            at.push_back(block(p, r * G1));
            p += r * G1;
        }
        align(4);
        size_t heads = pos();
        for (size_t r = 0; r < row_points.size(); r++) vec_header(at[r], row_points[r]);
        return heads;
    }
    void vec_header(size_t target, size_t len) {
        int64_t rel = (int64_t)target - (int64_t)pos();
        if (rel < INT32_MIN || rel > INT32_MAX || len > 0xffffffffull) throw Error("rkyv: archive too large for 32-bit relative pointers");
        int32_t r32 = (int32_t)rel;
        uint32_t l32 = (uint32_t)len;
        uint8_t b[8];
        std::memcpy(b, &r32, 4);
        std::memcpy(b + 4, &l32, 4);
        out.insert(out.end(), b, b + 8);
    }
    void vec_header_at(size_t field_pos, size_t target, size_t len) {   // into an already reserved struct
        int64_t rel = (int64_t)target - (int64_t)field_pos;
        if (rel < INT32_MIN || rel > INT32_MAX || len > 0xffffffffull) throw Error("rkyv: archive too large for 32-bit relative pointers");
        int32_t r32 = (int32_t)rel;
        uint32_t l32 = (uint32_t)len;
        std::memcpy(out.data() + field_pos, &r32, 4);
        std::memcpy(out.data() + field_pos + 4, &l32, 4);
    }
};
struct SigmaTables {   // the decoder's sections, host pointers (TKCRS001 section order)
    const uint8_t *g1_singles;   // G, x, y, delta, eta, lagrange_KL
    const uint8_t *xy_powers;
    size_t xy_points;
    const uint8_t *gamma;
    size_t gamma_points;
    const uint8_t *eta;
    std::vector<size_t> eta_rows;
    const uint8_t *delta;
    std::vector<size_t> delta_rows;
    const uint8_t *xh;
    std::vector<size_t> xh_rows;
    const uint8_t *xj;
    size_t xj_points;
    const uint8_t *yi;
    std::vector<size_t> yi_rows;
    const uint8_t *g2_points;    // H, alpha, alpha2, alpha3, alpha4, gamma, delta, eta, x, y
};
inline std::vector<uint8_t> encode_combined_sigma(const SigmaTables &t, FieldOrder order = FieldOrder::RustcSizeGroups) {
    SigmaLayout L(order);
    Writer w;
    // Serialize for SigmaRkyv: fields in declaration order; only sigma_1's vectors write anything before the root
    size_t xy = w.block(t.xy_powers, t.xy_points * G1);
    size_t gm = w.block(t.gamma, t.gamma_points * G1);
    size_t et = w.nested(t.eta, t.eta_rows);
    size_t dl = w.nested(t.delta, t.delta_rows);
    size_t xh = w.nested(t.xh, t.xh_rows);
    size_t xj = w.block(t.xj, t.xj_points * G1);
    size_t yi = w.nested(t.yi, t.yi_rows);
    w.align(L.root_align);
    const size_t root = w.pos();
    w.out.resize(root + L.root_size, 0);   // resolve() writes into zeroed memory: padding bytes are zero
    uint8_t *o = w.out.data();
    const size_t s1 = root + L.root[SigmaLayout::sigma_1].offset, s2 = root + L.root[SigmaLayout::sigma_2].offset;
    auto f1 = [&](int f) { return s1 + L.s1[f].offset; };
    std::memcpy(o + root + L.root[SigmaLayout::G].offset, t.g1_singles, G1);
    std::memcpy(o + f1(SigmaLayout::x), t.g1_singles + 1 * G1, G1);
    std::memcpy(o + f1(SigmaLayout::y), t.g1_singles + 2 * G1, G1);
    std::memcpy(o + f1(SigmaLayout::delta), t.g1_singles + 3 * G1, G1);
    std::memcpy(o + f1(SigmaLayout::eta), t.g1_singles + 4 * G1, G1);
    std::memcpy(o + root + L.root[SigmaLayout::lagrange_KL].offset, t.g1_singles + 5 * G1, G1);
    std::memcpy(o + root + L.root[SigmaLayout::H].offset, t.g2_points, G2);
    std::memcpy(o + s2, t.g2_points + G2, 9 * G2);
    w.vec_header_at(f1(SigmaLayout::xy_powers), xy, t.xy_points);
    w.vec_header_at(f1(SigmaLayout::gamma_inv_o_inst), gm, t.gamma_points);
    w.vec_header_at(f1(SigmaLayout::eta_inv_li_o_inter_alpha4_kj), et, t.eta_rows.size());
    w.vec_header_at(f1(SigmaLayout::delta_inv_li_o_prv), dl, t.delta_rows.size());
    w.vec_header_at(f1(SigmaLayout::delta_inv_alphak_xh_tx), xh, t.xh_rows.size());
    w.vec_header_at(f1(SigmaLayout::delta_inv_alpha4_xj_tx), xj, t.xj_points);
    w.vec_header_at(f1(SigmaLayout::delta_inv_alphak_yi_ty), yi, t.yi_rows.size());
    return std::move(w.out);
}
inline std::vector<uint8_t> encode_sigma_preprocess(const uint8_t *xy_powers, size_t xy_points, const uint8_t *gamma, size_t gamma_points) {
    Writer w;
    size_t xy = w.block(xy_powers, xy_points * G1), gm = w.block(gamma, gamma_points * G1);
    w.align(4);
    w.vec_header(xy, xy_points);
    w.vec_header(gm, gamma_points);
    return std::move(w.out);
}

}  // namespace rkyv
}  // namespace tkmk
