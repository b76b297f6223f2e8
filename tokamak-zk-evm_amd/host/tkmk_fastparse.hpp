// tkmk_fastparse.hpp — multi-threaded readers of the two large per-proof synthesizer documents `prove` takes:
//   placementVariables.json   [{"subcircuitId": k, "variables": ["0x..", ...]}, ...]   (PlacementVariables, libs/src/iotools/mod.rs:366-372;
//                             every string through HexString + ScalarField::from_hex, :126-146 — 115 MB of hex at 1024 placements)
//   permutation.json          [{"row": r, "col": c, "X": x, "Y": y}, ...]              (Permutation, libs/src/iotools/mod.rs:408-416)
// The reference parses both with serde on one thread inside Prover::init (prove/src/lib.rs:679-835).  Here the file is mapped,
// cut at object boundaries ('{' never occurs inside these documents' strings: keys are fixed words, values hex digits) and the
// objects are parsed by a pool of threads, each object with a strict scanner (any key order, any whitespace, anything unexpected
// is an error naming the byte).  Witness values go straight into one caller-provided (pinned) buffer in placement order, so the
// upload to the device is a single copy.  Same results as the serial scanner of tkmk_inputs.hpp (tests/host_cpp/inputs_driver.cpp).
#pragma once
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tkmk_fr.hpp"

namespace tkmk {

// threads this process may really use: affinity mask and cgroup CPU quota, capped (a one-GPU box gets a share of the host)
inline unsigned host_threads() {
    if (const char *e = std::getenv("TKMK_HOST_THREADS")) {
        int v = std::atoi(e);
        if (v >= 1) return (unsigned)(v > 64 ? 64 : v);
    }
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        unsigned c = (unsigned)CPU_COUNT(&set);
        if (c >= 1 && c < n) n = c;
    }
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        long long period = 0;
        if (std::fscanf(f, "%63s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
            long long quota = std::atoll(q);
            unsigned c = (unsigned)((quota + period - 1) / period);
            if (c >= 1 && c < n) n = c;
        }
        std::fclose(f);
    }
    return n > 32 ? 32 : n;
}

// read-only mapping of a whole file (pages come from the page cache as the parser threads touch them)
class MappedFile {
    const char *p_ = nullptr;
    size_t n_ = 0;

  public:
    explicit MappedFile(const std::string &path) {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw Error("cannot open " + path);
        struct stat st;
        if (::fstat(fd, &st) != 0) {
            ::close(fd);
            throw Error("cannot stat " + path);
        }
        n_ = (size_t)st.st_size;
        if (n_) {
            void *m = ::mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) {
                ::close(fd);
                throw Error("cannot map " + path);
            }
            p_ = static_cast<const char *>(m);
        }
        ::close(fd);
    }
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
    ~MappedFile() {
        if (p_) ::munmap(const_cast<char *>(p_), n_);
    }
    const char *data() const { return p_; }
    size_t size() const { return n_; }
};

namespace fastparse {

inline bool is_ws(char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t'; }

// runs fn(t), t in [0, threads), on a process-wide pool of parked worker threads (a proof calls this six times; starting and
// joining sixteen threads each time cost more than the work of the smaller calls); the first exception is rethrown on the caller.
// One parallel section at a time (callers serialise on the pool's mutex); sections do not nest.
class WorkerPool {
    std::mutex mu_, call_mu_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(unsigned)> *fn_ = nullptr;
    unsigned want_ = 0, next_ = 0, running_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false, failed_ = false;
    std::string err_;

    void run(unsigned t) {
        try {
            (*fn_)(t);
        } catch (const std::exception &ex) {
            std::lock_guard<std::mutex> lk(mu_);
            if (!failed_) failed_ = true, err_ = ex.what();
        }
    }
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            wake_.wait(lk, [&] { return stop_ || (generation_ != seen && next_ < want_); });
            if (stop_) return;
            if (generation_ == seen || next_ >= want_) continue;
            while (next_ < want_) {   // take indices of the current section until none is left
                unsigned t = next_++;
                running_++;
                lk.unlock();
                run(t);
                lk.lock();
                running_--;
            }
            seen = generation_;
            if (running_ == 0) done_.notify_all();
        }
    }

  public:
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto &w : workers_) w.join();
    }
    void parallel(unsigned threads, const std::function<void(unsigned)> &fn) {
        std::lock_guard<std::mutex> call(call_mu_);
        std::unique_lock<std::mutex> lk(mu_);
        while (workers_.size() + 1 < threads) workers_.emplace_back([this] { loop(); });   // the caller is the last worker
        fn_ = &fn, want_ = threads, next_ = 0, failed_ = false, err_.clear();
        generation_++;
        lk.unlock();
        wake_.notify_all();
        lk.lock();
        while (next_ < want_) {   // the calling thread works too
            unsigned t = next_++;
            running_++;
            lk.unlock();
            run(t);
            lk.lock();
            running_--;
        }
        done_.wait(lk, [&] { return running_ == 0; });
        fn_ = nullptr, want_ = 0;
        if (failed_) throw Error(err_);
    }
    static WorkerPool &instance() {
        static WorkerPool pool;
        return pool;
    }
};
inline void parallel(unsigned threads, const std::function<void(unsigned)> &fn) {
    if (threads <= 1) {
        fn(0);
        return;
    }
    WorkerPool::instance().parallel(threads, fn);
}

// positions of every '{' in [p, p + n), found by `threads` threads over equal slices
inline std::vector<size_t> object_starts(const char *p, size_t n, unsigned threads) {
    std::vector<std::vector<size_t>> part(threads);
    parallel(threads, [&](unsigned t) {
        size_t lo = n * t / threads, hi = n * (t + 1) / threads;
        const char *q = p + lo, *end = p + hi;
        while (q < end) {
            const char *b = static_cast<const char *>(std::memchr(q, '{', (size_t)(end - q)));
            if (!b) break;
            part[t].push_back((size_t)(b - p));
            q = b + 1;
        }
    });
    std::vector<size_t> all;
    size_t total = 0;
    for (auto &v : part) total += v.size();
    all.reserve(total);
    for (auto &v : part) all.insert(all.end(), v.begin(), v.end());
    return all;
}

// one JSON object of a flat document, scanned strictly
struct Scanner {
    const char *t;
    size_t i, n;
    const char *doc;   // name for messages
    [[noreturn]] void fail(const char *what) const { throw Error(std::string(doc) + ": " + what + " at byte " + std::to_string(i)); }
    void ws() {
        while (i < n && is_ws(t[i])) i++;
    }
    void expect(char c) {
        ws();
        if (i >= n || t[i] != c) fail("unexpected character");
        i++;
    }
    bool peek(char c) {
        ws();
        return i < n && t[i] == c;
    }
    void str(size_t &b, size_t &e) {   // plain strings only (no escapes occur in keys or hex text)
        expect('"');
        b = i;
        const char *q = static_cast<const char *>(std::memchr(t + i, '"', n - i));
        if (!q) fail("unterminated string");
        e = (size_t)(q - t);
        if (std::memchr(t + b, '\\', e - b)) fail("escape in string");
        i = e + 1;
    }
    uint64_t uint() {
        ws();
        size_t b = i;
        uint64_t v = 0;
        while (i < n && t[i] >= '0' && t[i] <= '9') {
            if (v > (UINT64_MAX - 9) / 10) fail("integer too large");
            v = v * 10 + (uint64_t)(t[i] - '0');
            i++;
        }
        if (b == i) fail("expected a non-negative integer");
        return v;
    }
    bool key_is(size_t b, size_t e, const char *k) const {
        size_t l = std::strlen(k);
        return e - b == l && std::memcmp(t + b, k, l) == 0;
    }
    void skip_scalar() {   // a number, a string, true / false / null
        ws();
        if (i < n && t[i] == '"') {
            size_t b, e;
            str(b, e);
            return;
        }
        size_t b = i;
        while (i < n && t[i] != ',' && t[i] != '}' && !is_ws(t[i])) {
            if (t[i] == '[' || t[i] == '{') fail("nested value under an unknown key");
            i++;
        }
        if (b == i) fail("expected a value");
    }
    // what may stand between the end of one object and the start of the next (or the end of the document)
    void between(bool last) {
        ws();
        if (last) {
            expect(']');
            ws();
            if (i != n) fail("trailing characters after the document");
        } else {
            expect(',');
            ws();
            if (i != n) fail("unexpected text between objects");
        }
    }
};

// hex digit values; 0xff = not a digit
inline const uint8_t *hex_lut() {
    static const struct Lut {
        uint8_t v[256];
        Lut() {
            std::memset(v, 0xff, sizeof v);
            for (int c = '0'; c <= '9'; c++) v[c] = (uint8_t)(c - '0');
            for (int c = 'a'; c <= 'f'; c++) v[c] = (uint8_t)(c - 'a' + 10);
            for (int c = 'A'; c <= 'F'; c++) v[c] = (uint8_t)(c - 'A' + 10);
        }
    } lut;
    return lut.v;
}
// ScalarField::from_hex on a HexString (libs/src/iotools/mod.rs:126-146): optional 0x, big-endian digits, reduced mod r
inline ScalarField fr_from_hex_fast(const char *h, size_t len) {
    const uint8_t *lut = hex_lut();
    size_t off = (len >= 2 && h[0] == '0' && (h[1] == 'x' || h[1] == 'X')) ? 2 : 0;
    size_t nd = len - off;
    if (nd > 64) throw Error("hex scalar longer than 32 bytes");
    uint64_t w[4] = {0, 0, 0, 0};
    const char *end = h + len;
    unsigned bad = 0;
    for (size_t k = 0; k < nd; k++) {   // digit k counted from the least significant end
        uint8_t v = lut[(uint8_t)end[-1 - (ptrdiff_t)k]];
        bad |= v;
        w[k >> 4] |= (uint64_t)(v & 0xf) << (4 * (k & 15));
    }
    if (bad & 0xf0) throw Error("invalid hex digit in scalar");
    frh::U256 v;
    std::memcpy(v.l, w, 32);
    while (frh::geq(v, frh::MOD)) v = frh::sub_raw(v, frh::MOD);
    return frh::store(v);
}

// 16 hexadecimal characters -> the 64-bit word they spell (first character most significant); false if one is not a hex digit.
// SSE2 (the x86-64 baseline): range tests give the nibble of every byte, adjacent nibbles are merged in 16-bit lanes and packed.
#if defined(__SSE2__)
inline bool hex16(const char *p, uint64_t &out) {
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p));
    const __m128i d = _mm_sub_epi8(v, _mm_set1_epi8('0'));
    const __m128i l = _mm_sub_epi8(_mm_or_si128(v, _mm_set1_epi8(0x20)), _mm_set1_epi8('a'));
    const __m128i is_d = _mm_cmpeq_epi8(_mm_min_epu8(d, _mm_set1_epi8(9)), d);
    const __m128i is_l = _mm_cmpeq_epi8(_mm_min_epu8(l, _mm_set1_epi8(5)), l);
    if (_mm_movemask_epi8(_mm_or_si128(is_d, is_l)) != 0xffff) return false;
    const __m128i nib = _mm_or_si128(_mm_and_si128(is_d, d), _mm_andnot_si128(is_d, _mm_add_epi8(l, _mm_set1_epi8(10))));
    // 16-bit lane = (second character's nibble << 8) | first character's nibble  ->  byte (first << 4) | second
    const __m128i b16 = _mm_or_si128(_mm_slli_epi16(_mm_and_si128(nib, _mm_set1_epi16(0x00ff)), 4), _mm_srli_epi16(nib, 8));
    const __m128i b8 = _mm_packus_epi16(b16, b16);   // 8 bytes, most significant first
    out = __builtin_bswap64((uint64_t)_mm_cvtsi128_si64(b8));
    return true;
}
#else
inline bool hex16(const char *p, uint64_t &out) {
    const uint8_t *lut = hex_lut();
    uint64_t w = 0;
    unsigned bad = 0;
    for (int k = 0; k < 16; k++) {
        uint8_t v = lut[(uint8_t)p[k]];
        bad |= v;
        w = (w << 4) | (v & 0xf);
    }
    out = w;
    return !(bad & 0xf0);
}
#endif
// fr_from_hex_fast for the bulk path: digits are right-aligned in a 64-character field of '0' and read 16 at a time
inline ScalarField fr_from_hex_bulk(const char *h, size_t len) {
    size_t off = (len >= 2 && h[0] == '0' && (h[1] == 'x' || h[1] == 'X')) ? 2 : 0;
    const size_t nd = len - off;
    if (nd > 64) throw Error("hex scalar longer than 32 bytes");
    uint64_t w[4] = {0, 0, 0, 0};
    if (nd <= 2) {   // "0x0" / "0x1" / small constants: most of a witness
        const uint8_t *lut = hex_lut();
        unsigned bad = 0;
        for (size_t k = 0; k < nd; k++) {
            uint8_t v = lut[(uint8_t)h[off + k]];
            bad |= v;
            w[0] = (w[0] << 4) | (v & 0xf);
        }
        if (bad & 0xf0) throw Error("invalid hex digit in scalar");
    } else {
        const char *d = h + off;
        const size_t full = nd / 16, rem = nd % 16;   // whole 16-digit words from the least significant end, then the short top one
        for (size_t k = 0; k < full; k++)
            if (!hex16(d + nd - 16 * (k + 1), w[k])) throw Error("invalid hex digit in scalar");
        if (rem) {
            char buf[16];
            std::memset(buf, '0', 16 - rem);
            std::memcpy(buf + 16 - rem, d, rem);
            if (!hex16(buf, w[full])) throw Error("invalid hex digit in scalar");
        }
    }
    frh::U256 v;
    std::memcpy(v.l, w, 32);
    while (frh::geq(v, frh::MOD)) v = frh::sub_raw(v, frh::MOD);
    return frh::store(v);
}

// the `variables` array of one placement, from just after its '[' to just after its ']': strings of hex digits separated by commas.
// Returns the position after ']'.  A tight loop of its own: 3.3 million values per configs[3] proof pass through here.
inline size_t read_hex_array(const char *t, size_t i, size_t n, ScalarField *dst, uint32_t want, uint32_t &count, const char *doc) {
    auto fail = [&](const char *what, size_t at) { throw Error(std::string(doc) + ": " + what + " at byte " + std::to_string(at)); };
    uint32_t cnt = 0;
    while (i < n && is_ws(t[i])) i++;
    if (i < n && t[i] == ']') {
        count = 0;
        return i + 1;
    }
    for (;;) {
        while (i < n && is_ws(t[i])) i++;
        if (i >= n || t[i] != '"') fail("unexpected character", i);
        const size_t b = ++i;
#if defined(__SSE2__)
        for (bool found = false; !found && i + 16 <= n;) {   // closing quote, 16 bytes at a time (never reading past the mapping)
            int m = _mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(t + i)), _mm_set1_epi8('"')));
            if (m) i += (size_t)__builtin_ctz((unsigned)m), found = true;
            else i += 16;
        }
#endif
        while (i < n && t[i] != '"') i++;   // escapes cannot pass the digit test below
        if (i >= n) fail("unterminated string", b);
        if (cnt >= want) throw Error("Corrupted placement variables.");   // more values than the subcircuit has wires
        dst[cnt++] = fr_from_hex_bulk(t + b, i - b);
        i++;
        while (i < n && is_ws(t[i])) i++;
        if (i < n && t[i] == ',') {
            i++;
            continue;
        }
        if (i < n && t[i] == ']') {
            i++;
            break;
        }
        fail("unexpected character", i);
    }
    count = cnt;
    return i;
}

}  // namespace fastparse

// ---- placementVariables.json ----------------------------------------------------------------------------------------------------
struct WitnessLayout {
    std::vector<uint32_t> id;      // subcircuitId of placement q
    std::vector<uint64_t> off;     // first variable of placement q in the value buffer (elements)
    uint64_t total = 0;            // values in the buffer
};

// n_wires[id] = variables a placement of kind `id` must carry (SubcircuitInfo::Nwires = flattenMap.len(); the reference checks it in
// gen_bXY and in the R1CS evaluation).  alloc(total) returns the buffer the values are written to.
// world / rank (a sharded prover, host/tkmk_service.hpp): only the placements q = rank mod world are converted and stored — off[] and
// total then describe the buffer of THOSE placements (off[q] of another rank's placement is meaningless); the kinds (id[]) of all
// placements are still read, the structure of the whole document is still checked.
inline WitnessLayout parse_placement_variables_fast(const char *t, size_t n, const std::vector<uint32_t> &n_wires,
                                                    const std::function<ScalarField *(uint64_t)> &alloc, unsigned threads, uint32_t world = 1,
                                                    uint32_t rank = 0) {
    using namespace fastparse;
    const char *doc = "placementVariables.json";
    WitnessLayout L;
    std::vector<size_t> start = object_starts(t, n, threads);
    const size_t P = start.size();
    {   // document head: '[' (and, for an empty document, ']')
        Scanner s{t, 0, P ? start[0] : n, doc};
        s.expect('[');
        s.ws();
        if (P == 0) {
            Scanner e{t, s.i, n, doc};
            e.expect(']');
            e.ws();
            if (e.i != n) e.fail("trailing characters after the document");
            alloc(0);
            return L;
        }
        if (s.i != start[0]) s.fail("unexpected text before the first placement");
    }
    // pass 1: the kind of every placement (first key in the documents the synthesizer writes; any position accepted)
    L.id.resize(P);
    std::atomic<size_t> next{0};
    parallel(threads, [&](unsigned) {
        for (;;) {
            size_t q0 = next.fetch_add(64);
            if (q0 >= P) break;
            for (size_t q = q0; q < q0 + 64 && q < P; q++) {
                size_t end = q + 1 < P ? start[q + 1] : n;
                Scanner s{t, start[q] + 1, end, doc};
                bool found = false;
                while (!found) {
                    size_t kb, ke;
                    s.str(kb, ke);
                    s.expect(':');
                    if (s.key_is(kb, ke, "subcircuitId")) {
                        uint64_t v = s.uint();
                        if (v >= n_wires.size()) throw Error("Invalid subcircuit id in placement_variables.");
                        L.id[q] = (uint32_t)v;
                        found = true;
                    } else if (s.peek('[')) {   // the variables array: jump over it
                        const char *c = static_cast<const char *>(std::memchr(t + s.i, ']', end - s.i));
                        if (!c) s.fail("unterminated array");
                        s.i = (size_t)(c - t) + 1;
                    } else {
                        s.skip_scalar();
                    }
                    if (!found) {
                        if (s.peek(',')) s.i++;
                        else s.fail("placement without subcircuitId");
                    }
                }
            }
        }
    });
    L.off.resize(P);
    for (size_t q = 0; q < P; q++) {
        L.off[q] = L.total;
        if (q % world == rank) L.total += n_wires[L.id[q]];
    }
    ScalarField *vars = alloc(L.total);
    // pass 2: every object in full; values straight to their final position
    next = 0;
    parallel(threads, [&](unsigned) {
        for (;;) {
            size_t q = next.fetch_add(1);
            if (q >= P) break;
            if (q % world != rank) continue;   // another rank's placement: that rank converts (and checks) its values
            size_t end = q + 1 < P ? start[q + 1] : n;
            Scanner s{t, start[q] + 1, end, doc};
            ScalarField *dst = vars + L.off[q];
            const uint32_t want = n_wires[L.id[q]];
            bool has_id = false, has_vars = false;
            for (;;) {
                size_t kb, ke;
                s.str(kb, ke);
                s.expect(':');
                if (s.key_is(kb, ke, "subcircuitId")) {
                    if (has_id) s.fail("duplicate key");
                    (void)s.uint();
                    has_id = true;
                } else if (s.key_is(kb, ke, "variables")) {
                    if (has_vars) s.fail("duplicate key");
                    s.expect('[');
                    uint32_t cnt = 0;
                    s.i = read_hex_array(t, s.i, end, dst, want, cnt, doc);
                    if (cnt != want) throw Error("Corrupted placement variables.");
                    has_vars = true;
                } else {
                    s.skip_scalar();
                }
                if (s.peek(',')) {
                    s.i++;
                    continue;
                }
                s.expect('}');
                break;
            }
            if (!has_id || !has_vars) s.fail("placement without subcircuitId / variables");
            s.between(q + 1 == P);
        }
    });
    return L;
}

// ---- permutation.json -----------------------------------------------------------------------------------------------------------
struct PermutationColumns {
    std::vector<uint32_t> row, col, X, Y;
    size_t size() const { return row.size(); }
};
inline PermutationColumns parse_permutation_fast(const char *t, size_t n, unsigned threads) {
    using namespace fastparse;
    const char *doc = "permutation.json";
    PermutationColumns out;
    std::vector<size_t> start = object_starts(t, n, threads);
    const size_t P = start.size();
    {
        Scanner s{t, 0, P ? start[0] : n, doc};
        s.expect('[');
        s.ws();
        if (P == 0) {
            Scanner e{t, s.i, n, doc};
            e.expect(']');
            e.ws();
            if (e.i != n) e.fail("trailing characters after the document");
            return out;
        }
        if (s.i != start[0]) s.fail("unexpected text before the first entry");
    }
    out.row.resize(P), out.col.resize(P), out.X.resize(P), out.Y.resize(P);
    std::atomic<size_t> next{0};
    parallel(threads, [&](unsigned) {
        for (;;) {
            size_t q0 = next.fetch_add(256);
            if (q0 >= P) break;
            for (size_t q = q0; q < q0 + 256 && q < P; q++) {
                size_t end = q + 1 < P ? start[q + 1] : n;
                Scanner s{t, start[q] + 1, end, doc};
                unsigned seen = 0;
                for (;;) {
                    size_t kb, ke;
                    s.str(kb, ke);
                    s.expect(':');
                    uint32_t *dst = nullptr;
                    unsigned bit = 0;
                    if (s.key_is(kb, ke, "row")) dst = &out.row[q], bit = 1;
                    else if (s.key_is(kb, ke, "col")) dst = &out.col[q], bit = 2;
                    else if (s.key_is(kb, ke, "X")) dst = &out.X[q], bit = 4;
                    else if (s.key_is(kb, ke, "Y")) dst = &out.Y[q], bit = 8;
                    if (dst) {
                        if (seen & bit) s.fail("duplicate key");
                        uint64_t v = s.uint();
                        if (v > 0xffffffffull) s.fail("integer too large");
                        *dst = (uint32_t)v;
                        seen |= bit;
                    } else {
                        s.skip_scalar();
                    }
                    if (s.peek(',')) {
                        s.i++;
                        continue;
                    }
                    s.expect('}');
                    break;
                }
                if (seen != 15) s.fail("permutation entry without row / col / X / Y");
                s.between(q + 1 == P);
            }
        }
    });
    return out;
}

}  // namespace tkmk
