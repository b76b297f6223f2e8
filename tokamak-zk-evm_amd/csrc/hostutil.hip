// Host-only helpers behind the C ABI (no device code).
//   tkmk_keccak256 — Keccak-256 with the original 0x01 padding, the hash of the prover's Fiat-Shamir transcript
//   (packages/backend/prove/src/lib.rs:3247-3394 calls tiny_keccak::Keccak::v256 at every state update and challenge:
//   ~90 permutations per proof).  The Python host side (tkmk/transcript.py) and any other binding use this entry.
#include "common.h"

#include <cstdint>
#include <cstring>

namespace {

inline uint64_t rol64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

void keccak_f1600(uint64_t a[25]) {   // lane (x, y) at a[x + 5 y]
    static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
                                    0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                                    0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                                    0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                                    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                                    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    // rotation offsets r[x + 5 y]
    static const int R[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) {
            uint64_t d = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
            for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
        }
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[x + 5 * y], R[x + 5 * y]);
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];
    }
}

}  // namespace

// Index of the constraints section of an iden3 .r1cs file (packages/backend/libs/src/iotools/mod.rs:613-650 walks the same
// layout): 3 linear combinations per constraint, each a u32 count followed by count x (u32 wire, field_size-byte coefficient).
// starts[3 r + m] = byte offset (from `section`) of the first record of matrix m in constraint r, counts[3 r + m] = its length.
// TKMK_SUCCESS when all 3 * n_constraints combinations lie inside the section (*consumed = bytes walked: the caller compares it
// with section_bytes to detect trailing bytes); TKMK_ERR_INVALID_ARGUMENT when the section ends inside the walk.
TK_API tkmk_error tkmk_r1cs_index(const uint8_t *section, size_t section_bytes, uint32_t n_constraints, uint32_t field_size,
                                  uint64_t *starts, uint32_t *counts, size_t *consumed) {
    if (!section || !starts || !counts) return TKMK_ERR_INVALID_POINTER;
    size_t off = 0;
    const size_t rec = 4 + (size_t)field_size;
    for (uint64_t k = 0; k < (uint64_t)n_constraints * 3; k++) {
        if (off + 4 > section_bytes) {
            if (consumed) *consumed = off;
            return TKMK_ERR_INVALID_ARGUMENT;
        }
        uint32_t cnt;
        std::memcpy(&cnt, section + off, 4);
        off += 4;
        starts[k] = off;
        counts[k] = cnt;
        if ((size_t)cnt > (section_bytes - off) / rec) {
            if (consumed) *consumed = off;
            return TKMK_ERR_INVALID_ARGUMENT;
        }
        off += (size_t)cnt * rec;
    }
    if (consumed) *consumed = off;
    return TKMK_SUCCESS;
}

TK_API tkmk_error tkmk_keccak256(const uint8_t *data, size_t len, uint8_t out[32]) {
    if ((!data && len) || !out) return TKMK_ERR_INVALID_ARGUMENT;
    const size_t rate = 136;
    uint64_t a[25] = {};
    uint8_t block[136];
    size_t off = 0;
    for (;;) {
        size_t take = len - off < rate ? len - off : rate;
        std::memset(block, 0, rate);
        if (take) std::memcpy(block, data + off, take);
        bool last = take < rate;
        if (last) {
            block[take] ^= 0x01;
            block[rate - 1] ^= 0x80;
        }
        for (size_t i = 0; i < rate / 8; i++) {
            uint64_t w;
            std::memcpy(&w, block + 8 * i, 8);      // little-endian host (x86-64)
            a[i] ^= w;
        }
        keccak_f1600(a);
        off += take;
        if (last) break;
    }
    std::memcpy(out, a, 32);
    return TKMK_SUCCESS;
}

// ScalarCfg::generate_random (icicle_core::traits::GenerateRandom; the prover's blinding scalars: prove/src/lib.rs:1040-1080): n uniform
// elements of Fr into a HOST buffer — 255 bits from the kernel's CSPRNG (getrandom), redrawn until below r (rejection sampling: no
// modular bias).  Host-only: no device is touched.
#include <errno.h>
#include <sys/random.h>
TK_API tkmk_error bls12_381_generate_scalars(tkmk_fr *out_host, size_t n) {
    if (!out_host && n) return TKMK_ERR_INVALID_POINTER;
    static const uint32_t R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    for (size_t i = 0; i < n;) {
        uint8_t b[32];
        size_t got = 0;
        while (got < sizeof b) {
            ssize_t k = getrandom(b + got, sizeof b - got, 0);
            if (k < 0) {
                if (errno == EINTR) continue;
                return TKMK_ERR_UNKNOWN;
            }
            got += (size_t)k;
        }
        b[31] &= 0x7f;
        uint32_t l[8];
        std::memcpy(l, b, 32);
        bool below = false;
        for (int k = 7; k >= 0; k--) {
            if (l[k] != R[k]) {
                below = l[k] < R[k];
                break;
            }
        }
        if (!below) continue;
        std::memcpy(out_host[i].limbs, l, 32);
        i++;
    }
    return TKMK_SUCCESS;
}
