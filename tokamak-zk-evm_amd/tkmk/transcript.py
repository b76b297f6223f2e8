"""Fiat-Shamir transcript of the prover: the work-alike of RollingKeccakTranscript / TranscriptManager
(packages/backend/prove/src/lib.rs:3211-3525; byte layout fixed by the Solidity verifier).  Host-side protocol glue
around the device path: it consumes the commitments encode_poly returns and produces the challenges the next round's
polynomial work is parameterised by.

Keccak-256 (original Keccak padding 0x01, not SHA3's 0x06) is implemented here because hashlib only carries SHA3;
it is pinned on the published known answers (tests/test_transcript.py).  The reference holds no transcript vectors, so
the state-update layout below is a restatement of the cited lines ("parity unpinned" beyond the hash itself)."""

_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M = (1 << 64) - 1


def _rol(x, n):
    return ((x << n) | (x >> (64 - n))) & _M if n else x


def _keccak_f(a):
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ (~b[(x + 1) % 5][y] & b[(x + 2) % 5][y] & _M) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def keccak256(data: bytes) -> bytes:
    """Keccak-256 through the library's host entry tkmk_keccak256 (a proof makes ~90 calls; the pure-Python permutation
    below costs 0.4 ms each and is kept as the cross-check, tests/test_transcript.py)"""
    import ctypes
    import tkmk
    data = bytes(data)
    out = ctypes.create_string_buffer(32)
    tkmk._check(tkmk.lib().tkmk_keccak256(data, ctypes.c_size_t(len(data)), out), "tkmk_keccak256")
    return out.raw


def keccak256_py(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        a = _keccak_f(a)
    return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


class RollingKeccakTranscript:
    """prove/src/lib.rs:3211-3400: two 32-byte state words, DST tags 0 / 1 (state update) and 2 (challenge)"""

    def __init__(self):
        self.state_0 = bytes(32)
        self.state_1 = bytes(32)
        self.challenge_counter = 0

    def update(self, data: bytes):
        if len(data) > 32:
            raise ValueError("Input must be 32 bytes or less")
        body = self.state_0 + self.state_1 + bytes(32 - len(data)) + bytes(data)   # value right-aligned in its 32-byte slot
        self.state_0, self.state_1 = keccak256(b"\x00\x00\x00\x00" + body), keccak256(b"\x00\x00\x00\x01" + body)

    def get_challenge_raw(self) -> bytes:
        buf = b"\x00\x00\x00\x02" + self.state_0 + self.state_1 + self.challenge_counter.to_bytes(4, "big")
        self.challenge_counter += 1
        return keccak256(buf)

    def get_challenge(self) -> int:
        """-> scalar as an int: top 3 bits of the big-endian hash cleared (FR_MASK), zero mapped to one.
        (ScalarField::from_bytes_le of the 253-bit value; it is below 2^253 < r, so no reduction happens.)"""
        raw = bytearray(self.get_challenge_raw())
        raw[0] &= 0x1F
        v = int.from_bytes(bytes(raw), "big")
        return v if v else 1

    def get_challenges(self, count):
        return [self.get_challenge() for _ in range(count)]

    def commit_scalar(self, value: int):
        """commit_field_as_bytes for an Fr element: 32 bytes big-endian (:3416-3426)"""
        self.update(int(value).to_bytes(32, "big"))

    def commit_base_field(self, value: int):
        """commit_bls12_381_field_element (:3429-3480): 48-byte big-endian value as (top 16 bytes left-padded to 32, low 32)"""
        be = int(value).to_bytes(48, "big")
        self.update(bytes(16) + be[:16])
        self.update(be[16:])

    def commit_g1(self, point96):
        """commit_g1_point (:3482-3500): x then y; point96 = 96-byte affine record (48-byte LE x, y)"""
        b = bytes(point96)
        self.commit_base_field(int.from_bytes(b[:48], "little"))
        self.commit_base_field(int.from_bytes(b[48:], "little"))


class TranscriptManager:
    """Commit order of the prover's rounds (prove/src/lib.rs:3528-3731)"""

    def __init__(self):
        self.transcript = RollingKeccakTranscript()

    def add_proof0(self, U, V, W, Q_AX, Q_AY, B):
        for p in (U, V, W, Q_AX, Q_AY, B):
            self.transcript.commit_g1(p)

    def get_thetas(self):
        return self.transcript.get_challenges(3)

    def add_proof1(self, R):
        self.transcript.commit_g1(R)

    def get_kappa0(self):
        return self.transcript.get_challenge()

    def add_proof2(self, Q_CX, Q_CY):
        self.transcript.commit_g1(Q_CX)
        self.transcript.commit_g1(Q_CY)

    def get_chi_zeta(self):
        return self.transcript.get_challenges(2)

    def add_proof3(self, V_eval, R_eval, R_omegaX_eval, R_omegaX_omegaY_eval):
        for s in (V_eval, R_eval, R_omegaX_eval, R_omegaX_omegaY_eval):
            self.transcript.commit_scalar(s)

    def get_kappa1(self):
        return self.transcript.get_challenge()
