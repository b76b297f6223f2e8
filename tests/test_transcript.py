"""not-gpu tier: host-side protocol glue either side of the device path: the Keccak transcript
(tkmk/transcript.py; reference prove/src/lib.rs:3211-3731) and the Solidity-verifier formatting (tkmk/proofio.py;
libs/src/iotools/mod.rs:1625-1700, prove/src/lib.rs:452-513).  Keccak-256 is pinned on published known answers; the
reference holds no transcript or proof vectors, so the layouts are checked against a byte-level restatement written
here from the cited lines (parity unpinned beyond the hash)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tokamak-zk-evm_amd"))
from tkmk import proofio  # noqa: E402
from tkmk.transcript import RollingKeccakTranscript, TranscriptManager, keccak256  # noqa: E402


def test_native_keccak_equals_python_permutation():
    """tkmk_keccak256 (csrc/hostutil.hip, host-only) against the pure-Python sponge on every block-boundary length"""
    import random
    from tkmk.transcript import keccak256_py
    rnd = random.Random(7)
    for n in list(range(0, 140)) + [271, 272, 273, 1000]:
        data = bytes(rnd.randrange(256) for _ in range(n))
        assert keccak256(data) == keccak256_py(data), n
    assert keccak256_py(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"


def test_keccak256_known_answers():
    assert keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    # the Solidity selector everyone knows: keccak256("transfer(address,uint256)")[:4] = a9059cbb
    assert keccak256(b"transfer(address,uint256)")[:4].hex() == "a9059cbb"
    # multi-block input (rate = 136 bytes): 200 x 'a', and the exact-rate boundary cases must differ and be stable
    assert len({keccak256(b"a" * n) for n in (135, 136, 137, 200, 272)}) == 5
    # differs from SHA3-256 (other padding byte)
    import hashlib
    assert keccak256(b"abc") != hashlib.sha3_256(b"abc").digest()


def test_transcript_layout_and_challenges():
    t = RollingKeccakTranscript()
    v = bytes(range(1, 21))                                    # 20-byte value: right-aligned in the last 32-byte slot
    t.update(v)
    body = bytes(64) + bytes(12) + v
    assert t.state_0 == keccak256(b"\x00\x00\x00\x00" + body) and t.state_1 == keccak256(b"\x00\x00\x00\x01" + body)
    s0, s1 = t.state_0, t.state_1
    raw0 = keccak256(b"\x00\x00\x00\x02" + s0 + s1 + (0).to_bytes(4, "big"))
    raw1 = keccak256(b"\x00\x00\x00\x02" + s0 + s1 + (1).to_bytes(4, "big"))
    c0, c1 = t.get_challenge(), t.get_challenge()
    assert c0 == int.from_bytes(bytes([raw0[0] & 0x1F]) + raw0[1:], "big") and c1 == int.from_bytes(bytes([raw1[0] & 0x1F]) + raw1[1:], "big")
    assert c0 < (1 << 253) and c0 != c1 and t.challenge_counter == 2
    assert (t.state_0, t.state_1) == (s0, s1)                  # challenges do not move the state
    with pytest.raises(ValueError):
        t.update(bytes(33))
    # both new state words derive from the OLD pair
    t2 = RollingKeccakTranscript()
    t2.update(b"\x01")
    t2.update(b"\x02")
    a0 = keccak256(b"\x00\x00\x00\x00" + bytes(64) + bytes(31) + b"\x01")
    a1 = keccak256(b"\x00\x00\x00\x01" + bytes(64) + bytes(31) + b"\x01")
    assert t2.state_1 == keccak256(b"\x00\x00\x00\x01" + a0 + a1 + bytes(31) + b"\x02")


def test_base_field_and_point_commit_split():
    x = int.from_bytes(bytes(range(100, 148)), "big")
    y = int.from_bytes(bytes(range(10, 58)), "big")
    pt = np.frombuffer(x.to_bytes(48, "little") + y.to_bytes(48, "little"), np.uint8)
    a, b = RollingKeccakTranscript(), RollingKeccakTranscript()
    a.commit_g1(pt)
    for c in (x, y):
        be = c.to_bytes(48, "big")
        b.update(bytes(16) + be[:16])
        b.update(be[16:])
    assert (a.state_0, a.state_1) == (b.state_0, b.state_1)
    s = RollingKeccakTranscript()
    s.commit_scalar(5)
    r = RollingKeccakTranscript()
    r.update(bytes(31) + b"\x05")
    assert s.state_0 == r.state_0


def test_round_order_is_deterministic_and_sensitive():
    pts = [np.frombuffer(bytes([k + 1]) * 96, np.uint8) for k in range(9)]

    def run(swap=False):
        m = TranscriptManager()
        p = list(pts)
        if swap:
            p[0], p[1] = p[1], p[0]
        m.add_proof0(*p[:6])
        th = m.get_thetas()
        m.add_proof1(p[6])
        k0 = m.get_kappa0()
        m.add_proof2(p[7], p[8])
        chi, zeta = m.get_chi_zeta()
        m.add_proof3(1, 2, 3, 4)
        return th + [k0, chi, zeta, m.get_kappa1()]

    a, b, c = run(), run(), run(swap=True)
    assert a == b and a != c and len(set(a)) == 7


def test_proof_formatting_round_trip():
    rng = np.random.default_rng(1)
    points = {k: rng.integers(0, 256, 96, dtype=np.uint8) for k in proofio.PROOF_POINT_ORDER}
    points["A_free"] = np.zeros(96, np.uint8)                  # G1serde::zero()
    scalars = {k: int(rng.integers(1, 1 << 62)) ** 4 for k in proofio.PROOF_SCALAR_ORDER}
    fmt = proofio.format_proof(points, scalars)
    assert len(fmt["proof_entries_part1"]) == 38 and len(fmt["proof_entries_part2"]) == 42
    assert all(len(e) == 2 + 32 for e in fmt["proof_entries_part1"]) and all(len(e) == 2 + 64 for e in fmt["proof_entries_part2"])
    x_be = bytes(points["U"][:48])[::-1]
    assert fmt["proof_entries_part1"][0] == "0x" + x_be[:16].hex() and fmt["proof_entries_part2"][0] == "0x" + x_be[16:].hex()
    y_be = bytes(points["U"][48:])[::-1]
    assert fmt["proof_entries_part1"][1] == "0x" + y_be[:16].hex()
    assert fmt["proof_entries_part2"][38] == "0x" + scalars["R_eval"].to_bytes(32, "big").hex()
    p2, s2 = proofio.recover_proof(fmt)
    assert all((p2[k] == points[k]).all() for k in points) and s2 == scalars
    pre = {k: points[k2] for k, k2 in zip(proofio.PREPROCESS_POINT_ORDER, ("U", "V", "W"))}
    f2 = proofio.format_preprocess(pre)
    assert len(f2["preprocess_entries_part1"]) == 6 and len(f2["preprocess_entries_part2"]) == 6
    back = proofio.recover_preprocess(f2)
    assert all((back[k] == pre[k]).all() for k in pre)
    with pytest.raises(ValueError):
        proofio.recover_preprocess({"preprocess_entries_part1": f2["preprocess_entries_part1"][:4], "preprocess_entries_part2": f2["preprocess_entries_part2"]})
