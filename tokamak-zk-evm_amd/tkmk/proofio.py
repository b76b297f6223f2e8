"""Solidity-verifier formatting of commitments and scalars: the work-alike of split_g1 / scalar_to_hex / split_push! /
pop_recover! (packages/backend/libs/src/iotools/mod.rs:1625-1700), FormattedProof (prove/src/lib.rs:452-513) and
FormattedPreprocess (preprocess/src/lib.rs:84-146).  Points are the 96-byte affine records (48-byte little-endian x, y)
the MSM returns; (0,0) is G1serde::zero()."""
import json

import numpy as np

PROOF_POINT_ORDER = ("U", "V", "W", "O_mid", "O_prv", "Q_AX", "Q_AY", "Q_CX", "Q_CY", "Pi_X", "Pi_Y", "B", "R", "M_Y", "M_X", "N_Y",
                     "N_X", "O_pub_free", "A_free")                                      # prove/src/lib.rs:460-501
PROOF_SCALAR_ORDER = ("R_eval", "R_omegaX_eval", "R_omegaX_omegaY_eval", "V_eval")        # :504-507
PREPROCESS_POINT_ORDER = ("s0", "s1", "O_pub_fix")                                        # preprocess/src/lib.rs:93-101


def split_g1(point96):
    """-> (x_part1, x_part2, y_part1, y_part2): big-endian, top 16 bytes / low 32 bytes (iotools/mod.rs:1625-1647)"""
    b = bytes(point96)
    out = []
    for c in (b[:48], b[48:]):
        be = c[::-1]
        out += ["0x" + be[:16].hex(), "0x" + be[16:].hex()]
    return tuple(out)


def scalar_to_hex(value: int):
    """32-byte big-endian (iotools/mod.rs:1649-1658)"""
    return "0x" + int(value).to_bytes(32, "big").hex()


def recover_g1(x1, x2, y1, y2):
    """next_point / recover_basefield (iotools/mod.rs:1675-1693)"""
    def base(p1, p2):
        be = bytes.fromhex(p1[2:] if p1.startswith("0x") else p1) + bytes.fromhex(p2[2:] if p2.startswith("0x") else p2)
        if len(be) != 48:
            raise ValueError("Invalid format")
        return be[::-1]
    return np.frombuffer(base(x1, x2) + base(y1, y2), np.uint8).copy()


def _format(points, order):
    p1, p2 = [], []
    for name in order:
        x1, x2, y1, y2 = split_g1(points[name])
        p1 += [x1, y1]
        p2 += [x2, y2]
    return p1, p2


def format_proof(points: dict, scalars: dict):
    """FormattedProof: 38 part-1 entries, 38 + 4 part-2 entries"""
    p1, p2 = _format(points, PROOF_POINT_ORDER)
    p2 += [scalar_to_hex(scalars[k]) for k in PROOF_SCALAR_ORDER]
    return {"proof_entries_part1": p1, "proof_entries_part2": p2}


def recover_proof(fmt: dict):
    p1, p2 = fmt["proof_entries_part1"], fmt["proof_entries_part2"]
    n = len(PROOF_POINT_ORDER)
    if len(p1) != 2 * n or len(p2) != 2 * n + len(PROOF_SCALAR_ORDER):
        raise ValueError("unexpected proof entry count")
    points = {name: recover_g1(p1[2 * i], p2[2 * i], p1[2 * i + 1], p2[2 * i + 1]) for i, name in enumerate(PROOF_POINT_ORDER)}
    scalars = {name: int(p2[2 * n + i], 16) for i, name in enumerate(PROOF_SCALAR_ORDER)}
    return points, scalars


def format_preprocess(points: dict):
    p1, p2 = _format(points, PREPROCESS_POINT_ORDER)
    return {"preprocess_entries_part1": p1, "preprocess_entries_part2": p2}


def recover_preprocess(fmt: dict):
    p1, p2 = fmt["preprocess_entries_part1"], fmt["preprocess_entries_part2"]
    n = len(PREPROCESS_POINT_ORDER)
    if len(p1) != 2 * n or len(p2) != 2 * n:            # assert_eq!(p1.len(), G1_CNT * 2) (preprocess/src/lib.rs:127-128)
        raise ValueError("unexpected preprocess entry count")
    return {name: recover_g1(p1[2 * i], p2[2 * i], p1[2 * i + 1], p2[2 * i + 1]) for i, name in enumerate(PREPROCESS_POINT_ORDER)}


def write_json(path, obj):
    with open(path, "w") as f:
        json.dump(obj, f, indent=2)
