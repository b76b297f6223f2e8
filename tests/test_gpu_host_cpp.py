"""gpu tier: the C++ host mirror (tokamak-zk-evm_amd/host/tkmk_host.hpp: DeviceVec, DensePolynomialExt, PolyExpr, Sigma1)
driven by tests/host_cpp/host_driver.cpp; every dumped result is compared with the oracle's restatement of the
reference (and the reference's own test identities).  This is the compiled-language host side above the C ABI."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(HERE, "host_cpp", "host_driver")


def _records(path):
    raw = open(path, "rb").read()
    out, off = {}, 0
    while off < len(raw):
        tag, n = struct.unpack_from("<IQ", raw, off)
        off += 12
        out[tag] = raw[off:off + n]
        off += n
    return out


def _poly(rec, tag):
    xs, ys, xd, yd = struct.unpack("<qqqq", rec[tag])
    return xs, ys, xd, yd, np.frombuffer(rec[tag + 1], np.uint8)


def test_cpp_host_mirror_against_oracle(gpu, oracle, tmp_path):
    if not os.path.exists(DRIVER):     # normally built by __graft_entry__.build(); same recipe
        pkg = os.path.join(os.path.dirname(HERE), "tokamak-zk-evm_amd")
        subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(pkg, "host"), DRIVER + ".cpp", "-o", DRIVER, "-L" + pkg, "-ltkmk_hip",
                        "-Wl,-rpath," + pkg], check=True)
    pins = json.load(open(os.path.join(HERE, "golden", "pins.json")))
    R = oracle.R_MOD
    xs, ys, bxs, bys, c, d, rs_x, rs_y = 16, 8, 8, 4, 4, 2, 16, 8
    a = oracle.fr_random(1, xs * ys).reshape(xs, ys, 32).copy()
    a[13:, :, :] = 0                      # degree (12, 6)
    a[:, 7:, :] = 0
    a = a.reshape(-1)
    b = oracle.fr_random(2, bxs * bys)
    sc = oracle.fr_random(3, 5)
    fx, fy, x, y, s = (sc[32 * i:32 * (i + 1)].copy() for i in range(5))
    tx, ty = int(pins["tau_x"], 16), int(pins["tau_y"], 16)
    g = oracle.to_bytes([int(pins["fixed_tau_g1_x"], 16), int(pins["fixed_tau_g1_y"], 16)], 48)
    mon = [pow(tx, i, R) * pow(ty, j, R) % R for i in range(rs_x) for j in range(rs_y)]
    crs = gpu.g1_batch_scalar_mul_device(gpu.DeviceBuffer.from_host(oracle.to_bytes(mon, 32)), g, rs_x * rs_y).to_host()
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<8I", xs, ys, bxs, bys, c, d, rs_x, rs_y))
        for arr in (a, b, sc, crs):
            f.write(arr.tobytes())
    gpu.release_scratch()
    subprocess.run([DRIVER, str(inp), str(outp)], check=True, timeout=300)
    rec = _records(outp)

    assert struct.unpack("<qq", rec[1]) == oracle.poly_find_degree(a, xs, ys) == (12, 6)
    oxs, oys, oxd, oyd, oc = _poly(rec, 10)                                   # optimize_size
    want, nx, ny = oracle.poly_resize(a, xs, ys, 13, 7)
    assert (oxs, oys, oxd, oyd) == (nx, ny, 12, 6) and (oc == want).all()
    assert (_poly(rec, 20)[4] == oracle.poly_scale_coeffs(a, xs, ys, fx, fy)).all()
    assert (np.frombuffer(rec[30], np.uint8) == oracle.poly_eval(a, xs, ys, x, y)).all()
    assert (_poly(rec, 32)[4] == oracle.poly_eval_x(a, xs, ys, x)).all() and _poly(rec, 32)[:2] == (1, ys)
    assert (_poly(rec, 34)[4] == oracle.poly_eval_y(a, xs, ys, y)).all() and _poly(rec, 34)[:2] == (xs, 1)
    # A * B  (mod.rs:1846-1996): product box (12+7+1, 6+3+1) -> 32 x 16
    pxs, pys, _, _, pc = _poly(rec, 40)
    ra, _, _ = oracle.poly_resize(a, xs, ys, 32, 16)
    rb, _, _ = oracle.poly_resize(b, bxs, bys, 32, 16)
    prod = oracle.bintt(oracle.fr_mul(oracle.bintt(ra, 32, 16), oracle.bintt(rb, 32, 16)), 32, 16, inverse=True)
    assert (pxs, pys) == (32, 16) and (pc == prod).all()
    # (A + B) - s*B on the common 16 x 8 shape
    rb2, _, _ = oracle.poly_resize(b, bxs, bys, xs, ys)
    assert (_poly(rec, 42)[4] == oracle.fr_sub(oracle.fr_add(a, rb2), oracle.fr_scalar_mul(s, rb2))).all()
    # coset evaluations and the way back
    assert (np.frombuffer(rec[50], np.uint8) == oracle.bintt(a, xs, ys, coset_x=fx, coset_y=fy)).all()
    assert (_poly(rec, 52)[4] == a).all()
    # divisions (the driver optimises the numerator first: 16 x 8 stays 16 x 8)
    qx, qy = oracle.poly_div_by_vanishing_opt(a, xs, ys, c, d)
    assert (_poly(rec, 60)[4] == qx).all() and (_poly(rec, 62)[4] == qy).all()
    assert _poly(rec, 60)[2:4] == (xs - c - 1, ys - 1) and _poly(rec, 62)[2:4] == (c - 1, ys - d - 1)
    rx, ry, rr = oracle.poly_div_by_ruffini(a, xs, ys, x, y)
    assert (_poly(rec, 70)[4] == rx).all() and (_poly(rec, 72)[4] == ry).all() and (np.frombuffer(rec[74], np.uint8) == rr).all()
    # fused expression A*B + s*(X-1)*B - (fx*A + fy*B), checked at a random point
    fxs, fys, _, _, fc = _poly(rec, 80)
    px, py = oracle.fr_random(9, 1), oracle.fr_random(10, 1)
    ea, eb = oracle.poly_eval(a, xs, ys, px, py), oracle.poly_eval(b, bxs, bys, px, py)
    one = oracle.to_bytes([1], 32)
    want = oracle.fr_sub(oracle.fr_add(oracle.fr_mul(ea, eb), oracle.fr_mul(s, oracle.fr_mul(oracle.fr_sub(px, one), eb))),
                         oracle.fr_add(oracle.fr_mul(fx, ea), oracle.fr_mul(fy, eb)))
    assert (oracle.poly_eval(fc.copy(), fxs, fys, px, py) == want).all() and (fxs, fys) == (32, 16)
    # commitment == [A(tau_x, tau_y)] G ; zero polynomial -> (0, 0)
    val = oracle.poly_eval(a, xs, ys, oracle.to_bytes([tx], 32), oracle.to_bytes([ty], 32))
    assert (np.frombuffer(rec[90], np.uint8) == oracle.g1_scalar_mul(val, g)).all()
    assert not np.frombuffer(rec[92], np.uint8).any()
    many = np.frombuffer(rec[94], np.uint8)                # encode_polys({A, 0, B})
    valb = oracle.poly_eval(b, bxs, bys, oracle.to_bytes([tx], 32), oracle.to_bytes([ty], 32))
    assert (many[:96] == oracle.g1_scalar_mul(val, g)).all() and not many[96:192].any()
    assert (many[192:] == oracle.g1_scalar_mul(valb, g)).all()
    assert struct.unpack("<I", rec[99])[0] == 7          # the three misuse cases raised tkmk::Error
