#!/usr/bin/env python3
"""Regenerates tests/golden/pins.json from the reference tree (run in the build container only).

Collects the few constants the reference pins for the MSM / NTT hot path as DATA:
  * the prime in the header of every committed .r1cs   (qap-compiler/subcircuits/library/r1cs)
  * the standard G1 generator limbs                     (setup/mpc-setup/src/conversions.rs:68-79)
  * the fixed-tau G1 generator and trapdoor scalars     (setup/trusted-setup/src/main.rs:71-74,
                                                         libs/src/field_structures/mod.rs:43-64)
  * production setup parameters                         (qap-compiler/subcircuits/library/setupParams.json)
Values are parsed out of those files' text/bytes; nothing from the reference is executed.
"""
import glob
import json
import os
import re
import struct

REF = "/root/reference/packages"
out = {}

primes = set()
files = sorted(glob.glob(REF + "/frontend/qap-compiler/subcircuits/library/r1cs/*.r1cs"))
for f in files:
    b = open(f, "rb").read()
    assert b[:4] == b"r1cs"
    nsec = struct.unpack_from("<I", b, 8)[0]
    off = 12
    for _ in range(nsec):
        typ, size = struct.unpack_from("<IQ", b, off)
        off += 12
        if typ == 1:
            fs = struct.unpack_from("<I", b, off)[0]
            primes.add(int.from_bytes(b[off + 4:off + 4 + fs], "little"))
        off += size
assert len(primes) == 1
out["r1cs_files"] = len(files)
out["r1cs_prime"] = hex(primes.pop())

src = open(REF + "/backend/setup/mpc-setup/src/conversions.rs").read()
m = re.search(r"fn icicle_g1_generator.*?x_limbs: \[u32; 12\] = \[(.*?)\];.*?y_limbs: \[u32; 12\] = \[(.*?)\];", src, re.S)
out["g1_generator_x_limbs"] = [int(t, 0) for t in re.findall(r"0x[0-9a-fA-F]+|\d+", m.group(1))]
out["g1_generator_y_limbs"] = [int(t, 0) for t in re.findall(r"0x[0-9a-fA-F]+|\d+", m.group(2))]

src = open(REF + "/backend/setup/trusted-setup/src/main.rs").read()
m = re.search(r"G1Affine::from_limbs\(\s*BaseField::from_hex\(\"(0x[0-9a-f]+)\"\).*?BaseField::from_hex\(\"(0x[0-9a-f]+)\"\)", src, re.S)
out["fixed_tau_g1_x"], out["fixed_tau_g1_y"] = m.group(1), m.group(2)
# the fixed G2 generator of the same recipe (main.rs:75-78): G2BaseField::from_hex over the whole 96-byte limb array
m = re.search(r"G2Affine::from_limbs\(\s*G2BaseField::from_hex\(\"(0x[0-9a-f]+)\"\).*?G2BaseField::from_hex\(\"(0x[0-9a-f]+)\"\)", src, re.S)
out["fixed_tau_g2_x"], out["fixed_tau_g2_y"] = m.group(1), m.group(2)

src = open(REF + "/backend/libs/src/field_structures/mod.rs").read()
body = src[src.index("pub fn gen_fixed"):]
for name in ("x", "y", "alpha", "gamma", "delta", "eta"):
    m = re.search(name + r": ScalarField::from_hex\(\s*\"(0x[0-9a-f]+)\"", body)
    out["tau_" + name] = m.group(1)

out["setup_params"] = json.load(open(REF + "/frontend/qap-compiler/subcircuits/library/setupParams.json"))

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pins.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1))

# Data fixtures for the R1CS reader / uvw evaluation tests: three small compiled subcircuits (binary iden3 .r1cs) and
# their subcircuitInfo entries, copied as DATA from the reference's committed library.
import shutil
lib = REF + "/frontend/qap-compiler/subcircuits/library/"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qap")
os.makedirs(dst + "/r1cs", exist_ok=True)
for i in (1, 2, 12):
    shutil.copy(lib + "r1cs/subcircuit%d.r1cs" % i, dst + "/r1cs/")
info = [e for e in json.load(open(lib + "subcircuitInfo.json")) if e["id"] in (1, 2, 12)]
json.dump(info, open(dst + "/subcircuitInfo.json", "w"))
json.dump(json.load(open(lib + "setupParams.json")), open(dst + "/setupParams.json", "w"))

# A second, larger set for the direct parity tests of the device-side constraint library (tests/test_gpu_witness.py,
# tests/test_r1cs_reader.py): four mid-size subcircuits of the same library (17 .. 154 KB each), same treatment — binary data files and
# their subcircuitInfo entries, nothing executed.
dst2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qap_more")
os.makedirs(dst2 + "/r1cs", exist_ok=True)
MORE = (0, 3, 7, 9)
for i in MORE:
    shutil.copy(lib + "r1cs/subcircuit%d.r1cs" % i, dst2 + "/r1cs/")
json.dump([e for e in json.load(open(lib + "subcircuitInfo.json")) if e["id"] in MORE], open(dst2 + "/subcircuitInfo.json", "w"))

# The rest of the production library (the seven largest subcircuits, 270-540 KB each) with the library's own subcircuitInfo.json and
# setupParams.json, so that the three directories together are the whole library the reference ships (tests/real_library.py
# assembles them into one directory for the reader / row-evaluation / setup -> preprocess -> prove tests).  Binary data, nothing executed.
dst3 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qap_rest")
os.makedirs(dst3 + "/r1cs", exist_ok=True)
REST = tuple(i for i in range(14) if i not in (1, 2, 12) + MORE)
for i in REST:
    shutil.copy(lib + "r1cs/subcircuit%d.r1cs" % i, dst3 + "/r1cs/")
shutil.copy(lib + "subcircuitInfo.json", dst3 + "/subcircuitInfo.json")
shutil.copy(lib + "setupParams.json", dst3 + "/setupParams.json")
