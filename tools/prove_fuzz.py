"""Randomized differential run of the GPU prover against the big-int restatement (tests/prove_ref.py) over random circuit shapes:
every proof point, evaluation and challenge must match, and the combined verifier equation must hold on the discrete logarithms.
usage: python tools/prove_fuzz.py [CASES] [FIRST_SEED] [native] [mid] [sharded]      (test infrastructure: uses the oracle)
With `native` the proof comes from tokamak-zk-evm_amd/bin/prove-testing over files (CRS written by Sigma.write, blinding through
--testing-mixer); with `sharded` (implies native) the same files also go through the sharded resident prover over the loopback
transport with 2, 4 or 8 virtual ranks (tkmk_prover_open_sharded), with and without the commit / Lagrange tables, and every rank's
document must be the binary's byte for byte."""
import json
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tokamak-zk-evm_amd"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import oracle  # noqa: E402
import prove_ref  # noqa: E402
import synth_circuit  # noqa: E402
import tkmk  # noqa: E402
from tkmk import g2, proofio  # noqa: E402
from tkmk.prove import Prover, random_mixer, run_rounds  # noqa: E402
from tkmk.setup import Sigma  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
sharded = "sharded" in sys.argv[3:]
native = "native" in sys.argv[3:] or sharded
mid = "mid" in sys.argv[3:]            # m_I = 64 .. 128 instead of 16 .. 64 (the restatement then needs 1 .. 15 s per case)
pins = json.load(open(os.path.join(ROOT, "tests", "golden", "pins.json")))
tau = {k: int(pins["tau_" + k], 16) for k in ("x", "y", "alpha", "gamma", "delta", "eta")}
g = oracle.to_bytes([int(pins["fixed_tau_g1_x"], 16), int(pins["fixed_tau_g1_y"], 16)], 48)
tkmk.set_device(0)
for seed in range(first, first + cases):
    rnd = random.Random(seed)
    s_max = rnd.choice([8, 8, 16])
    k_out, k_pub = rnd.choice([0, 1, 2]), rnd.choice([1, 2, 3])
    l_free = 4 if k_out + k_pub <= 4 else 8
    shape = dict(s_max=s_max, n_gate_kinds=rnd.choice([1, 2, 3]), n_out=rnd.choice([1, 2, 3]), n_in=rnd.choice([1, 2, 3]), n_prv=rnd.randrange(1, 13),
                 k_out=k_out, k_pub=k_pub, l_free=l_free, l_extra=rnd.choice([1, 2, 4, 5]), used_placements=rnd.randrange(5, s_max + 1),
                 bit_fraction=rnd.choice([0.0, 0.3, 0.9]))
    if mid:
        s_max = rnd.choice([16, 32])
        shape.update(s_max=s_max, n_gate_kinds=rnd.choice([4, 5, 6]), n_out=rnd.choice([3, 4, 6]), n_in=rnd.choice([3, 5, 8]), n_prv=rnd.randrange(4, 28),
                     used_placements=rnd.randrange(5, s_max + 1))
    d = tempfile.mkdtemp(prefix="tkmk_fuzz_")
    inst = synth_circuit.build(d, rnd, **shape)
    sp = inst["setup_params"]
    sigma = Sigma.gen(sp, tau, inst["qap"], inst["infos"], np.frombuffer(bytes(g), np.uint8), None)
    mixer = random_mixer(random.Random(seed))
    crs = prove_ref.sigma_gen(inst, tau)
    dlogs, ref_scalars, ref_ch, ref_p4t, rp = prove_ref.run(inst, crs, mixer, g)
    if native:
        sigma.write(os.path.join(d, "crs"))
        os.makedirs(os.path.join(d, "out"))
        hx = lambda v: [hx(e) for e in v] if isinstance(v, list) else "0x%x" % v      # noqa: E731
        json.dump({k: hx(v) for k, v in mixer.items()}, open(os.path.join(d, "mixer.json"), "w"))
        r = subprocess.run([os.path.join(ROOT, "tokamak-zk-evm_amd", "bin", "prove-testing"), "--crs", os.path.join(d, "crs"), "--synthesizer-stat", inst["synth"],
                            "--output", os.path.join(d, "out"), "--subcircuit-library", inst["qap"], "--testing-mixer", os.path.join(d, "mixer.json")],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (seed, shape, r.stderr)
        native_doc = json.load(open(os.path.join(d, "out", "proof.json")))
        points, scalars = proofio.recover_proof(native_doc)
        assert scalars == ref_scalars, (seed, shape)
        if sharded:
            from tkmk import dist, service
            fits = [w for w in (2, 4, 8) if w <= min(sp["n"], sp["l_D"] - sp["l"], sp["s_max"])]   # a power of two, at most the rows / columns there are
            if fits:
                world = rnd.choice(fits)
                os.environ["TKMK_PROVER_TABLE_C"] = rnd.choice(["12", "0"])
                comms = dist.loopback_comms(world)
                provers = dist.run_ranks(comms, lambda c: service.Prover(inst["qap"], os.path.join(d, "crs"), testing=True, comm=c))
                by_rank = {p.comm.rank: p for p in provers}
                docs = dist.run_ranks(comms, lambda c: by_rank[c.rank].prove(inst["synth"], None, testing_mixer_json=os.path.join(d, "mixer.json"))[0])
                for p in provers:
                    p.close()
                for c in comms:
                    c.close()
                assert all(doc == native_doc for doc in docs), (seed, shape, world)
    else:
        prover, binding = Prover.init(inst["qap"], inst["synth"], None, mixer=mixer, testing_mode=True, sigma=sigma.prover_view())
        points, scalars, challenges, p4t, _ = run_rounds(prover, binding)
        assert challenges == ref_ch and scalars == ref_scalars, (seed, shape)
    for name in proofio.PROOF_POINT_ORDER:
        assert (np.asarray(points[name]) == np.asarray(prove_ref.g1_of(dlogs[name], g))).all(), (seed, shape, name)
    pre = prove_ref.preprocess(rp, inst, crs)
    assert prove_ref.verify_snark(dlogs, ref_scalars, ref_ch, crs, sp, pre, rp.a_free, 1 + seed), (seed, shape)
    print(seed, "ok", sp, flush=True)
print("all", cases, "cases equal the restatement and verify")
