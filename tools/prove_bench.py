"""Full `prove` (init + prove0..prove4, tkmk/prove.py) on a synthetic circuit of the production shape, timed on one MI355X.

Shapes: --s-max 256 is the reference's production shape (n = 4096, m_I = 4096, s_max = 256: 2^20 constraint slots;
reference walls 45.70 s CPU / 21.08 s CUDA with 166 placements); --s-max 1024 is BASELINE.json configs[3]'s
"2^22-constraint circuit" (SURVEY.md §8d cfg 4).  The circuit comes from tools/synth_circuit.py (random satisfying
subcircuits with 60 % of the private wires constant bits / small values, as real witnesses are; iden3 .r1cs files on disk, synthesizer
documents handed over in memory); the CRS is the fixed-tau trusted setup of
that circuit, generated on the device by tkmk/setup.py (Sigma.gen: xy_powers and the QAP-derived binding tables).
--check runs the reference's testing-mode assertions (R1CS satisfaction, Lemma 3, quotient identities, zero remainders)
at full size and compares three commitments with [P(tau_x, tau_y)]G.

Prints one JSON line: seconds per stage (best of --repeat), constraint slots/s and real R1CS rows/s over init + rounds."""
import argparse
import json
import os
import random
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tokamak-zk-evm_amd"), os.path.join(ROOT, "tools")]

PINS = json.load(open(os.path.join(ROOT, "tests", "golden", "pins.json")))       # the fixed tau / G of setup/trusted-setup/src/main.rs:68-80
TAU_X, TAU_Y = int(PINS["tau_x"], 16), int(PINS["tau_y"], 16)


def stage_crs(tkmk, inst):
    """the reference string of the fixed-tau trusted setup for this circuit, generated on the device (tkmk/setup.py)"""
    from tkmk.setup import Sigma
    g = np.frombuffer(int(PINS["fixed_tau_g1_x"], 16).to_bytes(48, "little") + int(PINS["fixed_tau_g1_y"], 16).to_bytes(48, "little"), np.uint8).copy()
    tau = {k: int(PINS["tau_" + k], 16) for k in ("x", "y", "alpha", "gamma", "delta", "eta")}
    from tkmk import g2
    h2 = g2.from_hex_pair(PINS["fixed_tau_g2_x"], PINS["fixed_tau_g2_y"])      # Sigma2 (ten host-side G2 points): the verifier's side of the CRS
    return Sigma.gen(inst["setup_params"], tau, inst["qap"], inst["infos"], g, h2), g


# The production library's parameters that the synthetic full-size circuits share with it (packages/frontend/qap-compiler/subcircuits/library/
# setupParams.json = tests/golden/pins.json "setup_params"): s_D = 14 subcircuit kinds (4 public buffers + 10 gate kinds here), m_D = 26591
# global wires, i.e. 21767 private ones, spread over the gate kinds; l = 728, l_D = 4824, n = 4096 as before.
PRODUCTION_GATE_KINDS = 10
PRODUCTION_N_PRV = [2177] * 9 + [2174]


def run(s_max=256, placements=None, pool=24, n_prv=None, repeat=3, check=False, seed=0x746F6B616D616B04, profile_host=False,
        dist=None, comm_device="cuda"):
    """dist = torch.distributed with an initialised group: every rank calls run() with the same arguments, replicates the
    polynomial work and runs only the commitments it owns (Sigma1.dist, sharding.commits_balanced)"""
    import synth_circuit
    import tkmk
    from tkmk.prove import Prover, fr, random_mixer, run_rounds
    t = time.perf_counter()
    inst = synth_circuit.generate(random.Random(seed), s_max=s_max, n_gate_kinds=PRODUCTION_GATE_KINDS, n_out=80, n_in=170, n_prv=PRODUCTION_N_PRV if n_prv is None else n_prv, k_out=65, k_pub=20,
                                  l_free=128, l_extra=600, n=4096, m_i=4096, pool=pool, used_placements=placements, bit_fraction=0.6)
    tmp = tempfile.mkdtemp(prefix="tkmk_prove_bench_")
    synth_circuit.write(inst, tmp, synth_files=False)
    sp = inst["setup_params"]
    gen_s = time.perf_counter() - t
    t = time.perf_counter()
    sigma_obj, g = stage_crs(tkmk, inst)
    if dist is not None:
        sigma_obj.sigma1.dist, sigma_obj.sigma1.comm_device = dist, comm_device
    sigma = sigma_obj.prover_view()
    tkmk.synchronize()
    crs_s = time.perf_counter() - t
    inputs = {"setup_params": sp, "subcircuit_infos": inst["infos"], "placement_variables": inst["placement_variables"],
              "permutation": inst["permutation"], "instance": inst["instance"]}
    slots = sp["n"] * sp["s_max"]
    best, runs = None, []
    for rep in range(repeat):
        tkmk.synchronize()
        if dist is not None:
            dist.barrier()
        tkmk.stats_reset()
        t0 = time.perf_counter()
        prover, binding = Prover.init_from(inputs, inst["qap"], mixer=random_mixer(random.Random(rep)), testing_mode=check, sigma=sigma)
        tkmk.synchronize()
        init_s = time.perf_counter() - t0
        points, scalars, challenges, _, times = run_rounds(prover, binding)
        total = time.perf_counter() - t0
        rec = dict({k: round(v, 4) for k, v in prover.timing.items()}, **{k: round(v, 4) for k, v in times.items()})
        rec["msm_points"], rec["ntt_elements"] = tkmk.STATS["msm_points"], tkmk.STATS["ntt_elements"]
        rec["init"] = round(init_s, 4)
        rec["rounds"] = round(sum(times.values()), 4)
        rec["total"] = round(total, 4)
        runs.append(rec)
        if best is None or rec["total"] < best["total"]:
            best = rec
        if check and rep == 0:
            # commit identity on three of the proof's polynomials: [P(tau_x, tau_y)]G through a 1-point MSM
            for name, poly in (("B", prover.bXY + prover.cache["term_b_zk"]), ("R", None), ("A_free", prover.a_free_X)):
                if poly is None:
                    mx = prover.mixer
                    poly = prover.rXY + (prover.t_mi.mul_scalar(fr(mx["rR_X"])) + prover.t_smax.mul_scalar(fr(mx["rR_Y"])))
                want = tkmk.projective_to_affine_bytes(tkmk.msm(poly.eval(fr(TAU_X), fr(TAU_Y)), g))
                assert (np.asarray(points[name]) == np.asarray(want)).all(), "commit identity fails for " + name
        del prover
    if profile_host:                                       # where the host side of one more prove spends its time (stderr)
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        prover, binding = Prover.init_from(inputs, inst["qap"], mixer=random_mixer(random.Random(0)), sigma=sigma)
        run_rounds(prover, binding)
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
        del prover
    del sigma, sigma_obj
    return {"workload": "prove (init + prove0..4): synthetic circuit n=%d m_I=%d s_max=%d, %d placements, %d real R1CS rows" % (
        sp["n"], sp["l_D"] - sp["l"], sp["s_max"], len(inst["placement_variables"]), inst["r1cs_rows"]),
        "setup_params": sp, "constraint_slots": slots, "r1cs_rows": inst["r1cs_rows"], "seconds": best, "runs": runs,
        "constraint_slots_per_s": round(slots / best["total"]), "r1cs_rows_per_s": round(inst["r1cs_rows"] / best["total"]),
        "constraint_slots_per_s_rounds_only": round(slots / best["rounds"]),
        # SURVEY.md §8d algorithmic bytes of the proof's MSMs (128 B per point: scalar + base read once) and (bi)NTTs (64 B per
        # element: read once, written once) over the wall time, against 8 TB/s — both kernels are integer-VALU bound, so this is small
        "algorithmic_bytes": 128 * best["msm_points"] + 64 * best["ntt_elements"],
        "hbm_roofline": {"achieved_GBps": round((128 * best["msm_points"] + 64 * best["ntt_elements"]) / best["total"] / 1e9, 1),
                         "achieved_GBps_rounds_only": round((128 * best["msm_points"] + 64 * best["ntt_elements"]) / best["rounds"] / 1e9, 1),
                         "peak_GBps": 8000.0,
                         "frac": round((128 * best["msm_points"] + 64 * best["ntt_elements"]) / best["total"] / 1e9 / 8000.0, 5)},
        "generate_s": round(gen_s, 2), "stage_crs_s": round(crs_s, 2), "checked": bool(check)}


def run_native(s_max=256, placements=None, pool=24, n_prv=None, repeat=3, seed=0x746F6B616D616B04, compare=False):
    """the same workload through tokamak-zk-evm_amd/bin/prove (C++ host side): every input as a file in the reference's formats,
    the CRS as the TKCRS001 payload; times are the binary's own printout (inputs + CRS loading included in its total)"""
    import re
    import shutil
    import subprocess
    import synth_circuit
    import tkmk
    inst = synth_circuit.generate(random.Random(seed), s_max=s_max, n_gate_kinds=PRODUCTION_GATE_KINDS, n_out=80, n_in=170, n_prv=PRODUCTION_N_PRV if n_prv is None else n_prv, k_out=65, k_pub=20,
                                  l_free=128, l_extra=600, n=4096, m_i=4096, pool=pool, used_placements=placements, bit_fraction=0.6)
    tmp = tempfile.mkdtemp(prefix="tkmk_prove_native_")
    try:
        synth_circuit.write(inst, tmp)
        sp = inst["setup_params"]
        t_setup = time.perf_counter()
        sigma_obj, g = stage_crs(tkmk, inst)
        tkmk.synchronize()
        setup_s = time.perf_counter() - t_setup
        sigma_obj.write(os.path.join(tmp, "crs"))
        crs_bytes = os.path.getsize(os.path.join(tmp, "crs", "combined_sigma.tkcrs"))
        pv_bytes = os.path.getsize(os.path.join(inst["synth"], "placementVariables.json"))
        same = None
        if compare:      # the Python prover and the binary on the same inputs with the same blinding scalars: identical proof.json content
            from tkmk import proofio
            from tkmk.prove import Prover, random_mixer, run_rounds
            mixer = random_mixer(random.Random(7))
            hx = lambda v: [hx(e) for e in v] if isinstance(v, list) else "0x%x" % v      # noqa: E731
            json.dump({k: hx(v) for k, v in mixer.items()}, open(os.path.join(tmp, "mixer.json"), "w"))
            prover, binding = Prover.init(inst["qap"], inst["synth"], None, mixer=mixer, sigma=sigma_obj.prover_view())
            points, scalars, _, _, _ = run_rounds(prover, binding)
            want = proofio.format_proof(points, scalars)
            del prover
        del sigma_obj
        tkmk.release_scratch()
        binary = os.path.join(ROOT, "tokamak-zk-evm_amd", "bin", "prove")
        cmd = [binary, "--crs", os.path.join(tmp, "crs"), "--synthesizer-stat", inst["synth"], "--output", os.path.join(tmp, "out"),
               "--subcircuit-library", inst["qap"]]
        os.makedirs(os.path.join(tmp, "out"))
        if compare:
            r = subprocess.run([binary + "-testing"] + cmd[1:] + ["--testing-mixer", os.path.join(tmp, "mixer.json")], capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                raise RuntimeError(r.stderr)
            same = json.load(open(os.path.join(tmp, "out", "proof.json"))) == want
            if not same:
                raise RuntimeError("native proof.json differs from the Python prover's for the same blinding scalars")
        runs = []
        for _ in range(repeat):
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError(r.stderr)
            rec = {m.group(1): float(m.group(2)) for m in re.finditer(r"^(\S+)\s+([0-9.]+) s\b", r.stdout, re.M)}
            rec["total"] = float(re.search(r"Total elapsed time: ([0-9.]+)s", r.stdout).group(1))
            rec["rounds"] = round(sum(rec["prove%d" % k] for k in range(5)), 4)
            rec["process_wall"] = round(wall, 3)
            runs.append(rec)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    best = min(runs, key=lambda r: r["total"])
    slots = sp["n"] * sp["s_max"]
    compute = best["init.total"] + best["rounds"]
    return {"workload": "native bin/prove (files in, proof.json out): synthetic circuit n=%d m_I=%d s_max=%d, %d placements, %d real R1CS rows" % (
        sp["n"], sp["l_D"] - sp["l"], sp["s_max"], len(inst["placement_variables"]), inst["r1cs_rows"]),
        "constraint_slots": slots, "r1cs_rows": inst["r1cs_rows"], "seconds": best, "runs": runs,
        "constraint_slots_per_s": round(slots / best["total"]), "constraint_slots_per_s_init_plus_rounds": round(slots / compute),
        "crs_payload_bytes": crs_bytes, "placement_variables_json_bytes": pv_bytes, "sigma_gen_s": round(setup_s, 3),
        "equals_python_prover": same}


def stage_files(s_max=256, placements=None, pool=24, n_prv=None, seed=0x746F6B616D616B04, tmp=None):
    """everything `prove` reads, as files in the reference's formats: <tmp>/qap (subcircuit library), <tmp>/synth (the synthesizer's
    three documents), <tmp>/crs/combined_sigma.tkcrs (the fixed-tau trusted setup of that circuit, generated on the device by
    tkmk/setup.py).  -> dict(dirs, setup params, counts); the caller removes `tmp`"""
    import synth_circuit
    import tkmk
    t = time.perf_counter()
    inst = synth_circuit.generate(random.Random(seed), s_max=s_max, n_gate_kinds=PRODUCTION_GATE_KINDS, n_out=80, n_in=170,
                                  n_prv=PRODUCTION_N_PRV if n_prv is None else n_prv, k_out=65, k_pub=20,
                                  l_free=128, l_extra=600, n=4096, m_i=4096, pool=pool, used_placements=placements, bit_fraction=0.6)
    tmp = tempfile.mkdtemp(prefix="tkmk_prove_files_") if tmp is None else tmp
    synth_circuit.write(inst, tmp)
    gen_s = time.perf_counter() - t
    t = time.perf_counter()
    sigma_obj, _ = stage_crs(tkmk, inst)
    tkmk.synchronize()
    sigma_s = time.perf_counter() - t
    t = time.perf_counter()
    sigma_obj.write(os.path.join(tmp, "crs"))
    del sigma_obj
    tkmk.release_scratch()
    sp = inst["setup_params"]
    return {"tmp": tmp, "qap": inst["qap"], "synth": inst["synth"], "crs": os.path.join(tmp, "crs"), "setup_params": sp,
            "constraint_slots": sp["n"] * sp["s_max"], "r1cs_rows": inst["r1cs_rows"], "placements": len(inst["placement_variables"]),
            "generate_s": round(gen_s, 2), "sigma_gen_s": round(sigma_s, 2), "crs_write_s": round(time.perf_counter() - t, 2),
            "crs_payload_bytes": os.path.getsize(os.path.join(tmp, "crs", "combined_sigma.tkcrs")),
            "placement_variables_json_bytes": os.path.getsize(os.path.join(inst["synth"], "placementVariables.json")),
            "permutation_json_bytes": os.path.getsize(os.path.join(inst["synth"], "permutation.json")),
            "workload": "synthetic circuit n=%d m_I=%d s_max=%d, %d placements, %d real R1CS rows" % (
                sp["n"], sp["l_D"] - sp["l"], sp["s_max"], len(inst["placement_variables"]), inst["r1cs_rows"])}


def run_service(s_max=256, placements=None, pool=24, n_prv=None, repeat=5, warmup=1, seed=0x746F6B616D616B04, check=False):
    """the same workload through the resident prover (libtkmk_prover.so: host/tkmk_service.hpp): context opened once, then `repeat`
    proofs from the synthesizer's files to proof.json; per-stage seconds are the library's own (median over the repeats)"""
    import shutil
    import statistics
    import tkmk
    from tkmk import service
    files = stage_files(s_max, placements, pool, n_prv, seed)
    try:
        t = time.perf_counter()
        p = service.Prover(files["qap"], files["crs"])
        open_s = time.perf_counter() - t
        out_dir = os.path.join(files["tmp"], "out")
        runs = []
        for rep in range(warmup + repeat):
            tkmk.native_stats_reset()
            t0 = time.perf_counter()
            doc, tm = p.prove(files["synth"], out_dir, want_json=check and rep == 0)
            tm["wall_s"] = time.perf_counter() - t0
            tm.update(tkmk.native_stats())
            if rep >= warmup:
                runs.append(tm)
        verified = None
        if check:
            verified = _verify_from_files(files, out_dir)
        p.close()
    finally:
        shutil.rmtree(files["tmp"], ignore_errors=True)
    med = {k: statistics.median(r[k] for r in runs) for k in runs[0]}
    slots = files["constraint_slots"]
    return {"workload": "resident prover (files in, proof.json out): " + files["workload"], "constraint_slots": slots, "r1cs_rows": files["r1cs_rows"],
            "median": {k: round(v, 5) if isinstance(v, float) else v for k, v in med.items()}, "runs": runs, "open_context_s": round(open_s, 3),
            "constraint_slots_per_s": round(slots / med["wall_s"]), "r1cs_rows_per_s": round(files["r1cs_rows"] / med["wall_s"]),
            "init_fraction": round(med["init_s"] / med["wall_s"], 3), "verified_with_pairings": verified,
            "files": {k: files[k] for k in ("crs_payload_bytes", "placement_variables_json_bytes", "permutation_json_bytes", "sigma_gen_s", "crs_write_s")}}


def _verify_from_files(files, out_dir):
    """the proof the service wrote, checked from public files alone with real pairings (tests/prove_ref.verify_snark_pairing)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import verify_files
    return verify_files.verify(files["qap"], files["synth"], files["crs"], out_dir)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--service", action="store_true", help="time the resident prover (libtkmk_prover.so) on files")
    ap.add_argument("--native", action="store_true", help="time tokamak-zk-evm_amd/bin/prove on files instead of the Python prover")
    ap.add_argument("--compare", action="store_true", help="with --native: also check the binary's proof.json against the Python prover's")
    ap.add_argument("--s-max", type=int, default=256)
    ap.add_argument("--placements", type=int, default=None, help="used placements (default: all s_max; the reference's run has 166)")
    ap.add_argument("--pool", type=int, default=24, help="distinct gate witnesses")
    ap.add_argument("--n-prv", type=int, default=None, help="private wires per gate kind (default: the production library's 21767 over 10 kinds)")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--profile-host", action="store_true", help="cProfile of one more init + rounds, printed to stderr")
    ap.add_argument("--seed", type=int, default=0x746F6B616D616B04)
    args = ap.parse_args()
    import tkmk
    tkmk.set_device(0)
    if args.service:
        out = run_service(args.s_max, args.placements, args.pool, args.n_prv, args.repeat, 1, args.seed, args.check)
    elif args.native:
        out = run_native(args.s_max, args.placements, args.pool, args.n_prv, args.repeat, args.seed, args.compare)
    else:
        out = run(args.s_max, args.placements, args.pool, args.n_prv, args.repeat, args.check, args.seed, args.profile_host)
    out["reference_wall_s"] = {"cpu": 45.70, "cuda": 21.08, "note": "production shape, 166 placements, other hardware (BASELINE.md)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
