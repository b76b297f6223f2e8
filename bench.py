#!/usr/bin/env python3
"""bench.py — headline measurement of the hot path on MI355X (one process per GPU).

A "step" is one pass of the hot path over one batch of synthetic input resident in HBM:
  N = 1 : BASELINE.json configs[1] — one 2^24-point G1 Pippenger MSM.  (Curve: BLS12-381, the only curve
          the reference uses — SURVEY.md §0.2; BASELINE.json's "BN254" has no reference counterpart.)
  N > 1 : the same per-GPU shard on every rank (weak scaling, SURVEY.md §8e: the MSM shards by points),
          partial results exchanged with ONE RCCL all_gather of 144-byte points, then summed by a
          world_size-point MSM with unit scalars on every rank.
Inputs are generated on the device from a seed: scalars = splitmix64 stream mod r,
bases P_i = [h_i]G.  Prints ONE JSON line (rank 0).

value = "MSM field-adds/sec": group additions per second at the algorithmic count SURVEY.md §8d fixes for
this config (ceil(255/16) = 16 bucket-accumulate adds per point, independent of the window width the
kernel actually picks, so the figure is points/s x 16 and cannot be inflated by doing more work).

Secondary objects on the same line (N = 1 only; each records {"error": ...} instead of costing the line if it fails):
  cpu_baseline  the oracle's Pippenger on a 2^20-point sample of the same stream, on the box's host cores
  bn254_msm     the same MSM kernels over BN254 (BASELINE.json configs[1] as worded)
  ntt           BASELINE.json configs[2] (256 x 2^20 scalar-field NTTs) and the production _biNTT shapes
  prove         BASELINE.json's "constraints/sec (prove step)": the whole prover (init + prove0..4) on synthetic satisfying circuits at the
                reference's production shape (2^20 constraint slots) and at configs[3]'s 2^22 slots, the production shape through the native
                binary on files, the proof's algorithmic bytes against the HBM peak, and a CPU estimate from the oracle's measured rates
With --gpus N --prove-dist the production-shape proof is also timed with the commitments of each round spread over the ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MAD_PEAK_PER_S = 3.2e13          # measured v_mad_u64_u32 issue rate, lane-ops/s (profiles/r01_arith_microbench_v3.json)
MADS_PER_BUCKET_ADD = 7 * 392 + 196 + 2 * 301   # madd-2008-s on 14 x 29-bit limbs: 8 products (two share one reduction) + 2 squares (csrc/ffu.h, ec_u.h)
ADDS_PER_POINT = 16              # SURVEY.md §8d cfg 2: N * ceil(b/c) at c = 16, b = 255
ALG_BYTES_PER_POINT = 32 + 96    # each scalar and base read once
SEED = 0x746F6B616D616B00


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--logn", type=int, default=24, help="log2 of points per GPU (default: BASELINE configs[1])")
    ap.add_argument("--cpu-sample-logn", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true", help="skip the secondary NTT measurements (BASELINE.json configs[2], _biNTT)")
    ap.add_argument("--no-bn254", action="store_true", help="skip the secondary BN254 G1 MSM measurement (BASELINE.json configs[1] as worded)")
    ap.add_argument("--no-prove", action="store_true",
                    help="skip the full-prove measurements (BASELINE.json configs[3] and the reference's production shape)")
    ap.add_argument("--prove-dist", action="store_true",
                    help="N > 1 only: also time the production-shape prove with each round's commitments spread over the ranks")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI; default) or gloo (rehearsal of the N>1 path)")
    ap.add_argument("--share-gpu0", action="store_true", help="rehearsal only: every rank uses GPU 0 (needs --dist-backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal only: initialise the process group and run the partial-result exchange even with one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)" % (args.gpus, world))

    import torch
    import tkmk
    from tkmk import sharding
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    dev = 0 if args.share_gpu0 else local_rank
    torch.cuda.set_device(dev)
    tkmk.set_device(dev)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")     # only reached without a launcher (--force-dist rehearsal)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    comm_device = "cuda" if args.dist_backend == "nccl" else "cpu"

    n = 1 << args.logn
    # --- synthetic inputs, generated in HBM (untimed) ---
    scalars = tkmk.fr_random_device(SEED + 2 + 16 * rank, n)
    h = tkmk.fr_random_device(SEED + 3 + 16 * rank, n)
    g = np.frombuffer(bytes(_generator()), np.uint8).copy()
    bases = tkmk.g1_batch_scalar_mul_device(h, g, n)
    h.free()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        tkmk.synchronize()

    def step():
        return sharding.msm_sharded(tkmk, dist, scalars, bases, device=comm_device)   # 144-byte canonical projective

    for _ in range(args.warmup):
        step()
    if args.force_dist and world == 1:   # one-rank rehearsal of the exchange msm_sharded performs for N > 1
        part = tkmk.msm(scalars, bases)
        assert (sharding.combine_partials(tkmk, sharding.gather_partials(dist, part, comm_device)) == part).all()
    tkmk.profile_enable(True)
    tkmk.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
    barrier()
    elapsed = time.perf_counter() - t0
    tkmk.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    prove_dist = None
    if dist is not None and world > 1 and args.prove_dist:      # every rank takes part: replicated rounds, commitments by owner
        scalars.free()
        bases.free()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import prove_bench
        r = prove_bench.run(s_max=256, placements=166, repeat=3, dist=dist, comm_device=comm_device)
        prove_dist = {"workload": r["workload"] + ", commitments of each round spread over %d ranks" % world, "wall_s": r["seconds"]["total"],
                      "init_s": r["seconds"]["init"], "rounds_s": r["seconds"]["rounds"], "constraints_per_s": r["constraint_slots_per_s"],
                      "per_round_s": {k: r["seconds"][k] for k in ("prove0", "prove1", "prove2", "prove3", "prove4")}}

    acc_ms, acc_cnt = tkmk.profile_get("msm.accumulate")
    sections = {}
    for name in ("convert_bases", "digits", "hist", "scan", "scatter", "accumulate", "combine", "reduce_segments",
                 "reduce_windows"):
        ms, cnt = tkmk.profile_get("msm." + name)
        if cnt:
            sections[name] = round(ms / cnt, 4)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        points_per_s = n * world * args.steps / elapsed
        value = points_per_s * ADDS_PER_POINT
        # dominant kernel: k_accumulate (one launch per MSM of n points)
        # one big launch per step (for N > 1 the world-point combine MSM adds a negligible second one)
        kernel_ms = acc_ms / args.steps if acc_cnt else float("nan")
        alg_bytes = n * ALG_BYTES_PER_POINT
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = _traffic_from_profiles(args.logn)
        out = {
            "metric": "MSM field-adds/sec",
            "value": value,
            "unit": "group-adds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (381-bit Fq / 255-bit Fr modular integer arithmetic)",
            "data": "synthetic",
            "config": {"workload": "2^%d-point BLS12-381 G1 Pippenger MSM per GPU (BASELINE.json configs[1]), inputs resident in HBM"
                                   % args.logn,
                       "points_per_gpu": n, "sharding": "points" if world > 1 else "none"},
            "points_per_s": points_per_s,
            "roofline": {"bound": "hbm", "kernel": "k_accumulate_chunks", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "integer-VALU bound by construction (SURVEY.md §8d): ~375 int mul-adds per algorithmic byte"},
            "kernel_ms": sections,
            # the bound that actually applies: 32x32->64 integer multiply-add issue in k_accumulate_chunks
            "valu_roofline": {"kernel": "k_accumulate_chunks", "unit": "v_mad_u64_u32 lane-ops/s",
                              "achieved": n * ADDS_PER_POINT * MADS_PER_BUCKET_ADD / (kernel_ms * 1e-3),
                              "peak": MAD_PEAK_PER_S,
                              "frac": n * ADDS_PER_POINT * MADS_PER_BUCKET_ADD / (kernel_ms * 1e-3) / MAD_PEAK_PER_S},
        }
        if prove_dist is not None:
            out["prove_dist"] = prove_dist
        if world > 1:      # the CPU baseline and the secondary figures are N = 1 material; keep the scaling runs lean
            args.no_cpu_baseline = args.no_bn254 = args.no_ntt = args.no_prove = True
        def leg(key, fn):
            """a secondary figure must never cost the headline line: its failure is recorded under its key instead"""
            try:
                out[key] = fn()
            except Exception as e:      # noqa: BLE001
                out[key] = {"error": "%s: %s" % (type(e).__name__, e)}

        if not args.no_cpu_baseline:
            leg("cpu_baseline", lambda: _cpu_baseline(tkmk, args.cpu_sample_logn))
        if (not args.no_bn254 or not args.no_ntt) and prove_dist is None:
            scalars.free()
            bases.free()
        if not args.no_bn254:
            leg("bn254_msm", lambda: _bn254_secondary(tkmk, args.logn))
        if not args.no_ntt:
            leg("ntt", lambda: _ntt_secondary(tkmk))
        if not args.no_prove:
            leg("prove", lambda: _prove_secondary(tkmk))
            prod, cpu = out["prove"].get("production_2p20", {}), out.get("cpu_baseline", {})
            if "msm_points" in prod and "points_per_s" in cpu:
                try:
                    out["prove"]["cpu_estimate"] = _prove_cpu_estimate(tkmk, prod, cpu)
                except Exception as e:      # noqa: BLE001
                    out["prove"]["cpu_estimate"] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["result_x_lo"] = int.from_bytes(bytes(result[:8]), "little")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def _generator():
    # standard G1 generator, u32 LE limbs (reference pin: setup/mpc-setup/src/conversions.rs:68-79)
    x = [0xdb22c6bb, 0xfb3af00a, 0xf97a1aef, 0x6c55e83f, 0x171bac58, 0xa14e3a3f, 0x9774b905, 0xc3688c4f, 0x4fa9ac0f,
         0x2695638c, 0x3197d794, 0x17f1d3a7]
    y = [1187375073, 212476713, 2726857444, 3493644100, 738505709, 14358731, 3587181302, 4243972245, 1948093156,
         2694721773, 3819610353, 146011265]
    return b"".join(int(v).to_bytes(4, "little") for v in x + y)


def _bn254_secondary(tkmk, logn):
    """BASELINE.json configs[1] as worded ("2^24-point BN254 G1 Pippenger MSM"): the same kernels instantiated over the
    254-bit fields (csrc/msm_bn254.hip).  Secondary because the reference has no BN254 path."""
    n = 1 << logn
    s = tkmk.fr_random_device(SEED + 5, n, curve="bn254")
    h = tkmk.fr_random_device(SEED + 6, n, curve="bn254")
    g = np.zeros(64, np.uint8)
    g[0], g[32] = 1, 2
    b = tkmk.g1_batch_scalar_mul_device(h, g, n, curve="bn254")
    h.free()
    tkmk.msm(s, b, curve="bn254")
    tkmk.profile_enable(True)
    tkmk.profile_reset()
    tkmk.synchronize()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        tkmk.msm(s, b, curve="bn254")
    tkmk.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tkmk.profile_enable(False)
    acc_ms, acc_cnt = tkmk.profile_get("msm.accumulate")
    s.free()
    b.free()
    return {"workload": "2^%d-point BN254 G1 Pippenger MSM, inputs resident in HBM" % logn, "ms_per_msm": dt * 1e3,
            "points_per_s": n / dt, "group_adds_per_s": n / dt * ADDS_PER_POINT,
            "accumulate_kernel_ms": acc_ms / acc_cnt if acc_cnt else None,
            "hbm_frac_of_peak": (n * (32 + 64) / (acc_ms / acc_cnt * 1e-3) / 1e9 / HBM_PEAK_GBS) if acc_cnt else None}


def _traffic_from_profiles(logn):
    """HBM bytes per k_accumulate_chunks launch from the committed rocprofv3 PMC summary (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        key = "msm_accumulate_2^%d" % logn
        return t.get(key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def _ntt_secondary(tkmk):
    """Secondary, outside the timed region: BASELINE.json configs[2] (256 independent 2^20-point NTTs, resident) and the
    reference's largest production _biNTT (16384 x 512).  Algorithmic bytes = 64 B per element."""
    res = {}
    try:
        tkmk.init_ntt_domain_for_size(1 << 23)
        for name, n, batch, bi in (("rows_256x2^20", 1 << 20, 256, None), ("bintt_16384x512", None, None, (16384, 512))):
            elems = n * batch if bi is None else bi[0] * bi[1]
            a = tkmk.fr_random_device(SEED + 5, elems)
            o = tkmk.DeviceBuffer(32 * elems)
            fn = (lambda: tkmk.ntt(a, n, batch=batch, out=o)) if bi is None else (lambda: tkmk.bintt(a, bi[0], bi[1], out=o))
            fn()
            tkmk.synchronize()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            tkmk.synchronize()
            dt = (time.perf_counter() - t0) / reps
            # per-pass kernel times from HIP events on the launch stream (a separate, profiled repetition)
            tkmk.profile_enable(True)
            tkmk.profile_reset()
            fn()
            tkmk.profile_enable(False)
            passes = []
            for k in range(8):
                ms, cnt = tkmk.profile_get("ntt.pass%d" % k)
                if cnt:
                    passes.append(ms / cnt)
            moved = 64 * elems * len(passes)                   # every pass reads and writes each element once
            res[name] = {"ms": dt * 1e3, "elements_per_s": elems / dt, "algorithmic_GBps": 64 * elems / dt / 1e9,
                         "hbm_frac": 64 * elems / dt / 1e9 / HBM_PEAK_GBS, "passes_ms": [round(x, 4) for x in passes],
                         "moved_GBps": (moved / (sum(passes) * 1e-3) / 1e9) if passes else None,
                         "moved_hbm_frac": (moved / (sum(passes) * 1e-3) / 1e9 / HBM_PEAK_GBS) if passes else None,
                         "bound": "integer VALU (Fr products): see DESIGN.md section 4"}
            a.free()
            o.free()
    except Exception as e:  # secondary figure: never fail the headline line
        res["error"] = str(e)
    return res


def _prove_secondary(tkmk):
    """BASELINE.json's "constraints/sec (prove step)": the whole prover (tkmk/prove.py: init + prove0..prove4, every
    polynomial and commitment on the device) on synthetic satisfying circuits (tools/synth_circuit.py, fixed-tau CRS), at
    the reference's production shape with its placement count (2^20 constraint slots; reference walls 45.70 s CPU / 21.08 s
    CUDA on other hardware) and at configs[3]'s 2^22 slots (s_max = 1024, every placement used).
    constraints_per_s = constraint slots / (init + rounds) wall, host glue included."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import prove_bench
    tkmk.release_scratch()            # the arenas of the 2^24-point MSM / 16 GiB NTT legs go back to the driver first
    out = {}
    for key, kw in (("production_2p20", dict(s_max=256, placements=166, repeat=4)), ("configs3_2p22", dict(s_max=1024, repeat=2))):
        try:
            r = prove_bench.run(**kw)
        except Exception as e:      # noqa: BLE001
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            continue
        out[key] = {"workload": r["workload"], "constraints_per_s": r["constraint_slots_per_s"], "r1cs_rows_per_s": r["r1cs_rows_per_s"],
                    "wall_s": r["seconds"]["total"], "init_s": r["seconds"]["init"], "rounds_s": r["seconds"]["rounds"],
                    "per_round_s": {k: r["seconds"][k] for k in ("prove0", "prove1", "prove2", "prove3", "prove4")},
                    "constraint_slots": r["constraint_slots"], "r1cs_rows": r["r1cs_rows"],
                    "msm_points": r["seconds"]["msm_points"], "ntt_elements": r["seconds"]["ntt_elements"], "hbm_roofline": r["hbm_roofline"]}
        tkmk.release_scratch()
    # the same production-shape proof through the native binary (C++ host side): files in the reference's formats in, proof.json
    # out, a fresh process per run, CRS payload and JSON inputs loaded inside the timed total
    try:
        r = prove_bench.run_native(s_max=256, placements=166, repeat=2)
    except Exception as e:      # noqa: BLE001
        out["native_production_2p20"] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["reference_wall_s"] = {"cpu": 45.70, "cuda": 21.08, "note": "production shape, 166 placements, other hardware (BASELINE.md §1)"}
        return out
    out["native_production_2p20"] = {"workload": r["workload"], "constraints_per_s": r["constraint_slots_per_s"],
                                     "constraints_per_s_init_plus_rounds": r["constraint_slots_per_s_init_plus_rounds"],
                                     "wall_s": r["seconds"]["total"], "seconds": r["seconds"], "sigma_gen_s": r["sigma_gen_s"],
                                     "crs_payload_bytes": r["crs_payload_bytes"], "placement_variables_json_bytes": r["placement_variables_json_bytes"]}
    tkmk.release_scratch()
    out["reference_wall_s"] = {"cpu": 45.70, "cuda": 21.08, "note": "production shape, 166 placements, other hardware (BASELINE.md §1)"}
    return out


def _prove_cpu_estimate(tkmk, prod, msm_baseline):
    """What the MSM + NTT share of the production-shape proof costs on this box's host cores with the oracle's C port (the reference's
    Rust + ICICLE CPU prover cannot be built offline; its own published run is 45.7 s on ~8 threads of other hardware).  Not a
    measured proof: the proof's counted work (points committed, NTT elements) times the oracle's measured rates — the MSM rate
    from cpu_baseline's sample, the NTT rate from one 1024 x 1024 bivariate transform timed here and checked against the GPU.
    A lower bound for a CPU prover (polynomial bookkeeping, divisions and host glue come on top)."""
    import oracle
    threads = msm_baseline["cores"]
    xs = ys = 1024
    a = tkmk.fr_random_device(SEED + 9, xs * ys)
    ah = a.to_host()
    oracle.bintt(ah[:32 * 64 * 64], 64, 64)
    t0 = time.perf_counter()
    want = oracle.bintt(ah, xs, ys)
    dt = time.perf_counter() - t0
    ok = bool((np.asarray(tkmk.bintt(a, xs, ys).to_host()) == np.asarray(want)).all())
    ntt_rate = xs * ys / dt
    msm_s = prod["msm_points"] / msm_baseline["points_per_s"]
    ntt_s = prod["ntt_elements"] / ntt_rate
    return {"kind": "port, extrapolated from bounded samples (MSM + NTT share only)", "cores": threads, "msm_points_per_s": msm_baseline["points_per_s"],
            "ntt_elements_per_s": ntt_rate, "ntt_sample": "one 1024 x 1024 bivariate NTT, %.3f s, matches the GPU: %s" % (dt, ok),
            "msm_s": msm_s, "ntt_s": ntt_s, "seconds": msm_s + ntt_s, "gpu_wall_s": prod["wall_s"],
            "constraints_per_s": prod["constraint_slots"] / (msm_s + ntt_s)}


def _usable_cpus(omp_threads):
    """threads the CPU baseline may really use: OpenMP's default counts every hardware thread of the host, but a one-GPU
    box only gets a share of them (affinity mask and / or cgroup CPU quota); oversubscribing that share slows the run"""
    n = omp_threads
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_baseline(tkmk, sample_logn):
    """The oracle's Pippenger (a C port; the reference's Rust+ICICLE CPU path cannot be built offline) on a
    bounded sample of the same workload, all host threads, checked against the GPU on that sample."""
    import oracle
    m = 1 << sample_logn
    s = tkmk.fr_random_device(SEED + 2, m)
    hh = tkmk.fr_random_device(SEED + 3, m)
    g = np.frombuffer(bytes(_generator()), np.uint8).copy()
    b = tkmk.g1_batch_scalar_mul_device(hh, g, m)
    sh, bh = s.to_host(), b.to_host()
    threads = _usable_cpus(oracle.num_threads())
    t0 = time.perf_counter()
    want = oracle.g1_msm(sh, bh, threads=threads)
    dt = time.perf_counter() - t0
    got = tkmk.projective_to_affine_bytes(tkmk.msm(s, b))
    return {"value": m / dt * ADDS_PER_POINT, "unit": "group-adds/s", "cores": threads, "kind": "port",
            "sample": "one 2^%d-point MSM (first 2^%d points of the benchmark stream), %.2f s" % (sample_logn, sample_logn, dt),
            "points_per_s": m / dt, "matches_gpu": bool((got == want).all())}


if __name__ == "__main__":
    main()
