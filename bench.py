#!/usr/bin/env python3
"""bench.py — headline measurement of the hot path on MI355X (one process per GPU).

METRIC (BASELINE.json): constraints/sec for the full prove step on the 2^22-constraint synthesized circuit.

A "step" is ONE FULL PROOF of BASELINE.json configs[3] — Prover::init + prove0..prove4 + proof.json, the timed region of the
reference's own binary after check_device (packages/backend/prove/src/main.rs:41-84; its timing harness:
prove/optimization/tests/timing.rs:101-125) — through the resident native prover (libtkmk_prover.so: include/tkmk_prover.h,
host/tkmk_service.hpp, C++ over the C ABI of libtkmk_hip.so).  Nothing inside the step is Python: one ctypes call per proof.
  workload  : n = 4096, m_I = 4096, s_max = 1024 (2^22 constraint slots, every placement used; SURVEY.md section 8d cfg 4), synthetic
              satisfying circuit in the reference's file formats (tools/synth_circuit.py), fixed-tau CRS generated on the device
  per step  : reads <synth>/placementVariables.json (~115 MB of hex), permutation.json, instance.json from files; fresh random
              blinding scalars; writes proof.json
  resident  : what does not change between proofs of one circuit — the CRS in HBM (in the MSM's resident form), the subcircuit
              library as device CSR, the NTT domain — loaded once by tkmk_prover_open, untimed (the contract's "inputs already
              resident in HBM"); DESIGN.md section 5 gives the cold-start figures (process start, CRS load) separately
  N > 1     : every rank proves on its own GPU (independent proofs: weak scaling, no data-path collective; rank 0 stages the
              files, all ranks read them); the collective is only the timing barrier / max.  --one-proof instead has the N GPUs prove ONE
              circuit together (include/tkmk_prover.h tkmk_prover_open_sharded: every commit table and commitment sharded by grid row, one
              RCCL all-gather of 144 bytes per commitment of a round; strong scaling: value = 2^22 / time per proof).  --msm-sharded adds
              BASELINE.json configs[4]'s shape as a secondary: a point-sharded MSM with one RCCL all_gather of the 144-byte partial results.
value = constraint slots proved per second by the whole job = 2^22 * N * steps / max-over-ranks elapsed.

roofline (same JSON line): the dominant kernel of a proof, k_accumulate_chunks (bucket accumulation of every MSM).  Its launch
duration is measured live with HIP events recorded on the launch stream (tkmk_profile_*: recorded without synchronising) in a
SERIALISED PROFILING PASS run right after the timed region: the same proofs with the multi-MSM pipeline narrowed to one internal
stream (tkmk_msm_set_pipeline_streams(1)), so that no two kernels of a commit batch overlap and a section's time is the kernel's own
— what `rocprofv3 --kernel-trace --stats` reports per launch (profiles/r04_prove_configs3_kernel_stats_1stream.csv is the same
command with TKMK_MSM_STREAMS=1).  In the timed region itself three streams time-slice the device: events there bracket queueing
behind the other streams' kernels, so those figures are kept apart and labelled queue_inclusive.  algorithmic bytes = 128 B per
committed point (scalar + base read once) / launch time, against 8 TB/s; `traffic` from the committed
rocprofv3 PMC summary (profiles/traffic.json).  The kernel is integer-VALU bound by construction; valu_roofline gives the
fraction of the measured v_mad_u64_u32 rate.
cpu_baseline: the oracle (a C port — the reference's Rust + ICICLE CPU prover cannot be built offline) on a bounded sample: one
2^20-point MSM and one 1024 x 1024 bivariate NTT on the box's host cores, checked against the GPU, scaled by the proof's counted
MSM points / NTT elements: a LOWER bound on CPU prove time (polynomial bookkeeping and host glue come on top), labelled as such.

Secondary objects (N = 1; each records {"error": ...} instead of costing the line if it fails):
  production_2p20  the reference's production shape (s_max = 256, 166 placements; published walls 45.70 s CPU / 21.08 s CUDA)
  bin_prove_cold   the one-shot binary, process start -> proof.json (device init, CRS load, proof: what prove/src/main.rs:28-84 times
                   and the reference's published walls include), configs[3] from .tkcrs; production shape from .tkcrs and from .rkyv
  msm_2p24         BASELINE.json configs[1]: 2^24-point G1 MSM, BLS12-381 and (as worded) BN254
  msm_2p28_one_gpu BASELINE.json configs[4]'s operands on one GPU: a rank's 2^25-point shard and the whole 2^28-point MSM
  ntt              BASELINE.json configs[2]: 256 x 2^20 scalar-field NTTs, and the production _biNTT
"""
import argparse
import json
import os
import shutil
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MAD_PEAK_PER_S = 3.2e13          # measured v_mad_u64_u32 issue rate, lane-ops/s (profiles/r01_arith_microbench_v3.json)
MADS_PER_BUCKET_ADD = 7 * 392 + 196 + 2 * 301   # madd-2008-s on 14 x 29-bit limbs: 8 products (two share one reduction) + 2 squares (csrc/ffu.h, ec_u.h)
ADDS_PER_POINT = 16              # SURVEY.md §8d cfg 2: N * ceil(b/c) at c = 16, b = 255
ALG_BYTES_PER_POINT = 32 + 96    # each scalar and base read once
ALG_BYTES_PER_NTT_ELEMENT = 64   # read once, written once
SEED = 0x746F6B616D616B00
SECTIONS = ("convert_bases", "digits", "hist", "scan", "scatter", "prepare", "accumulate", "combine", "reduce_segments", "reduce_windows")


class _stdout_to_stderr:
    """RCCL prints a version banner on stdout when a communicator comes up; stdout carries the ONE JSON line and nothing else, so file
    descriptor 1 points at stderr while communicators are made"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def dist_size(comm):
    from tkmk import dist as tkdist
    return int(tkdist.lib().tkmk_comm_size(comm.handle))


def _per_rank(dist, comm_device, torch, values):
    """values: dict of numbers of THIS rank -> list of every rank's dict (rank order); one rank: [values]"""
    if dist is None:
        return [values]
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, values)
    return box


def _read_sections(tkmk, proofs):
    """per-proof section times of the event profiler since the last reset -> ({name: {ms_per_proof, launches_per_proof}}, ntt ms per proof,
    (accumulate ms, launches))"""
    sections = {}
    for name in SECTIONS:
        ms, cnt = tkmk.profile_get("msm." + name)
        if cnt:
            sections[name] = {"ms_per_proof": round(ms / proofs, 3), "launches_per_proof": cnt / proofs}
    ntt_ms = sum(tkmk.profile_get("ntt.pass%d" % k)[0] for k in range(8))
    return sections, ntt_ms / proofs, tkmk.profile_get("msm.accumulate")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--s-max", type=int, default=1024, help="1024 = BASELINE.json configs[3] (2^22 constraint slots); 256 = the reference's production shape")
    ap.add_argument("--placements", type=int, default=None, help="used placements (default: all s_max)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip production_2p20 / msm_2p24 / ntt / msm_2p28_one_gpu")
    ap.add_argument("--one-proof", action="store_true", help="N > 1: the N GPUs prove ONE circuit together (tkmk_prover_open_sharded: commit tables and "
                    "commitments sharded by grid row, one all-gather per round) instead of N independent proofs; value = 2^22 / time per proof, scaling strong")
    ap.add_argument("--msm-sharded", action="store_true", help="N > 1: also time the point-sharded MSM of BASELINE.json configs[4] (2^25 points per rank)")
    ap.add_argument("--msm-sharded-logn", type=int, default=25)
    ap.add_argument("--ntt-sharded", action="store_true", help="also time BASELINE.json configs[2] over the N GPUs: 256 x 2^20 NTTs split by batch index "
                    "(no collective), and ONE 2^25-point bivariate transform through the sharded prover's transform (one all-to-all); N = 1 runs over a "
                    "one-rank RCCL communicator")
    ap.add_argument("--ntt-sharded-logn", type=int, default=20, help="length of one NTT of the batch leg (2^20 = configs[2])")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI; default) or gloo (rehearsal of the N>1 path)")
    ap.add_argument("--share-gpu0", action="store_true", help="rehearsal only: every rank uses GPU 0 (needs --dist-backend gloo)")
    ap.add_argument("--no-serial-pass", action="store_true", help="skip the serialised profiling pass (for a rocprofv3 run whose kernel "
                    "statistics should hold the timed region's launches only); the roofline then carries the queue-inclusive event times")
    ap.add_argument("--workdir", default=None, help="where the staged files go (default: a fresh temporary directory)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)" % (args.gpus, world))

    import torch
    import tkmk
    from tkmk import service
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    dev = 0 if args.share_gpu0 else local_rank
    torch.cuda.set_device(dev)
    tkmk.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        with _stdout_to_stderr():
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
            else:
                dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
            dist.barrier()                                  # the communicator comes up here (and says so on stdout)
    comm_device = "cuda" if args.dist_backend == "nccl" else "cpu"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        tkmk.synchronize()

    # --- stage the files every rank reads (untimed): rank 0 generates the circuit, its witness and the CRS ---
    import prove_bench
    files = None
    if rank == 0:
        files = prove_bench.stage_files(s_max=args.s_max, placements=args.placements, tmp=args.workdir)
    if dist is not None:
        box = [files]
        dist.broadcast_object_list(box, src=0)
        files = box[0]
    slots = files["constraint_slots"]
    try:
        t = time.perf_counter()
        one_proof = args.one_proof          # N = 1: a one-rank RCCL communicator — the same code path, for rehearsing it on a one-GPU box
        if one_proof and comm_device != "cuda":
            raise SystemExit("--one-proof runs over libtkmk_dist.so (RCCL): needs --dist-backend nccl")
        shard_comm = None
        comm_report = None
        if (one_proof or args.msm_sharded or args.ntt_sharded) and comm_device == "cuda":
            # the communicator of libtkmk_dist.so (RCCL) for every sharded leg of this run, and ITS OWN account of who takes part: each
            # rank's line (size / rank as the communicator holds them, ncclCommCount, RCCL version, device PCI bus id and UUID) gathered
            # over the communicator itself — N ranks on N distinct devices, or the line says otherwise
            from tkmk import dist as tkdist
            with _stdout_to_stderr():
                shard_comm = tkdist.comm_from_torch(dist) if dist is not None else tkdist.Comm(tkdist.unique_id(), 1, 0)
            lines = shard_comm.describe_all()
            comm_report = {"size_reported_by_communicator": dist_size(shard_comm), "ranks": lines,
                           "distinct_devices": len({(l["pci_bus_id"], l["uuid"]) for l in lines}),
                           "gathered_with": "tkmk_comm_all_gather_host over this communicator"}
        prover = service.Prover(files["qap"], files["crs"], comm=shard_comm if one_proof else None)      # circuit-static state -> HBM, once (sharded: 1/N of the tables)
        open_s = time.perf_counter() - t
        jobs_in_flight = 1 if one_proof else world                                # proofs a step completes
        out_dir = os.path.join(files["tmp"], "out_rank%d" % rank)

        def step():
            return prover.prove(files["synth"], out_dir, want_json=False)[1]

        for _ in range(args.warmup):
            step()
        tkmk.profile_enable(True)
        tkmk.profile_reset()
        tkmk.native_stats_reset()
        barrier()
        t0 = time.perf_counter()
        timings = [step() for _ in range(args.steps)]
        barrier()
        elapsed = time.perf_counter() - t0
        tkmk.profile_enable(False)
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        stats = tkmk.native_stats()
        q_sections, q_ntt_ms, (q_acc_ms, q_acc_cnt) = _read_sections(tkmk, args.steps)      # queue-inclusive: three streams in flight
        # ---- serialised profiling pass (untimed for the headline): the same proofs, one internal stream, every kernel alone ----
        streams_default = tkmk.msm_get_pipeline_streams()
        serial = not args.no_serial_pass and streams_default > 1
        if serial:
            ser_proofs = max(1, min(args.steps, 3))
            tkmk.msm_set_pipeline_streams(1)
            step()
            tkmk.profile_enable(True)
            tkmk.profile_reset()
            tkmk.native_stats_reset()
            barrier()
            t1 = time.perf_counter()
            for _ in range(ser_proofs):
                step()
            barrier()
            ser_elapsed = time.perf_counter() - t1
            tkmk.profile_enable(False)
            ser_stats = tkmk.native_stats()
            sections, ntt_ms, (acc_ms, acc_cnt) = _read_sections(tkmk, ser_proofs)
            tkmk.msm_set_pipeline_streams(0)
        else:       # one stream throughout (TKMK_MSM_STREAMS=1): the timed region is already serialised; or the pass was declined
            ser_proofs, ser_elapsed, ser_stats = args.steps, elapsed, stats
            sections, ntt_ms, (acc_ms, acc_cnt) = q_sections, q_ntt_ms, (q_acc_ms, q_acc_cnt)
        measured_in = ("serialised profiling pass (1 internal stream): launch duration of the kernel running alone" if serial else
                       "timed region, 1 internal stream: launch duration of the kernel running alone" if streams_default == 1 else
                       "timed region, %d internal streams: QUEUE-INCLUSIVE event times, not kernel durations" % streams_default)
        prover.close()
        # what each rank did in the timed region, by its own counters (one process per GPU: the library's counters are the rank's)
        rank_work = _per_rank(dist, comm_device, torch, {"rank": rank, "msm_points": stats["msm.points"] / args.steps, "msm_calls": stats["msm.calls"] / args.steps,
                                                         "msm_bucket_additions": stats.get("msm.bucket_additions", 0) / args.steps,
                                                         "ntt_elements": stats["ntt.elements"] / args.steps, "ntt_calls": stats["ntt.calls"] / args.steps,
                                                         "per_proof_s": {k: round(statistics.median(t_[k] for t_ in timings), 5) for k in timings[0]}})

        msm_sharded = ntt_sharded = None
        if args.msm_sharded and (world > 1 or shard_comm is not None):         # every rank takes part
            msm_sharded = _msm_sharded_leg(tkmk, dist, comm_device, rank, world, args.msm_sharded_logn, barrier, torch, shard_comm)
        if args.ntt_sharded:
            ntt_sharded = _ntt_sharded_leg(tkmk, dist, comm_device, rank, world, args.ntt_sharded_logn, barrier, torch, shard_comm)
        if shard_comm is not None:
            shard_comm.close()

        if rank == 0:
            med = {k: statistics.median(t_[k] for t_ in timings) for k in timings[0]}
            points_per_proof = stats["msm.points"] / args.steps
            out = {
                "metric": "constraints/sec (prove step)",
                "value": slots * jobs_in_flight * args.steps / elapsed,
                "unit": "constraints/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True,
                "scaling": "strong" if one_proof else "weak",
                "vs_baseline": None,
                "dtype": "u32 limbs (255-bit Fr / 381-bit Fq modular integer arithmetic)",
                "data": "synthetic",
                "config": {"workload": "full prove (Prover::init + prove0..prove4 + proof.json) per GPU, BASELINE.json configs[3]: " + files["workload"] +
                                       "; synthesizer documents read from files every step; resident in HBM (untimed, open_context_s): the CRS with its commit "
                                       "table, the Lagrange-basis tables derived from it, the subcircuit library; the proof bytes are those of the reference's "
                                       "algorithm (DESIGN.md section 4 lists the algebraically equal forms used)",
                           "constraint_slots_per_proof": slots, "r1cs_rows_per_proof": files["r1cs_rows"], "proofs_per_step": jobs_in_flight,
                           "host_side": "native C++ (libtkmk_prover.so over the C ABI of libtkmk_hip.so)",
                           "sharding": ("ONE proof over %d GPUs (tkmk_prover_open_sharded): commit tables, transforms, streaming passes and the witness "
                                        "side divided by columns r mod N; one all-to-all per transform, one all-gather per commit batch; transcript replicated" % world) if one_proof
                           else "independent proofs per GPU" if world > 1 else "none"},
                "r1cs_rows_per_s": files["r1cs_rows"] * jobs_in_flight * args.steps / elapsed,
                "per_proof_s": {k: round(v, 5) for k, v in med.items()},
                "init_fraction": round(med["init_s"] / med["total_s"], 3),
                "open_context_s": round(open_s, 3),
                "staged_files": {k: files[k] for k in ("crs_payload_bytes", "placement_variables_json_bytes", "permutation_json_bytes", "generate_s",
                                                       "sigma_gen_s", "crs_write_s")},
                "work_per_proof": {"msm_points": points_per_proof, "msm_calls": stats["msm.calls"] / args.steps,
                                   "ntt_elements": stats["ntt.elements"] / args.steps, "ntt_calls": stats["ntt.calls"] / args.steps,
                                   "algorithmic_bytes": ALG_BYTES_PER_POINT * points_per_proof + ALG_BYTES_PER_NTT_ELEMENT * stats["ntt.elements"] / args.steps},
                # kernel times of one proof, each kernel running alone (serialised pass: one internal stream); their sum is below that
                # pass's own step time (the rest is polynomial passes that are not bracketed, host phases and launch gaps)
                "kernel_ms_per_proof": dict(sections, ntt_passes=round(ntt_ms, 3)),
                "serialised_pass": {"ran": serial, "pipeline_streams": 1 if serial else streams_default, "proofs": ser_proofs, "ms_per_step": ser_elapsed / ser_proofs * 1e3,
                                    "bracketed_kernels_ms_per_proof": round(sum(v["ms_per_proof"] for v in sections.values()) + ntt_ms, 3),
                                    "note": "untimed for the headline; same proofs with tkmk_msm_set_pipeline_streams(1): section times are "
                                            "kernel durations as rocprofv3 reports them (profiles/r04_prove_configs3_kernel_stats_1stream.csv)"},
                "kernel_ms_per_proof_queue_inclusive": dict(q_sections, ntt_passes=round(q_ntt_ms, 3), pipeline_streams=streams_default,
                                                            note="HIP events in the timed region: with several streams in flight an event pair "
                                                                 "also brackets the wait behind the other streams' kernels; NOT kernel time"),
            }
            if acc_cnt:
                avg_ms = acc_ms / acc_cnt
                alg = ALG_BYTES_PER_POINT * ser_stats["msm.points"] / acc_cnt      # per launch: one launch per MSM
                achieved = alg / (avg_ms * 1e-3) / 1e9
                # bucket additions the accumulate launches actually ran: the library counts the sorted-list lengths on the device (zero
                # digits of sparse / small scalars drop out, a table commit has 13 windows, a plain one 16)
                adds = ser_stats.get("msm.bucket_additions", 0)
                mads = adds * MADS_PER_BUCKET_ADD / (acc_ms * 1e-3)
                out["roofline"] = {"bound": "hbm", "kernel": "k_accumulate_chunks", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": achieved / HBM_PEAK_GBS, "traffic": _traffic_from_profiles("prove_accumulate_configs3_r04" if args.s_max == 1024 else None),
                                   "avg_launch_ms": avg_ms, "launches": acc_cnt, "launches_per_proof": acc_cnt / ser_proofs,
                                   "algorithmic_bytes_per_launch": alg, "share_of_step": (acc_ms / ser_proofs) / (elapsed / args.steps * 1e3),
                                   "measured_in": measured_in,
                                   "queue_inclusive_avg_launch_ms": (q_acc_ms / q_acc_cnt) if q_acc_cnt else None,
                                   "note": "integer-VALU bound by construction (SURVEY.md §8d: ~375 int mul-adds per algorithmic byte)"}
                out["valu_roofline"] = {"kernel": "k_accumulate_chunks", "unit": "v_mad_u64_u32 lane-ops/s", "achieved": mads, "peak": MAD_PEAK_PER_S,
                                        "frac": mads / MAD_PEAK_PER_S, "bucket_additions_per_proof": adds / ser_proofs,
                                        "note": "additions counted by the library (sorted-list lengths) x 3542 multiply-adds each, over the launch "
                                                "durations of the serialised pass (each launch alone on the device)"}
            out["hbm_roofline_whole_step"] = {"achieved_GBps": out["work_per_proof"]["algorithmic_bytes"] / (elapsed / args.steps) / 1e9, "peak_GBps": HBM_PEAK_GBS}
            if msm_sharded is not None:
                out["msm_sharded"] = msm_sharded
            if ntt_sharded is not None:
                out["ntt_sharded"] = ntt_sharded
            out["comm"] = comm_report if comm_report is not None else {"note": "no libtkmk_dist.so communicator in this run (independent proofs per GPU: "
                                                                                "torch.distributed carries the timing barrier only)"}
            out["per_rank"] = rank_work

            def leg(key, fn):
                """a secondary figure must never cost the headline line: its failure is recorded under its key instead"""
                try:
                    out[key] = fn()
                except Exception as e:      # noqa: BLE001
                    out[key] = {"error": "%s: %s" % (type(e).__name__, e)}

            tkmk.release_scratch()
            if world == 1 and not args.no_cpu_baseline:
                leg("cpu_baseline", lambda: _cpu_baseline(tkmk, out["work_per_proof"], slots, elapsed / args.steps))
            if world == 1 and not args.no_secondary:
                leg("bin_prove_cold", lambda: _cold_leg(prove_bench, files, args.s_max))
                leg("production_2p20", lambda: _production_leg(prove_bench))
                tkmk.release_scratch()
                leg("msm_2p24", lambda: _msm_leg(tkmk))
                tkmk.release_scratch()
                leg("ntt", lambda: _ntt_secondary(tkmk))
                tkmk.release_scratch()
                leg("msm_2p28_one_gpu", lambda: _msm_2p28_leg(tkmk))
            print(json.dumps(out), flush=True)
    finally:
        if dist is not None:
            dist.barrier()
        if rank == 0 and args.workdir is None:
            shutil.rmtree(files["tmp"], ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


def _traffic_from_profiles(key):
    """HBM bytes per k_accumulate_chunks launch from the committed rocprofv3 PMC summary (profiles/traffic.json), or None"""
    if key is None:
        return None
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def _generator():
    # standard G1 generator, u32 LE limbs (reference pin: setup/mpc-setup/src/conversions.rs:68-79)
    x = [0xdb22c6bb, 0xfb3af00a, 0xf97a1aef, 0x6c55e83f, 0x171bac58, 0xa14e3a3f, 0x9774b905, 0xc3688c4f, 0x4fa9ac0f,
         0x2695638c, 0x3197d794, 0x17f1d3a7]
    y = [1187375073, 212476713, 2726857444, 3493644100, 738505709, 14358731, 3587181302, 4243972245, 1948093156,
         2694721773, 3819610353, 146011265]
    return b"".join(int(v).to_bytes(4, "little") for v in x + y)


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


def _cpu_baseline(tkmk, work, slots, gpu_step_s):
    """The oracle (a C port; the reference's Rust + ICICLE CPU prover cannot be built offline — its own published run is 45.7 s for a
    2^20-slot proof on ~8 threads of other hardware) on a bounded sample of the step's work, on this box's host cores: one 2^20-point
    MSM and one 1024 x 1024 bivariate NTT from the benchmark's seeded streams, each checked against the GPU, then the proof's counted
    MSM points and NTT elements divided by those rates.  A LOWER bound on a CPU prover's time for the step (polynomial bookkeeping,
    divisions and host glue come on top), i.e. an upper bound on its constraints/s."""
    import oracle
    threads = _usable_cpus(oracle.num_threads())
    m = 1 << 20
    s = tkmk.fr_random_device(SEED + 2, m)
    hh = tkmk.fr_random_device(SEED + 3, m)
    g = np.frombuffer(bytes(_generator()), np.uint8).copy()
    b = tkmk.g1_batch_scalar_mul_device(hh, g, m)
    sh, bh = s.to_host(), b.to_host()
    t0 = time.perf_counter()
    want = oracle.g1_msm(sh, bh, threads=threads)
    msm_dt = time.perf_counter() - t0
    msm_ok = bool((tkmk.projective_to_affine_bytes(tkmk.msm(s, b)) == want).all())
    xs = ys = 1024
    tkmk.init_ntt_domain_for_size(xs * ys)
    a = tkmk.fr_random_device(SEED + 9, xs * ys)
    ah = a.to_host()
    oracle.bintt(ah[:32 * 64 * 64], 64, 64)
    t0 = time.perf_counter()
    want = oracle.bintt(ah, xs, ys)
    ntt_dt = time.perf_counter() - t0
    ntt_ok = bool((np.asarray(tkmk.bintt(a, xs, ys).to_host()) == np.asarray(want)).all())
    msm_rate, ntt_rate = m / msm_dt, xs * ys / ntt_dt
    msm_s, ntt_s = work["msm_points"] / msm_rate, work["ntt_elements"] / ntt_rate
    return {"value": slots / (msm_s + ntt_s), "unit": "constraints/s", "cores": threads, "kind": "port",
            "sample": "one 2^20-point MSM (%.2f s) + one 1024 x 1024 bivariate NTT (%.3f s) of the seeded streams on %d host threads; the step's "
                      "counted MSM points and NTT elements at those rates (MSM + NTT share only: a lower bound on CPU time)" % (msm_dt, ntt_dt, threads),
            "msm_points_per_s": msm_rate, "ntt_elements_per_s": ntt_rate, "matches_gpu": msm_ok and ntt_ok,
            "estimated_step_s": msm_s + ntt_s, "gpu_step_s": gpu_step_s,
            "reference_published_wall_s": {"cpu": 45.70, "cuda": 21.08, "note": "2^20-slot production shape, 166 placements, other hardware (BASELINE.md §1)"}}


def _production_leg(prove_bench):
    """the reference's production shape (n = 4096, m_I = 4096, s_max = 256, 166 placements: the run behind the published 45.70 s CPU /
    21.08 s CUDA walls) through the same resident prover"""
    r = prove_bench.run_service(s_max=256, placements=166, repeat=5, warmup=1)
    return {"workload": r["workload"], "constraints_per_s": r["constraint_slots_per_s"], "r1cs_rows_per_s": r["r1cs_rows_per_s"],
            "per_proof_s": r["median"], "init_fraction": r["init_fraction"], "open_context_s": r["open_context_s"],
            "reference_wall_s": {"cpu": 45.70, "cuda": 21.08, "note": "other hardware (BASELINE.md §1)"}}


def _cold_runs(files, crs_dir, reps=2):
    """bin/prove as tokamak-cli would spawn it (plus the library flag: the files live in a scratch directory), wall from fork to exit"""
    import re
    import subprocess
    out_dir = os.path.join(files["tmp"], "out_cold")
    os.makedirs(out_dir, exist_ok=True)
    cmd = [os.path.join(ROOT, "tokamak-zk-evm_amd", "bin", "prove"), "--crs", crs_dir, "--synthesizer-stat", files["synth"], "--output", out_dir,
           "--subcircuit-library", files["qap"]]
    walls, own = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        walls.append(time.perf_counter() - t0)
        if r.returncode != 0 or not os.path.exists(os.path.join(out_dir, "proof.json")):
            raise RuntimeError("bin/prove failed: " + r.stderr[-400:])
        own = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"^([a-z0-9. ]+?)\s+([0-9.]+) s", r.stdout, re.M)}
    return {"wall_s": round(min(walls), 3), "walls_s": [round(w, 3) for w in walls], "binary_breakdown_s": own}


def _cold_leg(prove_bench, files, s_max):
    """The like-for-like of the reference's published walls (prove/src/main.rs:28-84: total_start before check_device and the CRS
    load): ONE process per proof — device initialisation, subcircuit library + CRS from files into HBM (no commit table, no Lagrange
    tables: they pay only for a resident prover), the proof, proof.json."""
    import tkmk
    from tkmk import crs as crsmod
    from tkmk import rkyv
    res = {"what": "bin/prove, process start -> exit (fork + device init + CRS load + Prover::init + prove0..4 + proof.json); best of 2 (page cache warm)"}
    res["configs3_2p22" if s_max == 1024 else "headline_shape"] = dict(_cold_runs(files, files["crs"]), crs="combined_sigma.tkcrs (%.2f GB)" % (files["crs_payload_bytes"] / 1e9))
    prod = prove_bench.stage_files(s_max=256, placements=166)
    try:
        tkmk.release_scratch()
        res["production_2p20_tkcrs"] = dict(_cold_runs(prod, prod["crs"]), crs="combined_sigma.tkcrs (%.2f GB)" % (prod["crs_payload_bytes"] / 1e9))
        # the reference's own container for the same reference string (32-bit relative pointers: only below 2 GiB, i.e. not configs[3])
        arch = os.path.join(prod["tmp"], "crs_rkyv")
        os.makedirs(arch)
        sections = crsmod.read_payload(os.path.join(prod["crs"], "combined_sigma.tkcrs"))
        blob = rkyv.encode_combined_sigma(sections, rkyv.rows_for(prod["setup_params"]), "rustc_size_groups")
        open(os.path.join(arch, "combined_sigma.rkyv"), "wb").write(blob)
        del sections
        res["production_2p20_rkyv"] = dict(_cold_runs(prod, arch), crs="combined_sigma.rkyv (%.2f GB)" % (len(blob) / 1e9))
        del blob
        res["reference_wall_s"] = {"cpu": 45.70, "cuda": 21.08, "note": "production shape, other hardware (BASELINE.md §1)"}
    finally:
        shutil.rmtree(prod["tmp"], ignore_errors=True)
    return res


def _msm_leg(tkmk):
    """BASELINE.json configs[1]: one 2^24-point G1 Pippenger MSM with scalars and bases resident in HBM, on the reference's curve
    (BLS12-381) and on the curve the config names (BN254; no reference counterpart)"""
    res = {}
    n = 1 << 24
    for curve, gen in (("bls12_381", np.frombuffer(bytes(_generator()), np.uint8).copy()), ("bn254", None)):
        if gen is None:
            gen = np.zeros(64, np.uint8)
            gen[0], gen[32] = 1, 2
        s = tkmk.fr_random_device(SEED + 2, n, curve=curve)
        h = tkmk.fr_random_device(SEED + 3, n, curve=curve)
        b = tkmk.g1_batch_scalar_mul_device(h, gen, n, curve=curve)
        h.free()
        tkmk.msm(s, b, curve=curve)
        tkmk.profile_enable(True)
        tkmk.profile_reset()
        tkmk.synchronize()
        steps = 5
        t0 = time.perf_counter()
        for _ in range(steps):
            tkmk.msm(s, b, curve=curve)
        tkmk.synchronize()
        dt = (time.perf_counter() - t0) / steps
        tkmk.profile_enable(False)
        acc_ms, acc_cnt = tkmk.profile_get("msm.accumulate")
        pre = None
        if curve == "bls12_381":
            # the same MSM over a precomputed table (ICICLE's precompute_factor, which the reference leaves at 1): 13 levels of
            # 2^(20 j) multiples, one bucket set, 20-bit windows — what the resident prover commits with
            t0 = time.perf_counter()
            table = tkmk.msm_precompute_bases(b, n, 13, c=20)
            tkmk.synchronize()
            build_s = time.perf_counter() - t0
            first = tkmk.msm(s, table, msm_size=n, c=20, precompute_factor=13)
            same = bytes(first) == bytes(tkmk.msm(s, b))
            tkmk.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                tkmk.msm(s, table, msm_size=n, c=20, precompute_factor=13)
            tkmk.synchronize()
            pdt = (time.perf_counter() - t0) / steps
            table.free()
            pre = {"ms_per_msm": pdt * 1e3, "points_per_s": n / pdt, "table_build_s": build_s, "table_bytes": 96 * n * 13, "equals_plain_result": same}
        s.free()
        b.free()
        bytes_pt = 32 + (96 if curve == "bls12_381" else 64)
        res[curve] = {"workload": "2^24-point %s G1 Pippenger MSM, inputs resident in HBM" % curve, "ms_per_msm": dt * 1e3, "points_per_s": n / dt,
                      "group_adds_per_s": n / dt * ADDS_PER_POINT, "accumulate_kernel_ms": acc_ms / acc_cnt if acc_cnt else None,
                      "hbm_frac_of_peak": (n * bytes_pt / (acc_ms / acc_cnt * 1e-3) / 1e9 / HBM_PEAK_GBS) if acc_cnt else None}
        if pre:
            res[curve]["with_precomputed_table_c20"] = pre
        tkmk.release_scratch()
    return res


def _msm_2p28_leg(tkmk):
    """BASELINE.json configs[4]'s operands (2^28 points) on ONE GPU: a rank's 2^25-point shard, and the whole MSM in one call (34 GB of
    operands resident in HBM).  The 8-GPU job itself is `bench.py --gpus 8 --msm-sharded` (one all-gather of 144-byte partials)."""
    import ctypes
    n_shard, shards = 1 << 25, 8
    n = n_shard * shards
    gen = np.frombuffer(bytes(_generator()), np.uint8).copy()
    s = tkmk.fr_random_device(SEED + 7, n)
    b = tkmk.DeviceBuffer(96 * n)
    for j in range(shards):
        h = tkmk.fr_random_device(SEED + 8, n_shard, first=j * n_shard)
        pj = tkmk.g1_batch_scalar_mul_device(h, gen, n_shard)
        tkmk._check(tkmk.lib().tkmk_memcpy_d2d(ctypes.c_void_p(b.ptr + 96 * n_shard * j), ctypes.c_void_p(pj.ptr), ctypes.c_size_t(96 * n_shard)), "tkmk_memcpy_d2d")
        h.free()
        pj.free()
    res = {}
    s0, b0 = tkmk.DeviceBuffer(32 * n_shard), tkmk.DeviceBuffer(96 * n_shard)
    tkmk._check(tkmk.lib().tkmk_memcpy_d2d(ctypes.c_void_p(s0.ptr), ctypes.c_void_p(s.ptr), ctypes.c_size_t(32 * n_shard)), "tkmk_memcpy_d2d")
    tkmk._check(tkmk.lib().tkmk_memcpy_d2d(ctypes.c_void_p(b0.ptr), ctypes.c_void_p(b.ptr), ctypes.c_size_t(96 * n_shard)), "tkmk_memcpy_d2d")
    for key, sc, ba, pts, reps in (("shard_2p25", s0, b0, n_shard, 3), ("whole_2p28", s, b, n, 2)):
        tkmk.msm(sc, ba)
        tkmk.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            tkmk.msm(sc, ba)
        tkmk.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res[key] = {"ms_per_msm": dt * 1e3, "points_per_s": pts / dt}
    for x in (s, b, s0, b0):
        x.free()
    tkmk.release_scratch()
    res["workload"] = "BLS12-381 G1 MSM, operands resident in HBM; correctness of both shapes: tests/test_gpu_baseline_sizes.py"
    return res


def _ntt_secondary(tkmk):
    """BASELINE.json configs[2] (256 independent 2^20-point NTTs, resident) and the reference's largest production _biNTT
    (16384 x 512).  Algorithmic bytes = 64 B per element."""
    res = {}
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
                     "bound": "integer VALU (Fr products): see DESIGN.md section 4"}
        a.free()
        o.free()
    return res


def _ntt_sharded_leg(tkmk, dist, comm_device, rank, world, logn, barrier, torch, comm):
    """BASELINE.json configs[2] over the N GPUs of the job.
    batch: 256 independent length-2^logn forward NTTs, natural order, device resident, split by BATCH INDEX — rank r transforms the vectors
           b = r mod N (independent units, no collective); value = elements of the whole batch / max-over-ranks time.
    bivariate: ONE 2^25-point transform (16384 x 2048: the domain of prove2's p_comb) through the sharded prover's transform
           (tkmk_dist_fwd_cols_to_rows: X pass on the rank's columns, ONE all-to-all, Y pass on its rows) and back (tkmk_dist_inv_rows_to_cols);
           needs the libtkmk_dist.so communicator (RCCL), N = 1: a one-rank communicator."""
    out = {}
    n, batch = 1 << logn, 256
    mine = len(range(rank, batch, world))
    tkmk.init_ntt_domain_for_size(max(n, 1 << 25))
    data = tkmk.fr_random_device(SEED + 7 + 16 * rank, n * mine)
    res = tkmk.DeviceBuffer(32 * n * mine)
    tkmk.ntt(data, n, batch=mine, out=res)
    tkmk.native_stats_reset()
    barrier()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        tkmk.ntt(data, n, batch=mine, out=res)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=comm_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    dt /= steps
    st = tkmk.native_stats()
    data.free()
    res.free()
    out["batch"] = {"workload": "256 x 2^%d forward NTTs split by batch index over %d rank(s), no collective" % (logn, world),
                    "ms": dt * 1e3, "elements_per_s": batch * n / dt, "vectors_on_this_rank": mine,
                    "hbm": {"algorithmic_bytes": ALG_BYTES_PER_NTT_ELEMENT * batch * n, "achieved_GBps_whole_job": ALG_BYTES_PER_NTT_ELEMENT * batch * n / dt / 1e9,
                            "peak_GBps_per_gpu": HBM_PEAK_GBS, "frac_of_job_peak": ALG_BYTES_PER_NTT_ELEMENT * batch * n / dt / 1e9 / (HBM_PEAK_GBS * world)},
                    "per_rank_ntt_elements": [w["ntt_elements"] for w in _per_rank(dist, comm_device, torch, {"ntt_elements": st["ntt.elements"] / steps})]}
    if comm is None:
        out["bivariate"] = {"skipped": "needs the libtkmk_dist.so communicator (--dist-backend nccl)"}
        return out
    xs, ys = 16384, 2048
    lc, h = ys // world, xs // world
    coeff = tkmk.fr_random_device(SEED + 9 + 16 * rank, xs * lc)
    ev = comm.fwd_cols_to_rows(coeff, xs, ys, xs, ys)
    back = comm.inv_rows_to_cols(ev, xs, ys)
    same = bool((np.asarray(back.to_host()[:32 * 64]) == np.asarray(coeff.to_host()[:32 * 64])).all())     # round trip, first elements
    ev.free()
    back.free()
    tkmk.native_stats_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        ev = comm.fwd_cols_to_rows(coeff, xs, ys, xs, ys)
        tkmk.synchronize()
        ev.free()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=comm_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    dt /= steps
    st = tkmk.native_stats()
    coeff.free()
    out["bivariate"] = {"workload": "one %d x %d forward bivariate NTT over %d rank(s): columns r mod N -> X pass -> one all-to-all -> Y pass -> row slabs" % (xs, ys, world),
                        "entry": "tkmk_dist_fwd_cols_to_rows (C ABI, RCCL on device buffers)", "ms": dt * 1e3, "elements_per_s": xs * ys / dt,
                        "all_to_all_bytes_sent_per_rank": 32 * h * lc * (world - 1), "round_trip_equal": same,
                        "per_rank_ntt_elements": [w["ntt_elements"] for w in _per_rank(dist, comm_device, torch, {"ntt_elements": st["ntt.elements"] / steps})]}
    return out


def _msm_sharded_leg(tkmk, dist, comm_device, rank, world, logn, barrier, torch, comm=None):
    """BASELINE.json configs[4]'s shape: every rank holds 2^logn points of one MSM (generated in HBM from the seed), runs the full
    single-GPU pipeline on its shard, the 144-byte partial results meet in ONE all_gather (RCCL over xGMI) and every rank adds them"""
    from tkmk import dist as tkdist
    from tkmk import sharding
    n = 1 << logn
    scalars = tkmk.fr_random_device(SEED + 2 + 16 * rank, n)
    h = tkmk.fr_random_device(SEED + 3 + 16 * rank, n)
    g = np.frombuffer(bytes(_generator()), np.uint8).copy()
    bases = tkmk.g1_batch_scalar_mul_device(h, g, n)
    h.free()
    # the C-ABI entry (libtkmk_dist.so: RCCL all_gather on device buffers) when the ranks talk over RCCL; the torch helper for the
    # gloo rehearsal of the same partitioning
    run = (lambda: comm.msm_sharded(scalars, bases)) if comm is not None else (lambda: sharding.msm_sharded(tkmk, dist, scalars, bases, device=comm_device))
    run()
    tkmk.native_stats_reset()
    barrier()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        res = run()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=comm_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    dt /= steps
    st = tkmk.native_stats()
    scalars.free()
    bases.free()
    return {"per_rank": _per_rank(dist, comm_device, torch, {"rank": rank, "msm_points": st["msm.points"] / steps, "msm_bucket_additions": st.get("msm.bucket_additions", 0) / steps}),"workload": "2^%d-point BLS12-381 G1 MSM, 2^%d points per rank, one all_gather of 144-byte partial results" % (logn + (world - 1).bit_length(), logn),
            "entry": "tkmk_msm_sharded (C ABI, RCCL on device buffers)" if comm_device == "cuda" else "tkmk/sharding.py over torch.distributed (%s)" % comm_device,
            "ms_per_msm": dt * 1e3, "points_per_s": n * world / dt, "result_x_lo": int.from_bytes(bytes(res[:8]), "little")}


if __name__ == "__main__":
    main()
