/* tkmk_prover.h — C ABI of the resident prover (libtkmk_prover.so): the process/CLI boundary of `prove` and `preprocess`
 *  as library calls, for a host that produces many proofs of one circuit.
 *
 * Replaces, per call:
 *   tkmk_prover_open      the circuit-static part of Prover::init  (packages/backend/prove/src/lib.rs:675-835: setupParams.json,
 *                         subcircuitInfo.json, r1cs/subcircuit{id}.r1cs) + SigmaHolder::load (prove/src/sigma_source.rs:22-47:
 *                         <crs>/combined_sigma.rkyv; the flat <crs>/combined_sigma.tkcrs payload is accepted as a fast path)
 *                         + check_device (libs/src/utils/mod.rs:78-110)
 *   tkmk_prover_prove     main() of prove/src/main.rs:41-84: Prover::init on <synthesizer>/{placementVariables,permutation,
 *                         instance}.json, prove0..prove4 with the Fiat-Shamir transcript, <output>/proof.json in the
 *                         Solidity-verifier format (prove/src/lib.rs:452-513)
 *   tkmk_prover_close     process exit
 * A reference maintainer binds these three from Rust and keeps the `prove` argument surface.
 * One context per process and GPU; calls on one context must not overlap.  There is no CPU fallback: without a gfx950
 * device tkmk_prover_open returns TKMK_ERR_NO_DEVICE.  Errors: the tkmk_error code, text from tkmk_prover_last_error()
 * (the reference panics with the same messages). */
#ifndef TKMK_PROVER_H
#define TKMK_PROVER_H
#include "tkmk.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tkmk_prover tkmk_prover;

typedef struct {          /* seconds */
    double parse_s;       /* the three synthesizer documents -> host arrays (all cores) */
    double upload_s;      /* witness + index arrays -> HBM */
    double build_s;       /* u, v, w, b, s0, s1, a_free on the device */
    double binding_s;     /* A_free, O_pub_free, O_mid, O_prv */
    double init_s;        /* Prover::init = the four above */
    double prove_s[5];    /* prove0 .. prove4 */
    double write_s;       /* proof.json */
    double total_s;
} tkmk_prove_timing;

/* Loads the circuit-static state.  Because the context is meant to stay, xy_powers is expanded once into a precomputed commit
 * table (ICICLE's msm_precompute_bases; 20-bit windows, 13 x the table in HBM: 21 GB for the 2^24-point CRS of BASELINE.json
 * configs[3]) so that every large commit costs 13 instead of 16 bucket additions per point; commitments are bit-identical either
 * way.  Environment: TKMK_PROVER_TABLE_C=0 disables the table, 13..20 picks another window width.
 * Resident cost per circuit, in full (BASELINE.json configs[3], n = m_I = 4096, s_max = 1024; the production shape s_max = 256 is a
 * quarter of each): the commit table 21 GB; the Lagrange-basis twin of the n x s_max grid in the same 13-level form 5.2 GB; its prefix
 * sums in prove1's walk order 5.2 GB; the binding tables 4.2 GB; the subcircuit library and the NTT domain < 1 GB — 36 GB of the 288 —
 * and 7.6 s at open (2.7 s table expansion, 3.3 s group NTT behind the Lagrange table, 1.3 s prefix sums and their expansion; production
 * shape 1.9 s).  TKMK_PROVER_LAGRANGE=0 drops the second and third items (U, V, W, B, R are then committed from coefficients).  A
 * sharded context (tkmk_prover_open_sharded) keeps 1 / G of the first three per GPU and does 1 / G of their one-time work. */
tkmk_error tkmk_prover_open(const char *subcircuit_library_dir, const char *crs_dir, tkmk_prover **out);
/* ONE proof over the G GPUs of a node (the reference is single-device, so nothing is replaced).  G is a power of
 * two, at most min(n, m_I, s_max).  Rank r of the communicator `comm` (a tkmk_comm of include/tkmk_dist.h, made by the host: one process
 * per GPU over RCCL) holds the COLUMNS iy = r mod G of everything: of every commit table — xy_powers, the Lagrange-basis tables and their
 * prefix sums: resident HBM and the work at open (table expansion AND the group transforms behind the Lagrange tables) divide by G — and
 * of every coefficient matrix of a proof, so that the divisions, window sums and shifts along X, both passes of the vanishing division
 * and the commits are local to a rank; evaluations on the large domains live in row slabs, and a bivariate transform crosses between
 * the two layouts with ONE all-to-all.  The witness side divides by placement (a placement is a column of u, v, w, b): a rank converts,
 * uploads, evaluates and routes its own placements only.  Replicated: the Fiat-Shamir transcript (every rank derives the same challenges
 * from the same gathered commitments) and the small polynomials built from host values.  Exchanges per proof: one all-gather per commit
 * batch (144 B per commitment + a status block), one all-to-all per transform, a ring shift per Y-shifted term, all-gathers of a few
 * values (evaluation partials, remainder rows), one broadcast of rank 0's blinding scalars; DESIGN.md section 6 counts them.
 * Every rank must call _open_sharded and then every _prove / _prove_ex with the same arguments; every rank gets the same proof, byte for
 * byte the one the single-GPU context gives for the same blinding scalars (tests/test_gpu_sharded_prover.py, G = 2, 4, 8 over the
 * loopback transport); RANK 0 ALONE writes <output_dir>/proof.json (temporary file + rename), after all ranks agreed that they
 * finished.  An input error only one rank can see is agreed on and reported by all; a rank that fails for a reason its peers cannot share
 * (memory, device, transport) aborts the communicator (tkmk_comm_abort) so that they fail instead of waiting — the contexts of an aborted
 * communicator can only be closed.  libtkmk_dist.so must be in the process (it is: the host made `comm` with it). */
tkmk_error tkmk_prover_open_sharded(void *comm, const char *subcircuit_library_dir, const char *crs_dir, tkmk_prover **out);
int tkmk_prover_world_size(const tkmk_prover *p);   /* 1 for a context made by tkmk_prover_open */
/* output_dir may be NULL (no file is written); proof_json_out (optional) receives a malloc'ed copy of the document, to be
 * released with tkmk_prover_free_string.  testing_mixer_json: NULL in production (blinding scalars from getrandom());
 * a path to a JSON document with fixed blinding scalars makes the proof deterministic — for differential tests only,
 * a proof made with known blinding scalars is not zero-knowledge. */
tkmk_error tkmk_prover_prove(tkmk_prover *p, const char *synthesizer_dir, const char *output_dir, const char *testing_mixer_json,
                             tkmk_prove_timing *timing, char **proof_json_out);
/* The same with two switches for parity work and a record of what every commitment ran over.
 * flags: TKMK_PROVE_TEST_PARTS        prove4 commits Pi_AX, Pi_AY, Pi_CX, Pi_CY, Pi_B, M_X, N_X one by one and adds the points — the
 *                                     reference's own commit list (prove/src/lib.rs:2572-3184) — where the default adds the quotient
 *                                     polynomials first and commits Pi_X, Pi_Y once and N_X not at all (N_X = M_X);
 *        TKMK_PROVE_COEFFICIENT_BASIS U, V, W, B, R are committed from their coefficients (the reference's encode_poly) although the
 *                                     context holds the Lagrange-basis tables.
 * Neither changes a byte of the proof.  commit_boxes_json_out (optional, malloc'ed, tkmk_prover_free_string): a JSON array
 * [{"name", "x", "y", "basis": "coeff" | "evals"}] in commit order — (x_degree + 1) x (y_degree + 1) of encode_poly
 * (libs/src/iotools/mod.rs:2055-2060), the `msm=AxB` column of the reference's timing reports
 * (prove/optimization/timing.local.cpu.current.md "Encode Details"); pinned by tests/golden/encode_dims.json. */
#define TKMK_PROVE_TEST_PARTS 1
#define TKMK_PROVE_COEFFICIENT_BASIS 2
tkmk_error tkmk_prover_prove_ex(tkmk_prover *p, const char *synthesizer_dir, const char *output_dir, const char *testing_mixer_json, int flags,
                                tkmk_prove_timing *timing, char **proof_json_out, char **commit_boxes_json_out);
tkmk_error tkmk_prover_close(tkmk_prover *p);
void tkmk_prover_free_string(char *s);
const char *tkmk_prover_last_error(void);   /* message of the last failed call on this thread */
/* where the context's reference string came from: "combined_sigma.rkyv" or "combined_sigma.tkcrs" */
const char *tkmk_prover_crs_source(const tkmk_prover *p);

#ifdef __cplusplus
}
#endif
#endif
