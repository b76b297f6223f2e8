/*
 * tkmk.h — C ABI of libtkmk_hip.so, the MI355X (gfx950) backend for the Tokamak zk-EVM prover's
 * polynomial-commitment hot path: BLS12-381 G1 multi-scalar multiplication, scalar-field (Fr)
 * NTT / iNTT incl. the bivariate "_biNTT", and the element-wise Fr vector ops between them.
 *
 * WHAT THIS REPLACES.  The reference's Rust crate `libs` (packages/backend/libs) reaches the device
 * only through ICICLE v3.8.0's Rust wrappers (icicle_core / icicle_runtime / icicle_bls12_381, git tag
 * pinned at packages/backend/Cargo.toml:20-23), each a thin shim over one `extern "C"` symbol of the
 * ICICLE frontend library.  ICICLE is not vendored in the reference and is absent from this build
 * environment, so the symbol names and struct layouts below follow ICICLE v3's public C API as
 * documented/remembered and MUST be re-checked against the v3.8.0 headers
 * before claiming link-level compatibility; the semantics are fixed by the reference call sites
 * cited on every entry point.  INTEGRATION.md shows the Rust binding a maintainer adds.
 *
 * DATA ENCODING (pinned by packages/backend/setup/mpc-setup/src/conversions.rs:43-95 and
 * libs/src/iotools/mod.rs:1701-1706,1785-1816): every field element crosses this ABI as the PLAIN
 * (non-Montgomery) integer in little-endian u32 limbs — Fr 32 B, Fq 48 B; G1 affine = {x, y} 96 B with
 * (0,0) = point at infinity; G1 projective = {x, y, z} 144 B, homogeneous (X/Z, Y/Z), z = 0 = infinity.
 *
 * ERRORS: every function returns a tkmk_error (0 = success); nothing throws or aborts across the ABI.
 * THREADING (the contract, in full):
 *   - one host thread at a time PER STREAM: every entry keeps its scratch in a grow-only arena that belongs to the stream it is
 *     given (stream_handle / the `stream` argument; NULL = the default stream), so two threads may call concurrently only if they
 *     name DIFFERENT streams (the reference issues all device calls from its main thread: SURVEY.md section 8b);
 *   - one MSM BATCH at a time per CALLER STREAM: bls12_381_msm (batch or single), tkmk_msm_multi and tkmk_msm_multi_ex run their
 *     jobs over a set of pipeline streams and pinned result buffers that belongs to the stream the call names (cfg->stream_handle;
 *     NULL = the default stream), behind that set's lock, held from the first job's launch to the last job's result — a second
 *     thread's MSM call on the SAME stream waits for the whole batch of the first; calls that name DIFFERENT streams run
 *     concurrently and share the device (the resident prover's helper threads commit on their own streams beside the main
 *     thread's batches: results are unaffected, see tests/test_gpu_service.py);
 *   - process-global state: the NTT domain (like ICICLE's; initialise / grow it from one thread while no transform runs), the
 *     device binding (tkmk_set_device: one device per process), the allocator's cache, the pipeline width
 *     (tkmk_msm_set_pipeline_streams), the profile / stats counters;
 *   - libtkmk_prover.so: a context issues everything on the default stream, so the library serialises tkmk_prover_open / _prove /
 *     _close of UNSHARDED contexts of one process behind one lock (two contexts prove one after the other, byte for byte as they
 *     would alone); a sharded context's ranks are one process per GPU (or take turns through the loopback communicator).
 * There is NO CPU fallback: without a usable gfx950 device every compute entry returns
 * TKMK_ERR_NO_DEVICE.
 */
#ifndef TKMK_H
#define TKMK_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors icicle_runtime::errors::eIcicleError as matched by libs/src/vector_operations/mod.rs:386 */
typedef enum {
    TKMK_SUCCESS = 0,
    TKMK_ERR_INVALID_DEVICE = 1,
    TKMK_ERR_OUT_OF_MEMORY = 2,
    TKMK_ERR_INVALID_POINTER = 3,
    TKMK_ERR_ALLOCATION_FAILED = 4,
    TKMK_ERR_DEALLOCATION_FAILED = 5,
    TKMK_ERR_COPY_FAILED = 6,
    TKMK_ERR_SYNCHRONIZATION_FAILED = 7,
    TKMK_ERR_STREAM_CREATION_FAILED = 8,
    TKMK_ERR_STREAM_DESTRUCTION_FAILED = 9,
    TKMK_ERR_API_NOT_IMPLEMENTED = 10,
    TKMK_ERR_INVALID_ARGUMENT = 11,
    TKMK_ERR_NO_DEVICE = 12,
    TKMK_ERR_UNKNOWN = 999
} tkmk_error;

typedef struct { uint32_t limbs[8]; } tkmk_fr;            /* icicle_bls12_381::curve::ScalarField */
typedef struct { uint32_t limbs[12]; } tkmk_fq;           /* BaseField */
typedef struct { tkmk_fq x, y; } tkmk_g1_affine;          /* G1Affine  (96 B)  */
typedef struct { tkmk_fq x, y, z; } tkmk_g1_projective;   /* G1Projective (144 B) */
typedef void *tkmk_stream;                                /* IcicleStream handle (a hipStream_t) */

/* THE ONE DECLARED CONVENTION THAT IS AN INFERENCE, NOT A PIN ("parity unpinned"): the generator g of the
 * scalar field's two-adic subgroup, omega_{2^32} = g^((r-1)/2^32).  bls12_381_get_root_of_unity, the oracle, the Python
 * restatements and the setup all derive their root of unity from THIS constant.  The reference takes the root from ICICLE
 * (ntt::get_root_of_unity, libs/src/bivariate_polynomial/mod.rs:47-52; not in the tree).  5 = ffjavascript's rule (smallest
 * quadratic non-residue), which the reference's browser prover / verifier use on the native prover's outputs;
 * 7 = arkworks / zkcrypto's constant.  Both generate the subgroup and pass every self-consistency test; they order the domain
 * differently, so a wrong choice is silent within this repository and fatal against a reference-made CRS.  If ICICLE's constant
 * turns out to be 7^((r-1)/2^32): change this line (tests/test_root_convention.py proves the switch green), or set the
 * environment variable TKMK_FR_ROOT_GENERATOR=7 for a process (the same variable is honoured by the oracle and the Python refs). */
#define TKMK_BLS12_381_FR_ROOT_GENERATOR 5

/* ---------------------------------------------------------------------------------------------
 * Device runtime — replaces icicle_runtime::{memory::DeviceVec, stream::IcicleStream, Device}
 * used at ~40 sites, e.g. libs/src/bivariate_polynomial/mod.rs:446-457,1661-1662;
 * libs/src/utils/mod.rs:78-110 (check_device); libs/src/vector_operations/mod.rs:484-549.
 * --------------------------------------------------------------------------------------------- */
tkmk_error tkmk_device_count(int *count);                 /* is_device_available */
/* icicle_runtime::set_device (id is 0 in the reference: libs/src/utils/mod.rs:88-110).  The library's state (scratch arenas,
 * allocation cache, NTT domain, internal streams) lives on ONE device, the one current when the first entry point runs: call this
 * first to choose it.  A later call naming another device returns TKMK_ERR_INVALID_DEVICE — one process per GPU (tkmk_dist.h). */
tkmk_error tkmk_set_device(int device_id);
/* the verdict tkmk_set_device would give for (device the library is bound to or -1, requested device, devices present): no device
 * is touched — lets a host (and the not-gpu tests) check the rule */
tkmk_error tkmk_diag_device_switch(int bound_device, int requested_device, int device_count);
tkmk_error tkmk_get_available_memory(size_t *total, size_t *free_bytes); /* get_available_memory */
tkmk_error tkmk_malloc(void **ptr, size_t bytes);         /* DeviceVec::device_malloc */
tkmk_error tkmk_malloc_async(void **ptr, size_t bytes, tkmk_stream s);
/* Drop for DeviceVec.  The block goes to a reuse cache and is handed out again only after everything queued before this call has
 * finished — on the default stream AND on every stream made by tkmk_stream_create (no stream is named here, as in Drop). */
tkmk_error tkmk_free(void *ptr);
tkmk_error tkmk_free_async(void *ptr, tkmk_stream s);
tkmk_error tkmk_memcpy_h2d(void *dst, const void *src, size_t bytes);   /* copy_from_host */
tkmk_error tkmk_memcpy_d2h(void *dst, const void *src, size_t bytes);   /* copy_to_host   */
tkmk_error tkmk_memcpy_d2d(void *dst, const void *src, size_t bytes);   /* DeviceVec::copy */
tkmk_error tkmk_memcpy_h2d_async(void *dst, const void *src, size_t bytes, tkmk_stream s);
tkmk_error tkmk_memcpy_d2h_async(void *dst, const void *src, size_t bytes, tkmk_stream s);
/* strided device copy: `rows` rows of width_bytes (sub-grid gather of the CRS / coefficient box in encode_poly,
 * libs/src/iotools/mod.rs:2061-2088) */
tkmk_error tkmk_memcpy_2d_d2d(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t rows);
tkmk_error tkmk_memset(void *ptr, int value, size_t bytes);
tkmk_error tkmk_stream_create(tkmk_stream *s);            /* IcicleStream::create */
tkmk_error tkmk_stream_synchronize(tkmk_stream s);        /* IcicleStream::synchronize */
tkmk_error tkmk_stream_destroy(tkmk_stream s);            /* IcicleStream::destroy */
/* Marks a stream as BACKGROUND (on != 0) or foreground again: the accumulate kernels of MSM batches issued on a background stream hold one
 * workgroup per CU (half their usual share of a CU's registers), so that kernels issued elsewhere at the same time — the polynomial work
 * a round IS waiting for — find room on every CU.  For commitments nothing waits for until later.  Results are unaffected.  No ICICLE
 * counterpart (IcicleStream has no priorities: icicle_runtime::stream). */
tkmk_error tkmk_stream_set_background(tkmk_stream stream, int on);
tkmk_error tkmk_device_synchronize(void);
/* scratch used by MSM / NTT calls lives in per-stream grow-only arenas; this returns all of it to the driver */
tkmk_error tkmk_release_scratch(void);
const char *tkmk_error_string(tkmk_error e);
/* GenerateRandom (icicle_core::traits): n uniform scalars (the prover's blinding scalars, prove/src/lib.rs:1040-1080) / n random G1
 * points [k]G into HOST buffers; the scalars come from getrandom() with rejection sampling (host-only), the points are computed on
 * the device */
tkmk_error bls12_381_generate_scalars(tkmk_fr *out_host, size_t n);
tkmk_error bls12_381_generate_random_affine_points(tkmk_g1_affine *out_host, size_t n);
/* 1 if libtkmk_hip.so was built with gfx950 code objects (always, for this library) */
int tkmk_is_hip_build(void);
/* host-only: Keccak-256 (original 0x01 padding), the hash of the Fiat-Shamir transcript — replaces the tiny_keccak calls of
 * RollingKeccakTranscript::update / get_challenge_raw (prove/src/lib.rs:3247-3394); no device is touched */
tkmk_error tkmk_keccak256(const uint8_t *data, size_t len, uint8_t out[32]);
/* host-only: offsets and lengths of the 3 * n_constraints linear combinations of an iden3 .r1cs constraints section (the walk of
 * R1csBinary::scan_constraints, libs/src/iotools/mod.rs:613-650); *consumed = bytes walked (compare with section_bytes for
 * trailing bytes); INVALID_ARGUMENT if the section ends inside the walk */
tkmk_error tkmk_r1cs_index(const uint8_t *section, size_t section_bytes, uint32_t n_constraints, uint32_t field_size,
                           uint64_t *starts, uint32_t *counts, size_t *consumed);

/* ---------------------------------------------------------------------------------------------
 * MSM — replaces icicle_core::msm::msm<G1> (extern "C" bls12_381_msm in ICICLE v3) as called at
 *   libs/src/iotools/mod.rs:2093-2099          (encode_poly: one dense MSM over the CRS sub-grid)
 *   libs/src/group_structures/mod.rs:108-114   (macro twin over an owned Sigma1)
 *   libs/src/group_structures/mod.rs:127-143   (msm_g1_bases: gathered binding MSMs)
 *   libs/src/iotools/mod.rs:1113-1151,1239-1252 (setup: batched scalar-mul expressed as MSM batches)
 * Field order and defaults follow ICICLE v3 MSMConfig.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    tkmk_stream stream_handle;          /* NULL = default stream */
    int precompute_factor;              /* 1 = none; F > 1: `bases` is the table made by bls12_381_msm_precompute_bases */
    int c;                              /* window bits (2..18); 0 = choose from size */
    int bitsize;                        /* scalar bits to process; 0 = 255 */
    int batch_size;                     /* number of MSMs; results = batch_size points */
    bool are_points_shared_in_batch;    /* true: one bases array of msm_size; false: batch_size * msm_size */
    bool are_scalars_on_device;
    bool are_scalars_montgomery_form;
    bool are_points_on_device;
    bool are_points_montgomery_form;
    bool are_results_on_device;
    bool is_async;                      /* true: return before completion (device results only) */
    void *ext;                          /* must be NULL */
} tkmk_msm_config;

tkmk_msm_config tkmk_msm_default_config(void);            /* MSMConfig::default() */

/* results[b] = sum_i scalars[b*msm_size + i] * bases[(shared ? 0 : b*msm_size) + i].
 * msm_size = points per MSM.  Results are returned as canonical projective (x_affine, y_affine, 1)
 * or (0,1,0) for infinity, so that the reference's `G1Affine::from(projective)`
 * (libs/src/iotools/mod.rs:2112) yields the unique affine point. */
tkmk_error bls12_381_msm(const tkmk_fr *scalars, const tkmk_g1_affine *bases, int msm_size,
                         const tkmk_msm_config *cfg, tkmk_g1_projective *results);

/* Precomputed bases (ICICLE msm_precompute_bases; MSMConfig::precompute_factor): output_bases receives
 * msm_size * F' points (F' = min(cfg->precompute_factor, windows)) — the 2^(c W' j) multiples of every base, in the
 * library's converted form — on the device if cfg->are_results_on_device, else on the host.  A later bls12_381_msm /
 * tkmk_msm_multi call with the same c, bitsize and precompute_factor takes that table as `bases` and runs W' =
 * ceil(windows / F') windows over msm_size * F' points: the same number of bucket additions, no per-call base
 * conversion, 1/F' of the buckets.  The reference leaves precompute_factor at 1 (MSMConfig::default(),
 * libs/src/iotools/mod.rs:2096).  Measured gain on MI355X is modest (2^20 points: 4.5 -> 4.1 ms at F = 4) because the
 * bucket reduction is a latency chain whose length does not depend on the number of windows. */
tkmk_error bls12_381_msm_precompute_bases(const tkmk_g1_affine *bases, int msm_size, const tkmk_msm_config *cfg,
                                          tkmk_g1_affine *output_bases);

/* Several independent MSMs (own scalars, own bases, own size) in one call, e.g. the commits the prover issues
 * between two transcript challenges (prove/src/lib.rs prove0..prove4 call encode_poly back to back on
 * polynomials that do not depend on each other).  The jobs are pipelined over internal HIP streams so that one
 * job's latency-bound bucket reduction overlaps the next job's sort and accumulation; results[j] is exactly what
 * bls12_381_msm would return for job j.  cfg->batch_size must be 1; the *_on_device / *_montgomery_form flags
 * apply to every job.  A bls12_381_msm call with batch_size > 1 uses the same pipeline. */
typedef struct {
    const tkmk_fr *scalars;
    const tkmk_g1_affine *bases;
    int msm_size;
} tkmk_msm_job;
tkmk_error tkmk_msm_multi(const tkmk_msm_job *jobs, int n_jobs, const tkmk_msm_config *cfg, tkmk_g1_projective *results);

/* The same over VIEWS of device-resident tables, for the two shapes in which the reference builds MSM operands by copying:
 *   encode_poly (libs/src/iotools/mod.rs:2061-2088; macro twin libs/src/group_structures/mod.rs:59-119) trims the coefficient
 *     matrix to its degree box and copies the matching sub-grid of xy_powers, point by point, before every MSM:
 *       scalar_cols / scalar_stride and base_cols / base_stride address the box inside the resident matrix / CRS table
 *       (element k of the MSM = row k / cols, column k % cols of a stride-wide row-major table);
 *   msm_g1_bases over per-wire CRS rows (libs/src/group_structures/mod.rs:127-143,266-300):
 *       base_index[k] = row of the binding table that wire k commits against (device array of msm_size u32).
 * bases_form says what `bases` holds: TKMK_BASES_PLAIN / _MONTGOMERY as in tkmk_msm_config, or TKMK_BASES_CONVERTED = a
 * table that went through bls12_381_msm_convert_bases once (the CRS is fixed for its lifetime; the per-call conversion of
 * bls12_381_msm is then skipped).  Scalars and bases must be on the device (cfg->are_*_on_device = true).
 * base_table_len = number of records behind `bases` (0 = unchecked for strided views; with base_index it must be > 0 — a job
 * that leaves it 0 is refused with TKMK_ERR_INVALID_ARGUMENT — and an
 * entry >= base_table_len makes the call return TKMK_ERR_INVALID_ARGUMENT — the read is clamped on the device, never
 * issued out of bounds).  results[j] is bit-identical to bls12_381_msm on the gathered operands. */
#define TKMK_BASES_PLAIN 0
#define TKMK_BASES_MONTGOMERY 1
#define TKMK_BASES_CONVERTED 2
typedef struct {
    const tkmk_fr *scalars;
    const tkmk_g1_affine *bases;
    int msm_size;
    uint32_t scalar_cols, scalar_stride;   /* 0, 0 = contiguous */
    uint32_t base_cols, base_stride;       /* 0, 0 = contiguous */
    const uint32_t *base_index;            /* NULL = none; takes precedence over base_cols */
    uint64_t base_table_len;
    /* precomputed TABLE (0, 0 = none): `bases` was made by bls12_381_msm_precompute_bases over the WHOLE base table
     * (base_table_len rows) with cfg->c = table_c and cfg->precompute_factor = table_factor: level j of the table (rows
     * [j * base_table_len, (j + 1) * base_table_len)) holds the 2^(table_c * W' * j) multiples of level 0, W' = ceil(windows /
     * table_factor).  The MSM then runs W' windows of table_c bits over msm_size * table_factor entries — with table_factor =
     * windows a SINGLE bucket set, so that wide windows (table_c up to 20: 13 instead of 16 bucket additions per point) do not
     * multiply the bucket-reduction work.  Views address level 0; bases_form must be TKMK_BASES_CONVERTED.  Results are
     * bit-identical to the plain MSM.  A job with table_c > 16 needs msm_size * table_factor >= 2^18. */
    uint32_t table_c, table_factor;
} tkmk_msm_job_ex;
tkmk_error tkmk_msm_multi_ex(const tkmk_msm_job_ex *jobs, int n_jobs, const tkmk_msm_config *cfg, int bases_form,
                             tkmk_g1_projective *results);
/* bases (n points; host or device per cfg->are_points_on_device, plain or Montgomery per cfg->are_points_montgomery_form) ->
 * the library's resident form for TKMK_BASES_CONVERTED, written to `out` (host or device per cfg->are_results_on_device;
 * out == bases is allowed).  Only meaningful as the `bases` of a later tkmk_msm_multi_ex call. */
tkmk_error bls12_381_msm_convert_bases(const tkmk_g1_affine *bases, uint64_t n, const tkmk_msm_config *cfg, tkmk_g1_affine *out);

/* ---------------------------------------------------------------------------------------------
 * NTT — replaces icicle_core::ntt::{ntt, initialize_domain, release_domain, get_root_of_unity}
 * (extern "C" bls12_381_ntt, _ntt_init_domain, _ntt_release_domain, _get_root_of_unity) as called at
 *   libs/src/bivariate_polynomial/mod.rs:33-55      (init_ntt_domain_for_size: global, grow-only)
 *   libs/src/bivariate_polynomial/mod.rs:1449-1476  (_biNTT: 1-D, row batch, strided column batch)
 * --------------------------------------------------------------------------------------------- */
typedef enum { TKMK_NTT_FORWARD = 0, TKMK_NTT_INVERSE = 1 } tkmk_ntt_dir;   /* NTTDir::{kForward,kInverse} */
typedef enum { TKMK_ORDER_NN = 0 } tkmk_ntt_ordering;     /* only natural->natural is used by the reference */

typedef struct {
    tkmk_stream stream_handle;
    tkmk_fr coset_gen;                  /* 1 = no coset. forward: x[j] *= g^j first; inverse: undone last */
    int batch_size;
    bool columns_batch;                 /* false: vector b at [b*n, (b+1)*n); true: element i of vector b at i*batch+b */
    tkmk_ntt_ordering ordering;
    bool are_inputs_on_device;
    bool are_outputs_on_device;
    bool is_async;
    void *ext;                          /* must be NULL */
} tkmk_ntt_config;

typedef struct {
    tkmk_stream stream_handle;
    bool is_async;
    void *ext;
} tkmk_ntt_init_domain_config;

tkmk_ntt_config tkmk_ntt_default_config(void);            /* NTTConfig::default() */
tkmk_error bls12_381_get_root_of_unity(uint64_t max_size, tkmk_fr *rou_out);
/* primitive_root must have order 2^k; builds twiddles for sizes up to 2^k. Fails if a domain exists. */
tkmk_error bls12_381_ntt_init_domain(const tkmk_fr *primitive_root, const tkmk_ntt_init_domain_config *cfg);
tkmk_error bls12_381_ntt_release_domain(void);
/* size of the current domain, 0 = none (the domain is process-global: every host side in the process asks here before growing it) */
tkmk_error bls12_381_ntt_domain_size(uint64_t *size);
/* size = length n of ONE vector (power of two, <= domain); total elements = n * batch_size.
 * input may equal output (in place). */
tkmk_error bls12_381_ntt(const tkmk_fr *input, int size, tkmk_ntt_dir dir, const tkmk_ntt_config *cfg,
                         tkmk_fr *output);

/* The reference's own 2-D transform, restated as ONE device entry so the intermediate never leaves
 * HBM: DensePolynomialExt::_biNTT (libs/src/bivariate_polynomial/mod.rs:1422-1478) =
 * rows (length y_size, coset_y) then strided columns (length x_size, coset_x); element (ix,iy) at
 * ix*y_size + iy.  coset_x / coset_y may be NULL (= 1).  Pointers are device pointers when
 * on_device is true, host pointers otherwise. */
tkmk_error tkmk_bintt(const tkmk_fr *input, size_t x_size, size_t y_size, tkmk_ntt_dir dir,
                      const tkmk_fr *coset_x, const tkmk_fr *coset_y, bool on_device,
                      tkmk_stream stream, tkmk_fr *output);
/* Forward _biNTT of the zero-padded extension of a compact in_x x in_y coefficient matrix to x_size x y_size (powers of two,
 * in_x <= x_size, in_y <= y_size) without materialising the padding — replaces `resize` + `_biNTT` in to_rou_evals, _mul and the
 * fused evaluator's leaves (libs/src/bivariate_polynomial/mod.rs:1646-1674, 1920-1960, 459-502): the row pass runs over the
 * in_x existing rows only and the column pass reads the absent rows as zeros.  Same result as tkmk_bintt on the resized
 * matrix, bit for bit.  Device pointers; output must not alias input. */
tkmk_error tkmk_bintt_padded(const tkmk_fr *input, size_t in_x, size_t in_y, size_t x_size, size_t y_size, const tkmk_fr *coset_x,
                             const tkmk_fr *coset_y, tkmk_stream stream, tkmk_fr *output);

/* ---------------------------------------------------------------------------------------------
 * Vector ops — replaces icicle_core::vec_ops::VecOps<ScalarField> (extern "C" bls12_381_vector_add, …)
 * as used at libs/src/vector_operations/mod.rs:34-139,326-336,612-626 and
 * libs/src/bivariate_polynomial/mod.rs:332-435,835,935,1590-1611,1974,2180,2230,2276.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    tkmk_stream stream_handle;
    bool is_a_on_device;
    bool is_b_on_device;
    bool is_result_on_device;
    bool is_async;
    int batch_size;                     /* 1 (batched reductions: sum over batch_size vectors) */
    bool columns_batch;
    void *ext;
} tkmk_vecops_config;

tkmk_vecops_config tkmk_vecops_default_config(void);      /* VecOpsConfig::default() */
tkmk_error bls12_381_vector_add(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *cfg, tkmk_fr *out);
/* accumulate (icicle_core::vec_ops::accumulate_scalars): a[i] += b[i], in place in a */
tkmk_error bls12_381_vector_accumulate(tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *cfg);
tkmk_error bls12_381_vector_sub(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *cfg, tkmk_fr *out);
tkmk_error bls12_381_vector_mul(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *cfg, tkmk_fr *out);
tkmk_error bls12_381_vector_div(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *cfg, tkmk_fr *out);
tkmk_error bls12_381_vector_inv(const tkmk_fr *a, uint64_t n, const tkmk_vecops_config *cfg, tkmk_fr *out);
/* a is ONE scalar (a[0]); out[i] = a + b[i],  a - b[i] (ICICLE v3 scalar_sub_vec),  a * b[i] */
tkmk_error bls12_381_scalar_add_vec(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *cfg, tkmk_fr *out);
tkmk_error bls12_381_scalar_sub_vec(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *cfg, tkmk_fr *out);
tkmk_error bls12_381_scalar_mul_vec(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *cfg, tkmk_fr *out);
/* out[b] = sum / product of vector b (batch_size vectors of n) */
tkmk_error bls12_381_vector_sum(const tkmk_fr *a, uint64_t n, const tkmk_vecops_config *cfg, tkmk_fr *out);
tkmk_error bls12_381_vector_product(const tkmk_fr *a, uint64_t n, const tkmk_vecops_config *cfg, tkmk_fr *out);
/* out[i] = prod_{j > i} a[j], out[n-1] = 1 (device pointers, out != a): the running product of prove1, a serial host loop over
 * 2^20 elements in the reference (packages/backend/prove/src/lib.rs:1858-1862) */
tkmk_error tkmk_vec_suffix_product(const tkmk_fr *a_dev, uint64_t n, tkmk_fr *out_dev, tkmk_stream stream);
/* row-major rows x cols -> cols x rows */
tkmk_error bls12_381_matrix_transpose(const tkmk_fr *in, uint32_t rows, uint32_t cols, const tkmk_vecops_config *cfg, tkmk_fr *out);

/* ---------------------------------------------------------------------------------------------
 * Bivariate NTT over G1 POINTS: out[i][j] = sum_{a < x_size, b < y_size} w_x^(+-i a) w_y^(+-j b) in[a][b], natural order, roots of
 * unity as bls12_381_get_root_of_unity; the inverse direction is NOT divided by x_size * y_size.  No reference counterpart (the
 * reference commits in the coefficient basis only).  The inverse transform of the CRS sub-grid [tau_x^a tau_y^b]G is the Lagrange-basis
 * CRS N [L_i(tau_x) L_j(tau_y)]G: an MSM of a polynomial's EVALUATIONS over it, times 1/N, is encode_poly of its coefficients —
 * and the prover's u, v, w, b are evaluations, mostly zeros and small numbers.  A one-time cost per circuit (seconds at 2^22 points).
 * in: device, x_size rows of in_stride records (the first y_size of each row are used), bases_form = TKMK_BASES_*; out: device,
 * x_size * y_size plain affine records ((0, 0) = infinity).  Sizes are powers of two, x_size * y_size < 2^31.
 * --------------------------------------------------------------------------------------------- */
tkmk_error tkmk_g1_ntt(const tkmk_g1_affine *in_dev, int bases_form, uint32_t in_stride, uint32_t x_size, uint32_t y_size, tkmk_ntt_dir dir,
                       tkmk_g1_affine *out_dev, tkmk_stream stream);
/* the same with either pass switched off (axes = TKMK_G1_NTT_AXIS_X | _Y is tkmk_g1_ntt): _Y alone transforms the x_size rows, _X alone
 * the y_size columns — a sharded prover context runs each over its own part of the grid with a change of layout in between */
#define TKMK_G1_NTT_AXIS_X 1
#define TKMK_G1_NTT_AXIS_Y 2
tkmk_error tkmk_g1_ntt_axes(const tkmk_g1_affine *in_dev, int bases_form, uint32_t in_stride, uint32_t x_size, uint32_t y_size, tkmk_ntt_dir dir,
                            int axes, tkmk_g1_affine *out_dev, tkmk_stream stream);
/* out[i] = [scalar] in[i]: n plain affine records on the device (out_dev may be in_dev), one host scalar (plain integer below r).  No
 * reference counterpart (the reference multiplies single points on the host: G1serde `*`, libs/src/group_structures/mod.rs:916-931); here it
 * folds the 1 / N of an inverse transform into a resident Lagrange-basis table once per circuit. */
tkmk_error tkmk_g1_scale(const tkmk_g1_affine *in_dev, uint64_t n, const tkmk_fr *scalar, tkmk_g1_affine *out_dev, tkmk_stream stream);
/* Prefix sums of points: out[j] = sum_{j' <= j} in[idx(j')], idx(j) = j, or (j % rows) * cols + j / rows with `transposed` (the rows x cols
 * row-major table walked column by column).  With the Lagrange-basis points in the order of prove1's running product (lib.rs:1858-1866) this
 * is the table over which a PIECEWISE-CONSTANT evaluation vector commits as an MSM of its few jumps: sum_j r_j L_j = sum_j (r_j - r_{j+1}) S_j,
 * and r jumps only where the copy permutation is not the identity.  in: device, form TKMK_BASES_*; out: device, plain affine; out != in. */
tkmk_error tkmk_g1_prefix_sums(const tkmk_g1_affine *in_dev, int bases_form, uint32_t rows, uint32_t cols, int transposed,
                               tkmk_g1_affine *out_dev, tkmk_stream stream);

/* ---------------------------------------------------------------------------------------------
 * G2 MSM on BLS12-381 (the twist y^2 = x^3 + 4(1 + u) over Fp2 = Fq[u]/(u^2 + 1)) — ICICLE v3's `bls12_381_g2_msm`
 * (icicle_bls12_381::curve::G2CurveCfg).  The reference has no G2 MSM call site: G2 appears as nine scalar multiplications
 * of the generator in Sigma2::gen (packages/backend/libs/src/group_structures/mod.rs:752-777), which fit this entry as a
 * batch of one-point MSMs with shared points, and in the verifier's pairings.  Config fields as in bls12_381_msm except:
 * precompute_factor must be 0 or 1, c must be 0 (auto) or 2..12.  An affine record of all zeros is the point at infinity;
 * results are canonical projective, (x_affine, y_affine, 1) or (0, 1, 0), c0 before c1 in every Fp2 element.
 * --------------------------------------------------------------------------------------------- */
typedef struct { tkmk_fq c0, c1; } tkmk_fq2;               /* c0 + c1 u  (96 B) */
typedef struct { tkmk_fq2 x, y; } tkmk_g2_affine;          /* G2Affine (192 B) */
typedef struct { tkmk_fq2 x, y, z; } tkmk_g2_projective;   /* G2Projective (288 B) */
tkmk_error bls12_381_g2_msm(const tkmk_fr *scalars, const tkmk_g2_affine *bases, int msm_size, const tkmk_msm_config *cfg,
                            tkmk_g2_projective *results);

/* ---------------------------------------------------------------------------------------------
 * BN254 (alt_bn128) G1 MSM — the same Pippenger kernels instantiated over the 254-bit fields
 * (ICICLE v3 exports the per-curve twin `bn254_msm`).  The reference links only icicle-bls12-381
 * (packages/backend/Cargo.toml:23), so no reference call site exists for it; it is here
 * because BASELINE.json's configs name a 2^24-point BN254 G1 MSM.  Same config struct, flags, digit/bucket
 * pipeline and result convention ((x_affine, y_affine, 1) or (0, 1, 0)) as bls12_381_msm.
 * --------------------------------------------------------------------------------------------- */
typedef struct { uint32_t limbs[8]; } tkmk_bn254_fr;             /* icicle_bn254::curve::ScalarField */
typedef struct { uint32_t limbs[8]; } tkmk_bn254_fq;             /* BaseField */
typedef struct { tkmk_bn254_fq x, y; } tkmk_bn254_g1_affine;     /* 64 B, (0,0) = infinity */
typedef struct { tkmk_bn254_fq x, y, z; } tkmk_bn254_g1_projective; /* 96 B */
tkmk_error bn254_msm(const tkmk_bn254_fr *scalars, const tkmk_bn254_g1_affine *bases, int msm_size,
                     const tkmk_msm_config *cfg, tkmk_bn254_g1_projective *results);
typedef struct {
    const tkmk_bn254_fr *scalars;
    const tkmk_bn254_g1_affine *bases;
    int msm_size;
} tkmk_bn254_msm_job;
tkmk_error bn254_msm_precompute_bases(const tkmk_bn254_g1_affine *bases, int msm_size, const tkmk_msm_config *cfg,
                                      tkmk_bn254_g1_affine *output_bases);
tkmk_error tkmk_bn254_msm_multi(const tkmk_bn254_msm_job *jobs, int n_jobs, const tkmk_msm_config *cfg,
                                tkmk_bn254_g1_projective *results);
/* BN254 scalar-field NTT: the twins of bls12_381_get_root_of_unity / _ntt_init_domain / _ntt_release_domain / _ntt and
 * tkmk_bintt (declared below with their semantics); own process-global domain; two-adicity 28,
 * w_{2^28} = 5^((r-1)/2^28).  No reference call site (BASELINE.json configs[0] names the transform). */
tkmk_error bn254_get_root_of_unity(uint64_t max_size, tkmk_bn254_fr *rou_out);
tkmk_error bn254_ntt_init_domain(const tkmk_bn254_fr *primitive_root, const tkmk_ntt_init_domain_config *cfg);
tkmk_error bn254_ntt_release_domain(void);
tkmk_error bn254_ntt_domain_size(uint64_t *size);
tkmk_error bn254_ntt(const tkmk_bn254_fr *input, int size, tkmk_ntt_dir dir, const tkmk_ntt_config *cfg,   /* cfg->coset_gen: 8 limbs */
                     tkmk_bn254_fr *output);
tkmk_error tkmk_bn254_bintt(const tkmk_bn254_fr *input, size_t x_size, size_t y_size, tkmk_ntt_dir dir,
                            const tkmk_bn254_fr *coset_x, const tkmk_bn254_fr *coset_y, bool on_device, tkmk_stream stream,
                            tkmk_bn254_fr *output);
/* input generation twins of tkmk_fr_random_device / tkmk_g1_batch_scalar_mul_device (below) */
tkmk_error tkmk_bn254_fr_random_device(uint64_t seed, uint64_t first, uint64_t n, tkmk_bn254_fr *out_dev, tkmk_stream s);
tkmk_error tkmk_bn254_g1_batch_scalar_mul_device(const tkmk_bn254_fr *scalars_dev, const tkmk_bn254_g1_affine *base_host,
                                                 uint64_t n, tkmk_bn254_g1_affine *out_dev, tkmk_stream s);

/* ---------------------------------------------------------------------------------------------
 * Deterministic input generation on the device — used by bench.py and the tests to
 * build 2^24-point inputs without a multi-GiB fixture; also the "MSM as batched scalar-mul" setup
 * path of libs/src/iotools/mod.rs:1113-1151 (n results = n independent [s_i]P).
 * --------------------------------------------------------------------------------------------- */
/* out[i] = splitmix64(seed) stream element (first + i), reduced mod r; device pointer */
tkmk_error tkmk_fr_random_device(uint64_t seed, uint64_t first, uint64_t n, tkmk_fr *out_dev, tkmk_stream s);
/* dst[i] = src[idx[i]] for rows of row_bytes (multiple of 16; 96 = one G1 affine point): device-side gather of the
 * bases of the binding commitments (libs/src/group_structures/mod.rs:127-300 builds these lists on the host) */
tkmk_error tkmk_gather_rows_device(const void *src_dev, uint32_t row_bytes, const uint32_t *idx_dev, uint64_t n, void *dst_dev,
                                   tkmk_stream s);
/* out[i] = [scalars[i]] base  (affine results); all pointers device except `base` (host) */
tkmk_error tkmk_g1_batch_scalar_mul_device(const tkmk_fr *scalars_dev, const tkmk_g1_affine *base_host,
                                           uint64_t n, tkmk_g1_affine *out_dev, tkmk_stream s);

/* ---------------------------------------------------------------------------------------------
 * Bivariate coefficient-matrix helpers — device-resident replacements for the HOST loops of
 * DensePolynomialExt (libs/src/bivariate_polynomial/mod.rs), which copy the whole matrix D->H first
 * (SURVEY.md §8 rows a10-a12).  No ICICLE symbol corresponds to these; a maintainer calls them from
 * the same Rust methods.  Element (ix, iy) at ix*y_size + iy.  All matrix pointers are DEVICE pointers;
 * scalars (x, y, factors) are host pointers; kernels are enqueued on `stream` (results written to host
 * outputs imply a stream synchronisation).
 * --------------------------------------------------------------------------------------------- */
/* find_degree (mod.rs:1480-1515): largest row / column index with a non-zero coefficient, -1 if none */
tkmk_error tkmk_poly_find_degree(const tkmk_fr *coeffs_dev, uint32_t x_size, uint32_t y_size, int64_t *x_degree,
                                 int64_t *y_degree, tkmk_stream stream);
/* dst (dx x dy) = zeros with src (sx x sy) placed at (off_x, off_y), clipped: resize (mod.rs:1784-1806) is
 * off = (0,0); mul_monomial (mod.rs:1820-1844) is off = (x_exponent, y_exponent) */
tkmk_error tkmk_poly_place(const tkmk_fr *src_dev, uint32_t sx, uint32_t sy, tkmk_fr *dst_dev, uint32_t dx, uint32_t dy,
                           uint32_t off_x, uint32_t off_y, tkmk_stream stream);
/* dst[i][j] = src[i][j] * factor_x^i * factor_y^j (_scale_coeffs, mod.rs:1567-1613; NULL factor = 1); in place ok */
tkmk_error tkmk_poly_scale_coeffs(const tkmk_fr *src_dev, uint32_t x_size, uint32_t y_size, const tkmk_fr *factor_x,
                                  const tkmk_fr *factor_y, tkmk_fr *dst_dev, tkmk_stream stream);
/* out (out_x_size x y_size) = p * scale * (1 + X + ... + X^(m-1)) truncated to out_x_size rows, in coefficient space:
 * out[k][j] = scale * sum_{i = k-m+1 .. k} p[i][j].  With scale = 1/m that factor is the Lagrange polynomial K_0 of the m-th roots
 * of unity (unit evaluations at index 0), which prove2 / prove4 multiply large polynomials by (lib.rs:2238-2246, 3012-3040) — there
 * through three bivariate NTTs per product, here two running sums.  out != p. */
tkmk_error tkmk_poly_mul_ones_x(const tkmk_fr *p_dev, uint32_t x_size, uint32_t y_size, uint32_t m, const tkmk_fr *scale,
                                uint32_t out_x_size, tkmk_fr *out_dev, tkmk_stream stream);
/* dst[i][j] = evals[i][j] * (w_x^i - 1): PolyExpr::MulXMinusOne on the evaluation domain (mod.rs:372-378, 504-518) */
tkmk_error tkmk_poly_mul_x_minus_one_evals(const tkmk_fr *evals_dev, uint32_t x_size, uint32_t y_size, tkmk_fr *dst_dev,
                                           tkmk_stream stream);

/* One-pass evaluation of a pointwise expression over evaluation-domain matrices: the device side of
 * PolyExpr::evaluate_fused_with_domain / evaluate_on_domain (libs/src/bivariate_polynomial/mod.rs:227-260, 311-435), which in
 * the reference runs one pass (and one fresh 256 MiB buffer) per tree node — prove2's p_comb is ~15 passes over 2^23
 * elements.  `prog` is the tree in postfix order over a stack of at most 6 values:
 *   LEAF k   push leaves_dev[k][e]          CONST k  push consts[k]          ADD / SUB / MUL   (second op top)
 *   SCALE k  top *= consts[k]               MUL_X_MINUS_ONE  top *= (w_x^ix - 1), w_x of order x_size (mod.rs:372-378)
 * Every leaf is an x_size * y_size device matrix of evaluations (plain form, as tkmk_bintt writes them); consts are
 * host scalars; out_dev may alias a leaf.  At most 16 leaves, 16 constants, 128 steps after form conversions. */
typedef enum {
    TKMK_EXPR_LEAF = 0, TKMK_EXPR_CONST = 1, TKMK_EXPR_ADD = 2, TKMK_EXPR_SUB = 3, TKMK_EXPR_MUL = 4, TKMK_EXPR_SCALE = 5,
    TKMK_EXPR_MUL_X_MINUS_ONE = 6
} tkmk_expr_opcode;
typedef struct { uint8_t op; uint8_t arg; } tkmk_expr_instr;
tkmk_error tkmk_poly_expr_eval(const tkmk_expr_instr *prog, uint32_t n_instr, const tkmk_fr *const *leaves_dev, uint32_t n_leaves,
                               const tkmk_fr *consts, uint32_t n_consts, uint32_t x_size, uint32_t y_size, tkmk_fr *out_dev,
                               tkmk_stream stream);
/* The same evaluator over leaf VIEWS.  x_len is x_size or 1, y_len is y_size or 1 (a vector broadcast along the other axis: the
 * evaluations of an X-only or Y-only polynomial, which then cost a 1-D transform); element (i, j) of the domain reads
 * data[((i - rot_x) mod x_len) * y_len + ((j - rot_y) mod y_len)]: a rotation is the evaluation-domain form of
 * p(w_x^-rot_x X, w_y^-rot_y Y) (prove2's r(w^-1 X, Y), lib.rs:1969-1975, shares r's evaluations instead of a transform of its
 * own).  x_size and y_size must be powers of two. */
typedef struct {
    const tkmk_fr *data;          /* device */
    uint32_t x_len, y_len, rot_x, rot_y;
} tkmk_expr_leaf;
tkmk_error tkmk_poly_expr_eval_views(const tkmk_expr_instr *prog, uint32_t n_instr, const tkmk_expr_leaf *leaves, uint32_t n_leaves,
                                     const tkmk_fr *consts, uint32_t n_consts, uint32_t x_size, uint32_t y_size, tkmk_fr *out_dev,
                                     tkmk_stream stream);
/* The same on ONE ROW SLAB of a larger domain (a rank of the sharded prover, include/tkmk_dist.h ROWS layout): rows [x_first, x_first +
 * x_rows) of a domain with x_global rows; leaves are x_rows x y_size slabs or broadcast vectors (x_len in {x_rows, 1}); rot_x must be 0 (a
 * rotation along X crosses slabs: tkmk_dist_rows_rotate makes it beforehand).  Only MUL_X_MINUS_ONE depends on the global row index. */
tkmk_error tkmk_poly_expr_eval_views_slab(const tkmk_expr_instr *prog, uint32_t n_instr, const tkmk_expr_leaf *leaves, uint32_t n_leaves,
                                          const tkmk_fr *consts, uint32_t n_consts, uint32_t x_global, uint32_t x_first, uint32_t x_rows,
                                          uint32_t y_size, tkmk_fr *out_dev, tkmk_stream stream);
/* eval_x / eval_y / eval (mod.rs:1719-1750): out_dev has y_size / x_size elements; out_host one */
tkmk_error tkmk_poly_eval_x(const tkmk_fr *coeffs_dev, uint32_t x_size, uint32_t y_size, const tkmk_fr *x, tkmk_fr *out_dev,
                            tkmk_stream stream);
tkmk_error tkmk_poly_eval_y(const tkmk_fr *coeffs_dev, uint32_t x_size, uint32_t y_size, const tkmk_fr *y, tkmk_fr *out_dev,
                            tkmk_stream stream);
tkmk_error tkmk_poly_eval(const tkmk_fr *coeffs_dev, uint32_t x_size, uint32_t y_size, const tkmk_fr *x, const tkmk_fr *y,
                          tkmk_fr *out_host, tkmk_stream stream);
/* div_by_vanishing_opt (mod.rs:2284-2410): P = Q_X (X^c - 1) + Q_Y (Y^d - 1); c | x_size, d | y_size, powers of two;
 * quo_x is x_size x y_size, quo_y is c x y_size */
tkmk_error tkmk_poly_div_by_vanishing_opt(const tkmk_fr *p_dev, uint32_t x_size, uint32_t y_size, uint32_t c, uint32_t d,
                                          tkmk_fr *quo_x_dev, tkmk_fr *quo_y_dev, tkmk_stream stream);
/* Fused linear combination — poly_comb! (prove/src/lib.rs:30-38) and the operator chains the prover builds from `&a * &scalar`,
 * `&a + &b`, `&a - &b` and mul_monomial (libs/src/bivariate_polynomial/mod.rs:532-1281, 1820-1844), which in the reference cost one
 * pass and one temporary per operator: out (out_xs x out_ys, fully written) = sum_t coeffs[t] * X^off_x[t] Y^off_y[t] * polys[t],
 * one pass, every operand read once.  polys[t]: x_sizes[t] x y_sizes[t] coefficient matrix on the device; coeffs: host scalars;
 * off_x / off_y may be NULL; every shifted operand must fit into out; out must not alias an operand. */
tkmk_error tkmk_poly_lincomb(uint32_t n_terms, const tkmk_fr *coeffs_host, const tkmk_fr *const *polys_dev, const uint32_t *x_sizes,
                             const uint32_t *y_sizes, const uint32_t *off_x, const uint32_t *off_y, tkmk_fr *out_dev, uint32_t out_xs,
                             uint32_t out_ys, tkmk_stream stream);
/* div_by_ruffini (mod.rs:2412-2477): P = Q_X (X - x) + Q_Y (Y - y) + r; q_x is x_size x y_size, q_y has y_size elements */
tkmk_error tkmk_poly_div_by_ruffini(const tkmk_fr *p_dev, uint32_t x_size, uint32_t y_size, const tkmk_fr *x, const tkmk_fr *y,
                                    tkmk_fr *q_x_dev, tkmk_fr *q_y_dev, tkmk_fr *r_host, tkmk_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Opaque univariate polynomial — replaces icicle_bls12_381::polynomials::DensePolynomial, the storage object inside
 * DensePolynomialExt (libs/src/bivariate_polynomial/mod.rs:112-127): from_coeffs :1520,1542,1803, clone :523, copy_coeffs
 * :1485,1681,1706, get_coeff :1757, coeffs_mut_slice :127, eval :1725-1737, divide :2070 (+ the arithmetic ICICLE's object offers).
 * A handle owns one device buffer of plain Fr coefficients, low degree first; clone = device copy; delete releases it.
 * Results of add / subtract / multiply / multiply_by_scalar / slice / divide are NEW handles the caller deletes.
 * --------------------------------------------------------------------------------------------- */
typedef struct tkmk_polynomial tkmk_polynomial;
tkmk_error bls12_381_polynomial_create_from_coefficients(const tkmk_fr *coeffs, size_t n, bool on_device, tkmk_polynomial **out);
tkmk_error bls12_381_polynomial_create_from_rou_evaluations(const tkmk_fr *evals, size_t n, bool on_device, tkmk_polynomial **out);
tkmk_error bls12_381_polynomial_clone(const tkmk_polynomial *p, tkmk_polynomial **out);
tkmk_error bls12_381_polynomial_delete(tkmk_polynomial *p);
tkmk_error bls12_381_polynomial_nof_coeffs(const tkmk_polynomial *p, size_t *n);
tkmk_error bls12_381_polynomial_degree(const tkmk_polynomial *p, int64_t *degree);             /* -1 for the zero polynomial */
tkmk_error bls12_381_polynomial_copy_coeffs(const tkmk_polynomial *p, size_t start, size_t count, tkmk_fr *out, bool out_on_device);
tkmk_error bls12_381_polynomial_get_coeff(const tkmk_polynomial *p, size_t idx, tkmk_fr *out_host);
tkmk_error bls12_381_polynomial_coeffs_device_ptr(tkmk_polynomial *p, tkmk_fr **ptr, size_t *n);   /* coeffs_mut_slice */
tkmk_error bls12_381_polynomial_evaluate(const tkmk_polynomial *p, const tkmk_fr *x_host, tkmk_fr *out_host);
tkmk_error bls12_381_polynomial_add(const tkmk_polynomial *a, const tkmk_polynomial *b, tkmk_polynomial **out);
tkmk_error bls12_381_polynomial_subtract(const tkmk_polynomial *a, const tkmk_polynomial *b, tkmk_polynomial **out);
tkmk_error bls12_381_polynomial_multiply(const tkmk_polynomial *a, const tkmk_polynomial *b, tkmk_polynomial **out);   /* NTT domain must cover deg a + deg b */
tkmk_error bls12_381_polynomial_multiply_by_scalar(const tkmk_polynomial *a, const tkmk_fr *s_host, tkmk_polynomial **out);
/* out[i] = p[offset + i * stride], i < size */
tkmk_error bls12_381_polynomial_slice(const tkmk_polynomial *p, size_t offset, size_t stride, size_t size, tkmk_polynomial **out);
/* num = quot * den + rem with deg rem < deg den (long division); den must not be the zero polynomial */
tkmk_error bls12_381_polynomial_divide(const tkmk_polynomial *num, const tkmk_polynomial *den, tkmk_polynomial **quot, tkmk_polynomial **rem);

/* ---------------------------------------------------------------------------------------------
 * Witness side of the path: sparse R1CS rows x placement variables -> rows of the u / v / w evaluation
 * matrices.  Replaces eval_uvwxy_sparse_rows / eval_sparse_rows (libs/src/iotools/mod.rs:1426-1523,1590-1608), a host loop.
 * CSR of one matrix of one subcircuit; variables = n_placements x n_wires (plain Fr); out = s_max x n matrix, row
 * out_slot[p] receives placement p (rows >= n_rows stay as the caller zeroed them).  Device pointers throughout.
 * --------------------------------------------------------------------------------------------- */
tkmk_error tkmk_r1cs_eval_rows(const uint32_t *row_ptr_dev, const uint32_t *wire_dev, const tkmk_fr *coeff_dev, uint32_t n_rows,
                               uint32_t nnz, const tkmk_fr *variables_dev, uint32_t n_wires, uint32_t n_placements,
                               const uint32_t *out_slot_dev, uint32_t n, tkmk_fr *out_dev, tkmk_stream stream);

/* The same for a whole subcircuit library and every placement of a proof at once.  The library (CSR of A / B / C of every
 * subcircuit kind, read from <lib>/r1cs/subcircuit{id}.r1cs: SubcircuitR1CS::from_r1cs_sparse_only, libs/src/iotools/mod.rs:652-760)
 * is circuit-static: it is built once and kept on the device next to the CRS.  row_ptr / wire / coeff: 3 * n_sub HOST arrays in
 * the order [sub][A, B, C] (row_ptr[k]: n_rows[sub] + 1 entries; coeff plain Fr); wire indices are validated against n_wires.
 * _eval: placement p instantiates kind placement_id_dev[p] on vars_dev[placement_var_offset_dev[p] ..] (offsets in elements);
 * u / v / w are n x s_max evaluation matrices, element (row, placement), fully written — the layout from_rou_evals takes
 * (read_R1CS_gen_uvwXY transposes its s_max x n rows into it, libs/src/iotools/mod.rs:1391-1418).  Device pointers. */
typedef struct tkmk_r1cs_library tkmk_r1cs_library;
tkmk_error tkmk_r1cs_library_create(uint32_t n_sub, const uint32_t *n_rows, const uint32_t *n_wires, const uint32_t *const *row_ptr,
                                    const uint32_t *const *wire, const tkmk_fr *const *coeff, tkmk_r1cs_library **out);
tkmk_error tkmk_r1cs_library_destroy(tkmk_r1cs_library *lib);
tkmk_error tkmk_r1cs_library_eval(const tkmk_r1cs_library *lib, const tkmk_fr *vars_dev, const uint32_t *placement_id_dev,
                                  const uint64_t *placement_var_offset_dev, uint32_t n_placements, uint32_t n, uint32_t s_max,
                                  tkmk_fr *u_dev, tkmk_fr *v_dev, tkmk_fr *w_dev, tkmk_stream stream);
/* Routes witness values by a static (local wire, row) list of one subcircuit kind, for the n_placements placements of that kind
 * (variables at var_offset_dev[i], global placement index slot_dev[i]):
 *   matrix_dev      (optional): matrix_dev[row * matrix_stride + slot] = value — gen_bXY's interface-wire matrix
 *                               (libs/src/polynomial_structures/mod.rs:132-162; row = flattenMap[wire] - l)
 *   scalars_out_dev / index_out_dev (optional, together): entry i * n_list + e = value / row * index_inner + (index_add_slot ? slot : 0)
 *                               — the (scalar, CRS row) lists of encode_statement_common and encode_O_pub_free
 *                               (libs/src/group_structures/mod.rs:184-229, 266-300), consumed by tkmk_msm_multi_ex as base_index */
tkmk_error tkmk_witness_route(const tkmk_fr *vars_dev, const uint64_t *var_offset_dev, const uint32_t *slot_dev, uint32_t n_placements,
                              const uint32_t *list_wire_dev, const uint32_t *list_row_dev, uint32_t n_list, tkmk_fr *matrix_dev,
                              uint32_t matrix_stride, tkmk_fr *scalars_out_dev, uint32_t *index_out_dev, uint32_t index_inner,
                              int index_add_slot, tkmk_stream stream);
/* out_dev[dst_idx_dev[i]] = table_dev[src_idx_dev[i]], i < n; dst indices must be distinct — Permutation::to_poly's redirects
 * s0[row][col] = w_x^X, s1[row][col] = w_y^Y (libs/src/iotools/mod.rs:438-448).  table_len / out_len = elements behind the two
 * arrays: an entry with src >= table_len or dst >= out_len is skipped on the device (never dereferenced) and the call returns
 * TKMK_ERR_INVALID_ARGUMENT after the valid entries were written (the reference indexes a Vec and panics). */
tkmk_error tkmk_fr_scatter_table(const tkmk_fr *table_dev, uint64_t table_len, const uint32_t *src_idx_dev, const uint32_t *dst_idx_dev,
                                 uint64_t n, tkmk_fr *out_dev, uint64_t out_len, tkmk_stream stream);
/* pinned host memory for staging (HostSlice buffers the reference uploads from are pageable; pinned staging reaches link rate) */
tkmk_error tkmk_host_malloc(void **ptr, size_t bytes);
tkmk_error tkmk_host_free(void *ptr);

/* ---------------------------------------------------------------------------------------------
 * Measurement hooks (no reference counterpart; the reference's `timing` feature wraps host spans:
 * libs/src/lib.rs:11-141).  When enabled, launchers bracket each kernel with HIP events recorded on the
 * launch stream; names: "msm.digits|hist|scan|scatter|accumulate|reduce_segments|reduce_windows|
 * convert_bases", "ntt.pass<k>".  Recording does not wait for anything (the pipelined MSM entry keeps its overlap);
 * tkmk_profile_get waits for the recorded events and returns the section's summed time and launch count.
 * tkmk_stats_*: what the library was asked to do since the last reset — "msm.points", "msm.calls", "ntt.elements",
 * "ntt.calls" — for the algorithmic-byte figures of the roofline report (128 B per point, 64 B per element).
 * --------------------------------------------------------------------------------------------- */
/* Independent MSMs of one tkmk_msm_multi[_ex] call run round-robin on n internal streams (default 3, environment TKMK_MSM_STREAMS;
 * 1..8; 0 restores the default).  n = 1 issues every kernel of the batch in order on one stream: no two kernels of the batch
 * overlap, so event-bracketed section times and rocprofv3 kernel durations are those of each kernel running alone — the form
 * bench.py's profiling pass and profiles/r03_*_1stream.csv use.  Results never depend on n. */
tkmk_error tkmk_msm_set_pipeline_streams(int n);
int tkmk_msm_get_pipeline_streams(void);
tkmk_error tkmk_profile_enable(int on);
tkmk_error tkmk_profile_reset(void);
tkmk_error tkmk_profile_get(const char *name, double *sum_ms, int *count);
tkmk_error tkmk_stats_reset(void);
tkmk_error tkmk_stats_get(const char *name, uint64_t *value);
/* arithmetic micro-benchmarks (kind 0 Fr mul, 1 Fq mul, 2 v_mad_u64_u32, 3 G1 mixed add, 4 Fr add+sub, 5 Fq sqr) */
tkmk_error tkmk_diag_bench(int kind, uint32_t iters, uint32_t blocks, int reps, float *ms_out);
/* known-bytes gather probe (calibration of the FETCH_SIZE counter for the bucket-accumulation access pattern): n rows of row_bytes
 * (64 / 96 / 128) gathered from table_dev at idx_dev[i], one lane per row with consecutive 16-byte loads; *ms_out = mean launch time */
tkmk_error tkmk_diag_gather_probe(const void *table_dev, uint32_t row_bytes, const uint32_t *idx_dev, uint64_t n, int reps, float *ms_out);
/* out[i] = a[i]*b[i] through the device Montgomery product; field 0 = Fr (32 B), 1 = Fq (48 B); device pointers */
tkmk_error tkmk_diag_field_mul(int field, const void *a_dev, const void *b_dev, void *out_dev, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif /* TKMK_H */
