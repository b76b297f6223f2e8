/* tkmk_dist.h — C ABI of libtkmk_dist.so: the two sub-paths that shard across the GPUs of one node, for a host
 * in any language (one process per GPU).  The reference is single-device (device index 0 everywhere:
 * packages/backend/libs/src/utils/mod.rs:88-110), so there is no reference interface to replace; the entry points take the same
 * operands as their single-GPU twins in tkmk.h (bls12_381_msm, tkmk_bintt) plus a communicator.
 *
 * Collectives run through RCCL (xGMI between the GPUs of a node) on DEVICE buffers, and what they deliver is consumed on the device
 * (the gathered partial results are converted and summed by device kernels; only the final 144-byte result goes to the host).
 * The communicator is bootstrapped like NCCL's: rank 0 calls tkmk_comm_unique_id, hands the 128 bytes to the other ranks
 * over any channel the host already has (a file, MPI, torch.distributed, a socket), and every rank calls tkmk_comm_init.
 *
 * LOOPBACK transport (tkmk_comm_init_loopback): world_size VIRTUAL ranks inside one process on one GPU, one host thread per rank;
 * all_gather / all_to_all are device-to-device copies between the ranks' buffers behind a rendezvous, and the ranks take turns on the
 * device.  Same entry points, same code below the transport: this is how the G >= 2 index algebra of every entry (pack / place of
 * the transpose, gather + sum of partials, empty and infinite partials) runs on a one-GPU box (tests/test_gpu_dist.py).  Not a
 * production path: it adds no capacity.
 *
 * WHY all_gather + add and not a "bucket-sum reduce": RCCL has no reduction operator over 1152-bit group elements, and reducing
 * the 16 x 2^15 bucket sets of every rank (100 MB per GPU) would move 10^5 times more data than the 144-byte partial RESULTS
 * for the same answer (the MSM sum is associative: sum over ranks of (sum over the rank's points)).  Each rank therefore runs the
 * whole single-GPU pipeline on its shard and only the partial results meet: one ncclAllGather of 144 bytes per rank. */
#ifndef TKMK_DIST_H
#define TKMK_DIST_H
#include "tkmk.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tkmk_comm tkmk_comm;
#define TKMK_COMM_ID_BYTES 128

tkmk_error tkmk_comm_unique_id(uint8_t id[TKMK_COMM_ID_BYTES]);                 /* rank 0 only */
/* collective over all ranks; uses the calling thread's current device (tkmk_set_device) */
tkmk_error tkmk_comm_init(const uint8_t id[TKMK_COMM_ID_BYTES], int world_size, int rank, tkmk_comm **out);
/* world_size communicators (rank r in out_comms[r]) over the loopback transport; each is used from its own host thread, every
 * collective entry must be called by all of them, each is released with tkmk_comm_destroy */
tkmk_error tkmk_comm_init_loopback(int world_size, tkmk_comm **out_comms);
int tkmk_comm_is_loopback(const tkmk_comm *comm);
/* Loopback only (a no-op on an RCCL communicator): a caller that issues device work of its own between collective entries — the
 * sharded prover does — takes the device turn for the whole span (acquire = 1) and hands it back at the end (acquire = 0); the
 * entries give it up only while they wait for their peers. */
tkmk_error tkmk_comm_device_turn(tkmk_comm *comm, int acquire);
/* buf (host, bytes) of rank `root` -> buf of every rank: the few replicated host values a sharded computation must agree on (the
 * prover's blinding scalars).  One all-gather of `bytes` per rank. */
tkmk_error tkmk_comm_broadcast_host(tkmk_comm *comm, void *buf, size_t bytes, int root);
tkmk_error tkmk_comm_destroy(tkmk_comm *comm);
int tkmk_comm_rank(const tkmk_comm *comm);
int tkmk_comm_size(const tkmk_comm *comm);
const char *tkmk_dist_last_error(void);
/* One line of JSON text about THIS rank's side of the communicator (at most cap - 1 bytes, 512 is enough): transport, size / rank as held
 * by the communicator, ncclCommCount / ncclCommUserRank of the RCCL communicator underneath (-1 over the loopback), the RCCL version,
 * and the rank's device (HIP index, PCI bus id, UUID).  Gathered over the communicator itself (tkmk_comm_all_gather_host) the lines
 * show N ranks on N distinct devices. */
tkmk_error tkmk_comm_describe(const tkmk_comm *comm, char *out, size_t cap);

/* ---- host values and agreement ---- */
/* recv (host, world_size x bytes) <- send (host, bytes) of every rank, rank order: the few values a sharded computation reduces on the
 * host (degrees, evaluation partials, column totals).  One all-gather through a device staging pair kept with the communicator. */
tkmk_error tkmk_comm_all_gather_host(tkmk_comm *comm, const void *send, size_t bytes, void *recv);
tkmk_error tkmk_comm_all_gather_dev(tkmk_comm *comm, const void *send_dev, size_t bytes, void *recv_dev);
/* recv_dev (bytes) of rank r <- send_dev (bytes) of rank (r - distance) mod world_size, device to device: one place-shift along the ring
 * (grouped ncclSend / ncclRecv).  The sharded prover's Y shifts: Y^s p moves every rank's columns to the rank s places on. */
tkmk_error tkmk_comm_ring_shift(tkmk_comm *comm, const void *send_dev, size_t bytes, int distance, void *recv_dev);
/* every rank passes the status of a local step; ALL ranks return success only if every rank passed success (otherwise an error naming
 * the failed rank).  The sharded entries below use it between their local step and their exchange, and tkmk_msm_sharded /
 * tkmk_msm_multi_ex_sharded carry the same status inside the gathered record (16 bytes behind the partial results): a share that
 * failed on one rank is an error on every rank, never a sum that silently lacks it. */
tkmk_error tkmk_comm_agree(tkmk_comm *comm, tkmk_error local_status);
/* A rank that leaves a sharded computation early calls this so that its peers fail instead of waiting: the loopback group's pending and
 * later rendezvous fail at once; an RCCL communicator is aborted (ncclCommAbort).  The communicator only accepts tkmk_comm_destroy afterwards. */
tkmk_error tkmk_comm_abort(tkmk_comm *comm);

/* ---- the sharded prover's two matrix layouts and the transforms between them (world_size a power of two) ----
 * COLS: rank r holds all rows of the columns iy = r mod G (row-major x_size x (y_size / G); local column k = global column r + G k) —
 *       coefficient matrices and witness-native evaluations;  ROWS: rank r holds rows [r h, (r + 1) h), h = x_size / G, of all columns —
 *       evaluations on the large domains.  One bivariate transform = a pass on the local axis, ONE all-to-all, a pass on the other axis. */
#define TKMK_DIST_SKIP_X_PASS 1
#define TKMK_DIST_SKIP_Y_PASS 2
/* forward transform of the zero-padded extension of an in_x x in_y COLS coefficient matrix to the x_size x y_size domain -> ROWS
 * evaluations (h x y_size), natural order (= tkmk_bintt_padded on the whole matrix, bit for bit).  flags: skip either pass; both = a
 * change of layout only.  in_y >= world_size. */
tkmk_error tkmk_dist_fwd_cols_to_rows(tkmk_comm *comm, const tkmk_fr *in_cols_dev, size_t in_x, size_t in_y, size_t x_size, size_t y_size, int flags,
                                      tkmk_fr *out_rows_dev);
/* inverse transform of x_size x y_size ROWS evaluations -> COLS coefficients (x_size x (y_size / G)); in_rows_dev is overwritten.
 * (= tkmk_bintt inverse on the whole matrix).  flags as above. */
tkmk_error tkmk_dist_inv_rows_to_cols(tkmk_comm *comm, tkmk_fr *in_rows_dev, size_t x_size, size_t y_size, int flags, tkmk_fr *out_cols_dev);
/* the change of layout alone for records of any size that is a multiple of 16 bytes (G1 affine points: 96): COLS x_size x (y_size / G)
 * <-> ROWS (x_size / G) x y_size, one all-to-all each — the group transforms behind the Lagrange-basis tables use them at open */
tkmk_error tkmk_dist_relayout_cols_to_rows(tkmk_comm *comm, const void *in_cols_dev, size_t x_size, size_t y_size, size_t record_bytes, void *out_rows_dev);
tkmk_error tkmk_dist_relayout_rows_to_cols(tkmk_comm *comm, const void *in_rows_dev, size_t x_size, size_t y_size, size_t record_bytes, void *out_cols_dev);
/* out (h x y_size) = this rank's ROWS slab of the matrix rotated down by rot <= h rows, cyclically over all G h rows: the evaluations of
 * p(w^-rot X, Y) from those of p.  One all-gather of rot rows per rank. */
tkmk_error tkmk_dist_rows_rotate(tkmk_comm *comm, const tkmk_fr *slab_dev, size_t h, size_t y_size, size_t rot, tkmk_fr *out_dev);

/* One MSM whose points are sharded over the ranks: this rank holds msm_size points (scalars / bases as in bls12_381_msm, host or
 * device per cfg; msm_size may be 0 on some ranks).  Every rank gets the full result (canonical projective, host).  Exchange: ONE
 * ncclAllGather of the 144-byte partial results (device to device), then every rank adds the world_size partials on the device. */
tkmk_error tkmk_msm_sharded(tkmk_comm *comm, const tkmk_fr *scalars, const tkmk_g1_affine *bases, int msm_size, const tkmk_msm_config *cfg,
                            tkmk_g1_projective *result);

/* A BATCH of MSMs over views of row-sharded resident tables — the commit batch of one prover round:
 * jobs / cfg / bases_form exactly as tkmk_msm_multi_ex, each rank describing ITS share of every job (its rows of the table, the
 * matching strided view of the replicated scalars; a job may be empty on a rank).  Every rank runs its share through the pipelined
 * single-GPU entry, ONE ncclAllGather carries n_jobs x 144 bytes per rank, the world_size partials of every job are summed on the
 * device; every rank gets all n_jobs results (canonical projective, host). */
tkmk_error tkmk_msm_multi_ex_sharded(tkmk_comm *comm, const tkmk_msm_job_ex *jobs, int n_jobs, const tkmk_msm_config *cfg, int bases_form,
                                     tkmk_g1_projective *results);

/* One x_size x y_size bivariate NTT sharded over the ranks (both sizes multiples of world_size).
 *   in : this rank's x-slab, rows [rank * x_size / G, (rank + 1) * x_size / G) of the matrix (element (ix, iy) at ix * y_size + iy): device
 *   out: this rank's y-slab, ALL rows, columns [rank * y_size / G, (rank + 1) * y_size / G), row-major x_size x (y_size / G): device
 * Rows (length y_size, coset_y) are transformed locally, ONE ncclAllToAll moves block (rows of r) x (columns of q) to rank q,
 * columns (length x_size, coset_x) are transformed locally.  dir / cosets as tkmk_bintt.  in_slab_dev is overwritten. */
tkmk_error tkmk_bintt_sharded(tkmk_comm *comm, tkmk_fr *in_slab_dev, size_t x_size, size_t y_size, tkmk_ntt_dir dir, const tkmk_fr *coset_x,
                              const tkmk_fr *coset_y, tkmk_fr *out_slab_dev);

#ifdef __cplusplus
}
#endif
#endif
