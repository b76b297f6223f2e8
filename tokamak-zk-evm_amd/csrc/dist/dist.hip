// dist.hip — libtkmk_dist.so (include/tkmk_dist.h): the point-sharded MSM (one MSM, or a batch of MSMs over views of row-sharded
// tables) and the slab-sharded bivariate NTT, above the single-GPU C ABI of libtkmk_hip.so.  One process per GPU; collectives on
// device buffers over RCCL (xGMI inside a node).  The exchange steps go through a two-function transport (all_gather, all_to_all):
// RCCL in production; a LOOPBACK transport — world_size virtual ranks inside one process on one GPU, the collectives as
// device-to-device copies between the ranks' buffers behind a rendezvous — so that every line of the G >= 2 index algebra below
// (pack / place of the transpose, the gather + sum of the partial results, empty and infinite partials) runs under `-m gpu` on the
// one-GPU test box.  The entry points are the same functions for both.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/tkmk_dist.h"

#define TKD_API extern "C" __attribute__((visibility("default")))

// ---- loopback group: shared by the world_size communicators of one tkmk_comm_init_loopback call ----
struct loop_group {
    int world = 1;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void *> send;
    std::vector<void *> recv;
    // libtkmk_hip.so is called from one thread at a time (the reference issues all device calls from its main thread):
    // the virtual ranks — one host thread each — take turns on the device and give the turn up while they wait in a collective
    std::mutex device_turn;
    explicit loop_group(int w) : world(w), send(w, nullptr), recv(w, nullptr) {}
    bool broken = false;   // a rendezvous timed out: every later one fails at once instead of waiting again
    // false: a peer did not arrive within the limit (it failed before its collective, or never made the call) — the test transport
    // turns what would be a hang into an error on the ranks that did arrive
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return true;
        }
        if (!cv.wait_for(lk, std::chrono::seconds(rendezvous_timeout_s()), [&] { return generation != gen || broken; }) || broken) {
            broken = true;
            cv.notify_all();
            return false;
        }
        return true;
    }
    static int rendezvous_timeout_s() {   // TKMK_LOOPBACK_TIMEOUT_S (default 180), read at every rendezvous
        const char *e = getenv("TKMK_LOOPBACK_TIMEOUT_S");
        int x = e ? atoi(e) : 180;
        return x < 1 ? 1 : x;
    }
};

struct tkmk_comm {
    ncclComm_t nccl = nullptr;            // RCCL transport
    std::shared_ptr<loop_group> loop;     // loopback transport
    int world = 1, rank = 0;
    hipStream_t stream = nullptr;
    bool turn_held_by_caller = false;     // loopback: the calling thread took the device turn itself (tkmk_comm_device_turn)
};

static thread_local std::string g_err;
static tkmk_error fail(tkmk_error code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define TKD_NCCL(call)                                                                                          \
    do {                                                                                                        \
        ncclResult_t _r = (call);                                                                               \
        if (_r != ncclSuccess) return fail(TKMK_ERR_UNKNOWN, std::string(#call) + ": " + ncclGetErrorString(_r)); \
    } while (0)
#define TKD_HIP(call)                                                                                          \
    do {                                                                                                       \
        hipError_t _e = (call);                                                                                \
        if (_e != hipSuccess) return fail(TKMK_ERR_UNKNOWN, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)
#define TKD_TRY(call)                                                                                  \
    do {                                                                                               \
        tkmk_error _t = (call);                                                                        \
        if (_t != TKMK_SUCCESS) return fail(_t, std::string(#call) + ": " + tkmk_error_string(_t));    \
    } while (0)

// the calling rank's turn on the device (no-op for RCCL: one process per GPU)
struct device_turn {
    loop_group *g;
    bool held = false, callers = false;
    explicit device_turn(tkmk_comm *c) : g(c->loop.get()), callers(c->turn_held_by_caller) {
        if (callers) held = true;   // a caller that spans several entries (the sharded prover) already holds it
        else take();
    }
    ~device_turn() {
        if (callers) take();   // hand it back the way it came
        else give();
    }
    void take() {
        if (g && !held) g->device_turn.lock(), held = true;
    }
    void give() {
        if (g && held) g->device_turn.unlock(), held = false;
    }
};

// ---- transport: the two collectives the entries use.  Buffers are device memory; both return with the data in place. ----
// all_gather: recv[q * bytes ..] of every rank = send of rank q.
static tkmk_error transport_all_gather(tkmk_comm *c, device_turn &turn, const void *send, void *recv, size_t bytes) {
    if (c->loop) {
        loop_group &g = *c->loop;
        TKD_HIP(hipDeviceSynchronize());   // this rank's send buffer is complete before a peer reads it
        turn.give();
        g.send[c->rank] = send;
        bool met = g.barrier();             // every rank has published
        hipError_t e = hipSuccess;
        for (int q = 0; met && q < g.world && e == hipSuccess; q++)
            e = hipMemcpy((uint8_t *)recv + (size_t)q * bytes, g.send[q], bytes, hipMemcpyDeviceToDevice);
        met = met && g.barrier();           // every rank has read: the send buffers may be reused
        turn.take();
        if (!met) return fail(TKMK_ERR_UNKNOWN, "loopback all_gather: a peer rank did not reach the collective");
        TKD_HIP(e);
        return TKMK_SUCCESS;
    }
    TKD_NCCL(ncclAllGather(send, recv, bytes, ncclUint8, c->nccl, c->stream));
    TKD_HIP(hipStreamSynchronize(c->stream));
    return TKMK_SUCCESS;
}
// all_to_all: recv[q * bytes ..] of rank r = send[r * bytes ..] of rank q.
static tkmk_error transport_all_to_all(tkmk_comm *c, device_turn &turn, const void *send, void *recv, size_t bytes) {
    if (c->loop) {
        loop_group &g = *c->loop;
        TKD_HIP(hipDeviceSynchronize());
        turn.give();
        g.send[c->rank] = send;
        bool met = g.barrier();
        hipError_t e = hipSuccess;
        for (int q = 0; met && q < g.world && e == hipSuccess; q++)
            e = hipMemcpy((uint8_t *)recv + (size_t)q * bytes, (const uint8_t *)g.send[q] + (size_t)c->rank * bytes, bytes, hipMemcpyDeviceToDevice);
        met = met && g.barrier();
        turn.take();
        if (!met) return fail(TKMK_ERR_UNKNOWN, "loopback all_to_all: a peer rank did not reach the collective");
        TKD_HIP(e);
        return TKMK_SUCCESS;
    }
    TKD_NCCL(ncclAllToAll(send, recv, bytes, ncclUint8, c->nccl, c->stream));
    TKD_HIP(hipStreamSynchronize(c->stream));
    return TKMK_SUCCESS;
}

TKD_API const char *tkmk_dist_last_error(void) { return g_err.c_str(); }

TKD_API tkmk_error tkmk_comm_unique_id(uint8_t id[TKMK_COMM_ID_BYTES]) {
    if (!id) return TKMK_ERR_INVALID_POINTER;
    static_assert(sizeof(ncclUniqueId) == TKMK_COMM_ID_BYTES, "communicator id size");
    ncclUniqueId u;
    TKD_NCCL(ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return TKMK_SUCCESS;
}
TKD_API tkmk_error tkmk_comm_init(const uint8_t id[TKMK_COMM_ID_BYTES], int world_size, int rank, tkmk_comm **out) {
    if (!id || !out) return TKMK_ERR_INVALID_POINTER;
    if (world_size < 1 || rank < 0 || rank >= world_size) return TKMK_ERR_INVALID_ARGUMENT;
    int ndev = 0;
    if (tkmk_device_count(&ndev) != TKMK_SUCCESS || ndev < 1) return fail(TKMK_ERR_NO_DEVICE, "no HIP device");
    tkmk_comm *c = new tkmk_comm();
    c->world = world_size, c->rank = rank;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclResult_t r = ncclCommInitRank(&c->nccl, world_size, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(TKMK_ERR_UNKNOWN, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)ncclCommDestroy(c->nccl);
        delete c;
        return fail(TKMK_ERR_STREAM_CREATION_FAILED, "hipStreamCreate");
    }
    *out = c;
    return TKMK_SUCCESS;
}
TKD_API tkmk_error tkmk_comm_init_loopback(int world_size, tkmk_comm **out_comms) {
    if (!out_comms) return TKMK_ERR_INVALID_POINTER;
    if (world_size < 1 || world_size > 64) return TKMK_ERR_INVALID_ARGUMENT;
    int ndev = 0;
    if (tkmk_device_count(&ndev) != TKMK_SUCCESS || ndev < 1) return fail(TKMK_ERR_NO_DEVICE, "no HIP device");
    auto group = std::make_shared<loop_group>(world_size);
    for (int r = 0; r < world_size; r++) {
        tkmk_comm *c = new tkmk_comm();
        c->world = world_size, c->rank = r, c->loop = group;
        out_comms[r] = c;
    }
    return TKMK_SUCCESS;
}
TKD_API tkmk_error tkmk_comm_destroy(tkmk_comm *c) {
    if (!c) return TKMK_SUCCESS;
    if (c->nccl) {
        (void)hipStreamSynchronize(c->stream);
        (void)ncclCommDestroy(c->nccl);
        (void)hipStreamDestroy(c->stream);
    }
    delete c;
    return TKMK_SUCCESS;
}
TKD_API int tkmk_comm_rank(const tkmk_comm *c) { return c ? c->rank : -1; }
TKD_API int tkmk_comm_size(const tkmk_comm *c) { return c ? c->world : 0; }
TKD_API int tkmk_comm_is_loopback(const tkmk_comm *c) { return c && c->loop ? 1 : 0; }
TKD_API tkmk_error tkmk_comm_device_turn(tkmk_comm *c, int acquire) {
    if (!c) return TKMK_ERR_INVALID_POINTER;
    if (!c->loop) return TKMK_SUCCESS;   // one process per GPU: the device is this rank's alone
    if (acquire && !c->turn_held_by_caller) c->loop->device_turn.lock(), c->turn_held_by_caller = true;
    else if (!acquire && c->turn_held_by_caller) c->turn_held_by_caller = false, c->loop->device_turn.unlock();
    return TKMK_SUCCESS;
}
TKD_API tkmk_error tkmk_comm_broadcast_host(tkmk_comm *c, void *buf, size_t bytes, int root) {
    if (!c || (!buf && bytes)) return TKMK_ERR_INVALID_POINTER;
    if (root < 0 || root >= c->world) return TKMK_ERR_INVALID_ARGUMENT;
    if (bytes == 0 || c->world == 1) return TKMK_SUCCESS;
    device_turn turn(c);
    void *d_send = nullptr, *d_all = nullptr;
    TKD_TRY(tkmk_malloc(&d_send, bytes));
    tkmk_error e = tkmk_malloc(&d_all, bytes * (size_t)c->world);
    if (e == TKMK_SUCCESS) e = tkmk_memcpy_h2d(d_send, buf, bytes);
    if (e != TKMK_SUCCESS) {   // cannot take part: fail before a peer waits (allocation of a few bytes)
        (void)tkmk_free(d_send), (void)tkmk_free(d_all);
        return fail(e, "tkmk_comm_broadcast_host: staging");
    }
    e = transport_all_gather(c, turn, d_send, d_all, bytes);
    if (e == TKMK_SUCCESS) e = tkmk_memcpy_d2h(buf, (const uint8_t *)d_all + (size_t)root * bytes, bytes);
    (void)tkmk_free(d_send), (void)tkmk_free(d_all);
    return e;
}

// gathered[q][j] (canonical projective partials of rank q, job j: (x, y, 1) or (0, 1, 0)) -> affine[j][q] ((0, 0) = infinity): the
// operand layout of a batch of n_jobs MSMs with world points each
__global__ void k_partials_to_affine(const tkmk_g1_projective *__restrict__ gathered, tkmk_g1_affine *__restrict__ affine, uint32_t world,
                                     uint32_t n_jobs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= world * n_jobs) return;
    const uint32_t q = i / n_jobs, j = i - q * n_jobs;
    const tkmk_g1_projective p = gathered[i];
    bool inf = true;
    for (int l = 0; l < 12; l++) inf &= p.z.limbs[l] == 0;
    tkmk_g1_affine a;
    for (int l = 0; l < 12; l++) a.x.limbs[l] = inf ? 0u : p.x.limbs[l], a.y.limbs[l] = inf ? 0u : p.y.limbs[l];
    affine[(size_t)j * world + q] = a;
}

// d_part: this rank's n_jobs partial results on the device.  ONE all_gather of n_jobs * 144 bytes per rank; the world_size partials of
// every job are summed on the device (a batch of n_jobs unit-scalar MSMs of world_size points: one launch); results to the host.
static tkmk_error gather_and_sum(tkmk_comm *c, device_turn &turn, const void *d_part, int n_jobs, tkmk_stream stream, tkmk_g1_projective *results) {
    const size_t part_bytes = sizeof(tkmk_g1_projective) * (size_t)n_jobs;
    void *d_all = nullptr, *d_aff = nullptr;
    TKD_TRY(tkmk_malloc(&d_all, part_bytes * (size_t)c->world));
    tkmk_error e = tkmk_malloc(&d_aff, sizeof(tkmk_g1_affine) * (size_t)n_jobs * c->world);
    if (e != TKMK_SUCCESS) {
        (void)tkmk_free(d_all);
        return fail(e, "tkmk_malloc");
    }
    auto cleanup = [&] { (void)tkmk_free(d_all), (void)tkmk_free(d_aff); };
    e = transport_all_gather(c, turn, d_part, d_all, part_bytes);
    if (e != TKMK_SUCCESS) {
        cleanup();
        return e;
    }
    const uint32_t total = (uint32_t)c->world * (uint32_t)n_jobs;
    hipLaunchKernelGGL(k_partials_to_affine, (total + 127) / 128, 128, 0, 0, (const tkmk_g1_projective *)d_all, (tkmk_g1_affine *)d_aff, (uint32_t)c->world,
                       (uint32_t)n_jobs);
    hipError_t h = hipGetLastError();
    if (h == hipSuccess) h = hipDeviceSynchronize();
    if (h != hipSuccess) {
        cleanup();
        return fail(TKMK_ERR_UNKNOWN, std::string("k_partials_to_affine: ") + hipGetErrorString(h));
    }
    std::vector<tkmk_fr> ones((size_t)total);
    for (auto &o : ones) {
        std::memset(&o, 0, sizeof o);
        o.limbs[0] = 1;
    }
    tkmk_msm_config sum = tkmk_msm_default_config();
    sum.stream_handle = stream;
    sum.batch_size = n_jobs;
    sum.are_points_shared_in_batch = false;
    sum.are_points_on_device = true;
    e = bls12_381_msm(ones.data(), (const tkmk_g1_affine *)d_aff, c->world, &sum, results);
    cleanup();
    if (e != TKMK_SUCCESS) return fail(e, std::string("bls12_381_msm (sum of partials): ") + tkmk_error_string(e));
    return TKMK_SUCCESS;
}

TKD_API tkmk_error tkmk_msm_sharded(tkmk_comm *c, const tkmk_fr *scalars, const tkmk_g1_affine *bases, int msm_size, const tkmk_msm_config *cfg,
                                    tkmk_g1_projective *result) {
    if (!c || !cfg || !result) return TKMK_ERR_INVALID_POINTER;
    if (cfg->batch_size != 1) return TKMK_ERR_INVALID_ARGUMENT;
    device_turn turn(c);
    // 1. this rank's shard through the whole single-GPU pipeline; the partial result stays on the device
    void *d_part = nullptr;
    TKD_TRY(tkmk_malloc(&d_part, sizeof(tkmk_g1_projective)));
    tkmk_msm_config local = *cfg;
    local.are_results_on_device = true;
    local.is_async = false;
    tkmk_error e = bls12_381_msm(scalars, bases, msm_size, &local, (tkmk_g1_projective *)d_part);
    // 2. + 3. one all_gather of 144 bytes per rank, the partials summed on the device.  A rank whose own MSM failed still takes part in
    // the collective (with an infinite partial), so that its peers are not left waiting, and reports its error afterwards.
    if (e != TKMK_SUCCESS) (void)tkmk_memset(d_part, 0, sizeof(tkmk_g1_projective));
    tkmk_error e2 = gather_and_sum(c, turn, d_part, 1, cfg->stream_handle, result);
    (void)tkmk_free(d_part);
    if (e != TKMK_SUCCESS) return fail(e, std::string("bls12_381_msm: ") + tkmk_error_string(e));
    return e2;
}

TKD_API tkmk_error tkmk_msm_multi_ex_sharded(tkmk_comm *c, const tkmk_msm_job_ex *jobs, int n_jobs, const tkmk_msm_config *cfg, int bases_form,
                                             tkmk_g1_projective *results) {
    if (!c || !cfg) return TKMK_ERR_INVALID_POINTER;
    if (n_jobs < 0 || cfg->batch_size != 1) return TKMK_ERR_INVALID_ARGUMENT;
    if (n_jobs == 0) return TKMK_SUCCESS;
    if (!jobs || !results) return TKMK_ERR_INVALID_POINTER;
    device_turn turn(c);
    void *d_part = nullptr;
    TKD_TRY(tkmk_malloc(&d_part, sizeof(tkmk_g1_projective) * (size_t)n_jobs));
    tkmk_msm_config local = *cfg;
    local.are_results_on_device = true;
    local.is_async = false;
    tkmk_error e = tkmk_msm_multi_ex(jobs, n_jobs, &local, bases_form, (tkmk_g1_projective *)d_part);
    if (e != TKMK_SUCCESS) (void)tkmk_memset(d_part, 0, sizeof(tkmk_g1_projective) * (size_t)n_jobs);
    tkmk_error e2 = gather_and_sum(c, turn, d_part, n_jobs, cfg->stream_handle, results);
    (void)tkmk_free(d_part);
    if (e != TKMK_SUCCESS) return fail(e, std::string("tkmk_msm_multi_ex: ") + tkmk_error_string(e));
    return e2;
}

TKD_API tkmk_error tkmk_bintt_sharded(tkmk_comm *c, tkmk_fr *in_slab_dev, size_t x_size, size_t y_size, tkmk_ntt_dir dir, const tkmk_fr *coset_x,
                                      const tkmk_fr *coset_y, tkmk_fr *out_slab_dev) {
    if (!c || !in_slab_dev || !out_slab_dev) return TKMK_ERR_INVALID_POINTER;
    const size_t G = (size_t)c->world;
    if (x_size == 0 || y_size == 0 || x_size % G || y_size % G || (x_size & (x_size - 1)) || (y_size & (y_size - 1))) return TKMK_ERR_INVALID_ARGUMENT;
    device_turn turn(c);
    const size_t rows = x_size / G, cols = y_size / G;
    auto ntt_cfg = [&](const tkmk_fr *coset, size_t batch, bool columns) {
        tkmk_ntt_config n = tkmk_ntt_default_config();
        n.batch_size = (int)batch;
        n.columns_batch = columns;
        n.are_inputs_on_device = n.are_outputs_on_device = true;
        if (coset) n.coset_gen = *coset;
        return n;
    };
    // the send buffer first: a rank that cannot take part in the exchange must fail before any peer waits for it
    void *d_send = nullptr;
    TKD_TRY(tkmk_malloc(&d_send, rows * y_size * sizeof(tkmk_fr)));
    // 1. rows of this x-slab (length y_size, coset_y), in place
    tkmk_ntt_config rc = ntt_cfg(coset_y, rows, false);
    tkmk_error e = bls12_381_ntt(in_slab_dev, (int)y_size, dir, &rc, in_slab_dev);
    // 2. pack block (my rows) x (columns of q) contiguously for every q, then ONE all-to-all, device to device; the block from
    //    rank q (its rows, my columns) is a contiguous run of the y-slab (rows q * rows .. of width cols), so it is received in place
    for (size_t q = 0; q < G && e == TKMK_SUCCESS; q++)
        e = tkmk_memcpy_2d_d2d((uint8_t *)d_send + q * rows * cols * sizeof(tkmk_fr), cols * sizeof(tkmk_fr), (const uint8_t *)in_slab_dev + q * cols * sizeof(tkmk_fr),
                               y_size * sizeof(tkmk_fr), cols * sizeof(tkmk_fr), rows);
    if (e == TKMK_SUCCESS) e = tkmk_device_synchronize();   // the packing ran on the default stream; the collective runs on the communicator's
    // a rank whose local step failed still enters the exchange (its peers would wait for it otherwise) and reports afterwards
    tkmk_error e2 = transport_all_to_all(c, turn, d_send, out_slab_dev, rows * cols * sizeof(tkmk_fr));
    (void)tkmk_free(d_send);
    if (e != TKMK_SUCCESS) return fail(e, std::string("rows / pack: ") + tkmk_error_string(e));
    if (e2 != TKMK_SUCCESS) return e2;
    // 3. columns of the y-slab (length x_size, coset_x): x_size x cols matrix, element (ix, j) at ix * cols + j
    tkmk_ntt_config cc = ntt_cfg(coset_x, cols, cols > 1);
    TKD_TRY(bls12_381_ntt(out_slab_dev, (int)x_size, dir, &cc, out_slab_dev));
    return TKMK_SUCCESS;
}
