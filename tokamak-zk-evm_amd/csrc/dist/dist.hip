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
    // libtkmk_hip.so takes one host thread at a time PER STREAM (include/tkmk.h, THREADING) and the virtual ranks — one host thread each —
    // all issue on the default stream: they take turns on the device and give the turn up while they wait in a collective
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
    void *small_send = nullptr, *small_recv = nullptr;   // device staging of tkmk_comm_all_gather_host (SMALL_BYTES per rank), made on first use
    bool aborted = false;                 // tkmk_comm_abort: every later collective on this communicator fails at once
};
static const size_t SMALL_BYTES = 4096;

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
    if (c->aborted) return fail(TKMK_ERR_UNKNOWN, "communicator aborted");
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
    if (c->aborted) return fail(TKMK_ERR_UNKNOWN, "communicator aborted");
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
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->small_send) (void)tkmk_free(c->small_send);
    if (c->small_recv) (void)tkmk_free(c->small_recv);
    delete c;
    return TKMK_SUCCESS;
}
TKD_API int tkmk_comm_rank(const tkmk_comm *c) { return c ? c->rank : -1; }
TKD_API int tkmk_comm_size(const tkmk_comm *c) { return c ? c->world : 0; }
TKD_API int tkmk_comm_is_loopback(const tkmk_comm *c) { return c && c->loop ? 1 : 0; }
// What THIS rank's side of the communicator is, as one line of text (at most cap - 1 bytes): transport, size / rank as the communicator
// holds them, the RCCL version in use, and the device the rank runs on (PCI bus id + UUID).  A launcher gathers the lines over the
// communicator itself (tkmk_comm_all_gather_host) to show N ranks on N distinct devices (bench.py puts them into its JSON line).
TKD_API tkmk_error tkmk_comm_describe(const tkmk_comm *c, char *out, size_t cap) {
    if (!c || !out || cap < 2) return TKMK_ERR_INVALID_POINTER;
    int dev = -1, ver = 0;
    char bus[32] = "?";
    std::string uuid = "?";
    if (hipGetDevice(&dev) == hipSuccess) {
        (void)hipDeviceGetPCIBusId(bus, sizeof bus, dev);
        hipUUID u;
        if (hipDeviceGetUuid(&u, dev) == hipSuccess) {
            static const char *hx = "0123456789abcdef";
            uuid.clear();
            for (unsigned char b : u.bytes) uuid.push_back(hx[b >> 4]), uuid.push_back(hx[b & 15]);
        }
    }
    (void)ncclGetVersion(&ver);
    int nccl_count = -1, nccl_rank = -1;
    if (c->nccl) (void)ncclCommCount(c->nccl, &nccl_count), (void)ncclCommUserRank(c->nccl, &nccl_rank);
    std::string line = std::string("{\"transport\": \"") + (c->loop ? "loopback" : "rccl") + "\", \"size\": " + std::to_string(c->world) + ", \"rank\": " +
                       std::to_string(c->rank) + ", \"nccl_comm_count\": " + std::to_string(nccl_count) + ", \"nccl_comm_user_rank\": " + std::to_string(nccl_rank) +
                       ", \"rccl_version\": " + std::to_string(ver) + ", \"hip_device\": " + std::to_string(dev) + ", \"pci_bus_id\": \"" + bus + "\", \"uuid\": \"" + uuid + "\"}";
    if (line.size() + 1 > cap) return TKMK_ERR_INVALID_ARGUMENT;
    std::memcpy(out, line.c_str(), line.size() + 1);
    return TKMK_SUCCESS;
}
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

// host values of every rank to every rank: recv[q * bytes ..] = send of rank q.  Small payloads (degrees, evaluation partials, column
// totals, status words) go through a staging pair kept with the communicator; anything larger through a pair of its own.
static tkmk_error all_gather_host(tkmk_comm *c, device_turn &turn, const void *send, size_t bytes, void *recv) {
    if (c->aborted) return fail(TKMK_ERR_UNKNOWN, "communicator aborted");
    if (bytes == 0) return TKMK_SUCCESS;
    if (c->world == 1) {
        std::memcpy(recv, send, bytes);
        return TKMK_SUCCESS;
    }
    void *d_send = nullptr, *d_all = nullptr;
    const bool small = bytes <= SMALL_BYTES;
    if (small) {
        if (!c->small_send) {
            TKD_TRY(tkmk_malloc(&c->small_send, SMALL_BYTES));
            TKD_TRY(tkmk_malloc(&c->small_recv, SMALL_BYTES * (size_t)c->world));
        }
        d_send = c->small_send, d_all = c->small_recv;
    } else {
        TKD_TRY(tkmk_malloc(&d_send, bytes));
        tkmk_error e = tkmk_malloc(&d_all, bytes * (size_t)c->world);
        if (e != TKMK_SUCCESS) {
            (void)tkmk_free(d_send);
            return fail(e, "tkmk_malloc");
        }
    }
    tkmk_error e = tkmk_memcpy_h2d(d_send, send, bytes);
    if (e == TKMK_SUCCESS) e = transport_all_gather(c, turn, d_send, d_all, bytes);
    if (e == TKMK_SUCCESS) e = tkmk_memcpy_d2h(recv, d_all, bytes * (size_t)c->world);
    if (!small) (void)tkmk_free(d_send), (void)tkmk_free(d_all);
    return e;
}
// Every rank contributes the status of its local step; all ranks get the same verdict: success only if every rank succeeded.  This is what
// keeps a one-sided failure (a refused job, an allocation that failed on one GPU) from becoming a wrong sum on the ranks that did not fail.
static tkmk_error agree(tkmk_comm *c, device_turn &turn, tkmk_error mine, const char *what) {
    if (c->world == 1) return mine == TKMK_SUCCESS ? TKMK_SUCCESS : fail(mine, std::string(what) + ": " + tkmk_error_string(mine));
    int32_t st = (int32_t)mine;
    std::vector<int32_t> all((size_t)c->world);
    tkmk_error e = all_gather_host(c, turn, &st, sizeof st, all.data());
    if (e != TKMK_SUCCESS) return e;
    if (mine != TKMK_SUCCESS) return fail(mine, std::string(what) + ": " + tkmk_error_string(mine));
    for (int q = 0; q < c->world; q++)
        if (all[(size_t)q] != 0)
            return fail(TKMK_ERR_UNKNOWN, std::string(what) + ": rank " + std::to_string(q) + " failed (" + tkmk_error_string((tkmk_error)all[(size_t)q]) + "); no rank keeps a result");
    return TKMK_SUCCESS;
}
TKD_API tkmk_error tkmk_comm_all_gather_host(tkmk_comm *c, const void *send, size_t bytes, void *recv) {
    if (!c || ((!send || !recv) && bytes)) return TKMK_ERR_INVALID_POINTER;
    device_turn turn(c);
    return all_gather_host(c, turn, send, bytes, recv);
}
TKD_API tkmk_error tkmk_comm_agree(tkmk_comm *c, tkmk_error local_status) {
    if (!c) return TKMK_ERR_INVALID_POINTER;
    device_turn turn(c);
    return agree(c, turn, local_status, "tkmk_comm_agree");
}
TKD_API tkmk_error tkmk_comm_all_gather_dev(tkmk_comm *c, const void *send_dev, size_t bytes, void *recv_dev) {
    if (!c || ((!send_dev || !recv_dev) && bytes)) return TKMK_ERR_INVALID_POINTER;
    if (c->aborted) return fail(TKMK_ERR_UNKNOWN, "communicator aborted");
    if (bytes == 0) return TKMK_SUCCESS;
    device_turn turn(c);
    TKD_TRY(tkmk_device_synchronize());   // the caller's producers ran on its own streams
    return transport_all_gather(c, turn, send_dev, recv_dev, bytes);
}
// A rank that leaves a sharded computation early (an exception above the collectives) calls this so that its peers do not wait for it:
// loopback: every pending and later rendezvous of the group fails at once; RCCL: ncclCommAbort tears this rank's side down (its peers'
// pending collectives then end in an error once RCCL notices; they must not be left to a hang that only a watchdog would end).
TKD_API tkmk_error tkmk_comm_abort(tkmk_comm *c) {
    if (!c) return TKMK_ERR_INVALID_POINTER;
    c->aborted = true;
    if (c->loop) {
        std::lock_guard<std::mutex> lk(c->loop->mu);
        c->loop->broken = true;
        c->loop->cv.notify_all();
    } else if (c->nccl) {
        (void)ncclCommAbort(c->nccl);
        c->nccl = nullptr;
    }
    return TKMK_SUCCESS;
}

// recv (device, bytes) of rank r <- send (device, bytes) of rank (r - distance) mod world: every rank passes its block `distance` places
// along the ring (the sharded prover's Y shifts: column j of a COLS matrix belongs to rank j mod G, so Y^s p moves every rank's columns
// to the rank s places on).  RCCL: one grouped send + receive per rank; loopback: a copy out of the source rank's buffer.
TKD_API tkmk_error tkmk_comm_ring_shift(tkmk_comm *c, const void *send_dev, size_t bytes, int distance, void *recv_dev) {
    if (!c || ((!send_dev || !recv_dev) && bytes)) return TKMK_ERR_INVALID_POINTER;
    if (distance < 0 || send_dev == recv_dev) return TKMK_ERR_INVALID_ARGUMENT;
    if (c->aborted) return fail(TKMK_ERR_UNKNOWN, "communicator aborted");
    if (bytes == 0) return TKMK_SUCCESS;
    const int W = c->world, d = distance % W;
    if (d == 0) return tkmk_memcpy_d2d(recv_dev, send_dev, bytes);
    device_turn turn(c);
    const int to = (c->rank + d) % W, from = (c->rank - d + W) % W;
    if (c->loop) {
        loop_group &g = *c->loop;
        TKD_HIP(hipDeviceSynchronize());
        turn.give();
        g.send[c->rank] = send_dev;
        bool met = g.barrier();
        hipError_t e = hipSuccess;
        if (met) e = hipMemcpy(recv_dev, g.send[from], bytes, hipMemcpyDeviceToDevice);
        met = met && g.barrier();
        turn.take();
        if (!met) return fail(TKMK_ERR_UNKNOWN, "loopback ring_shift: a peer rank did not reach the collective");
        TKD_HIP(e);
        return TKMK_SUCCESS;
    }
    TKD_TRY(tkmk_device_synchronize());   // the caller's producer ran on its own stream
    TKD_NCCL(ncclGroupStart());
    ncclResult_t rs = ncclSend(send_dev, bytes, ncclUint8, to, c->nccl, c->stream);
    ncclResult_t rr = ncclRecv(recv_dev, bytes, ncclUint8, from, c->nccl, c->stream);
    TKD_NCCL(ncclGroupEnd());
    TKD_NCCL(rs);
    TKD_NCCL(rr);
    TKD_HIP(hipStreamSynchronize(c->stream));
    return TKMK_SUCCESS;
}

// the gathered record of one rank: its n_jobs partial results followed by ONE status block (16 bytes, first word = the rank's tkmk_error)
static inline size_t part_record_bytes(int n_jobs) { return sizeof(tkmk_g1_projective) * (size_t)n_jobs + 16; }

// d_part: this rank's record (part_record_bytes(n_jobs): n_jobs partial results, then the status block, already written) on the device.
// ONE all_gather of that record per rank; if ANY rank's status is non-zero every rank returns an error and no result; otherwise the
// world_size partials of every job are summed on the device (a batch of n_jobs unit-scalar MSMs of world_size points: one launch).
static tkmk_error gather_and_sum(tkmk_comm *c, device_turn &turn, const void *d_part, int n_jobs, tkmk_stream stream, tkmk_g1_projective *results) {
    const size_t rec = part_record_bytes(n_jobs), part_bytes = sizeof(tkmk_g1_projective) * (size_t)n_jobs;
    void *d_all = nullptr, *d_aff = nullptr, *d_packed = nullptr;
    TKD_TRY(tkmk_malloc(&d_all, rec * (size_t)c->world));
    tkmk_error e = tkmk_malloc(&d_aff, sizeof(tkmk_g1_affine) * (size_t)n_jobs * c->world);
    if (e == TKMK_SUCCESS) e = tkmk_malloc(&d_packed, part_bytes * (size_t)c->world);
    auto cleanup = [&] { (void)tkmk_free(d_all), (void)tkmk_free(d_aff), (void)tkmk_free(d_packed); };
    // (an allocation of a few KB that fails here fails before the collective: this rank's peers would wait — it is reported, and the
    // caller's tkmk_comm_abort is what releases them)
    if (e != TKMK_SUCCESS) {
        cleanup();
        return fail(e, "tkmk_malloc");
    }
    e = transport_all_gather(c, turn, d_part, d_all, rec);
    if (e != TKMK_SUCCESS) {
        cleanup();
        return e;
    }
    // the status words of all ranks (world x 4 bytes out of the gathered records), then the partials packed back to back
    std::vector<int32_t> status((size_t)c->world);
    e = tkmk_memcpy_2d_d2d(d_packed, 4, (const uint8_t *)d_all + part_bytes, rec, 4, (size_t)c->world);
    if (e == TKMK_SUCCESS) e = tkmk_memcpy_d2h(status.data(), d_packed, 4 * (size_t)c->world);
    if (e == TKMK_SUCCESS) e = tkmk_memcpy_2d_d2d(d_packed, part_bytes, d_all, rec, part_bytes, (size_t)c->world);
    if (e != TKMK_SUCCESS) {
        cleanup();
        return fail(e, "gather_and_sum: unpack");
    }
    for (int q = 0; q < c->world; q++)
        if (status[(size_t)q] != 0) {
            cleanup();
            return fail(q == c->rank ? (tkmk_error)status[(size_t)q] : TKMK_ERR_UNKNOWN,
                        "sharded MSM: rank " + std::to_string(q) + " failed its share (" + tkmk_error_string((tkmk_error)status[(size_t)q]) + "); no rank keeps a sum");
        }
    const uint32_t total = (uint32_t)c->world * (uint32_t)n_jobs;
    hipLaunchKernelGGL(k_partials_to_affine, (total + 127) / 128, 128, 0, 0, (const tkmk_g1_projective *)d_packed, (tkmk_g1_affine *)d_aff, (uint32_t)c->world,
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
// writes the status block behind the partial results of a record (and clears the partials of a failed share: never a half-written sum)
static tkmk_error seal_record(void *d_part, int n_jobs, tkmk_error status) {
    const size_t part_bytes = sizeof(tkmk_g1_projective) * (size_t)n_jobs;
    int32_t block[4] = {(int32_t)status, 0, 0, 0};
    if (status != TKMK_SUCCESS) (void)tkmk_memset(d_part, 0, part_bytes);
    return tkmk_memcpy_h2d((uint8_t *)d_part + part_bytes, block, sizeof block);
}

TKD_API tkmk_error tkmk_msm_sharded(tkmk_comm *c, const tkmk_fr *scalars, const tkmk_g1_affine *bases, int msm_size, const tkmk_msm_config *cfg,
                                    tkmk_g1_projective *result) {
    if (!c || !cfg || !result) return TKMK_ERR_INVALID_POINTER;
    if (cfg->batch_size != 1) return TKMK_ERR_INVALID_ARGUMENT;
    if (c->aborted) return fail(TKMK_ERR_UNKNOWN, "communicator aborted");
    device_turn turn(c);
    // 1. this rank's shard through the whole single-GPU pipeline; the partial result stays on the device
    void *d_part = nullptr;
    TKD_TRY(tkmk_malloc(&d_part, part_record_bytes(1)));
    tkmk_msm_config local = *cfg;
    local.are_results_on_device = true;
    local.is_async = false;
    tkmk_error e = bls12_381_msm(scalars, bases, msm_size, &local, (tkmk_g1_projective *)d_part);
    // 2. + 3. one all_gather of 144 + 16 bytes per rank.  A rank whose own MSM failed still takes part in the collective, with its error
    // in the status block: EVERY rank then returns an error (a sum that silently lacks a share is never returned).
    tkmk_error es = seal_record(d_part, 1, e);
    if (es != TKMK_SUCCESS && e == TKMK_SUCCESS) e = es;
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
    if (c->aborted) return fail(TKMK_ERR_UNKNOWN, "communicator aborted");
    device_turn turn(c);
    void *d_part = nullptr;
    TKD_TRY(tkmk_malloc(&d_part, part_record_bytes(n_jobs)));
    tkmk_msm_config local = *cfg;
    local.are_results_on_device = true;
    local.is_async = false;
    tkmk_error e = tkmk_msm_multi_ex(jobs, n_jobs, &local, bases_form, (tkmk_g1_projective *)d_part);
    tkmk_error es = seal_record(d_part, n_jobs, e);
    if (es != TKMK_SUCCESS && e == TKMK_SUCCESS) e = es;
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
    // the ranks agree on the local step first: if one failed, none exchanges a slab (a garbage slab is never transformed further)
    tkmk_error ea = agree(c, turn, e, "tkmk_bintt_sharded: rows / pack");
    if (ea != TKMK_SUCCESS) {
        (void)tkmk_free(d_send);
        return ea;
    }
    tkmk_error e2 = transport_all_to_all(c, turn, d_send, out_slab_dev, rows * cols * sizeof(tkmk_fr));
    (void)tkmk_free(d_send);
    if (e2 != TKMK_SUCCESS) return e2;
    // 3. columns of the y-slab (length x_size, coset_x): x_size x cols matrix, element (ix, j) at ix * cols + j
    tkmk_ntt_config cc = ntt_cfg(coset_x, cols, cols > 1);
    TKD_TRY(bls12_381_ntt(out_slab_dev, (int)x_size, dir, &cc, out_slab_dev));
    return TKMK_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The two layouts of a sharded prover's matrices (host/tkmk_host.hpp DistCtx; world_size G a power of two):
//   COLS  column-interleaved: rank r holds ALL rows of the columns iy = r mod G, row-major x_size x (y_size / G); local column k is
//         global column r + G k.  Coefficient matrices live in this layout: every X-direction recurrence of the prover (div_by_ruffini's
//         Horner scan, the K0 window sums, X shifts, both passes of div_by_vanishing_opt whose Y stride s_max is a multiple of G) and the
//         commit against a column-interleaved table are then local.  Witness-native evaluations (one column per placement) too.
//   ROWS  contiguous row slabs: rank r holds rows [r h, (r + 1) h), h = x_size / G, of all columns.  Evaluations on the large domains
//         live in this layout (pointwise work; a root shift is a rotation by a few rows = a halo from the previous rank).
// A bivariate transform crosses from one to the other with ONE all-to-all:
//   forward  COLS coefficients -> X pass on local columns -> all-to-all -> Y pass on local rows -> ROWS evaluations
//   inverse  ROWS evaluations  -> Y pass on local rows    -> all-to-all -> X pass on local columns -> COLS coefficients
// `flags` switch either pass off (TKMK_DIST_SKIP_X_PASS | TKMK_DIST_SKIP_Y_PASS = a pure change of layout).
// ---------------------------------------------------------------------------------------------------------------------------------
typedef uint4 fr_half;   // a field element travels as two 16-byte halves

// out (h x Y, ROWS) from the received blocks: block p = rank p's (h x lyb) piece of my rows; element (i, j): j < in_y ? B[j % G][i][j / G] : 0.
// An element is a record of `hv` 16-byte pieces (2: a field element; 6: a G1 affine point).
__global__ void k_unpack_cols_to_rows(const fr_half *__restrict__ blocks, fr_half *__restrict__ out, uint32_t h, uint32_t Y, uint32_t in_y, uint32_t G,
                                      uint32_t lyb, uint32_t hv) {
    const uint64_t total = (uint64_t)h * Y * hv;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t e = t / hv;
        const uint32_t piece = (uint32_t)(t - e * hv);
        const uint32_t i = (uint32_t)(e / Y), j = (uint32_t)(e - (uint64_t)i * Y);
        fr_half v = make_uint4(0, 0, 0, 0);
        if (j < in_y) v = blocks[(((uint64_t)(j % G) * h + i) * lyb + j / G) * hv + piece];
        out[t] = v;
    }
}
// send blocks from a ROWS slab (h x Y): block p = (h x ly) piece with my rows and the columns p + G k
__global__ void k_pack_rows_to_cols(const fr_half *__restrict__ slab, fr_half *__restrict__ blocks, uint32_t h, uint32_t Y, uint32_t G, uint32_t ly, uint32_t hv) {
    const uint64_t total = (uint64_t)h * Y * hv;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        // t enumerates the DESTINATION (p, i, k, piece) so that stores are contiguous; loads stride by G records
        const uint64_t e = t / hv;
        const uint32_t piece = (uint32_t)(t - e * hv);
        const uint32_t p = (uint32_t)(e / ((uint64_t)h * ly));
        const uint64_t rem = e - (uint64_t)p * h * ly;
        const uint32_t i = (uint32_t)(rem / ly), k = (uint32_t)(rem - (uint64_t)i * ly);
        blocks[t] = slab[((uint64_t)i * Y + p + (uint64_t)G * k) * hv + piece];
    }
}
static unsigned copy_grid(uint64_t total) {
    uint64_t g = (total + 255) / 256;
    return (unsigned)(g > 256 * 16 ? 256 * 16 : (g ? g : 1));
}
static bool pow2(size_t v) { return v && !(v & (v - 1)); }

TKD_API tkmk_error tkmk_dist_fwd_cols_to_rows(tkmk_comm *c, const tkmk_fr *in_cols_dev, size_t in_x, size_t in_y, size_t x_size, size_t y_size, int flags,
                                              tkmk_fr *out_rows_dev) {
    if (!c || !in_cols_dev || !out_rows_dev) return TKMK_ERR_INVALID_POINTER;
    const size_t G = (size_t)c->world;
    if (!pow2(G) || !pow2(in_x) || !pow2(in_y) || !pow2(x_size) || !pow2(y_size) || in_x > x_size || in_y > y_size || in_y < G || x_size < G ||
        (flags & ~(TKMK_DIST_SKIP_X_PASS | TKMK_DIST_SKIP_Y_PASS)))
        return TKMK_ERR_INVALID_ARGUMENT;
    device_turn turn(c);
    const size_t lyb = in_y / G, h = x_size / G;
    // 1. X pass over the local columns, zero-padded from in_x to x_size rows
    void *d_a = nullptr, *d_b = nullptr;
    const tkmk_fr *a = in_cols_dev;
    tkmk_error e = TKMK_SUCCESS;
    const bool x_pass = !(flags & TKMK_DIST_SKIP_X_PASS);
    if (in_x < x_size || x_pass) {
        e = tkmk_malloc(&d_a, x_size * lyb * sizeof(tkmk_fr));
        if (e == TKMK_SUCCESS && in_x < x_size) e = tkmk_memset((uint8_t *)d_a + in_x * lyb * sizeof(tkmk_fr), 0, (x_size - in_x) * lyb * sizeof(tkmk_fr));
        if (e == TKMK_SUCCESS) e = tkmk_memcpy_d2d(d_a, in_cols_dev, in_x * lyb * sizeof(tkmk_fr));
        if (e == TKMK_SUCCESS && x_pass) {
            tkmk_ntt_config n = tkmk_ntt_default_config();
            n.batch_size = (int)lyb, n.columns_batch = lyb > 1, n.are_inputs_on_device = n.are_outputs_on_device = true;
            e = bls12_381_ntt((const tkmk_fr *)d_a, (int)x_size, TKMK_NTT_FORWARD, &n, (tkmk_fr *)d_a);
        }
        a = (const tkmk_fr *)d_a;
    }
    if (e == TKMK_SUCCESS) e = tkmk_malloc(&d_b, x_size * lyb * sizeof(tkmk_fr));
    if (e == TKMK_SUCCESS) e = tkmk_device_synchronize();
    // 2. ONE all-to-all: rows [q h, (q + 1) h) of the local matrix are a contiguous block for rank q
    tkmk_error ea = agree(c, turn, e, "tkmk_dist_fwd_cols_to_rows: X pass");
    if (ea == TKMK_SUCCESS) ea = transport_all_to_all(c, turn, a, d_b, h * lyb * sizeof(tkmk_fr));
    if (ea != TKMK_SUCCESS) {
        (void)tkmk_free(d_a), (void)tkmk_free(d_b);
        return ea;
    }
    // 3. interleave the received column sets into whole rows (columns past in_y are the zero padding), 4. Y pass over the local rows
    hipLaunchKernelGGL(k_unpack_cols_to_rows, copy_grid((uint64_t)h * y_size * 2), 256, 0, 0, (const fr_half *)d_b, (fr_half *)out_rows_dev, (uint32_t)h, (uint32_t)y_size,
                       (uint32_t)in_y, (uint32_t)G, (uint32_t)lyb, 2u);
    hipError_t he = hipGetLastError();
    if (he == hipSuccess) he = hipDeviceSynchronize();
    (void)tkmk_free(d_a), (void)tkmk_free(d_b);
    if (he != hipSuccess) return fail(TKMK_ERR_UNKNOWN, std::string("k_unpack_cols_to_rows: ") + hipGetErrorString(he));
    if (!(flags & TKMK_DIST_SKIP_Y_PASS)) {
        tkmk_ntt_config n = tkmk_ntt_default_config();
        n.batch_size = (int)h, n.are_inputs_on_device = n.are_outputs_on_device = true;
        TKD_TRY(bls12_381_ntt(out_rows_dev, (int)y_size, TKMK_NTT_FORWARD, &n, out_rows_dev));
    }
    return TKMK_SUCCESS;
}

TKD_API tkmk_error tkmk_dist_inv_rows_to_cols(tkmk_comm *c, tkmk_fr *in_rows_dev, size_t x_size, size_t y_size, int flags, tkmk_fr *out_cols_dev) {
    if (!c || !in_rows_dev || !out_cols_dev) return TKMK_ERR_INVALID_POINTER;
    const size_t G = (size_t)c->world;
    if (!pow2(G) || !pow2(x_size) || !pow2(y_size) || y_size < G || x_size < G || (flags & ~(TKMK_DIST_SKIP_X_PASS | TKMK_DIST_SKIP_Y_PASS)))
        return TKMK_ERR_INVALID_ARGUMENT;
    device_turn turn(c);
    const size_t ly = y_size / G, h = x_size / G;
    void *d_send = nullptr;
    tkmk_error e = tkmk_malloc(&d_send, h * y_size * sizeof(tkmk_fr));
    // 1. Y pass over the local rows (in place: the evaluations are consumed)
    if (e == TKMK_SUCCESS && !(flags & TKMK_DIST_SKIP_Y_PASS)) {
        tkmk_ntt_config n = tkmk_ntt_default_config();
        n.batch_size = (int)h, n.are_inputs_on_device = n.are_outputs_on_device = true;
        e = bls12_381_ntt(in_rows_dev, (int)y_size, TKMK_NTT_INVERSE, &n, in_rows_dev);
    }
    if (e == TKMK_SUCCESS) e = tkmk_device_synchronize();
    // 2. pack the column set of every destination, ONE all-to-all; the block from rank q is rows [q h, (q + 1) h) of my columns: in place
    if (e == TKMK_SUCCESS) {
        hipLaunchKernelGGL(k_pack_rows_to_cols, copy_grid((uint64_t)h * y_size * 2), 256, 0, 0, (const fr_half *)in_rows_dev, (fr_half *)d_send, (uint32_t)h, (uint32_t)y_size,
                           (uint32_t)G, (uint32_t)ly, 2u);
        hipError_t he = hipGetLastError();
        if (he == hipSuccess) he = hipDeviceSynchronize();
        if (he != hipSuccess) e = TKMK_ERR_UNKNOWN;
    }
    tkmk_error ea = agree(c, turn, e, "tkmk_dist_inv_rows_to_cols: Y pass / pack");
    if (ea == TKMK_SUCCESS) ea = transport_all_to_all(c, turn, d_send, out_cols_dev, h * ly * sizeof(tkmk_fr));
    (void)tkmk_free(d_send);
    if (ea != TKMK_SUCCESS) return ea;
    // 3. X pass over the local columns
    if (!(flags & TKMK_DIST_SKIP_X_PASS)) {
        tkmk_ntt_config n = tkmk_ntt_default_config();
        n.batch_size = (int)ly, n.columns_batch = ly > 1, n.are_inputs_on_device = n.are_outputs_on_device = true;
        TKD_TRY(bls12_381_ntt(out_cols_dev, (int)x_size, TKMK_NTT_INVERSE, &n, out_cols_dev));
    }
    return TKMK_SUCCESS;
}

// out (h x y_size) = this rank's slab of the matrix ROTATED down by `rot` rows (cyclically over all x_size = G h rows): out row i is
// global row (rank h + i - rot) mod x_size — the evaluation-domain form of p(w^-rot X, Y) on a ROWS slab.  rot <= h: the first rot rows
// come from the end of the previous rank's slab (ONE all-gather of rot rows per rank), the rest is a shifted copy of the own slab.
TKD_API tkmk_error tkmk_dist_rows_rotate(tkmk_comm *c, const tkmk_fr *slab_dev, size_t h, size_t y_size, size_t rot, tkmk_fr *out_dev) {
    if (!c || !slab_dev || !out_dev) return TKMK_ERR_INVALID_POINTER;
    if (!h || !y_size || rot > h || slab_dev == out_dev) return TKMK_ERR_INVALID_ARGUMENT;
    device_turn turn(c);
    const size_t row = y_size * sizeof(tkmk_fr);
    if (rot == 0) return tkmk_memcpy_d2d(out_dev, slab_dev, h * row);
    void *d_all = nullptr;
    tkmk_error e = tkmk_malloc(&d_all, (size_t)c->world * rot * row);
    if (e == TKMK_SUCCESS) e = tkmk_device_synchronize();
    tkmk_error ea = agree(c, turn, e, "tkmk_dist_rows_rotate");
    if (ea == TKMK_SUCCESS) ea = transport_all_gather(c, turn, (const uint8_t *)slab_dev + (h - rot) * row, d_all, rot * row);
    if (ea == TKMK_SUCCESS) {
        const size_t prev = (size_t)(c->rank + c->world - 1) % (size_t)c->world;
        ea = tkmk_memcpy_d2d(out_dev, (const uint8_t *)d_all + prev * rot * row, rot * row);
        if (ea == TKMK_SUCCESS && h > rot) ea = tkmk_memcpy_d2d((uint8_t *)out_dev + rot * row, slab_dev, (h - rot) * row);
    }
    (void)tkmk_free(d_all);
    return ea;
}

// The same change of layout for records of any size that is a multiple of 16 bytes (a G1 affine point: 96) — what the group transforms
// behind the Lagrange-basis commit tables move at open (host/tkmk_service.hpp): COLS x_size x (y_size / G) -> ROWS (x_size / G) x y_size
// and back, one all-to-all each, no transform.
TKD_API tkmk_error tkmk_dist_relayout_cols_to_rows(tkmk_comm *c, const void *in_cols_dev, size_t x_size, size_t y_size, size_t record_bytes, void *out_rows_dev) {
    if (!c || !in_cols_dev || !out_rows_dev) return TKMK_ERR_INVALID_POINTER;
    const size_t G = (size_t)c->world;
    if (!pow2(G) || !pow2(x_size) || !pow2(y_size) || x_size < G || y_size < G || !record_bytes || record_bytes % 16) return TKMK_ERR_INVALID_ARGUMENT;
    device_turn turn(c);
    const size_t ly = y_size / G, h = x_size / G, hv = record_bytes / 16;
    void *d_b = nullptr;
    tkmk_error e = tkmk_malloc(&d_b, x_size * ly * record_bytes);
    if (e == TKMK_SUCCESS) e = tkmk_device_synchronize();
    tkmk_error ea = agree(c, turn, e, "tkmk_dist_relayout_cols_to_rows");
    if (ea == TKMK_SUCCESS) ea = transport_all_to_all(c, turn, in_cols_dev, d_b, h * ly * record_bytes);
    if (ea != TKMK_SUCCESS) {
        (void)tkmk_free(d_b);
        return ea;
    }
    hipLaunchKernelGGL(k_unpack_cols_to_rows, copy_grid((uint64_t)h * y_size * hv), 256, 0, 0, (const fr_half *)d_b, (fr_half *)out_rows_dev, (uint32_t)h, (uint32_t)y_size,
                       (uint32_t)y_size, (uint32_t)G, (uint32_t)ly, (uint32_t)hv);
    hipError_t he = hipGetLastError();
    if (he == hipSuccess) he = hipDeviceSynchronize();
    (void)tkmk_free(d_b);
    if (he != hipSuccess) return fail(TKMK_ERR_UNKNOWN, std::string("k_unpack_cols_to_rows: ") + hipGetErrorString(he));
    return TKMK_SUCCESS;
}
TKD_API tkmk_error tkmk_dist_relayout_rows_to_cols(tkmk_comm *c, const void *in_rows_dev, size_t x_size, size_t y_size, size_t record_bytes, void *out_cols_dev) {
    if (!c || !in_rows_dev || !out_cols_dev) return TKMK_ERR_INVALID_POINTER;
    const size_t G = (size_t)c->world;
    if (!pow2(G) || !pow2(x_size) || !pow2(y_size) || x_size < G || y_size < G || !record_bytes || record_bytes % 16) return TKMK_ERR_INVALID_ARGUMENT;
    device_turn turn(c);
    const size_t ly = y_size / G, h = x_size / G, hv = record_bytes / 16;
    void *d_send = nullptr;
    tkmk_error e = tkmk_malloc(&d_send, h * y_size * record_bytes);
    if (e == TKMK_SUCCESS) {
        hipLaunchKernelGGL(k_pack_rows_to_cols, copy_grid((uint64_t)h * y_size * hv), 256, 0, 0, (const fr_half *)in_rows_dev, (fr_half *)d_send, (uint32_t)h, (uint32_t)y_size,
                           (uint32_t)G, (uint32_t)ly, (uint32_t)hv);
        hipError_t he = hipGetLastError();
        if (he == hipSuccess) he = hipDeviceSynchronize();
        if (he != hipSuccess) e = TKMK_ERR_UNKNOWN;
    }
    tkmk_error ea = agree(c, turn, e, "tkmk_dist_relayout_rows_to_cols");
    if (ea == TKMK_SUCCESS) ea = transport_all_to_all(c, turn, d_send, out_cols_dev, h * ly * record_bytes);
    (void)tkmk_free(d_send);
    return ea;
}
