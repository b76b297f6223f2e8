// dist.hip — libtkmk_dist.so (include/tkmk_dist.h): the point-sharded MSM and the slab-sharded bivariate NTT over RCCL, above the
// single-GPU C ABI of libtkmk_hip.so.  One process per GPU; collectives on device buffers (xGMI inside a node).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../../include/tkmk_dist.h"

#define TKD_API extern "C" __attribute__((visibility("default")))

struct tkmk_comm {
    ncclComm_t nccl = nullptr;
    int world = 1, rank = 0;
    hipStream_t stream = nullptr;
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
TKD_API tkmk_error tkmk_comm_destroy(tkmk_comm *c) {
    if (!c) return TKMK_SUCCESS;
    (void)hipStreamSynchronize(c->stream);
    (void)ncclCommDestroy(c->nccl);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return TKMK_SUCCESS;
}
TKD_API int tkmk_comm_rank(const tkmk_comm *c) { return c ? c->rank : -1; }
TKD_API int tkmk_comm_size(const tkmk_comm *c) { return c ? c->world : 0; }

TKD_API tkmk_error tkmk_msm_sharded(tkmk_comm *c, const tkmk_fr *scalars, const tkmk_g1_affine *bases, int msm_size, const tkmk_msm_config *cfg,
                                    tkmk_g1_projective *result) {
    if (!c || !cfg || !result) return TKMK_ERR_INVALID_POINTER;
    if (cfg->batch_size != 1) return TKMK_ERR_INVALID_ARGUMENT;
    // 1. this rank's shard through the whole single-GPU pipeline; the partial result stays on the device
    void *d_part = nullptr, *d_all = nullptr;
    TKD_TRY(tkmk_malloc(&d_part, sizeof(tkmk_g1_projective)));
    tkmk_error e = tkmk_malloc(&d_all, sizeof(tkmk_g1_projective) * (size_t)c->world);
    if (e != TKMK_SUCCESS) {
        (void)tkmk_free(d_part);
        return fail(e, "tkmk_malloc");
    }
    auto cleanup = [&] { (void)tkmk_free(d_part), (void)tkmk_free(d_all); };
    tkmk_msm_config local = *cfg;
    local.are_results_on_device = true;
    local.is_async = false;
    e = bls12_381_msm(scalars, bases, msm_size, &local, (tkmk_g1_projective *)d_part);
    if (e != TKMK_SUCCESS) {
        cleanup();
        return fail(e, std::string("bls12_381_msm: ") + tkmk_error_string(e));
    }
    // 2. ONE all_gather of 144 bytes per rank, device to device
    ncclResult_t r = ncclAllGather(d_part, d_all, sizeof(tkmk_g1_projective), ncclUint8, c->nccl, c->stream);
    hipError_t h = r == ncclSuccess ? hipStreamSynchronize(c->stream) : hipSuccess;
    if (r != ncclSuccess || h != hipSuccess) {
        cleanup();
        return fail(TKMK_ERR_UNKNOWN, r != ncclSuccess ? std::string("ncclAllGather: ") + ncclGetErrorString(r) : std::string("hipStreamSynchronize: ") + hipGetErrorString(h));
    }
    // 3. every rank adds the world_size partials (a world_size-point MSM with unit scalars: the combine also runs on the device)
    std::vector<tkmk_g1_projective> parts((size_t)c->world);
    e = tkmk_memcpy_d2h(parts.data(), d_all, sizeof(tkmk_g1_projective) * parts.size());
    cleanup();
    if (e != TKMK_SUCCESS) return fail(e, "tkmk_memcpy_d2h");
    std::vector<tkmk_fr> ones((size_t)c->world);
    std::vector<tkmk_g1_affine> pts((size_t)c->world);
    for (int q = 0; q < c->world; q++) {
        std::memset(&ones[q], 0, sizeof(tkmk_fr));
        ones[q].limbs[0] = 1;
        bool inf = true;
        for (uint32_t l : parts[q].z.limbs) inf &= l == 0;
        std::memset(&pts[q], 0, sizeof(tkmk_g1_affine));      // (0, 0) = infinity
        if (!inf) pts[q].x = parts[q].x, pts[q].y = parts[q].y;   // canonical (x, y, 1)
    }
    tkmk_msm_config sum = tkmk_msm_default_config();
    sum.stream_handle = cfg->stream_handle;
    TKD_TRY(bls12_381_msm(ones.data(), pts.data(), c->world, &sum, result));
    return TKMK_SUCCESS;
}

TKD_API tkmk_error tkmk_bintt_sharded(tkmk_comm *c, tkmk_fr *in_slab_dev, size_t x_size, size_t y_size, tkmk_ntt_dir dir, const tkmk_fr *coset_x,
                                      const tkmk_fr *coset_y, tkmk_fr *out_slab_dev) {
    if (!c || !in_slab_dev || !out_slab_dev) return TKMK_ERR_INVALID_POINTER;
    const size_t G = (size_t)c->world;
    if (x_size == 0 || y_size == 0 || x_size % G || y_size % G || (x_size & (x_size - 1)) || (y_size & (y_size - 1))) return TKMK_ERR_INVALID_ARGUMENT;
    const size_t rows = x_size / G, cols = y_size / G;
    auto ntt_cfg = [&](const tkmk_fr *coset, size_t batch, bool columns) {
        tkmk_ntt_config n = tkmk_ntt_default_config();
        n.batch_size = (int)batch;
        n.columns_batch = columns;
        n.are_inputs_on_device = n.are_outputs_on_device = true;
        if (coset) n.coset_gen = *coset;
        return n;
    };
    // 1. rows of this x-slab (length y_size, coset_y), in place
    tkmk_ntt_config rc = ntt_cfg(coset_y, rows, false);
    TKD_TRY(bls12_381_ntt(in_slab_dev, (int)y_size, dir, &rc, in_slab_dev));
    // 2. pack block (my rows) x (columns of q) contiguously for every q, then ONE all-to-all, device to device; the block from
    //    rank q (its rows, my columns) is a contiguous run of the y-slab (rows q * rows .. of width cols), so it is received in place
    void *d_send = nullptr;
    TKD_TRY(tkmk_malloc(&d_send, rows * y_size * sizeof(tkmk_fr)));
    tkmk_error e = TKMK_SUCCESS;
    for (size_t q = 0; q < G && e == TKMK_SUCCESS; q++)
        e = tkmk_memcpy_2d_d2d((uint8_t *)d_send + q * rows * cols * sizeof(tkmk_fr), cols * sizeof(tkmk_fr), (const uint8_t *)in_slab_dev + q * cols * sizeof(tkmk_fr),
                               y_size * sizeof(tkmk_fr), cols * sizeof(tkmk_fr), rows);
    if (e == TKMK_SUCCESS) e = tkmk_device_synchronize();   // the packing ran on the default stream; the collective runs on the communicator's
    if (e != TKMK_SUCCESS) {
        (void)tkmk_free(d_send);
        return fail(e, "pack");
    }
    ncclResult_t r = ncclAllToAll(d_send, out_slab_dev, rows * cols * sizeof(tkmk_fr), ncclUint8, c->nccl, c->stream);
    hipError_t h = r == ncclSuccess ? hipStreamSynchronize(c->stream) : hipSuccess;
    (void)tkmk_free(d_send);
    if (r != ncclSuccess) return fail(TKMK_ERR_UNKNOWN, std::string("ncclAllToAll: ") + ncclGetErrorString(r));
    if (h != hipSuccess) return fail(TKMK_ERR_UNKNOWN, std::string("hipStreamSynchronize: ") + hipGetErrorString(h));
    // 3. columns of the y-slab (length x_size, coset_x): x_size x cols matrix, element (ix, j) at ix * cols + j
    tkmk_ntt_config cc = ntt_cfg(coset_x, cols, cols > 1);
    TKD_TRY(bls12_381_ntt(out_slab_dev, (int)x_size, dir, &cc, out_slab_dev));
    return TKMK_SUCCESS;
}
