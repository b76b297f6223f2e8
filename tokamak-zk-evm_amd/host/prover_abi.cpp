// prover_abi.cpp — libtkmk_prover.so: the C ABI of include/tkmk_prover.h over host/tkmk_service.hpp (ProverContext).
// Built twice: libtkmk_prover.so (production: blinding scalars always from getrandom(), a testing_mixer_json argument is refused)
// and, with -DTKMK_TESTING_MODE, libtkmk_prover_testing.so for the differential tests (the reference's `testing-mode` feature).
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/tkmk_prover.h"
#include "tkmk_crs_load.hpp"
#include "tkmk_service.hpp"

using namespace tkmk;

struct tkmk_prover {
    std::unique_ptr<ProverContext> ctx;
};

static thread_local std::string g_last_error;
// libtkmk_hip.so takes calls from ONE host thread at a time per stream (include/tkmk.h "Threading"), and a prover context issues all its
// work on the default stream: two contexts of one process therefore take turns, whole calls at a time.  (The virtual ranks of a sharded
// prover over the loopback transport take turns through the communicator's device turn instead; RCCL ranks are one process per GPU.)
static std::mutex g_default_stream_mu;
struct DefaultStreamTurn {
    std::unique_lock<std::mutex> lk;
    explicit DefaultStreamTurn(bool sharded) : lk(g_default_stream_mu, std::defer_lock) {
        if (!sharded) lk.lock();
    }
};

template <class F>
static tkmk_error guarded(F &&fn) {
    try {
        fn();
        return TKMK_SUCCESS;
    } catch (const Error &e) {
        g_last_error = e.what();
        return e.code == TKMK_SUCCESS ? TKMK_ERR_UNKNOWN : e.code;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return TKMK_ERR_INVALID_ARGUMENT;
    }
}

#define TKP_API extern "C" __attribute__((visibility("default")))

TKP_API tkmk_error tkmk_prover_open(const char *subcircuit_library_dir, const char *crs_dir, tkmk_prover **out) {
    if (!subcircuit_library_dir || !crs_dir || !out) return TKMK_ERR_INVALID_POINTER;
    return guarded([&] {
        int ndev = 0;
        if (tkmk_device_count(&ndev) != TKMK_SUCCESS || ndev < 1) throw Error(TKMK_ERR_NO_DEVICE, "tkmk_prover_open: no HIP device (the MI355X backend has no CPU fallback)");
        std::string crs = crs_dir;
        DefaultStreamTurn turn(false);
        std::unique_ptr<tkmk_prover> p(new tkmk_prover());
        p->ctx = ProverContext::open(subcircuit_library_dir, crs, [&](const SetupParams &sp, std::string &source) { return load_prover_sigma(crs, sp, source, resident_table_c(sp)); });
        *out = p.release();
    });
}

static char *dup_string(const std::string &doc) {
    char *s = (char *)std::malloc(doc.size() + 1);
    if (!s) throw Error("out of memory");
    std::memcpy(s, doc.c_str(), doc.size() + 1);
    return s;
}

// the entries of libtkmk_dist.so a sharded context calls, resolved at run time: the prover library itself does not link RCCL, and the
// host that made the communicator has the library in the process already
static ShardLink link_for(void *comm) {
    void *h = dlopen("libtkmk_dist.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) {   // not loaded under that name: next to this library
        Dl_info info;
        if (dladdr((void *)&link_for, &info) && info.dli_fname) {
            std::string dir = info.dli_fname;
            size_t s = dir.rfind('/');
            dir = s == std::string::npos ? "." : dir.substr(0, s);
            h = dlopen((dir + "/libtkmk_dist.so").c_str(), RTLD_NOW);
        }
    }
    if (!h) throw Error("tkmk_prover_open_sharded: libtkmk_dist.so is not loaded and was not found next to libtkmk_prover.so");
    ShardLink l;
    l.comm = comm;
    auto sym = [&](const char *name) {
        void *p = dlsym(h, name);
        if (!p) throw Error(std::string("tkmk_prover_open_sharded: libtkmk_dist.so lacks ") + name);
        return p;
    };
    l.multi_ex_sharded = (decltype(l.multi_ex_sharded))sym("tkmk_msm_multi_ex_sharded");
    l.broadcast_host = (decltype(l.broadcast_host))sym("tkmk_comm_broadcast_host");
    l.device_turn = (decltype(l.device_turn))sym("tkmk_comm_device_turn");
    l.all_gather_host = (decltype(l.all_gather_host))sym("tkmk_comm_all_gather_host");
    l.agree = (decltype(l.agree))sym("tkmk_comm_agree");
    l.abort = (decltype(l.abort))sym("tkmk_comm_abort");
    l.fwd_cols_to_rows = (decltype(l.fwd_cols_to_rows))sym("tkmk_dist_fwd_cols_to_rows");
    l.inv_rows_to_cols = (decltype(l.inv_rows_to_cols))sym("tkmk_dist_inv_rows_to_cols");
    l.rows_rotate = (decltype(l.rows_rotate))sym("tkmk_dist_rows_rotate");
    l.ring_shift = (decltype(l.ring_shift))sym("tkmk_comm_ring_shift");
    l.relayout_cols_to_rows = (decltype(l.relayout_cols_to_rows))sym("tkmk_dist_relayout_cols_to_rows");
    l.relayout_rows_to_cols = (decltype(l.relayout_rows_to_cols))sym("tkmk_dist_relayout_rows_to_cols");
    int world = ((int (*)(const void *))sym("tkmk_comm_size"))(comm), rank = ((int (*)(const void *))sym("tkmk_comm_rank"))(comm);
    if (world < 1 || rank < 0 || rank >= world) throw Error("tkmk_prover_open_sharded: invalid communicator");
    l.shard = Shard{(uint32_t)world, (uint32_t)rank};
    return l;
}

TKP_API tkmk_error tkmk_prover_open_sharded(void *comm, const char *subcircuit_library_dir, const char *crs_dir, tkmk_prover **out) {
    if (!comm || !subcircuit_library_dir || !crs_dir || !out) return TKMK_ERR_INVALID_POINTER;
    return guarded([&] {
        int ndev = 0;
        if (tkmk_device_count(&ndev) != TKMK_SUCCESS || ndev < 1) throw Error(TKMK_ERR_NO_DEVICE, "tkmk_prover_open_sharded: no HIP device (the MI355X backend has no CPU fallback)");
        ShardLink link = link_for(comm);
        std::string crs = crs_dir;
        std::unique_ptr<tkmk_prover> p(new tkmk_prover());
        p->ctx = ProverContext::open(
            subcircuit_library_dir, crs, [&](const SetupParams &sp, std::string &source) { return load_prover_sigma(crs, sp, source, resident_table_c(sp, link.shard.world), link.shard); }, link);
        *out = p.release();
    });
}
TKP_API int tkmk_prover_world_size(const tkmk_prover *p) { return p && p->ctx->link ? (int)p->ctx->link.shard.world : 1; }

TKP_API tkmk_error tkmk_prover_prove(tkmk_prover *p, const char *synthesizer_dir, const char *output_dir, const char *testing_mixer_json,
                                     tkmk_prove_timing *timing, char **proof_json_out) {
    return tkmk_prover_prove_ex(p, synthesizer_dir, output_dir, testing_mixer_json, 0, timing, proof_json_out, nullptr);
}

TKP_API tkmk_error tkmk_prover_prove_ex(tkmk_prover *p, const char *synthesizer_dir, const char *output_dir, const char *testing_mixer_json, int flags,
                                        tkmk_prove_timing *timing, char **proof_json_out, char **commit_boxes_json_out) {
    if (!p || !synthesizer_dir) return TKMK_ERR_INVALID_POINTER;
    if (flags & ~(TKMK_PROVE_TEST_PARTS | TKMK_PROVE_COEFFICIENT_BASIS)) return TKMK_ERR_INVALID_ARGUMENT;
    if (proof_json_out) *proof_json_out = nullptr;
    if (commit_boxes_json_out) *commit_boxes_json_out = nullptr;
    std::vector<CommitBox> boxes;
    struct SinkGuard {   // the sink is thread-local and must not outlive `boxes`
        explicit SinkGuard(std::vector<CommitBox> *s) { commit_box_sink() = s; }
        ~SinkGuard() { commit_box_sink() = nullptr; }
    } sink_guard(commit_boxes_json_out ? &boxes : nullptr);
    return guarded([&] {
#ifdef TKMK_TESTING_MODE
        Mixer mixer = testing_mixer_json ? mixer_from_json(json::read_file(testing_mixer_json)) : Mixer::random();
#else
        // the reference gates fixed blinding scalars behind the compile-time feature `testing-mode`; so does this library
        if (testing_mixer_json) throw Error("tkmk_prover_prove: testing_mixer_json needs the testing-mode build (libtkmk_prover_testing.so); this library always draws its blinding scalars from getrandom()");
        Mixer mixer = Mixer::random();
#endif
        ProveTiming tm;
        DefaultStreamTurn turn((bool)p->ctx->link);
        Proof proof = p->ctx->prove(synthesizer_dir, output_dir ? output_dir : "", mixer, &tm, flags);
        if (timing) {
            timing->parse_s = tm.parse, timing->upload_s = tm.upload, timing->build_s = tm.build, timing->binding_s = tm.binding;
            timing->init_s = tm.init, timing->write_s = tm.write, timing->total_s = tm.total;
            for (int k = 0; k < 5; k++) timing->prove_s[k] = tm.prove[k];
        }
        if (proof_json_out) *proof_json_out = dup_string(proof.to_json());
        if (commit_boxes_json_out) {
            std::string doc = "[";
            for (size_t k = 0; k < boxes.size(); k++)
                doc += std::string(k ? ", " : "") + "{\"name\": \"" + boxes[k].name + "\", \"x\": " + std::to_string(boxes[k].x) + ", \"y\": " +
                       std::to_string(boxes[k].y) + ", \"basis\": \"" + boxes[k].basis + "\"}";
            *commit_boxes_json_out = dup_string(doc + "]");
        }
    });
}

TKP_API tkmk_error tkmk_prover_close(tkmk_prover *p) {
    if (!p) return TKMK_SUCCESS;
    return guarded([&] {
        DefaultStreamTurn turn((bool)p->ctx->link);
        delete p;
    });
}
TKP_API void tkmk_prover_free_string(char *s) { std::free(s); }
TKP_API const char *tkmk_prover_last_error(void) { return g_last_error.c_str(); }
TKP_API const char *tkmk_prover_crs_source(const tkmk_prover *p) { return p ? p->ctx->crs_source.c_str() : ""; }
