// tkmk_crs_load.hpp — where `prove` / `preprocess` get the reference string from: <crs>/combined_sigma.rkyv resp.
// <crs>/sigma_preprocess.rkyv, the reference's own archives (prove/src/sigma_source.rs:22-32, preprocess/src/main.rs:47-53), through
// tkmk_rkyv.hpp; or, when present, the flat TKCRS001 payload <crs>/combined_sigma.tkcrs (the form the reference's decoder derives
// from the archive: backend-wasm/tools/rkyv-decoder-wasm/src/lib.rs:118-140) as a fast path that needs no validation pass.
#pragma once
#include "tkmk_prover.hpp"
#include "tkmk_rkyv.hpp"

namespace tkmk {

inline bool file_exists(const std::string &path) {
    struct stat st;
    return ::stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
// read-only mapping of a file -> (view, owner)
inline std::pair<CrsPayload::View, std::shared_ptr<void>> map_file(const std::string &path) {
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error("cannot open " + path);
    struct stat st;
    if (::fstat(fd, &st) != 0 || st.st_size <= 0) {
        ::close(fd);
        throw Error("cannot stat " + path);
    }
    size_t n = (size_t)st.st_size;
    void *m = ::mmap(nullptr, n, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) throw Error("cannot map " + path);
    return {CrsPayload::View{static_cast<const uint8_t *>(m), n}, std::shared_ptr<void>(m, [n](void *q) { ::munmap(q, n); })};
}
inline rkyv::Expect expect_for(const SetupParams &sp) {
    size_t m_i = sp.l_D - sp.l;
    rkyv::Expect ex;
    ex.rs_y = 2 * sp.s_max;
    ex.xy_powers = std::max(2 * sp.n, 2 * m_i) * ex.rs_y;
    ex.gamma = sp.l, ex.eta = m_i * sp.s_max, ex.delta = (sp.m_D - sp.l_D) * sp.s_max;
    return ex;
}
// the nine sections of the combined reference string, whichever container holds it
inline CrsPayload load_combined_sigma(const std::string &crs_dir, const SetupParams &sp) {
    const std::string flat = crs_dir + "/combined_sigma.tkcrs", archive = crs_dir + "/combined_sigma.rkyv";
    if (file_exists(flat)) return CrsPayload::read(flat);
    if (file_exists(archive)) {
        auto m = map_file(archive);
        return rkyv::decode_combined_sigma(m.first, m.second, expect_for(sp));
    }
    throw Error("No reference string is found. Run the Setup first (expected " + archive + ").");
}
// table_c: window width of the precomputed commit table (0 = none): worth its one-time cost (seconds) and memory (13 x xy_powers at
// 20 bits) only for a prover that stays resident
inline std::unique_ptr<ProverSigma> load_prover_sigma(const std::string &crs_dir, const SetupParams &sp, std::string &source, uint32_t table_c = 0,
                                                      Shard shard = Shard{}, std::unique_ptr<Sigma1> *whole_grid = nullptr) {
    CrsPayload crs = load_combined_sigma(crs_dir, sp);
    source = crs.container;
    return std::unique_ptr<ProverSigma>(new ProverSigma(ProverSigma::from_payload(crs, sp, table_c, shard, whole_grid)));
}
// the resident prover's default: 20-bit windows once xy_powers is large enough for the wide sort to pay (>= 2^20 points);
// TKMK_PROVER_TABLE_C = 0 turns the table off, 13..20 picks another width.
// world: the ranks of a sharded context.  A rank's share of a commit has 1 / world of the points but the whole bucket set, and what a
// commit costs per bucket (fragment combine, bucket reduction) does not divide: measured over the loopback transport at configs[3]
// (tools/one_proof_loopback.py, profiles/r04_one_proof_loopback_table_window_by_world.json) 20-bit windows stay best at 2 and 4 ranks
// (141 against 153 ms and 96 against 97 ms of device work per rank) and 16-bit windows win at 8 (64.6 against 70.5 ms per rank: 2^15
// buckets instead of 2^19 for shares of 2 x 10^6 points).
inline uint32_t resident_table_c(const SetupParams &sp, uint32_t world = 1) {
    if (const char *e = std::getenv("TKMK_PROVER_TABLE_C")) {
        int v = std::atoi(e);
        return v >= 2 && v <= 20 ? (uint32_t)v : 0;
    }
    size_t m_i = sp.l_D - sp.l, points = std::max(2 * sp.n, 2 * m_i) * 2 * sp.s_max;
    if (points < (1u << 20)) return 0u;
    return world >= 8 ? 16u : 20u;
}

}  // namespace tkmk
