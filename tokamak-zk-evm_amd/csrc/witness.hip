// witness.hip — the witness side of Prover::init on the device: everything the reference does per proof
// in host loops over placements x wires, as a handful of launches over ONE resident copy of the parsed witness.
//
//   tkmk_r1cs_library_*      circuit-static CSR of every subcircuit's A / B / C (coefficients kept in Montgomery form), built once
//                            per subcircuit library; _eval = eval_uvwxy_sparse_rows for ALL placements in one launch
//                            (packages/backend/libs/src/iotools/mod.rs:1426-1523, 1581-1608: a rayon loop over placements), writing
//                            the n x s_max evaluation matrices directly (no s_max x n intermediate, no transposes)
//   tkmk_witness_route       gen_bXY's interface-wire scatter (libs/src/polynomial_structures/mod.rs:132-162) and the
//                            (scalar, CRS row) lists of encode_statement_common / encode_O_pub_free
//                            (libs/src/group_structures/mod.rs:184-229, 266-300) from a static per-subcircuit wire list
//   tkmk_fr_scatter_table    Permutation::to_poly's redirects (libs/src/iotools/mod.rs:438-448): dst[idx] = table[src]
//   tkmk_host_malloc/_free   pinned host staging for the parsed witness (H2D at link rate)
// Every entry works on device pointers and a stream; nothing here touches the oracle or the host CPU for arithmetic.
#include <mutex>
#include <vector>

#include "common.h"

struct tkmk_r1cs_library {
    uint32_t n_sub = 0, max_rows = 0;
    uint32_t *d_rowptr = nullptr;    // concatenated row_ptr arrays, (n_rows + 1) each, order [sub][matrix]
    uint32_t *d_wire = nullptr;      // concatenated wire arrays
    fr_t *d_coeff = nullptr;         // concatenated coefficients, Montgomery form
    uint32_t *d_desc = nullptr;      // per (sub, matrix): {rowptr base, entry base}; per sub: n_rows, n_wires  -> 8 u32 per sub
    std::vector<uint32_t> n_rows, n_wires;
};

// desc layout per subcircuit: [0..2] rowptr base of A,B,C; [3..5] entry base of A,B,C; [6] n_rows; [7] n_wires
__global__ __launch_bounds__(256) void k_r1cs_library_eval(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ wire,
                                                          const fr_t *__restrict__ coeff_mont, const uint32_t *__restrict__ desc,
                                                          const fr_t *__restrict__ vars, const uint32_t *__restrict__ pl_id,
                                                          const uint64_t *__restrict__ pl_off, uint32_t s_max, fr_t *__restrict__ u,
                                                          fr_t *__restrict__ v, fr_t *__restrict__ w) {
    const uint32_t p = blockIdx.z, m = blockIdx.y;
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t *d = desc + 8 * pl_id[p];
    if (row >= d[6]) return;   // rows past the subcircuit's constraint count stay zero (the matrices are pre-zeroed)
    const uint32_t *rp = rowptr + d[m];
    const uint32_t eb = d[3 + m];
    const fr_t *x = vars + pl_off[p];
    fr_t acc = Fr::zero();
    for (uint32_t k = rp[row]; k < rp[row + 1]; k++)
        acc = Fr::add(acc, Fr::mul(Fr::canon(tk_load(x + wire[eb + k])), tk_load(coeff_mont + eb + k)));   // plain * Montgomery -> plain
    fr_t *out = m == 0 ? u : m == 1 ? v : w;
    tk_store(out + (uint64_t)row * s_max + p, acc);   // evaluation matrix, element (row, placement)
}
__global__ __launch_bounds__(256) void k_fr_to_mont(const fr_t *__restrict__ in, fr_t *__restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tk_store(out + i, Fr::to_mont(Fr::canon(tk_load(in + i))));
}

// row_ptr / wire / coeff: 3 * n_sub host arrays in the order [sub][A, B, C]; row_ptr[k] has n_rows[sub] + 1 entries, wire[k] and
// coeff[k] have row_ptr[k][n_rows[sub]] entries (plain Fr).  Wire indices are checked against n_wires here, once, so that _eval
// cannot read outside a placement's variables.
TK_API tkmk_error tkmk_r1cs_library_create(uint32_t n_sub, const uint32_t *n_rows, const uint32_t *n_wires, const uint32_t *const *row_ptr,
                                           const uint32_t *const *wire, const tkmk_fr *const *coeff, tkmk_r1cs_library **out) {
    if (!out || !n_rows || !n_wires || !row_ptr || !wire || !coeff) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    std::vector<uint32_t> rp_all, wire_all, desc(8 * (size_t)n_sub);
    std::vector<tkmk_fr> coeff_all;
    uint32_t max_rows = 0;
    for (uint32_t s = 0; s < n_sub; s++) {
        for (int m = 0; m < 3; m++) {
            const uint32_t *rp = row_ptr[3 * s + m];
            if (!rp) return TKMK_ERR_INVALID_POINTER;
            const uint32_t nnz = rp[n_rows[s]];
            if (rp[0] != 0) return TKMK_ERR_INVALID_ARGUMENT;
            for (uint32_t r = 0; r < n_rows[s]; r++)
                if (rp[r] > rp[r + 1]) return TKMK_ERR_INVALID_ARGUMENT;
            if (nnz && (!wire[3 * s + m] || !coeff[3 * s + m])) return TKMK_ERR_INVALID_POINTER;
            for (uint32_t k = 0; k < nnz; k++)
                if (wire[3 * s + m][k] >= n_wires[s]) return TKMK_ERR_INVALID_ARGUMENT;
            desc[8 * s + m] = (uint32_t)rp_all.size();
            desc[8 * s + 3 + m] = (uint32_t)wire_all.size();
            rp_all.insert(rp_all.end(), rp, rp + n_rows[s] + 1);
            wire_all.insert(wire_all.end(), wire[3 * s + m], wire[3 * s + m] + nnz);
            coeff_all.insert(coeff_all.end(), coeff[3 * s + m], coeff[3 * s + m] + nnz);
        }
        desc[8 * s + 6] = n_rows[s];
        desc[8 * s + 7] = n_wires[s];
        if (n_rows[s] > max_rows) max_rows = n_rows[s];
    }
    if (wire_all.empty()) {   // keep the device pointers valid
        wire_all.push_back(0);
        coeff_all.push_back(tkmk_fr{});
    }
    tkmk_r1cs_library *lib = new tkmk_r1cs_library();
    lib->n_sub = n_sub;
    lib->max_rows = max_rows;
    lib->n_rows.assign(n_rows, n_rows + n_sub);
    lib->n_wires.assign(n_wires, n_wires + n_sub);
    auto up = [&](void **dst, const void *src, size_t bytes) -> tkmk_error {
        TK_HIP(hipMalloc(dst, bytes ? bytes : 4));
        if (bytes) TK_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return TKMK_SUCCESS;
    };
    tkmk_error e = up((void **)&lib->d_rowptr, rp_all.data(), rp_all.size() * 4);
    if (e == TKMK_SUCCESS) e = up((void **)&lib->d_wire, wire_all.data(), wire_all.size() * 4);
    if (e == TKMK_SUCCESS) e = up((void **)&lib->d_coeff, coeff_all.data(), coeff_all.size() * sizeof(tkmk_fr));
    if (e == TKMK_SUCCESS) e = up((void **)&lib->d_desc, desc.data(), desc.size() * 4);
    if (e == TKMK_SUCCESS) {
        hipLaunchKernelGGL(k_fr_to_mont, tk_div_up(coeff_all.size(), 256), 256, 0, 0, (const fr_t *)lib->d_coeff, lib->d_coeff,
                           (uint64_t)coeff_all.size());
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) e = TKMK_ERR_UNKNOWN;
    }
    if (e != TKMK_SUCCESS) {
        (void)hipFree(lib->d_rowptr), (void)hipFree(lib->d_wire), (void)hipFree(lib->d_coeff), (void)hipFree(lib->d_desc);
        delete lib;
        return e;
    }
    *out = lib;
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_r1cs_library_destroy(tkmk_r1cs_library *lib) {
    if (!lib) return TKMK_SUCCESS;
    (void)hipDeviceSynchronize();
    (void)hipFree(lib->d_rowptr), (void)hipFree(lib->d_wire), (void)hipFree(lib->d_coeff), (void)hipFree(lib->d_desc);
    delete lib;
    return TKMK_SUCCESS;
}

// placement p (p < n_placements <= s_max) instantiates subcircuit placement_id_dev[p] on the variables
// vars_dev[placement_var_offset_dev[p] .. + n_wires).  u / v / w: n x s_max matrices, fully written (zero outside the rows and
// placements in use).  The caller guarantees that every id < n_sub and that each placement's variable range lies inside vars_dev
// (host-checked by the service before the upload).
TK_API tkmk_error tkmk_r1cs_library_eval(const tkmk_r1cs_library *lib, const tkmk_fr *vars_dev, const uint32_t *placement_id_dev,
                                         const uint64_t *placement_var_offset_dev, uint32_t n_placements, uint32_t n, uint32_t s_max,
                                         tkmk_fr *u_dev, tkmk_fr *v_dev, tkmk_fr *w_dev, tkmk_stream stream) {
    if (!lib || !u_dev || !v_dev || !w_dev) return TKMK_ERR_INVALID_POINTER;
    if (n_placements && (!vars_dev || !placement_id_dev || !placement_var_offset_dev)) return TKMK_ERR_INVALID_POINTER;
    if (n_placements > s_max || lib->max_rows > n) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    hipStream_t s = tk_stream(stream);
    const size_t bytes = (size_t)n * s_max * sizeof(fr_t);
    TK_HIP(hipMemsetAsync(u_dev, 0, bytes, s));
    TK_HIP(hipMemsetAsync(v_dev, 0, bytes, s));
    TK_HIP(hipMemsetAsync(w_dev, 0, bytes, s));
    if (n_placements && lib->max_rows) {
        hipLaunchKernelGGL(k_r1cs_library_eval, dim3(tk_div_up(lib->max_rows, 256), 3, n_placements), 256, 0, s, lib->d_rowptr, lib->d_wire,
                           (const fr_t *)lib->d_coeff, lib->d_desc, (const fr_t *)vars_dev, placement_id_dev, placement_var_offset_dev,
                           s_max, (fr_t *)u_dev, (fr_t *)v_dev, (fr_t *)w_dev);
        TK_HIP(hipGetLastError());
    }
    return TKMK_SUCCESS;
}

// For the n_placements placements of ONE subcircuit kind (variables at var_offset_dev[i], global placement index slot_dev[i])
// and a static list of n_list (local wire, row) pairs of that kind:
//   matrix_dev != NULL :  matrix_dev[row * matrix_stride + slot]     = variable      (gen_bXY: row = flattenMap - l)
//   scalars_out != NULL:  scalars_out[i * n_list + e]                = variable
//                         index_out  [i * n_list + e]                = row * index_inner + (index_add_slot ? slot : 0)
__global__ __launch_bounds__(256) void k_witness_route(const fr_t *__restrict__ vars, const uint64_t *__restrict__ var_off,
                                                      const uint32_t *__restrict__ slot, uint32_t n_placements,
                                                      const uint32_t *__restrict__ list_wire, const uint32_t *__restrict__ list_row,
                                                      uint32_t n_list, fr_t *__restrict__ matrix, uint32_t matrix_stride,
                                                      fr_t *__restrict__ scalars_out, uint32_t *__restrict__ index_out, uint32_t index_inner,
                                                      int index_add_slot) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (uint64_t)n_placements * n_list) return;
    uint32_t i = (uint32_t)(g / n_list), e = (uint32_t)(g - (uint64_t)i * n_list);
    fr_t val = tk_load(vars + var_off[i] + list_wire[e]);
    const uint32_t row = list_row[e], sl = slot[i];
    if (matrix) tk_store(matrix + (uint64_t)row * matrix_stride + sl, val);
    if (scalars_out) {
        tk_store(scalars_out + g, val);
        index_out[g] = row * index_inner + (index_add_slot ? sl : 0u);
    }
}
TK_API tkmk_error tkmk_witness_route(const tkmk_fr *vars_dev, const uint64_t *var_offset_dev, const uint32_t *slot_dev, uint32_t n_placements,
                                     const uint32_t *list_wire_dev, const uint32_t *list_row_dev, uint32_t n_list, tkmk_fr *matrix_dev,
                                     uint32_t matrix_stride, tkmk_fr *scalars_out_dev, uint32_t *index_out_dev, uint32_t index_inner,
                                     int index_add_slot, tkmk_stream stream) {
    if (n_placements == 0 || n_list == 0) return TKMK_SUCCESS;
    if (!vars_dev || !var_offset_dev || !slot_dev || !list_wire_dev || !list_row_dev) return TKMK_ERR_INVALID_POINTER;
    if ((scalars_out_dev == nullptr) != (index_out_dev == nullptr)) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    hipStream_t s = tk_stream(stream);
    hipLaunchKernelGGL(k_witness_route, tk_div_up((uint64_t)n_placements * n_list, 256), 256, 0, s, (const fr_t *)vars_dev, var_offset_dev,
                       slot_dev, n_placements, list_wire_dev, list_row_dev, n_list, (fr_t *)matrix_dev, matrix_stride, (fr_t *)scalars_out_dev,
                       index_out_dev, index_inner, index_add_slot);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}

// out[dst_idx[i]] = table[src_idx[i]], i < n: distinct dst_idx required (the caller resolves duplicates: last writer wins in the
// reference's serial loop).  An index past its array is never dereferenced: the entry is skipped and the call reports it.
__global__ __launch_bounds__(256) void k_fr_scatter_table(const fr_t *__restrict__ table, uint64_t table_len, const uint32_t *__restrict__ src_idx,
                                                         const uint32_t *__restrict__ dst_idx, uint64_t n, fr_t *__restrict__ out, uint64_t out_len,
                                                         uint32_t *__restrict__ bad) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t sidx = src_idx[i], didx = dst_idx[i];
    if (sidx >= table_len || didx >= out_len) {
        *bad = 1u;
        return;
    }
    tk_store(out + didx, tk_load(table + sidx));
}
TK_API tkmk_error tkmk_fr_scatter_table(const tkmk_fr *table_dev, uint64_t table_len, const uint32_t *src_idx_dev, const uint32_t *dst_idx_dev,
                                        uint64_t n, tkmk_fr *out_dev, uint64_t out_len, tkmk_stream stream) {
    if (n == 0) return TKMK_SUCCESS;
    if (!table_dev || !src_idx_dev || !dst_idx_dev || !out_dev) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    tk_scratch d_bad;
    TK_TRY(d_bad.alloc(4, s));
    TK_HIP(hipMemsetAsync(d_bad.p, 0, 4, s));
    hipLaunchKernelGGL(k_fr_scatter_table, tk_div_up(n, 256), 256, 0, s, (const fr_t *)table_dev, table_len, src_idx_dev, dst_idx_dev, n,
                       (fr_t *)out_dev, out_len, d_bad.as<uint32_t>());
    TK_HIP(hipGetLastError());
    uint32_t bad = 0;
    TK_HIP(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, s));
    TK_HIP(hipStreamSynchronize(s));
    return bad ? TKMK_ERR_INVALID_ARGUMENT : TKMK_SUCCESS;
}

// pinned (page-locked) host memory: the parsed witness is written here by the parser threads and uploaded in one copy
TK_API tkmk_error tkmk_host_malloc(void **ptr, size_t bytes) {
    if (!ptr) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    TK_HIP(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_host_free(void *ptr) {
    if (!ptr) return TKMK_SUCCESS;
    TK_HIP(hipHostFree(ptr));
    return TKMK_SUCCESS;
}
