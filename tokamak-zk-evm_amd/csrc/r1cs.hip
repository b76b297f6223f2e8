// r1cs.hip — sparse R1CS row evaluation on the device: the witness-side producer of the u/v/w evaluation matrices
// (SURVEY.md §8f-3).  Reference: eval_uvwxy_sparse_rows / build_d_vec / eval_sparse_rows
// (packages/backend/libs/src/iotools/mod.rs:1426-1523, 1581-1608), a rayon loop over placements on the host.
// One lane per (placement, constraint row): out[p*n + row] = sum_e coeff[e] * variables[p][wire[e]] over the row's
// CSR entries; rows past the subcircuit's constraint count and rows without entries are zero.
#include "common.h"

__global__ __launch_bounds__(256) void k_r1cs_eval_rows(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ wire,
                                                       const fr_t *__restrict__ coeff_mont, uint32_t n_rows,
                                                       const fr_t *__restrict__ variables, uint32_t n_wires, uint32_t n_placements,
                                                       const uint32_t *__restrict__ out_slot, uint32_t n, fr_t *__restrict__ out) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)n_placements * n_rows) return;
    uint32_t p = (uint32_t)(e / n_rows), row = (uint32_t)(e - (uint64_t)p * n_rows);
    const fr_t *v = variables + (uint64_t)p * n_wires;
    fr_t acc = Fr::zero();
    for (uint32_t k = row_ptr[row]; k < row_ptr[row + 1]; k++)
        acc = Fr::add(acc, Fr::mul(Fr::canon(tk_load(v + wire[k])), tk_load(coeff_mont + k)));  // plain * Montgomery -> plain
    if (row < n) tk_store(out + (uint64_t)out_slot[p] * n + row, acc);
}
__global__ __launch_bounds__(256) void k_to_mont(const fr_t *__restrict__ in, fr_t *__restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tk_store(out + i, Fr::to_mont(Fr::canon(tk_load(in + i))));
}

// CSR of one matrix (A, B or C) of one subcircuit: row_ptr[n_rows + 1], wire[nnz] (wire index inside the subcircuit),
// coeff[nnz] (plain Fr).  variables: n_placements x n_wires plain Fr (the placements that instantiate this subcircuit);
// out_slot[p] = index of placement p in the global placement list; out: s_max x n matrix (pre-zeroed by the caller),
// row p receives the evaluations of placement p.  All pointers are device pointers.
TK_API tkmk_error tkmk_r1cs_eval_rows(const uint32_t *row_ptr_dev, const uint32_t *wire_dev, const tkmk_fr *coeff_dev, uint32_t n_rows,
                                      uint32_t nnz, const tkmk_fr *variables_dev, uint32_t n_wires, uint32_t n_placements,
                                      const uint32_t *out_slot_dev, uint32_t n, tkmk_fr *out_dev, tkmk_stream stream) {
    if (!row_ptr_dev || !out_slot_dev || !out_dev || !variables_dev) return TKMK_ERR_INVALID_POINTER;
    if (nnz && (!wire_dev || !coeff_dev)) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    if (n_placements == 0 || n_rows == 0) return TKMK_SUCCESS;
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    tk_scratch cm;
    TK_TRY(cm.alloc((size_t)(nnz ? nnz : 1) * sizeof(fr_t), s));
    if (nnz) hipLaunchKernelGGL(k_to_mont, tk_div_up(nnz, 256), 256, 0, s, (const fr_t *)coeff_dev, cm.as<fr_t>(), (uint64_t)nnz);
    hipLaunchKernelGGL(k_r1cs_eval_rows, tk_div_up((uint64_t)n_placements * n_rows, 256), 256, 0, s, row_ptr_dev, wire_dev,
                       (const fr_t *)cm.p, n_rows, (const fr_t *)variables_dev, n_wires, n_placements, out_slot_dev, n, (fr_t *)out_dev);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}
