// Diagnostic: does a kernel boundary make one XCD's stores visible to readers on another XCD that hold an
// older copy of the line, for (1) hipMalloc memory, (2) hipMallocAsync pool memory, (3) pool memory that is
// freed / re-allocated between iterations?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_write(uint32_t *x, uint32_t per_block, uint32_t val) {
    uint32_t *p = x + (size_t)blockIdx.x * per_block;
    for (uint32_t i = threadIdx.x; i < per_block; i += blockDim.x) p[i] = val + i;
}
__global__ void k_read(const uint32_t *x, uint32_t per_block, uint32_t val, uint32_t shift, uint32_t *bad) {
    uint32_t src = (blockIdx.x + shift) % gridDim.x;
    const uint32_t *p = x + (size_t)src * per_block;
    uint32_t n = 0;
    for (uint32_t i = threadIdx.x; i < per_block; i += blockDim.x) n += p[i] != val + i;
    if (n) atomicAdd(bad, n);
}
int main() {
    const uint32_t nb = 1024, per = 4096;  // 16 MiB
    uint32_t *bad; CK(hipMalloc(&bad, 4));
    for (int mode = 0; mode < 3; mode++) {
        uint32_t *x = nullptr; uint32_t total_bad = 0;
        if (mode == 0) CK(hipMalloc(&x, (size_t)nb * per * 4));
        if (mode == 1) CK(hipMallocAsync(&x, (size_t)nb * per * 4, 0));
        for (int it = 0; it < 20; it++) {
            uint32_t *junk = nullptr;
            if (mode == 2) {
                // varying-size junk allocation first so the pool hands out shifted addresses
                CK(hipMallocAsync(&junk, (size_t)(1 + it % 3) * 1000000, 0));
                CK(hipMallocAsync(&x, (size_t)nb * per * 4, 0));
            }
            CK(hipMemsetAsync(bad, 0, 4, 0));
            hipLaunchKernelGGL(k_write, nb, 256, 0, 0, x, per, 1000u * it);
            hipLaunchKernelGGL(k_read, nb, 256, 0, 0, x, per, 1000u * it, 1u + it % 7, bad);
            uint32_t h = 0; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
            total_bad += h;
            if (mode == 2) { CK(hipFreeAsync(x, 0)); CK(hipFreeAsync(junk, 0)); }
        }
        printf("mode %d: stale words over 20 iterations = %u\n", mode, total_bad);
        if (mode == 0) CK(hipFree(x));
        if (mode == 1) CK(hipFreeAsync(x, 0));
        CK(hipDeviceSynchronize());
    }
    return 0;
}
