// Microbenchmark: rate of no-return 16-byte-record float atomics (4 lanes per record, the scatter's
// request shape) versus the FOOTPRINT the records are spread over, and versus a two-region mix.
// Usage: ./atomic_footprint_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

__global__ void __launch_bounds__(256) k(float *tab, uint32_t n_rows, int iters) {
    const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t rec = lane >> 2, sub = lane & 3;
    for (int it = 0; it < iters; it++) {
        const uint32_t h = hash32(gw * 7919u + it * 104729u + rec * 2654435761u);
        atomicAdd(tab + (size_t)(h % n_rows) * 4 + sub, 1.0f);
    }
}

int main() {
    const size_t max_rows = 64u * 1024 * 1024;       // 1 GiB of 16-byte rows
    float *tab; (void)hipMalloc(&tab, max_rows * 16); (void)hipMemset(tab, 0, max_rows * 16);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256, iters = 2000;
    for (size_t rows : {size_t(512), size_t(4096), size_t(65536), size_t(524288), size_t(1) << 21, size_t(6300000), size_t(1) << 24, max_rows}) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, (uint32_t)rows, 50);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, (uint32_t)rows, iters);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double recs = (double)blocks * 4 * iters * 16;
        printf("footprint %10.2f MB : %7.2f G records/s (%.2f ms)\n", rows * 16 / 1e6, recs / ms / 1e6, ms);
    }
    // is the limit per CU or chip-wide?  one 4-wave block per CU on `nb` CUs, 100 MB footprint
    for (int nb : {16, 32, 64, 128, 256, 512, 1024}) {
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, tab, 6300000u, 50);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, tab, 6300000u, iters);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double recs = (double)nb * 4 * iters * 16;
        printf("blocks %5d : %7.2f G records/s  (%.3f G/s per block)\n", nb, recs / ms / 1e6, recs / ms / 1e6 / nb);
    }
    (void)hipFree(tab);
    return 0;
}
