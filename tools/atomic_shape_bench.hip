// Microbenchmark: chip-wide rate of no-return global float atomics by request shape.
// LANES_PER_REC consecutive lanes add to consecutive dwords of one random 64-byte-aligned-or-not
// "record"; records are spread uniformly over a 100 MB table (the fp32 gradient-table size).
// Usage: ./atomic_shape_bench  -> prints G records/s, G lane-atomics/s, GB/s of added bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

template <int LPR>   // lanes per record: 1,4,8,16,32,64
__global__ void __launch_bounds__(256) k(float *tab, uint32_t n_slots, int iters, int align_dw) {
    const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t rec = lane / LPR, sub = lane % LPR;
    for (int it = 0; it < iters; it++) {
        uint32_t h = hash32(gw * 7919u + it * 104729u + rec * 2654435761u);
        // slot index in units of align_dw dwords
        uint32_t slot = h % n_slots;
        atomicAdd(tab + (size_t)slot * align_dw + sub, 1.0f);
    }
}

template <int LPR>
double run(float *tab, size_t n_dw, int align_dw, int iters, int blocks = 256 * 8) {
    const uint32_t n_slots = (uint32_t)(n_dw / align_dw) - 64;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<LPR>, dim3(blocks), dim3(256), 0, 0, tab, n_slots, 8, align_dw);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<LPR>, dim3(blocks), dim3(256), 0, 0, tab, n_slots, iters, align_dw);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * 4 * iters;
    const double recs = instr * (64 / LPR);
    printf("blocks %4d lanes/record %2d (%3d B)  align %3d B : %7.2f G records/s  %7.2f G lane-atomics/s  %7.1f GB/s  (%.2f ms)\n", blocks, LPR,
           LPR * 4, align_dw * 4, recs / ms / 1e6, instr * 64 / ms / 1e6, instr * 256 / ms / 1e6, ms);
    return ms;
}

int main() {
    const size_t n_dw = 25u * 1000 * 1000;   // 100 MB
    float *tab; hipMalloc(&tab, n_dw * 4); hipMemset(tab, 0, n_dw * 4);
    const int iters = 400;
    run<1>(tab, n_dw, 1, iters);
    run<4>(tab, n_dw, 4, iters);     // 16-B records, 16-B aligned (the current scatter)
    run<8>(tab, n_dw, 8, iters);     // 32-B aligned pairs
    run<8>(tab, n_dw, 4, iters);     // 32-B pairs at 16-B alignment (may straddle a 64-B line)
    run<16>(tab, n_dw, 16, iters);
    run<32>(tab, n_dw, 32, iters);
    run<64>(tab, n_dw, 64, iters);
    // small hot table (contention like coarse levels): 4096 records of 16 B
    run<4>(tab, 4096 * 4 + 64 * 4, 4, iters);
    run<4>(tab, 512 * 4 + 64 * 4, 4, iters);
    // occupancy sweep for the 16-B record shape: 1, 2, 4 blocks (4 waves each) per CU
    run<4>(tab, n_dw, 4, iters * 4, 256);
    run<4>(tab, n_dw, 4, iters * 4, 512);
    run<4>(tab, n_dw, 4, iters * 2, 1024);
    hipFree(tab);
    return 0;
}
