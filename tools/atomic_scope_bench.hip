// Microbenchmark (round 3): rate of no-return 16-byte-record float atomics by MEMORY SCOPE.  Agent (device) scope atomics are
// executed at the memory side on a multi-XCD part (the eight L2s are not coherent with each other: `sc1`), 21 G requests/s
// chip-wide (atomic_footprint_bench).  Workgroup-scope atomics carry no sc1 and can be executed in the issuing XCD's L2.
// If every XCD adds into its OWN copy of the table (8 x the footprint), L2-local atomics are correct: is their rate higher?
// Also verifies the sums: every record adds 1.0 -- the grand total over the copies must equal the number of adds.
// Usage: ./atomic_scope_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xFu;
}

template <int SCOPE>   // 0 = agent, 1 = workgroup into the XCD's private copy
__global__ void __launch_bounds__(256) k(float *tab, uint32_t n_rows, int iters, uint32_t *xcd_seen) {
    const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t rec = lane >> 2, sub = lane & 3;
    const uint32_t xcd = xcc_id();
    if (threadIdx.x == 0) atomicOr(xcd_seen + (blockIdx.x & 7u), 1u << xcd);
    float *base = SCOPE == 1 ? tab + (size_t)xcd * n_rows * 4 : tab;
    for (int it = 0; it < iters; it++) {
        const uint32_t h = hash32(gw * 7919u + it * 104729u + rec * 2654435761u);
        float *p = base + (size_t)(h % n_rows) * 4 + sub;
        if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void k_sum(const float *tab, size_t n, double *out) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += tab[i];
    atomicAdd(out, acc);
}

int main() {
    const size_t rows = 6300000;                    // the gradient table: 100.8 MB per copy
    float *tab; (void)hipMalloc(&tab, rows * 16 * 8);
    uint32_t *seen; (void)hipMalloc(&seen, 32);
    double *sum; (void)hipMalloc(&sum, 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int scope = 0; scope < 2; scope++) {
        for (int blocks : {256, 1024, 4096}) {
            const int iters = 2000 * 256 / blocks * (blocks > 256 ? 2 : 1);
            (void)hipMemset(tab, 0, rows * 16 * 8); (void)hipMemset(seen, 0, 32); (void)hipMemset(sum, 0, 8);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            if (scope) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, tab, (uint32_t)rows, iters, seen);
            else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, tab, (uint32_t)rows, iters, seen);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            hipLaunchKernelGGL(k_sum, dim3(2048), dim3(256), 0, 0, tab, rows * 4 * 8, sum);
            double h; (void)hipMemcpy(&h, sum, 8, hipMemcpyDeviceToHost);
            uint32_t hs[8]; (void)hipMemcpy(hs, seen, 32, hipMemcpyDeviceToHost);
            const double recs = (double)blocks * 4 * iters * 16, adds = recs * 4;
            printf("%s scope, %5d blocks: %7.2f G records/s (%.2f ms)  sum %.0f of %.0f %s   xcd masks by blockIdx%%8:", scope ? "workgroup" : "agent    ",
                   blocks, recs / ms / 1e6, ms, h, adds, h == adds ? "OK" : "MISMATCH");
            for (int i = 0; i < 8; i++) printf(" %02x", hs[i]);
            printf("\n");
        }
    }
    return 0;
}
