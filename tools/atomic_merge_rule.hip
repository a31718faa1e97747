// Which lanes of one no-return float-atomic wave-instruction are merged into one 64-byte request?
// 16 records of 16 B (4 lanes each) per instruction; patterns place records of one 64-B line at
// different lane distances / orders.  Time per instruction ~ number of requests (21 G requests/s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// pattern -> (line slot index, row within line 0..3) for record t of the instruction
template <int P>
__device__ __forceinline__ void place(uint32_t t, uint32_t &line_id, uint32_t &row) {
    if (P == 0) { line_id = t; row = t & 3; }                                  // 16 distinct lines
    if (P == 1) { line_id = t >> 1; row = t & 1; }                             // pairs, adjacent, ascending
    if (P == 2) { line_id = t >> 1; row = 1 - (t & 1); }                       // pairs, adjacent, descending
    if (P == 3) { line_id = t & 7; row = t >> 3; }                             // pairs, 8 records apart
    if (P == 4) { line_id = t >> 2; row = t & 3; }                             // quads ascending (whole line)
    if (P == 5) { line_id = t >> 2; const uint32_t perm[4] = {2, 0, 3, 1}; row = perm[t & 3]; }   // quads permuted
    if (P == 6) { line_id = t >> 1; row = 0; }                                 // exact duplicates, adjacent
    if (P == 7) { line_id = t & 3; row = t >> 2; }                             // quads, members 4 records apart
    if (P == 8) { line_id = t >> 1; row = (t & 1) * 2; }                       // pairs adjacent, rows 0 and 2 (gap)
}
template <int P>
__global__ void __launch_bounds__(256) k(float *tab, uint32_t n_lines, int iters) {
    const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t t = lane >> 2, i = lane & 3;
    uint32_t lid, row;
    place<P>(t, lid, row);
    for (int it = 0; it < iters; it++) {
        const uint32_t line = hash32(gw * 7919u + it * 104729u + lid * 2654435761u) % n_lines;
        atomicAdd(tab + (size_t)line * 16 + row * 4 + i, 1.0f);
    }
}
template <int P> void run(float *tab, uint32_t n_lines, const char *name) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<P>, dim3(2048), dim3(256), 0, 0, tab, n_lines, 8);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<P>, dim3(2048), dim3(256), 0, 0, tab, n_lines, 400);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double instr = 2048.0 * 4 * 400;
    printf("P%-2s %-44s %6.2f ms  -> %5.2f requests/instruction (at 21.1 G req/s)\n", name + 0, name + 3, ms, ms * 1e-3 * 21.1e9 / instr);
}
int main() {
    const uint32_t n_lines = 1500000;   // 96 MB
    float *tab; (void)hipMalloc(&tab, (size_t)n_lines * 64 + 4096); (void)hipMemset(tab, 0, (size_t)n_lines * 64);
    run<0>(tab, n_lines, "0  16 distinct lines");
    run<1>(tab, n_lines, "1  pairs adjacent ascending");
    run<2>(tab, n_lines, "2  pairs adjacent descending");
    run<3>(tab, n_lines, "3  pairs 8 records apart");
    run<4>(tab, n_lines, "4  quads ascending (whole line)");
    run<5>(tab, n_lines, "5  quads permuted");
    run<6>(tab, n_lines, "6  exact duplicate rows adjacent");
    run<7>(tab, n_lines, "7  quads, members 4 records apart");
    run<8>(tab, n_lines, "8  pairs adjacent, rows 0 and 2");
    return 0;
}
