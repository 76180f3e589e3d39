// hbm_write_pat4.hip -- does the NUMBER OF STORES A WAVE KEEPS IN FLIGHT change the rate of a store stream?
// Same shapes as hbm_write_pat3.hip (4 GB, four waves of a workgroup interleaved in 1 KB pieces inside a chunk), with an
// `s_waitcnt vmcnt(T)` behind every store: T = -1 no wait (the hardware queue decides), 0 one store at a time, ...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ unsigned char dyn_lds[];
template <int T>
__global__ __launch_bounds__(256) void fill_chunks(u32x4* dst, size_t chunk_vec, size_t nchunks, int touch) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    if (touch) dyn_lds[threadIdx.x] = 1;
    const size_t npieces = chunk_vec / 64;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t base = c * chunk_vec;
        for (size_t p = wave; p < npieces; p += 4) {
            dst[base + p * 64 + lane] = v; v.x++;
            if (T == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (T == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            if (T == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (T == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (T == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
}
template <int T>
static float run(u32x4* d, size_t bytes, size_t chunk, unsigned grid, unsigned lds) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const size_t chunk_vec = chunk / 16, nchunks = bytes / chunk;
    if (grid == 0) grid = (unsigned)nchunks;
    (void)hipFuncSetAttribute((const void*)fill_chunks<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(fill_chunks<T>, dim3(grid), dim3(256), lds, 0, d, chunk_vec, nchunks, lds ? 1 : 0);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    return (bytes - bytes % chunk) / (best * 1e-3) / 1e12;
}
template <int T>
static void row(u32x4* d, size_t bytes) {
    printf("%-8d  %-12.2f  %-12.2f  %-12.2f  %-12.2f  %-12.2f  %-12.2f\n", T,
           run<T>(d, bytes, (size_t)973 << 10, 0, 40960), run<T>(d, bytes, (size_t)973 << 10, 0, 81920), run<T>(d, bytes, (size_t)1 << 20, 0, 40960),
           run<T>(d, bytes, (size_t)4 << 10, 0, 40960), run<T>(d, bytes, (size_t)4 << 10, 1024, 40960), run<T>(d, bytes, (size_t)64 << 10, 0, 40960));
    fflush(stdout);
}
int main() {
    const size_t bytes = (size_t)4 << 30;
    u32x4* d; if (hipMalloc((void**)&d, bytes) != hipSuccess) return 1;
    printf("TB/s; group memory 40 KB per workgroup (4 per CU) unless said\n");
    printf("%-8s  %-12s  %-12s  %-12s  %-12s  %-12s  %-12s\n", "vmcnt", "973K chunks", "973K, 2/CU", "1M chunks", "4K chunks", "4K sweep1024", "64K chunks");
    row<-1>(d, bytes); row<0>(d, bytes); row<1>(d, bytes); row<2>(d, bytes); row<4>(d, bytes); row<8>(d, bytes);
    return 0;
}
