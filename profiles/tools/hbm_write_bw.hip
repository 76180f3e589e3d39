// hbm_write_bw.hip -- ceiling probe for the enumerate kernel: how fast can MI355X take a pure 16-byte-per-lane store
// stream?  Variants: plain / nontemporal stores, workgroup-contiguous chunks of several sizes.
// Build: hipcc -O3 --offload-arch=gfx950 -o hbm_write_bw hbm_write_bw.hip ; run: ./hbm_write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(256) void fill_kernel(u32x4* dst, size_t chunk_vec, size_t total_vec) {
    // every workgroup owns one contiguous chunk (like one unit's order table); waves split it in quarters
    const size_t base = (size_t)blockIdx.x * chunk_vec;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t per = chunk_vec / 4;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (size_t i = lane; i < per; i += 64) {
        const size_t at = base + wave * per + i;
        if (at < total_vec) {
            if (NT) __builtin_nontemporal_store(v, dst + at);
            else dst[at] = v;
        }
        v.x += 1;
    }
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    u32x4* d;
    if (hipMalloc((void**)&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const size_t total_vec = bytes / 16;
    for (int nt = 0; nt < 2; nt++) {
        for (size_t chunk_kb : {256, 1024, 4096}) {
            const size_t chunk_vec = chunk_kb * 1024 / 16;
            const unsigned grid = (unsigned)(total_vec / chunk_vec);
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                if (nt) hipLaunchKernelGGL(fill_kernel<1>, dim3(grid), dim3(256), 0, 0, d, chunk_vec, total_vec);
                else hipLaunchKernelGGL(fill_kernel<0>, dim3(grid), dim3(256), 0, 0, d, chunk_vec, total_vec);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms = 0;
                hipEventElapsedTime(&ms, a, b);
                if (rep == 2) printf("nt=%d chunk=%zuKB grid=%u: %.3f ms  %.1f GB/s\n", nt, chunk_kb, grid, ms, bytes / (ms * 1e-3) / 1e9);
            }
        }
    }
    hipFree(d);
    return 0;
}
