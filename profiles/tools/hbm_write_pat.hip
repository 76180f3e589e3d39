// hbm_write_pat.hip -- does the ORDER in which a store stream covers memory matter?  4 GB of 16-byte-per-lane stores:
//   P0  a workgroup owns a contiguous 1 MB chunk, its four waves a quarter each (the enumerate kernel's shape)
//   P1  the same chunk, the four waves interleaved in 1 KB pieces (one compact window per workgroup)
//   P2  the same chunk, waves interleaved in 4 KB pieces
//   P3  grid-stride: workgroup i writes the 4 KB pieces i, i + grid, ... (all workgroups sweep memory together)
//   P4  hipMemsetAsync (the runtime's fill kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int P>
__global__ __launch_bounds__(256) void fill(u32x4* dst, size_t total_vec, size_t chunk_vec) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    if (P == 0) {
        const size_t base = (size_t)blockIdx.x * chunk_vec, per = chunk_vec / 4;
        for (size_t i = lane; i < per; i += 64) { dst[base + wave * per + i] = v; v.x++; }
    } else if (P == 1 || P == 2) {
        const size_t piece = P == 1 ? 64 : 256;                       // vectors per piece: 1 KB / 4 KB
        const size_t base = (size_t)blockIdx.x * chunk_vec, npieces = chunk_vec / piece;
        for (size_t p = wave; p < npieces; p += 4)
            for (size_t i = lane; i < piece; i += 64) { dst[base + p * piece + i] = v; v.x++; }
    } else {
        const size_t piece = 256, npieces = total_vec / piece;
        for (size_t p = blockIdx.x; p < npieces; p += gridDim.x) { dst[p * piece + threadIdx.x] = v; v.x++; }
    }
}
template <int P> float run(u32x4* d, size_t bytes, unsigned grid) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const size_t total_vec = bytes / 16, chunk_vec = total_vec / grid;
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(fill<P>, dim3(grid), dim3(256), 0, 0, d, total_vec, chunk_vec);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}
int main() {
    const size_t bytes = (size_t)4 << 30;
    u32x4* d; if (hipMalloc((void**)&d, bytes) != hipSuccess) return 1;
    const char* nm[4] = {"P0 quarters", "P1 1 KB interleave", "P2 4 KB interleave", "P3 grid-stride 4 KB"};
    float ms[4] = {run<0>(d, bytes, 4096), run<1>(d, bytes, 4096), run<2>(d, bytes, 4096), run<3>(d, bytes, 4096)};
    for (int i = 0; i < 4; i++) printf("%-22s %.3f ms  %.1f GB/s\n", nm[i], ms[i], bytes / (ms[i] * 1e-3) / 1e9);
    printf("%-22s %.3f ms\n", "P3 grid 1024", run<3>(d, bytes, 1024));
    printf("%-22s %.3f ms\n", "P3 grid 16384", run<3>(d, bytes, 16384));
    printf("%-22s %.3f ms\n", "P3 grid 65536", run<3>(d, bytes, 65536));
    printf("%-22s %.3f ms\n", "P3 grid 262144", run<3>(d, bytes, 262144));
    printf("%-22s %.3f ms\n", "P3 grid 1048576 (one 4 KB piece per workgroup)", run<3>(d, bytes, 1048576));
    printf("%-22s %.3f ms\n", "P2 grid 16384 (256 KB chunks)", run<2>(d, bytes, 16384));
    printf("%-22s %.3f ms\n", "P2 grid 65536 (64 KB chunks)", run<2>(d, bytes, 65536));
    printf("%-22s %.3f ms\n", "P0 grid 16384 (256 KB chunks)", run<0>(d, bytes, 16384));
    { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); float best = 1e9f;
      for (int rep = 0; rep < 4; rep++) { (void)hipEventRecord(a); (void)hipMemsetAsync(d, 1, bytes, 0); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float t; (void)hipEventElapsedTime(&t, a, b); if (rep && t < best) best = t; }
      printf("%-22s %.3f ms  %.1f GB/s\n", "P4 hipMemsetAsync", best, bytes / (best * 1e-3) / 1e9); }
    return 0;
}
