// hbm_write_bw2.hip -- store-stream ceiling by store width: 4, 8 and 16 bytes per lane, contiguous per wave-instruction,
// a workgroup owns a contiguous 1 MB chunk (four waves, a quarter each).  ./hbm_write_bw2
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T>
__global__ __launch_bounds__(256) void fill(T* dst, size_t chunk_elems, size_t total) {
    const size_t base = (size_t)blockIdx.x * chunk_elems;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t per = chunk_elems / 4;
    T v; __builtin_memset(&v, 0, sizeof(T)); reinterpret_cast<unsigned*>(&v)[0] = threadIdx.x;
    for (size_t i = lane; i < per; i += 64) {
        const size_t at = base + wave * per + i;
        if (at < total) dst[at] = v;
        reinterpret_cast<unsigned*>(&v)[0] += 1;
    }
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <class T> void run(const char* name, void* d, size_t bytes) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t total = bytes / sizeof(T), chunk = (1u << 20) / sizeof(T);
    const unsigned grid = (unsigned)(total / chunk);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(fill<T>, dim3(grid), dim3(256), 0, 0, (T*)d, chunk, total);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("%-14s %.3f ms  %.1f GB/s\n", name, best, bytes / (best * 1e-3) / 1e9);
}
int main() {
    const size_t bytes = (size_t)4 << 30;
    void* d; if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    run<unsigned>("4 B per lane", d, bytes);
    run<u32x2>("8 B per lane", d, bytes);
    run<u32x4>("16 B per lane", d, bytes);
    { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float best = 1e9f;
      for (int rep = 0; rep < 4; rep++) { hipEventRecord(a); hipMemsetAsync(d, 1, bytes, 0); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms; }
      printf("%-14s %.3f ms  %.1f GB/s\n", "hipMemsetAsync", best, bytes / (best * 1e-3) / 1e9); }
    return 0;
}
