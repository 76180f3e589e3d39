// Store-stream ceiling for the shape of the ILP fill kernel: N entries as two arrays (int32 column + f64 coefficient), a workgroup
// of 256 threads per 1024 consecutive entries, every thread one 16-byte store into the first array and two into the second --
// against the same bytes into ONE array.   hipcc --offload-arch=gfx950 -O3 -o hbm_write_two_streams hbm_write_two_streams.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void two(int32_t* col, double* val, int64_t n) {
    const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p + 4 > n) return;
    const int v = (int)p;
    *reinterpret_cast<int4*>(col + p) = make_int4(v, v + 1, v + 2, v + 3);
    *reinterpret_cast<double2*>(val + p) = make_double2(1.0, 2.0);
    *reinterpret_cast<double2*>(val + p + 2) = make_double2(0.5, -1.0);
}
__global__ __launch_bounds__(256) void one(int4* out, int64_t n16) {   // the same 48 bytes per thread, contiguous per workgroup
    const int64_t g = (int64_t)blockIdx.x * 768 + threadIdx.x;
    const int v = (int)g;
    for (int k = 0; k < 3; k++) if (g + 256 * k < n16) out[g + 256 * k] = make_int4(v, v + 1, v + 2, v + 3);
}
__global__ __launch_bounds__(256) void only_val(double* val, int64_t n) {
    const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p + 4 > n) return;
    *reinterpret_cast<double2*>(val + p) = make_double2(1.0, 2.0);
    *reinterpret_cast<double2*>(val + p + 2) = make_double2(0.5, -1.0);
}
int main() {
    const int64_t n = 56546944;   // entries (a multiple of 1024)
    int32_t* col; double* val; int4* flat;
    hipMalloc(&col, n * 4); hipMalloc(&val, n * 8); hipMalloc(&flat, n * 12);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned grid = (unsigned)(n / 1024);
    for (int which = 0; which < 3; which++) {
        float best = 1e9f;
        for (int r = 0; r < 6; r++) {
            hipEventRecord(a);
            if (which == 0) hipLaunchKernelGGL(two, dim3(grid), dim3(256), 0, 0, col, val, n);
            else if (which == 1) hipLaunchKernelGGL(one, dim3(grid), dim3(256), 0, 0, flat, n * 12 / 16);
            else hipLaunchKernelGGL(only_val, dim3(grid), dim3(256), 0, 0, val, n);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (r > 0 && ms < best) best = ms;
        }
        const double bytes = which == 2 ? n * 8.0 : n * 12.0;
        printf("%-34s %.4f ms  %.0f GB/s\n", which == 0 ? "two arrays (int32 + f64)" : which == 1 ? "one array, same bytes" : "the f64 array alone", best, bytes / (best * 1e-3) / 1e9);
    }
    return 0;
}
