// hbm_write_pat6.hip -- store-stream rate of the PERSISTENT shape considered for the order-table kernel (round 3): G workgroups
// stay resident (G = CUs x wg_per_cu), each takes 1 MB chunks in turn (chunk c, c + G, ...); S waves of the workgroup store
// (interleaved in 1 KB pieces, as the emission loop does), the other waves of the workgroup idle (they stand for the waves that
// build the next unit's image).  Arguments: triples  wg_per_cu threads storing_waves ; lds bytes per workgroup = 150 KB / wg_per_cu.
// Reference line: the non-persistent shape of hbm_write_pat5 (one chunk per workgroup in dispatch order, 4 per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ unsigned char dyn_lds[];
__global__ __launch_bounds__(1024) void fill_persist(u32x4* dst, size_t chunk_vec, size_t nchunks, int storing) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    dyn_lds[threadIdx.x] = 1;
    if (wave >= storing) return;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t base = c * chunk_vec;
        for (size_t i = (size_t)wave * 64 + lane; i < chunk_vec; i += (size_t)storing * 64) { dst[base + i] = v; v.x++; }
    }
}
__global__ __launch_bounds__(256) void fill_chunks(u32x4* dst, size_t chunk_vec, size_t nchunks) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    dyn_lds[threadIdx.x] = 1;
    const size_t base = (size_t)blockIdx.x * chunk_vec;
    for (size_t i = wave * 64 + lane; i < chunk_vec; i += 256) { dst[base + i] = v; v.x++; }
}
int main(int argc, char** argv) {
    const size_t bytes = (size_t)4 << 30, chunk = 1 << 20, nchunks = bytes / chunk;
    u32x4* d; if (hipMalloc((void**)&d, bytes + (1 << 20)) != hipSuccess) return 1;
    (void)hipFuncSetAttribute((const void*)fill_persist, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)fill_chunks, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int round = 0; round < 2; round++) {
        {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(fill_chunks, dim3((unsigned)nchunks), dim3(256), 40960, 0, d, chunk / 16, nchunks);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("reference (1 chunk per workgroup, 4 per CU, dispatch order)   %.2f TB/s\n", bytes / (best * 1e-3) / 1e12);
        }
        for (int i = 1; i + 2 < argc; i += 3) {
            const int per = atoi(argv[i]), threads = atoi(argv[i + 1]), storing = atoi(argv[i + 2]);
            const int lds = (150 * 1024 / per) & ~255;
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(fill_persist, dim3((unsigned)(cus * per)), dim3(threads), lds, 0, d, chunk / 16, nchunks, storing);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("persistent: %d workgroup(s) per CU x %4d threads, %d storing waves, %6d B LDS   %.2f TB/s\n", per, threads, storing, lds, bytes / (best * 1e-3) / 1e12);
            fflush(stdout);
        }
    }
    return 0;
}
