// hbm_write_pat5.hip -- store-stream rate against the ALIGNMENT / size of the contiguous chunk a workgroup owns (chunk
// sizes in bytes on the command line; one chunk per workgroup in dispatch order, 40 KB of group memory = 4 per CU,
// four waves interleaved in 1 KB pieces).  Optional first argument "-o BYTES": the whole stream starts BYTES into the buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ unsigned char dyn_lds[];
__global__ __launch_bounds__(256) void fill_chunks(u32x4* dst, size_t chunk_vec, size_t nchunks) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    dyn_lds[threadIdx.x] = 1;
    const size_t base = (size_t)blockIdx.x * chunk_vec;
    for (size_t i = wave * 64 + lane; i < chunk_vec; i += 256) { dst[base + i] = v; v.x++; }
}
int main(int argc, char** argv) {
    const size_t bytes = (size_t)4 << 30;
    u32x4* d; if (hipMalloc((void**)&d, bytes + (1 << 20)) != hipSuccess) return 1;
    (void)hipFuncSetAttribute((const void*)fill_chunks, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    size_t off = 0;
    int a = 1;
    if (argc > 2 && !strcmp(argv[1], "-o")) { off = strtoull(argv[2], nullptr, 10); a = 3; }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int round = 0; round < 2; round++)
        for (int i = a; i < argc; i++) {
            const size_t chunk = strtoull(argv[i], nullptr, 10), nchunks = bytes / chunk;
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(fill_chunks, dim3((unsigned)nchunks), dim3(256), 40960, 0, d + off / 16, chunk / 16, nchunks);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("chunk %9zu (+%zu)  %.2f TB/s\n", chunk, off, nchunks * chunk / (best * 1e-3) / 1e12); fflush(stdout);
        }
    return 0;
}
