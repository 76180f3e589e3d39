// hbm_write_pat3.hip -- store-stream rate against (a) the size of the contiguous chunk a workgroup owns, (b) how many
// workgroups a CU holds (dynamic group memory as the limiter, as in the enumerate kernel), (c) persistent workgroups
// that sweep memory together in chunks.  4 GB of 16-byte-per-lane stores, four waves of a workgroup interleaved in 1 KB
// pieces inside the chunk.
//   C  chunk per workgroup, grid = bytes / chunk (dispatch order = address order)
//   S  `grid` persistent workgroups, workgroup i writes chunks i, i + grid, ...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ unsigned char dyn_lds[];
__global__ __launch_bounds__(256) void fill_chunks(u32x4* dst, size_t chunk_vec, size_t nchunks, int touch) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    if (touch) dyn_lds[threadIdx.x] = 1;
    const size_t npieces = chunk_vec / 64;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t base = c * chunk_vec;
        for (size_t p = wave; p < npieces; p += 4) { dst[base + p * 64 + lane] = v; v.x++; }
    }
}
static float run(u32x4* d, size_t bytes, size_t chunk, unsigned grid, unsigned lds) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const size_t chunk_vec = chunk / 16, nchunks = bytes / chunk;
    if (grid == 0) grid = (unsigned)nchunks;
    (void)hipFuncSetAttribute((const void*)fill_chunks, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(fill_chunks, dim3(grid), dim3(256), lds, 0, d, chunk_vec, nchunks, lds ? 1 : 0);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}
int main() {
    const size_t bytes = (size_t)4 << 30;
    u32x4* d; if (hipMalloc((void**)&d, bytes) != hipSuccess) return 1;
    const unsigned ldss[4] = {0, 40 * 1024, 80 * 1024, 160 * 1024 - 512};
    const size_t chunks[6] = {(size_t)1 << 20, (size_t)256 << 10, (size_t)64 << 10, (size_t)16 << 10, (size_t)4 << 10, (size_t)973 << 10};
    printf("C: one chunk per workgroup, dispatch order (TB/s)\n%-10s", "chunk");
    for (unsigned l : ldss) printf("  lds %-6u", l);
    printf("\n");
    for (size_t c : chunks) {
        printf("%-10zu", c);
        for (unsigned l : ldss) { const float ms = run(d, bytes - bytes % c, c, 0, l); printf("  %-10.2f", (bytes - bytes % c) / (ms * 1e-3) / 1e12); }
        printf("\n"); fflush(stdout);
    }
    printf("S: persistent workgroups sweep together (TB/s)\n%-10s", "chunk");
    const unsigned grids[3] = {512, 1024, 2048};
    for (unsigned g : grids) printf("  grid %-5u", g);
    printf("  (lds: 80K, 40K, 20K)\n");
    for (size_t c : chunks) {
        printf("%-10zu", c);
        for (int i = 0; i < 3; i++) { const float ms = run(d, bytes - bytes % c, c, grids[i], (80 * 1024) >> i); printf("  %-10.2f", (bytes - bytes % c) / (ms * 1e-3) / 1e12); }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
