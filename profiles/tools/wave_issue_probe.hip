// wave_issue_probe.hip -- how fast does ONE wavefront, alone on a CU, get through serial code?  (MI355X, gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o wave_issue_probe wave_issue_probe.hip && ./wave_issue_probe
// Every test runs a loop of `iters` iterations x 16 operations inside one wavefront and reports shader-clock cycles
// (s_memtime) per operation: dependent VALU adds, independent VALU adds, dependent SALU adds, a v_readlane -> VALU
// chain, a dependent LDS load chain (pointer chasing), an LDS CAS chain, and a dependent global-load chain through L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(long long* out, int iters, int* chase_g) {
    __shared__ int lds[1024];
    __shared__ unsigned long long lds64[256];
    const int t = threadIdx.x;
    for (int i = t; i < 1024; i += 64) lds[i] = (i * 17 + 5) & 1023;
    for (int i = t; i < 256; i += 64) lds64[i] = ~0ull;
    __syncthreads();
    long long c0, c1;
    int x = t, y = t + 1, z = t + 2, w = t + 3;
    // 1. dependent VALU
    c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) x = x * 3 + k;   // v_mad / v_mul_lo chain -> use add/xor instead below
    }
    c1 = clock64();
    if (t == 0) out[0] = c1 - c0;
    // 1b. dependent VALU add/xor only
    c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) y = (y ^ (k + 1)) + x;
    }
    c1 = clock64();
    if (t == 0) out[1] = c1 - c0;
    // 2. four independent VALU chains
    c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) { x += k; y ^= x >> 31; z += 3; w ^= 5; x += 1; y += 1; z ^= k; w += 7; }
    }
    c1 = clock64();
    if (t == 0) out[2] = c1 - c0;   // 32 ops per iteration
    // 3. readlane -> VALU chain
    c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) { int s = __builtin_amdgcn_readlane(x, k); x = (x & ~s) + 1; }
    }
    c1 = clock64();
    if (t == 0) out[3] = c1 - c0;   // 16 x (readlane + 2 valu)
    // 4. dependent LDS loads
    int p = t & 1023;
    c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) p = lds[p];
    }
    c1 = clock64();
    if (t == 0) out[4] = c1 - c0;
    // 5. LDS 64-bit CAS chain
    unsigned long long key = t;
    c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) key = atomicCAS(&lds64[(t * 4 + k) & 255], ~0ull, key + 1) + k;
    }
    c1 = clock64();
    if (t == 0) out[5] = c1 - c0;
    // 6. dependent global loads (L2 hits)
    int q = t;
    c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) q = chase_g[q];
    }
    c1 = clock64();
    if (t == 0) out[6] = c1 - c0;
    // 7. uniform branchy scalar code: a data-dependent loop on scalar values (SALU)
    int s = iters;
    c0 = clock64();
    int acc = 0;
    for (int i = 0; i < iters * 16; i++) { acc += (i & s) ? 3 : 1; s ^= acc; }
    c1 = clock64();
    if (t == 0) out[7] = c1 - c0;
    out[8 + t] = x + y + z + w + p + (int)key + q + acc;
}

int main() {
    long long* d; int* g;
    hipMalloc(&d, 128 * sizeof(long long));
    std::vector<int> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = (i * 29 + 7) & 4095;
    hipMalloc(&g, 4096 * sizeof(int));
    hipMemcpy(g, h.data(), 4096 * sizeof(int), hipMemcpyHostToDevice);
    const int iters = 256;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, iters, g);
    hipDeviceSynchronize();
    long long o[8];
    hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
    const char* nm[8] = {"dependent v_mul/v_mad", "dependent v_xor+v_add (2 ops)", "4 independent VALU chains (32 ops)", "v_readlane -> 2 VALU", "dependent LDS load",
                         "LDS 64-bit CAS (returning)", "dependent global load (L2)", "scalar loop iteration (~5 SALU)"};
    const double per[8] = {16, 32, 32, 16, 16, 16, 16, 16};
    for (int i = 0; i < 8; i++) printf("%-40s %8.1f cycles per op\n", nm[i], (double)o[i] / (iters * per[i]));
    return 0;
}
