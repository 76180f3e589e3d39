// cu_mask_probe.hip -- does a stream created with hipExtStreamCreateWithCUMask confine its kernels to the masked CUs on this
// chip, and how do mask bits map to (XCC, SE, CU)?  Every workgroup records the hardware ids it ran on.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
__global__ void where(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
        for (int i = 0; i < 2000; i++) asm volatile("s_sleep 10");   // keep the workgroup resident for a while
    }
}
static int run(hipStream_t s, unsigned* d, std::vector<unsigned>& h, int n) {
    hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, s, d);
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus;
    for (int i = 0; i < n; i++) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
    }
    return (int)cus.size();
}
int main(int argc, char** argv) {
    const int n = 8192;
    unsigned* d; (void)hipMalloc((void**)&d, 2 * n * 4);
    std::vector<unsigned> h(2 * n);
    hipStream_t plain; (void)hipStreamCreate(&plain);
    printf("no mask: %d distinct (xcc, se, sh, cu)\n", run(plain, d, h, n));
    for (int variant = 0; variant < 4; variant++) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const char* what = "";
        if (variant == 0) { mask[0] = 0xFFFFFFFFu; what = "bits 0-31"; }
        if (variant == 1) { for (int i = 0; i < 8; i++) mask[i] = 0x000000FFu; what = "bits 0-7 of every dword"; }
        if (variant == 2) { for (int i = 0; i < 8; i++) mask[i] = 0x11111111u; what = "every 4th bit"; }
        if (variant == 3) { mask[0] = mask[1] = 0xFFFFFFFFu; what = "bits 0-63"; }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", what, hipGetErrorString(e)); continue; }
        const int c = run(s, d, h, n);
        std::set<unsigned> xccs; for (int i = 0; i < n; i++) xccs.insert(h[2 * i + 1] & 0xF);
        printf("%-26s: %d distinct CUs on %zu XCCs\n", what, c, xccs.size());
        (void)hipStreamDestroy(s);
    }
    return 0;
}
