// Development probe: which compute unit does bit b of a stream's CU mask (hipExtStreamCreateWithCUMask) select?
// For every bit: a stream with only that bit, one small kernel that records (XCC_ID, HW_ID) of its wavefronts.
//   hipcc --offload-arch=gfx950 -O3 -o tools/cumask_probe tools/cumask_probe.hip && tools/cumask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void whereami(unsigned *out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

int main() {
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int words = (cus + 31) / 32;
    unsigned *out;
    CK(hipMalloc(&out, 8 * 64));
    printf("%d CUs, mask of %d words\n", cus, words);
    for (int b = 0; b < cus; ++b) {
        std::vector<uint32_t> mask(words, 0);
        mask[b / 32] = 1u << (b % 32);
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, words, mask.data());
        if (e != hipSuccess) { printf("bit %3d: create failed: %s\n", b, hipGetErrorString(e)); continue; }
        CK(hipMemsetAsync(out, 0xff, 8 * 64, st));
        hipLaunchKernelGGL(whereami, dim3(16), dim3(64), 0, st, out);
        CK(hipStreamSynchronize(st));
        unsigned h[32];
        CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        // HW_ID (gfx9): [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se
        bool same = true;
        for (int k = 1; k < 16; ++k) same = same && h[2 * k] == h[0] && ((h[2 * k + 1] >> 8) & 0xff) == ((h[1] >> 8) & 0xff);
        printf("bit %3d -> xcc %u  se %u sh %u cu %2u %s\n", b, h[0] & 0xf, (h[1] >> 13) & 7, (h[1] >> 12) & 1, (h[1] >> 8) & 0xf, same ? "" : "(blocks landed on several CUs!)");
        CK(hipStreamDestroy(st));
    }
    return 0;
}
