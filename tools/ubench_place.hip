// (diagnosis) where the waves of a workgroup land: SIMD and CU id per wave for workgroups of 448 / 512 threads with the
// dynamic LDS the cooperative tracking launch asks for (one workgroup per CU).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_place.hip -o /tmp/ubench_place && /tmp/ubench_place
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void place_kernel(unsigned* out) {
    extern __shared__ char pad[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = (hw & 0xFFFFu) | ((xcc & 0xF) << 16);
    if (threadIdx.x == 0) pad[0] = 1;
}

int main() {
    for (int threads : {448, 512}) {
        unsigned* d;
        const int nb = 160;
        (void)hipMalloc(&d, nb * 16 * 4);
        (void)hipMemset(d, 0xFF, nb * 16 * 4);
        (void)hipFuncSetAttribute((const void*)place_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 90112);
        place_kernel<<<nb, threads, 90112>>>(d);
        (void)hipDeviceSynchronize();
        static unsigned h[160 * 16];
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int hist[16][4] = {};
        int rr = 0;
        for (int b = 0; b < nb; ++b) {
            bool ok = true;
            for (int w = 0; w < threads / 64; ++w) {
                const unsigned v = h[b * 16 + w];
                const int simd = (v >> 4) & 3;
                hist[w][simd]++;
                if (simd != (int)(((h[b * 16] >> 4) & 3) + w) % 4) ok = false;
            }
            rr += ok;
            if (b < 6) {
                printf("threads %d block %3d xcc %u cu %2u:", threads, b, (h[b * 16] >> 16) & 0xF, (h[b * 16] >> 8) & 0xF);
                for (int w = 0; w < threads / 64; ++w) printf(" w%d:simd%u", w, (h[b * 16 + w] >> 4) & 3);
                printf("\n");
            }
        }
        printf("threads %d: %d of %d workgroups place wave w on SIMD (first + w) mod 4; histogram wave x simd:\n", threads, rr, nb);
        for (int w = 0; w < threads / 64; ++w) printf("   wave %d: %3d %3d %3d %3d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        (void)hipFree(d);
    }
    return 0;
}
