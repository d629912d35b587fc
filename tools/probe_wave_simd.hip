// Which SIMD does wave w of a workgroup run on?  (HW_REG_HW_ID, gfx9: bits 5:4 = SIMD, 3:0 = wave slot, 11:8 = CU.)  One workgroup per CU
// (150 KB of LDS), 9 or 12 waves with the limb GEMM's register footprint (amdgpu_waves_per_eu(3,3)).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_wave_simd.hip -o tools/probe_wave_simd
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(3, 3))) void k(unsigned* out) {
    extern __shared__ unsigned lds[];
    if (threadIdx.x == 0) lds[0] = 1;
    __syncthreads();
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * WAVES + (threadIdx.x >> 6)] = id;
}
template <int WAVES> void run() {
    unsigned* d; hipMalloc(&d, 64 * WAVES * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    k<WAVES><<<8, 64 * WAVES, 150 * 1024>>>(d);
    unsigned h[8 * WAVES]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) {
        printf("%d waves, workgroup %d: simd of wave 0..%d =", WAVES, b, WAVES - 1);
        for (int w = 0; w < WAVES; ++w) printf(" %u", (h[b * WAVES + w] >> 4) & 3);
        printf("   (cu %u)\n", (h[b * WAVES] >> 8) & 15);
    }
    hipFree(d);
}
int main() { run<9>(); run<12>(); run<10>(); run<11>(); return 0; }
