// How does one SIMD's ChaCha20 (per-lane block form, the limb GEMM's draw pass) throughput depend on the number of resident
// waves?  One workgroup per CU (dynamic LDS pad), W waves per SIMD, every lane computes N dependent blocks.  If 1 wave / SIMD
// already reaches the rate of 2, the draw pass is latency-free VALU work and a wave that stages ALONE runs twice as fast as
// one that shares its SIMD - the premise of splitting the limb-GEMM workgroup into two half-workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_chacha_waves tools/microbench_chacha_waves.hip && tools/microbench_chacha_waves
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../sda_amd/csrc/chacha.hpp"
using namespace sda;

template <int ROUNDS>
__global__ void k(uint32_t* out, int n) {
    extern __shared__ uint8_t pad[];
    const uint32_t key[8] = {1, 2, 3, 4, 5, 6, 7, threadIdx.x};
    uint32_t c = blockIdx.x, acc = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t o[16];
        chacha_block_lane<ROUNDS>(key, c, acc, 7, 9, o);
        acc ^= o[0] ^ o[5] ^ o[10] ^ o[15];
        c += o[1] & 1;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc + pad[0];
}

int main() {
    uint32_t* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 2000, cus = 256;
    hipFuncSetAttribute((const void*)k<20>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int wps = 1; wps <= 4; ++wps) {
        const int threads = 256 * wps;                       // wps waves per SIMD, one workgroup per CU (100 KB of LDS)
        k<20><<<cus, threads, 100 * 1024>>>(d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<20><<<cus, threads, 100 * 1024>>>(d, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double blocks_per_simd = (double)n * 64 * wps;  // lanes x blocks on one SIMD
        printf("waves/SIMD %d: %.3f ms, %.1f ns per block per lane-slot (SIMD: %.2f us per 64 blocks), blocks/s/SIMD %.3e\n", wps, ms,
               ms * 1e6 / n, ms * 1e3 / n / wps, blocks_per_simd / (ms * 1e-3));
    }
    return 0;
}
