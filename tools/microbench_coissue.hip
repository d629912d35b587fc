// Do matrix-core work and vector-ALU work of DIFFERENT waves on one SIMD overlap on gfx950?  Workgroups of 512 threads (two
// waves per SIMD, one workgroup per CU): role M waves issue independent v_mfma_i32_16x16x64_i8, role V waves issue vector
// integer instructions (add / alignbit / xor: the ChaCha20 mix).  Times: M alone, V alone, both side by side, and ONE wave per
// SIMD doing both interleaved in program order (3 vector instructions behind every MFMA).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_coissue.hip -o tools/microbench_coissue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int ITERS = 2048;

// MODE 0: waves 0-3 MFMA, waves 4-7 idle (exit).  1: waves 0-3 exit, 4-7 VALU.  2: both.  3: waves 0-3 do MFMA + VALU interleaved, 4-7 exit
// 4: like 2 but the V waves run 2x the VALU work (is the result max(M, V) or M + V?)
template <int MODE>
__global__ __launch_bounds__(512) void co(uint32_t* out, uint32_t seed) {
    const uint32_t wave = threadIdx.x >> 6;
    const bool mrole = wave < 4;
    v4i a = {(int)seed, (int)threadIdx.x, 3, 4}, b = {5, 6, (int)seed, 8};
    v4i acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v4i{i, 0, 0, 0};
    uint32_t x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = seed + i * 77 + threadIdx.x;
    if (mrole && (MODE == 0 || MODE == 2 || MODE == 4)) {
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
        }
    } else if (!mrole && (MODE == 1 || MODE == 2 || MODE == 4)) {
        const int reps = MODE == 4 ? 2 : 1;
        for (int it = 0; it < ITERS * reps; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {          // 3 vector instructions per slot: 24 per iteration, like the M role's 8 MFMAs x 3
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) % 12]));
                asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x[(i + 4) % 12]));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[(i + 8) % 12]) : "v"(x[i]));
            }
        }
    } else if (mrole && MODE == 3) {
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) % 12]));
                asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x[(i + 4) % 12]));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[(i + 8) % 12]) : "v"(x[i]));
            }
        }
    } else return;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (uint32_t)(acc[i].x + acc[i].y + acc[i].z + acc[i].w);
#pragma unroll
    for (int i = 0; i < 12; ++i) s ^= x[i];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int MODE> static int run(const char* what, uint32_t* d) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    co<MODE><<<256 * 4, 512>>>(d, 1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    co<MODE><<<256 * 4, 512>>>(d, 1);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: 4 workgroups in turn, each ITERS x 8 slots
    const double slots = 4.0 * ITERS * 8;
    printf("%-62s %8.3f ms  %6.2f ns per slot (1 MFMA / 3 VALU)\n", what, ms, ms * 1e6 / slots);
    return 0;
}

int main() {
    uint32_t* d;
    CHK(hipMalloc(&d, 4096));
    if (run<0>("M: one wave per SIMD, 8 independent MFMA 16x16x64 i8 chains", d)) return 1;
    if (run<1>("V: one wave per SIMD, add / alignbit / xor", d)) return 1;
    if (run<2>("M + V on the same SIMD, different waves", d)) return 1;
    if (run<4>("M + 2 x V on the same SIMD, different waves", d)) return 1;
    if (run<3>("one wave per SIMD: MFMA, 3 vector instructions, MFMA, ...", d)) return 1;
    return 0;
}
