// Probe of the gfx950 semantics the limb-GEMM share-gen kernel relies on (run on the GPU box):
//   1. v_permlane32_swap / v_permlane16_swap: which lanes trade places
//   2. v_mfma_i32_16x16x64_i8: D[i][j] = sum over the 64 k-slots of A[i][k] * B[k][j] with lane (x = lane & 15, g = lane >> 4)
//      holding k-slots (g, byte 0..15) of row/column x, and D row 4 g + reg, column lane & 15
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__global__ void probe(int* out_swap, int* out_d, const int8_t* A, const int8_t* B) {
    const int lane = threadIdx.x;
    unsigned x = lane, y = 100 + lane;
    v2u r32 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    v2u r16 = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    out_swap[lane] = r32.x; out_swap[64 + lane] = r32.y; out_swap[128 + lane] = r16.x; out_swap[192 + lane] = r16.y;
    v4i a, b, c = {0, 0, 0, 0};
    const int* pa = reinterpret_cast<const int*>(A + lane * 16);
    const int* pb = reinterpret_cast<const int*>(B + lane * 16);
    a.x = pa[0]; a.y = pa[1]; a.z = pa[2]; a.w = pa[3];
    b.x = pb[0]; b.y = pb[1]; b.z = pb[2]; b.w = pb[3];
    v4i d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    out_d[lane * 4 + 0] = d.x; out_d[lane * 4 + 1] = d.y; out_d[lane * 4 + 2] = d.z; out_d[lane * 4 + 3] = d.w;
}

int main() {
    int8_t hA[64 * 16], hB[64 * 16];
    uint32_t s = 12345;
    for (int i = 0; i < 64 * 16; ++i) { s = s * 1664525u + 1013904223u; hA[i] = (int8_t)(s >> 24); s = s * 1664525u + 1013904223u; hB[i] = (int8_t)(s >> 24); }
    int8_t *dA, *dB; int *dS, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dS, 256 * 4); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dS, dD, dA, dB);
    int hS[256], hD[256];
    hipMemcpy(hS, dS, sizeof hS, hipMemcpyDeviceToHost); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane32_swap .x (first operand after)", "permlane32_swap .y (second operand after)",
                            "permlane16_swap .x", "permlane16_swap .y"};
    for (int k = 0; k < 4; ++k) { printf("%s:\n ", names[k]); for (int l = 0; l < 64; ++l) printf("%d%s", hS[64 * k + l], l % 16 == 15 ? "\n " : " "); printf("\n"); }
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int reg = 0; reg < 4; ++reg) {
            const int j = lane & 15, i = 4 * (lane >> 4) + reg;
            long want = 0;
            for (int g = 0; g < 4; ++g) for (int q = 0; q < 16; ++q) want += (long)hA[(i + 16 * g) * 16 + q] * hB[(j + 16 * g) * 16 + q];
            if (want != hD[lane * 4 + reg]) { if (bad < 5) printf("D mismatch lane %d reg %d: got %d want %ld\n", lane, reg, hD[lane * 4 + reg], want); ++bad; }
        }
    printf("mfma_i32_16x16x64_i8 layout assumption: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    return 0;
}
