// Where does global_load_lds_dwordx4 put the 16 bytes of lane i?  (gfx950 LDS-DMA layout probe)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_glds.hip -o tools/probe_glds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const uint32_t* g, uint32_t* out) {
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 0xFFFFFFFFu;
    __syncthreads();
    if (wave == 1) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + lane * 4),
                                         (__attribute__((address_space(3))) void*)(lds + 256), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = lds[i];
}
int main() {
    uint32_t h[256], *d, *o, r[1024];
    for (int i = 0; i < 256; ++i) h[i] = i;        // dword j of lane i = 4 i + j
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 128, 4096>>>(d, o);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int first = -1, linear = 1;
    for (int i = 0; i < 1024; ++i) if (r[i] != 0xFFFFFFFFu) { if (first < 0) first = i; }
    printf("first written dword index %d (expected 256)\n", first);
    for (int i = 0; i < 256; ++i) if (r[256 + i] != (uint32_t)i) linear = 0;
    printf("layout lane-contiguous 16 bytes (base + 16 lane): %s\n", linear ? "yes" : "no");
    for (int i = 0; i < 16; ++i) printf("%u ", r[256 + i]); printf("\n");
    for (int i = 64; i < 80; ++i) printf("%u ", r[256 + i]); printf("\n");
    return 0;
}
