// How soon after a 16-byte vector store may its data register be written again?  (round 5, after the limb GEMM's intermittent
// wrong shares: DESIGN.md 4 "Narrow limb GEMM" (5))
//
// Each thread stores {x, 0, y, 0} for it = 0 .. ITERS - 1 to a different place, with x = value(it); K wait states after the store
// the SAME register receives value(it + 1) - one inline-asm block, so that nothing but s_nop K - 1 stands between the store and
// the v_mov.  A store that reads its data late leaves value(it + 1) in memory.  The ISA manual asks for 1 wait state after a store
// of more than 64 bits (2 with an SGPR offset); the compiler inserts exactly that.  Run with enough threads to back the memory
// pipeline up.   hipcc --offload-arch=gfx950 -O2 -o microbench_store_war microbench_store_war.hip && ./microbench_store_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__host__ __device__ __forceinline__ uint32_t value(uint32_t gid, uint32_t it) { return gid * 2654435761u + it * 40503u + 1u; }

template <int K, int FORM>       // FORM 0: buffer_store_dwordx4 offen + SGPR soffset, nt; 1: global_store_dwordx4 nt; 2: buffer store, soffset 0
__global__ __launch_bounds__(256) void war_kernel(v4u* out, uint32_t iters, uint32_t threads) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint64_t base = (uint64_t)out;
    v4i rs = {(int)(uint32_t)base, (int)((uint32_t)(base >> 32) & 0xFFFFu), (int)0xFFFFFFFFu, (int)0x00020000};
    rs.x = __builtin_amdgcn_readfirstlane(rs.x); rs.y = __builtin_amdgcn_readfirstlane(rs.y);
    const uint32_t voff = gid * 16u;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t x = value(gid, it), nx = value(gid, it + 1);
        const uint32_t soff = __builtin_amdgcn_readfirstlane(it * threads * 16u);      // (iters * threads * 16 < 4 GiB)
        // the data lives in v[20:23] by name, so that the overwrite of its first register can stand exactly K wait states behind
        // the store (no compiler between them)
#define SETUP "v_mov_b32 v20, %[x]\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, %[y]\n\tv_mov_b32 v23, 0\n\ts_nop 4\n\t"
        if (FORM == 0 || FORM == 2) {
            const uint32_t so = FORM == 0 ? soff : 0u;
            const uint32_t vo = FORM == 0 ? voff : voff + soff;
            if (K == 0)
                asm volatile(SETUP "buffer_store_dwordx4 v[20:23], %[vo], %[rs], %[so] offen nt\n\tv_mov_b32 v20, %[nx]"
                             :: [x] "v"(x), [y] "v"(~x), [vo] "v"(vo), [rs] "s"(rs), [so] "s"(so), [nx] "v"(nx) : "memory", "v20", "v21", "v22", "v23");
            else
                asm volatile(SETUP "buffer_store_dwordx4 v[20:23], %[vo], %[rs], %[so] offen nt\n\ts_nop %[k]\n\tv_mov_b32 v20, %[nx]"
                             :: [x] "v"(x), [y] "v"(~x), [vo] "v"(vo), [rs] "s"(rs), [so] "s"(so), [nx] "v"(nx), [k] "n"(K > 0 ? K - 1 : 0) : "memory", "v20", "v21", "v22", "v23");
        } else if (FORM == 3 || FORM == 4) {                        // 8-byte stores {x, ~x} (no documented hazard)
            v4u* p = out + (size_t)it * threads + gid;
            if (FORM == 3) {
                if (K == 0)
                    asm volatile(SETUP "global_store_dwordx2 %[p], v[20:21], off nt\n\tv_mov_b32 v20, %[nx]"
                                 :: [x] "v"(x), [y] "v"(~x), [p] "v"(p), [nx] "v"(nx) : "memory", "v20", "v21", "v22", "v23");
                else
                    asm volatile(SETUP "global_store_dwordx2 %[p], v[20:21], off nt\n\ts_nop %[k]\n\tv_mov_b32 v20, %[nx]"
                                 :: [x] "v"(x), [y] "v"(~x), [p] "v"(p), [nx] "v"(nx), [k] "n"(K > 0 ? K - 1 : 0) : "memory", "v20", "v21", "v22", "v23");
            } else {
                if (K == 0)
                    asm volatile(SETUP "buffer_store_dwordx2 v[20:21], %[vo], %[rs], %[so] offen nt\n\tv_mov_b32 v20, %[nx]"
                                 :: [x] "v"(x), [y] "v"(~x), [vo] "v"(voff), [rs] "s"(rs), [so] "s"(soff), [nx] "v"(nx) : "memory", "v20", "v21", "v22", "v23");
                else
                    asm volatile(SETUP "buffer_store_dwordx2 v[20:21], %[vo], %[rs], %[so] offen nt\n\ts_nop %[k]\n\tv_mov_b32 v20, %[nx]"
                                 :: [x] "v"(x), [y] "v"(~x), [vo] "v"(voff), [rs] "s"(rs), [so] "s"(soff), [nx] "v"(nx), [k] "n"(K > 0 ? K - 1 : 0) : "memory", "v20", "v21", "v22", "v23");
            }
        } else {
            v4u* p = out + (size_t)it * threads + gid;
            if (K == 0)
                asm volatile(SETUP "global_store_dwordx4 %[p], v[20:23], off nt\n\tv_mov_b32 v20, %[nx]"
                             :: [x] "v"(x), [y] "v"(~x), [p] "v"(p), [nx] "v"(nx) : "memory", "v20", "v21", "v22", "v23");
            else
                asm volatile(SETUP "global_store_dwordx4 %[p], v[20:23], off nt\n\ts_nop %[k]\n\tv_mov_b32 v20, %[nx]"
                             :: [x] "v"(x), [y] "v"(~x), [p] "v"(p), [nx] "v"(nx), [k] "n"(K > 0 ? K - 1 : 0) : "memory", "v20", "v21", "v22", "v23");
        }
    }
}

template <int K, int FORM> int run(v4u* d_out, std::vector<v4u>& h, uint32_t iters, uint32_t threads) {
    CK(hipMemset(d_out, 0, (size_t)iters * threads * 16));
    hipLaunchKernelGGL((war_kernel<K, FORM>), dim3(threads / 256), dim3(256), 0, 0, d_out, iters, threads);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d_out, (size_t)iters * threads * 16, hipMemcpyDeviceToHost));
    size_t next = 0, other = 0;
    for (uint32_t it = 0; it < iters; ++it)
        for (uint32_t g = 0; g < threads; ++g) {
            const v4u v = h[(size_t)it * threads + g];
            if (v.x == value(g, it)) continue;
            if (v.x == value(g, it + 1)) ++next; else ++other;
        }
    printf("  %-46s wait states %2d: %zu stores carry the NEXT value, %zu something else (of %zu)\n",
           FORM == 0 ? "buffer_store_dwordx4 offen, SGPR soffset, nt" : FORM == 1 ? "global_store_dwordx4 nt" : FORM == 2 ? "buffer_store_dwordx4 offen, soffset 0, nt" :
           FORM == 3 ? "global_store_dwordx2 nt" : "buffer_store_dwordx2 offen, SGPR soffset, nt", K, next, other,
           (size_t)iters * threads);
    return 0;
}

int main() {
    const uint32_t threads = 256 * 256 * 8, iters = 256;          // 2 Gi bytes per run
    v4u* d_out;
    CK(hipMalloc(&d_out, (size_t)iters * threads * 16));
    std::vector<v4u> h((size_t)iters * threads);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>(d_out, h, iters, threads); run<1, 0>(d_out, h, iters, threads); run<2, 0>(d_out, h, iters, threads); run<3, 0>(d_out, h, iters, threads);
        run<4, 0>(d_out, h, iters, threads); run<8, 0>(d_out, h, iters, threads); run<16, 0>(d_out, h, iters, threads);
        run<0, 2>(d_out, h, iters, threads); run<1, 2>(d_out, h, iters, threads); run<2, 2>(d_out, h, iters, threads); run<4, 2>(d_out, h, iters, threads);
        run<0, 1>(d_out, h, iters, threads); run<1, 1>(d_out, h, iters, threads); run<2, 1>(d_out, h, iters, threads); run<4, 1>(d_out, h, iters, threads);
        run<0, 3>(d_out, h, iters, threads); run<1, 3>(d_out, h, iters, threads); run<2, 3>(d_out, h, iters, threads);
        run<0, 4>(d_out, h, iters, threads); run<1, 4>(d_out, h, iters, threads); run<2, 4>(d_out, h, iters, threads);
    }
    return 0;
}
