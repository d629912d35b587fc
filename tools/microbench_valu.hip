// VALU instruction-rate microbenchmark for gfx950: how many wave64 instructions per cycle per SIMD
// does each integer op sustain?  Decides how the 62-bit modular arithmetic is decomposed.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o tools/microbench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

template <int OP>
__global__ __launch_bounds__(256) void bench(uint32_t* out, uint32_t seed) {
    uint32_t a[CHAINS], b[CHAINS];
    uint64_t w[CHAINS];
    uint64_t mask64 = __builtin_amdgcn_read_exec() ^ seed;
    uint32_t t32 = 0, t33 = 0; uint64_t m2 = 0; (void)t32; (void)t33; (void)m2;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) { a[i] = seed + threadIdx.x * 7 + i; b[i] = seed * 3 + i + 1; w[i] = ((uint64_t)a[i] << 32) | b[i]; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            if (OP == 4) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 6) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i + 1) % CHAINS]));
            if (OP == 7) asm volatile("v_alignbit_b32 %0, %0, %0, 16" : "+v"(a[i]));
            if (OP == 8) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 9) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 10) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            if (OP == 11) { double d; asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(d) : "v"(w[i])); w[i] = __double_as_longlong(d); }
            if (OP == 12) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 13) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 14) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 15) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            if (OP == 16) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(seed));
            if (OP == 17) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : );
            if (OP == 18) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 19) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            if (OP == 20) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
            if (OP == 21) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 22) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            if (OP == 23) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 24) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
            if (OP == 25) asm volatile("v_lshrrev_b64 %0, 31, %0" : "+v"(w[i]));
            if (OP == 26) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
            if (OP == 27) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(w[i]), "v"(w[(i+1)%CHAINS]) : "vcc");
            if (OP == 28) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(mask64));
            if (OP == 29) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 30) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 31) asm volatile("v_mad_u64_u32 %0, %3, %1, %2, %0" : "+v"(w[i]) : "s"(seed), "v"(b[i]), "s"(mask64));
            if (OP == 32) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(a[i]), "+v"(b[(i+1)%CHAINS]) : "v"(b[i]) : "vcc");
            if (OP == 33) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 34) asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 35) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 36) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            if (OP == 37) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 39) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            if (OP == 40) asm volatile("v_sub_co_u32 %2, vcc, %0, %1\n\tv_cndmask_b32 %0, %2, %0, vcc" : "+v"(a[i]), "+v"(b[i]), "=&v"(t32) : : "vcc");
            if (OP == 41) asm volatile("v_sub_co_u32_e64 %2, %3, %0, %1\n\tv_cndmask_b32_e64 %0, %2, %0, %3" : "+v"(a[i]), "+v"(b[i]), "=&v"(t32), "=&s"(m2) : : );
            if (OP == 42) asm volatile("v_sub_co_u32 %2, vcc, %0, %1\n\tv_subb_co_u32 %3, vcc, %4, %1, vcc\n\tv_cndmask_b32 %0, %2, %0, vcc\n\tv_cndmask_b32 %4, %3, %4, vcc" : "+v"(a[i]), "+v"(b[i]), "=&v"(t32), "=&v"(t33), "+v"(a[(i+1)%CHAINS]) : : "vcc");
            if (OP == 43) asm volatile("v_mad_i64_i32 %0, %3, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "s"(seed), "s"(mask64));
            if (OP == 44) asm volatile("v_ashrrev_i64 %0, 31, %0" : "+v"(w[i]));
            if (OP == 45) asm volatile("v_bfe_i32 %0, %0, 0, 31" : "+v"(a[i]));
            if (OP == 46) asm volatile("v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            if (OP == 47) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "s"(mask64));
            if (OP == 48) asm volatile("v_cmp_ge_u64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc" : : "v"(w[i]), "v"(w[(i+1)%CHAINS]), "v"(a[i]), "v"(b[i]) : "vcc");
            if (OP == 49) asm volatile("v_alignbyte_b32 %0, %0, %0, 2" : "+v"(a[i]));
            if (OP == 50) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(a[i]) : "s"(seed));
            if (OP == 51) asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %2, %2, %0\n\tv_perm_b32 %2, %2, %2, %3" : "+v"(a[i]), "+v"(b[i]) : "v"(seed), "s"(seed));
            if (OP == 52) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 53) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 38) asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %2, %2, %0\n\tv_alignbit_b32 %2, %2, %2, 16" : "+v"(a[i]), "+v"(b[i]) : "v"(seed));
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
int run(const char* name, uint32_t* d_out, int blocks, double clk_ghz) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    bench<OP><<<blocks, 256>>>(d_out, 12345u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    bench<OP><<<blocks, 256>>>(d_out, 12345u);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    double waves = (double)blocks * 4;
    double insts = waves * ITERS * CHAINS;                 // wave-instructions
    double per_simd_per_s = insts / (ms * 1e-3) / 1024.0;  // 256 CU x 4 SIMD
    printf("%-28s %8.3f ms  %7.3f wave-inst/ns/SIMD  => %6.2f cycles per wave64 instruction at %.2f GHz\n", name, ms,
           per_simd_per_s * 1e-9, clk_ghz * 1e9 / per_simd_per_s, clk_ghz);
    return 0;
}

int main() {
    uint32_t* d_out;
    const int blocks = 256 * 8;      // 8 blocks of 256 per CU = 8 waves per SIMD
    CHK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    double clk = prop.clockRate * 1e-6;
    printf("%s  CUs %d  clock %.2f GHz\n", prop.gcnArchName, prop.multiProcessorCount, clk);
    run<0>("v_add_u32", d_out, blocks, clk);
    run<1>("v_mul_lo_u32", d_out, blocks, clk);
    run<2>("v_mul_hi_u32", d_out, blocks, clk);
    run<3>("v_mad_u64_u32 (acc)", d_out, blocks, clk);
    run<15>("v_mad_u64_u32 (c=0)", d_out, blocks, clk);
    run<19>("v_mad_i64_i32 (acc)", d_out, blocks, clk);
    run<4>("v_mul_u32_u24", d_out, blocks, clk);
    run<12>("v_mul_hi_u32_u24", d_out, blocks, clk);
    run<5>("v_mad_u32_u24", d_out, blocks, clk);
    run<6>("v_lshl_add_u64", d_out, blocks, clk);
    run<10>("v_add_co_u32", d_out, blocks, clk);
    run<13>("v_add3_u32", d_out, blocks, clk);
    run<14>("v_xad_u32", d_out, blocks, clk);
    run<7>("v_alignbit_b32", d_out, blocks, clk);
    run<8>("v_xor_b32", d_out, blocks, clk);
    run<17>("v_cndmask_b32", d_out, blocks, clk);
    run<9>("s_nop1 + v_mov_b32_dpp", d_out, blocks, clk);
    run<11>("v_fma_f64", d_out, blocks, clk);
    run<16>("v_mul_lo_u32 (sgpr opnd)", d_out, blocks, clk);
    run<18>("v_pk_mul_lo_u16", d_out, blocks, clk);
    run<20>("v_mov_b32", d_out, blocks, clk);
    run<21>("v_sub_u32", d_out, blocks, clk);
    run<22>("v_addc_co_u32", d_out, blocks, clk);
    run<36>("v_sub_co_u32", d_out, blocks, clk);
    run<32>("add_co + addc (2 inst)", d_out, blocks, clk);
    run<23>("v_and_b32", d_out, blocks, clk);
    run<24>("v_lshlrev_b32", d_out, blocks, clk);
    run<25>("v_lshrrev_b64", d_out, blocks, clk);
    run<26>("v_cmp_lt_u32 vcc", d_out, blocks, clk);
    run<27>("v_cmp_lt_u64 vcc", d_out, blocks, clk);
    run<28>("v_cndmask_b32_e64 sgpr", d_out, blocks, clk);
    run<29>("v_max_u32", d_out, blocks, clk);
    run<30>("v_and_or_b32", d_out, blocks, clk);
    run<31>("v_mad_u64_u32 sgpr,vgpr", d_out, blocks, clk);
    run<33>("v_fma_f32", d_out, blocks, clk);
    run<34>("v_add_u32_dpp (no nop)", d_out, blocks, clk);
    run<35>("v_perm_b32", d_out, blocks, clk);
    run<37>("v_pk_add_u16", d_out, blocks, clk);
    run<38>("ARX triple (3 inst)", d_out, blocks, clk);
    run<51>("ARX triple with v_perm (3)", d_out, blocks, clk);
    run<49>("v_alignbyte_b32", d_out, blocks, clk);
    run<50>("v_perm_b32 sgpr selector", d_out, blocks, clk);
    run<52>("v_min_u32", d_out, blocks, clk);
    run<53>("v_mul_hi_i32", d_out, blocks, clk);
    run<39>("cmp vcc + cndmask vcc (2)", d_out, blocks, clk);
    run<40>("sub_co vcc + cndmask vcc (2)", d_out, blocks, clk);
    run<41>("sub_co sgpr + cndmask e64 (2)", d_out, blocks, clk);
    run<42>("condsub64 vcc (4 inst)", d_out, blocks, clk);
    run<43>("v_mad_i64_i32 vgpr,sgpr", d_out, blocks, clk);
    run<44>("v_ashrrev_i64", d_out, blocks, clk);
    run<45>("v_bfe_i32", d_out, blocks, clk);
    run<46>("v_subb_co_u32", d_out, blocks, clk);
    run<47>("v_lshl_add_u64 sgpr", d_out, blocks, clk);
    run<48>("cmp_ge_u64 + cndmask (2)", d_out, blocks, clk);
    return 0;
}
