// The reference's OWN representatives (value mode SDA_VALUES_RUST_SIGNED): Rust's `%` on i64 is the truncated remainder
// (sign follows the dividend), so the reference's intermediate values live in (-q, q) and their signs depend on the
// history of the running sums.  The throughput kernels emit canonical residues in [0, q), equal
// to these modulo q; the kernels below reproduce the reference's values BIT FOR BIT for the paths whose arithmetic is
// visible in /root/reference:
//     additive.rs:42-47      shares = [r_1 .. r_{n-1}, fold(secret, |acc, r| (acc - r) % q)]
//     combiner.rs:20-26      result[ix] = (result[ix] + share[ix]) % q, participant after participant
//     additive.rs:62-69      the same loop over the clerks' vectors (reconstruct)
//     full.rs:30, :46-48, :62   (s + m) % q;  the same loop over masks;  (ms - m) % q
//     chacha.rs:41-44, :88   (s + m) % q;  (ms - m) % q
// Inputs may be ANY i64 (the reference never range-checks); sums are formed in 128 bits, so nothing wraps where debug Rust
// would panic and release Rust would wrap - for |values| < 2^62 (every value the protocol produces) all three agree.
// This is a fidelity mode: one lane per column, participants strictly in order - no row splits, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"
#include "signed_rem.hpp"

namespace sda {

static constexpr int kSignedThreads = 256;

// additive.rs:42-47 for element i of participant p: out[j][i] = rand[i (n-1) + j] (untouched), out[n-1][i] = the fold
__global__ __launch_bounds__(kSignedThreads) void signed_additive_gen_kernel(GenLayout L, uint32_t n, int64_t q) {
    const size_t i = (size_t)blockIdx.x * kSignedThreads + threadIdx.x, p = blockIdx.y;
    if (i >= L.len) return;
    const int64_t* rp = L.rand + p * L.rand_stride + i * (size_t)(n - 1);
    int64_t* op = L.out + p * L.out_stride_participant + i;
    int64_t acc = L.secrets[p * L.secrets_stride + i];
    for (uint32_t j = 0; j + 1 < n; ++j) {
        const int64_t r = rp[j];
        op[(size_t)j * L.out_stride_clerk] = r;
        acc = trunc_rem128((__int128)acc - r, q);
    }
    op[(size_t)(n - 1) * L.out_stride_clerk] = acc;
}

// combiner.rs:20-26: state[job][col] = (state + row[col]) % q for the rows in order
__global__ __launch_bounds__(kSignedThreads) void signed_combine_update_kernel(int64_t* __restrict__ state,
                                                                               const int64_t* __restrict__ shares, size_t job_stride,
                                                                               size_t n_rows, size_t row_stride, size_t dimension,
                                                                               int64_t q) {
    const size_t col = (size_t)blockIdx.x * kSignedThreads + threadIdx.x, job = blockIdx.y;
    if (col >= dimension) return;
    const int64_t* base = shares + job * job_stride + col;
    int64_t r = state[job * dimension + col];
    for (size_t row = 0; row < n_rows; ++row) r = trunc_rem128((__int128)r + base[row * row_stride], q);
    state[job * dimension + col] = r;
}

// full.rs:30 / chacha.rs:43 (a + b) % q and full.rs:62 / chacha.rs:88 (a - b) % q
__global__ __launch_bounds__(kSignedThreads) void signed_addsub_kernel(const int64_t* __restrict__ a, const int64_t* __restrict__ b,
                                                                       size_t len, bool subtract, int64_t q, int64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kSignedThreads + threadIdx.x;
    if (i >= len) return;
    const __int128 x = subtract ? (__int128)a[i] - b[i] : (__int128)a[i] + b[i];
    out[i] = trunc_rem128(x, q);
}

static inline uint64_t sdiv(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

hipError_t launch_additive_generate_signed(const GenLayout& L, uint32_t n, const ModParams& mod, const DrbgKey& key, int rounds,
                                           hipStream_t s) {
    if (L.participants == 0 || L.len == 0) return hipSuccess;
    if (n < 1) return hipErrorInvalidValue;
    // no injected draws: the kernel draws them itself, from the streams the canonical kernel would use (sda_kernels.hip) - no
    // participants x len x (n - 1) scratch
    if (!L.rand && n > 1) return launch_additive_generate_signed_drbg(L, n, mod, key, rounds, s);
    const int64_t q = (int64_t)mod.m;
    const uint64_t blocks = sdiv(L.len, kSignedThreads);
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    for (size_t p0 = 0; p0 < L.participants; p0 += 65535) {
        GenLayout S = L;
        S.secrets = L.secrets + p0 * L.secrets_stride;
        if (L.rand) S.rand = L.rand + p0 * L.rand_stride;
        S.out = L.out + p0 * L.out_stride_participant;
        S.participants = L.participants - p0 < 65535 ? L.participants - p0 : 65535;
        note_kernel("signed_additive_gen_kernel");
        signed_additive_gen_kernel<<<dim3((unsigned)blocks, (unsigned)S.participants), dim3(kSignedThreads), 0, s>>>(S, n, q);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

hipError_t launch_combine_update_signed(int64_t* d_state, const int64_t* d_shares, size_t jobs, size_t job_stride, size_t n_rows,
                                        size_t row_stride, size_t dimension, int64_t q, hipStream_t s) {
    if (jobs == 0 || n_rows == 0 || dimension == 0) return hipSuccess;
    const uint64_t blocks = sdiv(dimension, kSignedThreads);
    if (blocks > 0x7FFFFFFFull || jobs > 65535) return hipErrorInvalidConfiguration;
    signed_combine_update_kernel<<<dim3((unsigned)blocks, (unsigned)jobs), dim3(kSignedThreads), 0, s>>>(d_state, d_shares, job_stride, n_rows,
                                                                                                       row_stride, dimension, q);
    return hipGetLastError();
}

hipError_t launch_addsub_signed(const int64_t* d_a, const int64_t* d_b, size_t len, bool subtract, int64_t q, int64_t* d_out,
                                hipStream_t s) {
    if (len == 0) return hipSuccess;
    const uint64_t blocks = sdiv(len, kSignedThreads);
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    signed_addsub_kernel<<<dim3((unsigned)blocks), dim3(kSignedThreads), 0, s>>>(d_a, d_b, len, subtract, q, d_out);
    return hipGetLastError();
}

}  // namespace sda
