// The no-arithmetic HBM floor of the dual-role launch for a given SHAPE: K secrets per batch read (16 K bytes per lane: two adjacent
// batches), N clerk rows written (one 16-byte non-temporal store per lane and row), every PERIOD-th workgroup sums 512 columns x 500
// rows of a second share buffer instead.  tools/microbench_hbm.hip measures (K, N) = (3, 8) only - config 3's pattern; this one says
// what the 26 row streams of config 4 / the (8, 7, 26) shapes can reach.
// (this file: the SAME kernel on buffers carved out of one slab at chosen offsets - does the duration of config 3's traffic depend on
// where the buffers lie?)  Build: hipcc --offload-arch=gfx950 -O3 tools/archive/experiments_r06/microbench_hbm_place.hip -o /tmp/microbench_hbm_place
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef long long ll2 __attribute__((ext_vector_type(2)));

template <int K, int N>
__global__ __launch_bounds__(256) void kmix(const long long* __restrict__ in, long long* __restrict__ out, const long long* __restrict__ prev,
                                            size_t dim, size_t B, size_t Bs, size_t P, size_t chunks, unsigned long long n_gen,
                                            unsigned long long n_comb, unsigned col_blocks, unsigned period, unsigned rows, long long* sink) {
    const unsigned long long b = blockIdx.x;
    const unsigned long long q = b / period, rem = b - q * period;
    if (rem == 0 && q < n_comb) {
        const size_t bx = q % col_blocks, t = q / col_blocks, job = t % N, split = t / N;
        const size_t c0 = 2 * (bx * 256 + threadIdx.x);
        if (c0 + 1 >= B) return;
        const long long* base = prev + job * P * Bs + c0;
        long long a = 0, c = 0;
        for (size_t r = split * rows; r < (split + 1) * rows && r < P; r += 16) {
            ll2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = r + u < (split + 1) * rows && r + u < P ? __builtin_nontemporal_load(reinterpret_cast<const ll2*>(base + (r + u) * Bs)) : ll2{0, 0};
#pragma unroll
            for (int u = 0; u < 16; ++u) { a += v[u].x; c += v[u].y; }
        }
        if ((a ^ c) == 0x1234567) *sink = a;
        return;
    }
    const unsigned long long before = q + (rem ? 1 : 0);
    const unsigned long long idx = b - (before < n_comb ? before : n_comb);
    if (idx >= n_gen) return;
    const size_t p = idx / chunks, chunk = idx - p * chunks;
    const size_t pair = chunk * 256 + threadIdx.x, b0 = 2 * pair;
    if (b0 + 1 >= B || (b0 + 2) * K > dim) return;
    long long acc = 0;
    const long long* sp = in + p * dim + b0 * K;
#pragma unroll
    for (int i = 0; i < K; ++i) { const ll2 v = *reinterpret_cast<const ll2*>(sp + 2 * i); acc += v.x ^ v.y; }
    long long* op = out + p * Bs + b0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        ll2 w; w.x = acc + j; w.y = acc - j;
        __builtin_nontemporal_store(w, reinterpret_cast<ll2*>(op + (size_t)j * P * Bs));
    }
}


int main() {
    const int K = 3, N = 8;
    const size_t P = 2500, dim = 1 << 20, B = (dim + K - 1) / K, Bs = (B + 15) / 16 * 16;
    const size_t e_in = P * dim, e_sh = (size_t)N * P * Bs;
    const size_t slab_bytes = 250ull << 30;
    char* slab; long long* sink;
    CHK(hipMalloc(&slab, slab_bytes)); CHK(hipMalloc(&sink, 8));
    CHK(hipMemset(slab, 1, slab_bytes));
    printf("slab at %p\n", (void*)slab);
    const size_t chunks = (B / 2 + 255) / 256;
    const unsigned col_blocks = (unsigned)((B / 2 + 255) / 256);
    const unsigned rows = 512;
    const unsigned long long splits = (P + rows - 1) / rows, n_gen = chunks * P, n_comb = (unsigned long long)col_blocks * N * splits;
    unsigned period = (unsigned)(n_gen / n_comb + 1);
    if (period % 2 == 0) period = period > 2 ? period - 1 : 3;
    unsigned long long grid = n_gen + n_comb;
    if ((n_comb - 1) * period + 1 > grid) grid = (n_comb - 1) * period + 1;
    const double bytes = (double)P * dim * 8 + 2.0 * N * P * B * 8;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto measure = [&](size_t base, size_t gap, const char* tag) -> int {
        long long* in = (long long*)(slab + base);
        long long* out = (long long*)(slab + base + e_in * 8 + gap);
        long long* prev = (long long*)(slab + base + e_in * 8 + gap + e_sh * 8 + gap);
        if (base + e_in * 8 + 2 * (e_sh * 8 + gap) > slab_bytes) return 0;
        float best = 1e9f, worst = 0;
        for (int r = 0; r < 6; ++r) {
            // alternate the roles of the two share buffers like the bench does
            long long* w = (r & 1) ? prev : out; long long* rd = (r & 1) ? out : prev;
            CHK(hipEventRecord(e0));
            kmix<K, N><<<dim3((unsigned)grid), dim3(256)>>>(in, w, rd, dim, B, Bs, P, chunks, n_gen, n_comb, col_blocks, period, rows, sink);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { if (ms < best) best = ms; if (ms > worst) worst = ms; }
        }
        printf("%s base %7.3f GiB gap %10zu: %.3f - %.3f ms = %.2f TB/s\n", tag, base / 1073741824.0, gap, best, worst, bytes / (best * 1e-3) / 1e12);
        return 0;
    };
    for (int rep = 0; rep < 2; ++rep) {
        for (size_t g = 0; g <= 112; g += 8) if (measure(g << 30, 1 << 20, "base sweep")) return 1;
        for (size_t m = 0; m <= 14; ++m) if (measure((size_t)((4ull << 30) + (m << 26)), 1 << 20, "fine base ")) return 1;
        const size_t gaps[] = {0, 4096, 65536, 1 << 20, 2 << 20, 16 << 20, 256 << 20, 1ull << 30, (1ull << 30) + (1 << 20), 3ull << 30};
        for (size_t gp : gaps) if (measure(0, gp, "gap sweep ")) return 1;
    }
    return 0;
}
