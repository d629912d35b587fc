// HBM floor of the share-gen access pattern on gfx950: each lane reads 48 contiguous bytes (3 x 16 B) and
// writes one 16-byte non-temporal store to each of 8 row streams (rows 128-byte aligned), no arithmetic.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_hbm.hip -o tools/microbench_hbm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef long long ll2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: gen pattern, 1: writes only, 2: reads only (sum), 3: gen pattern with plain (non-NT) stores
__global__ __launch_bounds__(256) void k(const long long* __restrict__ in, long long* __restrict__ out, size_t dim,
                                         size_t B, size_t Bs, size_t P, size_t chunks, long long* sink) {
    const size_t p = blockIdx.x / chunks, chunk = blockIdx.x - p * chunks;
    const size_t pair = chunk * 256 + threadIdx.x, b0 = 2 * pair;
    if (b0 + 1 >= B || b0 * 3 + 6 > dim) return;
    ll2 v[3];
    long long acc = 0;
    if (MODE != 1) {
        const long long* sp = in + p * dim + b0 * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) { v[i] = *reinterpret_cast<const ll2*>(sp + 2 * i); acc += v[i].x ^ v[i].y; }
    } else { acc = (long long)b0; }
    if (MODE == 2) { if (acc == 0x1234567) *sink = acc; return; }
    long long* op = out + p * Bs + b0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ll2 w; w.x = acc + j; w.y = acc - j;
        if (MODE == 3) *reinterpret_cast<ll2*>(op + (size_t)j * P * Bs) = w;
        else __builtin_nontemporal_store(w, reinterpret_cast<ll2*>(op + (size_t)j * P * Bs));
    }
}

// MODE 4: the dual-role launch's traffic without its arithmetic - every 15th workgroup sums 512 columns x 500 rows of a
// second share buffer (16 x 16-byte nt loads in flight per lane), the others run the gen pattern on `out`
template <int NTLOAD, int NTSTORE, int NTSECRET>
__global__ __launch_bounds__(256) void kmix(const long long* __restrict__ in, long long* __restrict__ out,
                                            const long long* __restrict__ prev, size_t dim, size_t B, size_t Bs, size_t P,
                                            size_t chunks, unsigned long long n_gen, unsigned long long n_comb,
                                            unsigned col_blocks, long long* sink, unsigned G) {
    // G = 1: every 15th workgroup is a clerk-sum item; G > 1: runs of G clerk-sum items followed by 14 G share-gen chunks
    const unsigned long long b = blockIdx.x, period = 15ull * G;
    const unsigned long long grp = b / period, off = b - grp * period;
    const unsigned long long q = grp * G + (off < G ? off : G), rem = off < G ? 0 : 1;
    if (rem == 0 && q < n_comb) {
        const size_t bx = q % col_blocks, t = q / col_blocks, job = t % 8, split = t / 8;
        const size_t c0 = 2 * (bx * 256 + threadIdx.x);
        if (c0 + 1 >= B) return;
        const long long* base = prev + job * P * Bs + c0;
        long long a = 0, c = 0;
        for (size_t r = split * 500; r < (split + 1) * 500; r += 16) {
            ll2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = r + u < (split + 1) * 500 ? (NTLOAD ? __builtin_nontemporal_load(reinterpret_cast<const ll2*>(base + (r + u) * Bs)) : *reinterpret_cast<const ll2*>(base + (r + u) * Bs)) : ll2{0, 0};
#pragma unroll
            for (int u = 0; u < 16; ++u) { a += v[u].x; c += v[u].y; }
        }
        if ((a ^ c) == 0x1234567) *sink = a;
        return;
    }
    const unsigned long long before = q;                         // clerk-sum positions below b
    const unsigned long long idx = b - (before < n_comb ? before : n_comb);
    if (idx >= n_gen) return;
    const size_t p = idx / chunks, chunk = idx - p * chunks;
    const size_t pair = chunk * 256 + threadIdx.x, b0 = 2 * pair;
    if (b0 + 1 >= B || b0 * 3 + 6 > dim) return;
    ll2 v[3];
    long long acc = 0;
    const long long* sp = in + p * dim + b0 * 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) { v[i] = NTSECRET ? __builtin_nontemporal_load(reinterpret_cast<const ll2*>(sp + 2 * i)) : *reinterpret_cast<const ll2*>(sp + 2 * i); acc += v[i].x ^ v[i].y; }
    long long* op = out + p * Bs + b0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ll2 w; w.x = acc + j; w.y = acc - j;
        if (NTSTORE) __builtin_nontemporal_store(w, reinterpret_cast<ll2*>(op + (size_t)j * P * Bs));
        else *reinterpret_cast<ll2*>(op + (size_t)j * P * Bs) = w;
    }
}

// the same mix with a workgroup covering NP x 512 batches of one participant (NP consecutive 4 KB pieces per clerk row)
template <int NP, int PER>
__global__ __launch_bounds__(256) void kmix_wide(const long long* __restrict__ in, long long* __restrict__ out,
                                                 const long long* __restrict__ prev, size_t dim, size_t B, size_t Bs, size_t P,
                                                 size_t chunks, unsigned long long n_gen, unsigned long long n_comb,
                                                 unsigned col_blocks, long long* sink) {
    // every PER-th workgroup is a clerk-sum item until they run out; PER is ODD (workgroup b runs on XCD b mod 8: an even
    // period parks every clerk-sum item on one or two XCDs) and at most the natural ratio 14 / NP + 1
    const unsigned long long b = blockIdx.x, per = PER;
    const unsigned long long grp = b / per, off = b - grp * per;
    if (off == 0 && grp < n_comb) {
        const size_t bx = grp % col_blocks, t = grp / col_blocks, job = t % 8, split = t / 8;
        const size_t c0 = 2 * (bx * 256 + threadIdx.x);
        if (c0 + 1 >= B) return;
        const long long* base = prev + job * P * Bs + c0;
        long long a = 0, c = 0;
        for (size_t r = split * 500; r < (split + 1) * 500; r += 16) {
            ll2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = r + u < (split + 1) * 500 ? __builtin_nontemporal_load(reinterpret_cast<const ll2*>(base + (r + u) * Bs)) : ll2{0, 0};
#pragma unroll
            for (int u = 0; u < 16; ++u) { a += v[u].x; c += v[u].y; }
        }
        if ((a ^ c) == 0x1234567) *sink = a;
        return;
    }
    const unsigned long long before = grp + (off > 0 ? 1 : 0);
    const unsigned long long idx = b - (before < n_comb ? before : n_comb);
    if (idx >= n_gen) return;
    const size_t p = idx / chunks, chunk = idx - p * chunks;
#pragma unroll
    for (int h = 0; h < NP; ++h) {
        const size_t pair = (chunk * NP + h) * 256 + threadIdx.x, b0 = 2 * pair;
        if (b0 + 1 >= B || b0 * 3 + 6 > dim) continue;
        ll2 v[3];
        long long acc = 0;
        const long long* sp = in + p * dim + b0 * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) { v[i] = *reinterpret_cast<const ll2*>(sp + 2 * i); acc += v[i].x ^ v[i].y; }
        long long* op = out + p * Bs + b0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ll2 w; w.x = acc + j; w.y = acc - j;
            __builtin_nontemporal_store(w, reinterpret_cast<ll2*>(op + (size_t)j * P * Bs));
        }
    }
}

template <int MODE>
int run(const char* name, const long long* in, long long* out, size_t dim, size_t B, size_t Bs, size_t P, long long* sink, double bytes) {
    const size_t chunks = (B / 2 + 255) / 256;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<MODE><<<dim3((unsigned)(chunks * P)), dim3(256)>>>(in, out, dim, B, Bs, P, chunks, sink);
    CHK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CHK(hipEventRecord(e0));
        k<MODE><<<dim3((unsigned)(chunks * P)), dim3(256)>>>(in, out, dim, B, Bs, P, chunks, sink);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-44s %7.3f ms  %6.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    const size_t P = 2000, dim = 1 << 20, B = 349526, Bs = 349536;
    long long *in, *out, *sink;
    CHK(hipMalloc(&in, P * dim * 8)); CHK(hipMalloc(&out, 8 * P * Bs * 8)); CHK(hipMalloc(&sink, 8));
    CHK(hipMemset(in, 1, P * dim * 8));
    const double rd = (double)P * dim * 8, wr = 8.0 * P * B * 8;
    run<0>("gen pattern: 16.8 GB read + 44.7 GB NT write", in, out, dim, B, Bs, P, sink, rd + wr);
    run<3>("gen pattern, plain stores", in, out, dim, B, Bs, P, sink, rd + wr);
    run<1>("writes only (44.7 GB, NT)", in, out, dim, B, Bs, P, sink, wr);
    run<2>("reads only (16.8 GB)", in, out, dim, B, Bs, P, sink, rd);
    {   // dual-role traffic: gen pattern on `out` + clerk-sum reads of a second share buffer, one grid
        long long* prev;
        CHK(hipMalloc(&prev, 8 * P * Bs * 8));
        CHK(hipMemset(prev, 1, 8 * P * Bs * 8));
        const size_t chunks = (B / 2 + 255) / 256;
        const unsigned col_blocks = (unsigned)((B / 2 + 255) / 256);
        const unsigned long long n_gen = chunks * P, n_comb = (unsigned long long)col_blocks * 8 * 4;
        const unsigned long long grid = n_gen + n_comb;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        const unsigned Gs[] = {1, 9, 33, 129, 1025};
        for (unsigned gi = 0; gi < 5; ++gi) {
            float best = 1e9f;
            for (int r = 0; r < 5; ++r) {
                CHK(hipEventRecord(e0));
                kmix<1, 1, 0><<<dim3((unsigned)grid), dim3(256)>>>(in, out, prev, dim, B, Bs, P, chunks, n_gen, n_comb, col_blocks, sink, Gs[gi]);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
            }
            char name[96];
            snprintf(name, sizeof name, "dual-role traffic, runs of %u (106.2 GB)", Gs[gi]);
            printf("%-44s %7.3f ms  %6.2f TB/s\n", name, best, (rd + 2 * wr) / (best * 1e-3) / 1e12);
        }
        // cache-policy variants of the same traffic (strict alternation): which of the three streams is non-temporal
#define VARIANT(L, S, C)                                                                                                       \
        {                                                                                                                      \
            float best = 1e9f;                                                                                                 \
            for (int r = 0; r < 5; ++r) {                                                                                      \
                CHK(hipEventRecord(e0));                                                                                       \
                kmix<L, S, C><<<dim3((unsigned)grid), dim3(256)>>>(in, out, prev, dim, B, Bs, P, chunks, n_gen, n_comb, col_blocks, sink, 1); \
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));                                                         \
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;                                \
            }                                                                                                                  \
            printf("share reads nt=%d, share writes nt=%d, secret reads nt=%d: %7.3f ms  %6.2f TB/s\n", L, S, C, best,        \
                   (rd + 2 * wr) / (best * 1e-3) / 1e12);                                                                      \
        }
        VARIANT(1, 1, 0) VARIANT(0, 1, 0) VARIANT(1, 0, 0) VARIANT(0, 0, 0) VARIANT(1, 1, 1) VARIANT(0, 1, 1)
#undef VARIANT
#define WIDE(NP, PER)                                                                                                               \
        {                                                                                                                      \
            const size_t wchunks = (chunks + NP - 1) / NP;                                                                     \
            const unsigned long long wn_gen = wchunks * P, wgrid = wn_gen + n_comb;                                            \
            float best = 1e9f;                                                                                                 \
            for (int r = 0; r < 5; ++r) {                                                                                      \
                CHK(hipEventRecord(e0));                                                                                       \
                kmix_wide<NP, PER><<<dim3((unsigned)wgrid), dim3(256)>>>(in, out, prev, dim, B, Bs, P, wchunks, wn_gen, n_comb, col_blocks, sink); \
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));                                                         \
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;                                \
            }                                                                                                                  \
            printf("share-gen workgroup covers %d x 512 batches, period %d: %7.3f ms  %6.2f TB/s\n", NP, PER, best, (rd + 2 * wr) / (best * 1e-3) / 1e12); \
        }
        WIDE(1, 15) WIDE(1, 13) WIDE(2, 7) WIDE(2, 5) WIDE(4, 3)
#undef WIDE
    }
    return 0;
}
