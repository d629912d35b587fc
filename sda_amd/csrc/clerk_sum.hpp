// The clerk sum of one item (two adjacent columns per lane over a range of rows of one job), shared by the clerk-sum kernels and
// the dual-role launches of sda_kernels.hip and by the dual-role form of the narrow limb GEMM (ngemm_kernels.hip).
// combiner.rs:15-29: exact 128-bit column sums over rows, reduced once at finish.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime.h>

#include "modarith.hpp"

namespace sda {

typedef long long ll2 __attribute__((ext_vector_type(2)));

// `pair` = index of the lane's column pair (columns 2 pair, 2 pair + 1); UNROLL independent 16-byte loads in flight
template <bool VEC, int UNROLL>
__device__ __forceinline__ void combine_pair(uint64_t* __restrict__ acc_lo, int64_t* __restrict__ acc_hi,
                                             const int64_t* __restrict__ shares, size_t job_stride, size_t n_rows,
                                             size_t row_stride, size_t dimension, size_t rows_per_split, bool atomic,
                                             size_t pair, size_t by, size_t bz) {
    const size_t c0 = 2 * pair;
    if (c0 >= dimension) return;
    const bool two = c0 + 1 < dimension;
    const size_t job = by;
    const size_t r_begin = bz * rows_per_split;
    size_t r_end = r_begin + rows_per_split;
    if (r_end > n_rows) r_end = n_rows;
    const int64_t* base = shares + job * job_stride + c0;

    uint64_t lo0 = 0, lo1 = 0;
    int64_t hi0 = 0, hi1 = 0;
    size_t r = r_begin;
    if (VEC && two) {
        for (; r + UNROLL <= r_end; r += UNROLL) {
            ll2 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(base + (r + u) * row_stride));
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { acc_add(lo0, hi0, v[u].x); acc_add(lo1, hi1, v[u].y); }
        }
        for (; r < r_end; ++r) {
            ll2 v = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(base + r * row_stride));
            acc_add(lo0, hi0, v.x); acc_add(lo1, hi1, v.y);
        }
    } else {
        for (; r < r_end; ++r) {
            acc_add(lo0, hi0, base[r * row_stride]);
            if (two) acc_add(lo1, hi1, base[r * row_stride + 1]);
        }
    }

    const size_t idx = job * dimension + c0;
    if (atomic) {
        acc_atomic_add(acc_lo + idx, acc_hi + idx, lo0, hi0);
        if (two) acc_atomic_add(acc_lo + idx + 1, acc_hi + idx + 1, lo1, hi1);
    } else {
        uint64_t l = acc_lo[idx]; int64_t h = acc_hi[idx];
        uint64_t nl = l + lo0; h += hi0 + (nl < l ? 1 : 0);
        acc_lo[idx] = nl; acc_hi[idx] = h;
        if (two) {
            l = acc_lo[idx + 1]; h = acc_hi[idx + 1];
            nl = l + lo1; h += hi1 + (nl < l ? 1 : 0);
            acc_lo[idx + 1] = nl; acc_hi[idx + 1] = h;
        }
    }
}

}  // namespace sda
