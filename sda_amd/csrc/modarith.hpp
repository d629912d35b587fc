// 64-bit modular arithmetic for gfx950 (and the host-side constant precomputation).
//
// CDNA4 has no 64-bit integer multiplier: a 64x64->128 product is four v_mad_u64_u32
// (32x32+64->64) plus carries, so every formula below is arranged to minimise the number of
// 32-bit multiplies:
//   * matrix-vector products accumulate un-reduced 128-bit sums of (Montgomery-form constant) x
//     (residue) and pay ONE Montgomery REDC per output,
//   * uniform sampling uses Lemire's multiply-shift (one 64x64->128 product, no division),
//   * the rand-0.3-compatible `v % range` uses Barrett with a precomputed floor(2^64/range).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SDA_HD __host__ __device__ __forceinline__
#define SDA_D __device__ __forceinline__
#else
#define SDA_HD inline
#define SDA_D inline
#endif

namespace sda {

typedef unsigned __int128 u128;

struct U128 {
    uint64_t lo, hi;
};

// ---- 64x64 -> 128 ---------------------------------------------------------------------------
SDA_HD U128 mul64x64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    U128 r;
    r.lo = a * b;
    r.hi = __umul64hi(a, b);
    return r;
#else
    u128 p = (u128)a * b;
    return U128{(uint64_t)p, (uint64_t)(p >> 64)};
#endif
}

SDA_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((u128)a * b) >> 64);
#endif
}

// acc += a*b  (128-bit accumulate, caller guarantees no overflow past 2^128)
SDA_HD void mac128(U128& acc, uint64_t a, uint64_t b) {
    U128 p = mul64x64(a, b);
    uint64_t lo = acc.lo + p.lo;
    acc.hi += p.hi + (lo < acc.lo ? 1u : 0u);
    acc.lo = lo;
}

// ---- Montgomery (R = 2^64), odd modulus p < 2^62 ---------------------------------------------
struct MontCtx {
    uint64_t p;      // modulus
    uint64_t pinv;   // -p^{-1} mod 2^64
    uint64_t r2;     // 2^128 mod p  (to_mont(x) = redc(x * r2))
};

// REDC: T < p * 2^64  ->  T * 2^-64 mod p, canonical [0, p)
SDA_HD uint64_t mont_redc(U128 T, uint64_t p, uint64_t pinv) {
    uint64_t m = T.lo * pinv;
    // T + m*p is divisible by 2^64; its low word cancels, carry out of the low word is (T.lo != 0)
    uint64_t t = T.hi + mulhi64(m, p) + (T.lo != 0 ? 1u : 0u);
    return t >= p ? t - p : t;
}

// keep a 128-bit accumulator below p * 2^64 (call when it may have reached < 2 * p * 2^64)
SDA_HD void mont_acc_condsub(U128& acc, uint64_t p) {
    if (acc.hi >= p) acc.hi -= p;
}

// ---- Barrett for u64 % m with mu = floor(2^64 / m)  (m >= 2) --------------------------------
SDA_HD uint64_t barrett_mod64(uint64_t v, uint64_t m, uint64_t mu) {
    uint64_t qhat = mulhi64(v, mu);
    uint64_t r = v - qhat * m;          // true quotient - qhat in {0,1,2}
    if (r >= m) r -= m;
    if (r >= m) r -= m;
    return r;
}

// any i64 -> canonical [0, m)
SDA_HD uint64_t canon_i64(int64_t x, uint64_t m, uint64_t mu) {
    if ((uint64_t)x < m) return (uint64_t)x;           // the common case: already canonical
    if (x >= 0) return barrett_mod64((uint64_t)x, m, mu);
    uint64_t mag = (uint64_t)0 - (uint64_t)x;          // |x| <= 2^63
    uint64_t r = barrett_mod64(mag, m, mu);
    return r == 0 ? 0 : m - r;
}

SDA_HD uint64_t addmod(uint64_t a, uint64_t b, uint64_t m) {   // a,b in [0,m), m < 2^63
    uint64_t s = a + b;
    return s >= m ? s - m : s;
}

SDA_HD uint64_t submod(uint64_t a, uint64_t b, uint64_t m) {
    return a >= b ? a - b : a + m - b;
}

// signed 128-bit value (two's complement hi:lo) -> canonical residue mod m (m < 2^63).
// Bit-serial; used once per output column, never in a streaming loop.
SDA_HD uint64_t mod_i128(uint64_t lo, int64_t hi, uint64_t m, uint64_t mu) {
    bool neg = hi < 0;
    uint64_t mh = (uint64_t)hi, ml = lo;
    if (neg) {                                   // magnitude = -(hi:lo)
        ml = ~ml + 1;
        mh = ~mh + (ml == 0 ? 1u : 0u);
    }
    uint64_t r = barrett_mod64(mh, m, mu);
    for (int i = 63; i >= 0; --i) {
        r = (r << 1) | ((ml >> i) & 1u);         // < 2m + 1 <= 2^64 - 1
        if (r >= m) r -= m;
    }
    return (neg && r != 0) ? m - r : r;
}

#if defined(__HIPCC__)
// ---- exact 128-bit column accumulators (lo u64, hi i64 two's complement) -------------------------------
SDA_D void acc_add(uint64_t& lo, int64_t& hi, int64_t v) {
    const uint64_t nl = lo + (uint64_t)v;
    hi += (v >> 63) + (nl < lo ? 1 : 0);
    lo = nl;
}
// accumulator in memory += (hi:lo), carry propagated from the value the low-word atomic returns
SDA_D void acc_atomic_add(uint64_t* lo_p, int64_t* hi_p, uint64_t lo, int64_t hi) {
    if (lo != 0) {
        const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(lo_p), (unsigned long long)lo);
        if (old + lo < old) hi += 1;
    }
    if (hi != 0) atomicAdd(reinterpret_cast<unsigned long long*>(hi_p), (unsigned long long)hi);
}
#endif

// ---- Lemire uniform sampling in [0, m) from a 64-bit word ------------------------------------
// returns true (accepted) and the value; rejected iff lo(x*m) < (2^64 mod m) = lemire_thr
SDA_HD bool lemire_sample(uint64_t x, uint64_t m, uint64_t lemire_thr, uint64_t& out) {
    U128 pr = mul64x64(x, m);
    out = pr.hi;
    return pr.lo >= lemire_thr;
}

// ---- the PAIRED rule of sda-drbg-v1 (round 5) for moduli m <= kDrbgPairedMax (m m < 2^46): ONE 64-bit candidate word yields TWO
// draws.  x m m = (ra m + rb) 2^64 + lo is Lemire's method with range m^2 - accept iff lo >= 2^64 mod m^2, then ra m + rb is uniform
// in [0, m^2) - read off in two steps, x m = ra 2^64 + l1 and l1 m = rb 2^64 + lo ("batched" bounded integers): (ra, rb) is a
// uniform pair in [0, m)^2, with a rejection probability below 2^-18.  Draws 2j and 2j + 1 of a batch are such a pair; the block
// counter of draw pair j of batch group g is g * ceil(T / 2) + j.  ModParams::lemire_thr2 holds 2^64 mod m^2 for these moduli
// (lemire_thr stays 2^64 mod m for every modulus).
// 155 draws per batch of tss's PSS_155_728_100 then cost 78 candidate words instead of 155 (the ChaCha20 blocks are a third of
// that kernel's vector work).  Spec: DESIGN.md (sda-drbg-v1); vectors: tests/golden/drbg.json.
static constexpr uint64_t kDrbgPairedMax = 0x7F7F7Full;
SDA_HD bool drbg_paired(uint64_t m) { return m <= kDrbgPairedMax; }
SDA_HD bool lemire_pair(uint64_t x, uint32_t m, uint64_t thr2, uint32_t& ra, uint32_t& rb) {
    const uint64_t t0 = (uint64_t)(uint32_t)x * m;
    const uint64_t t1 = (uint64_t)(uint32_t)(x >> 32) * m + (t0 >> 32);                 // x m = t1 2^32 + lo32(t0), t1 < 2^55
    ra = (uint32_t)(t1 >> 32);
    const uint64_t u0 = (uint64_t)(uint32_t)t0 * m;                                     // l1 = lo32(t1) 2^32 + lo32(t0)
    const uint64_t u1 = (uint64_t)(uint32_t)t1 * m + (u0 >> 32);
    rb = (uint32_t)(u1 >> 32);
    return ((u1 << 32) | (uint32_t)u0) >= thr2;
}

// ---- host-only helpers (plain host functions: parsed but never emitted in the device pass) ------
inline uint64_t h_mulmod(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)(((u128)a * b) % m); }

inline uint64_t h_powmod(uint64_t b, uint64_t e, uint64_t m) {
    uint64_t r = 1 % m;
    b %= m;
    while (e) {
        if (e & 1) r = h_mulmod(r, b, m);
        b = h_mulmod(b, b, m);
        e >>= 1;
    }
    return r;
}

// modular inverse by extended Euclid; returns false if gcd(a, m) != 1
inline bool h_invmod(uint64_t a, uint64_t m, uint64_t& out) {
    __int128 r0 = m, r1 = a % m, t0 = 0, t1 = 1;
    while (r1 != 0) {
        __int128 q = r0 / r1;
        __int128 r2 = r0 - q * r1; r0 = r1; r1 = r2;
        __int128 t2 = t0 - q * t1; t0 = t1; t1 = t2;
    }
    if (r0 != 1) return false;
    if (t0 < 0) t0 += m;
    out = (uint64_t)t0;
    return true;
}

inline uint64_t h_canon(int64_t x, uint64_t m) {
    __int128 r = (__int128)x % (__int128)m;
    if (r < 0) r += m;
    return (uint64_t)r;
}

inline uint64_t h_barrett_mu(uint64_t m) {           // floor(2^64 / m), m >= 2
    return (uint64_t)((((u128)1) << 64) / m);
}

inline uint64_t h_lemire_thr(uint64_t m) {           // 2^64 mod m
    return (uint64_t)((((u128)1) << 64) % m);
}
inline uint64_t h_lemire_thr2(uint64_t m) {          // the paired rule's threshold 2^64 mod m^2; 0 outside its domain
    return drbg_paired(m) ? (uint64_t)((((u128)1) << 64) % ((u128)m * m)) : 0;
}

inline MontCtx h_mont_ctx(uint64_t p) {              // p odd
    MontCtx c;
    c.p = p;
    uint64_t inv = p;                                // Newton: inv = p^{-1} mod 2^64
    for (int i = 0; i < 6; ++i) inv *= 2 - p * inv;
    c.pinv = (uint64_t)0 - inv;
    u128 r = (((u128)1) << 64) % p;
    c.r2 = (uint64_t)((r * r) % p);
    return c;
}

inline uint64_t h_to_mont(uint64_t x, uint64_t p) {  // x * 2^64 mod p
    return (uint64_t)((((u128)x) << 64) % p);
}

}  // namespace sda
