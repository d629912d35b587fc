// Host-side number theory for handle construction: primality, Lagrange evaluation matrices in
// Montgomery form.  Runs once per handle / per clerk-index set, never per element.
#pragma once
#include <stdint.h>
#include <vector>

#include "modarith.hpp"

namespace sda {

inline bool h_is_prime(uint64_t n) {
    if (n < 2) return false;
    static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (uint64_t p : small) {
        if (n == p) return true;
        if (n % p == 0) return false;
    }
    uint64_t d = n - 1;
    int r = 0;
    while ((d & 1) == 0) { d >>= 1; ++r; }
    for (uint64_t a : small) {               // deterministic for n < 2^64
        uint64_t x = h_powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool composite = true;
        for (int i = 1; i < r; ++i) {
            x = h_mulmod(x, x, n);
            if (x == n - 1) { composite = false; break; }
        }
        if (composite) return false;
    }
    return true;
}

inline bool h_all_distinct(const std::vector<uint64_t>& v) {
    for (size_t i = 0; i < v.size(); ++i)
        for (size_t j = i + 1; j < v.size(); ++j)
            if (v[i] == v[j]) return false;
    return true;
}

// L[e][i] = l_i(evals[e]) for the Lagrange basis l_i on `nodes` (distinct), all mod prime p.
// Column `drop_col` (the node whose value is fixed to 0) is omitted; entries are returned in
// Montgomery form (x * 2^64 mod p), row-major [evals.size()][nodes.size() - 1].
inline bool h_lagrange_matrix_mont(const std::vector<uint64_t>& nodes, const std::vector<uint64_t>& evals,
                                   size_t drop_col, uint64_t p, std::vector<uint64_t>& out) {
    const size_t m = nodes.size();
    std::vector<uint64_t> den_inv(m);
    for (size_t i = 0; i < m; ++i) {
        uint64_t den = 1;
        for (size_t l = 0; l < m; ++l)
            if (l != i) den = h_mulmod(den, submod(nodes[i], nodes[l], p), p);
        if (!h_invmod(den, p, den_inv[i])) return false;
    }
    out.assign(evals.size() * (m - 1), 0);
    std::vector<uint64_t> pre(m + 1), suf(m + 1);
    for (size_t e = 0; e < evals.size(); ++e) {
        const uint64_t y = evals[e];
        pre[0] = 1;
        for (size_t l = 0; l < m; ++l) pre[l + 1] = h_mulmod(pre[l], submod(y, nodes[l], p), p);
        suf[m] = 1;
        for (size_t l = m; l-- > 0;) suf[l] = h_mulmod(suf[l + 1], submod(y, nodes[l], p), p);
        size_t col = 0;
        for (size_t i = 0; i < m; ++i) {
            if (i == drop_col) continue;
            const uint64_t num = h_mulmod(pre[i], suf[i + 1], p);
            out[e * (m - 1) + col++] = h_to_mont(h_mulmod(num, den_inv[i], p), p);
        }
    }
    return true;
}

}  // namespace sda
