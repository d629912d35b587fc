"""ctypes binding of the C oracle (oracle/sda_oracle.c).  TEST INFRASTRUCTURE ONLY - see that
file's header: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsda_oracle.so")
_lib = None

I64P = C.POINTER(C.c_int64)
U64P = C.POINTER(C.c_uint64)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "sda_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.sdao_baseline_pass.restype = C.c_size_t
    return _lib


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(I64P)


def positive(values, modulus):
    v, vp = _i64(values)
    out = np.empty_like(v)
    lib().sdao_positive(vp, C.c_size_t(v.size), C.c_int64(modulus), out.ctypes.data_as(I64P))
    return out


def additive_generate(q, n, secrets, rand, mode=0):
    s, sp = _i64(secrets)
    r, rp = _i64(rand)
    assert r.size == s.size * (n - 1)
    out = np.empty((n, s.size), dtype=np.int64)
    st = lib().sdao_additive_generate(C.c_int64(q), n, sp, C.c_size_t(s.size), rp, out.ctypes.data_as(I64P), mode)
    assert st == 0, st
    return out


def combine(q, shares, mode=0):
    sh = np.ascontiguousarray(shares, dtype=np.int64)
    assert sh.ndim == 2
    P, L = sh.shape
    out = np.empty(L, dtype=np.int64)
    st = lib().sdao_combine(C.c_int64(q), sh.ctypes.data_as(I64P), C.c_size_t(P), C.c_size_t(L), C.c_size_t(L),
                            out.ctypes.data_as(I64P), mode)
    assert st == 0, st
    return out


def packed_share_matrix(p, k, t, n, w2, w3):
    M = np.empty((n, k + t), dtype=np.uint64)
    st = lib().sdao_packed_share_matrix(C.c_int64(p), k, t, n, C.c_int64(w2), C.c_int64(w3), M.ctypes.data_as(U64P))
    assert st == 0, st
    return M


def packed_generate(p, k, t, n, w2, w3, secrets, rand):
    s, sp = _i64(secrets)
    B = (s.size + k - 1) // k
    r, rp = _i64(rand)
    assert r.size == B * t
    out = np.empty((n, B), dtype=np.int64)
    st = lib().sdao_packed_generate(C.c_int64(p), k, t, n, C.c_int64(w2), C.c_int64(w3), sp, C.c_size_t(s.size), rp,
                                    out.ctypes.data_as(I64P))
    assert st == 0, st
    return out


def packed_generate_systematic(p, k, t, n, w2, w3, secrets, draws, want_implied=False):
    """the library's CSPRNG share map (include/sda_hip.h): draws = shares 0..t-1, the rest by interpolation through
    (1, 0), the secrets at w2^i and the draws at w3^(j+1).  -> [n][B] (and, on request, the [B * t] randomness tss's own
    share() would need for the same shares)"""
    s, sp = _i64(secrets)
    r, rp = _i64(draws)
    B = (s.size + k - 1) // k
    assert r.size == B * t
    out = np.empty((n, B), dtype=np.int64)
    imp = np.empty(B * t, dtype=np.int64) if want_implied else None
    st = lib().sdao_packed_generate_systematic(C.c_int64(p), k, t, n, C.c_int64(w2), C.c_int64(w3), sp, C.c_size_t(s.size), rp,
                                               out.ctypes.data_as(I64P), imp.ctypes.data_as(I64P) if want_implied else None)
    if st != 0:
        raise ValueError(f"sdao_packed_generate_systematic failed ({st})")
    return (out, imp) if want_implied else out


def packed_generate_csprng(p, k, t, n, w2, w3, secrets, draws, share_map=1):
    """what a generator WITHOUT injected randomness produces from its CSPRNG draws: share_map 1 = systematic (the
    matrix-form kernels' default), 0 = tss's nodes (the transform kernel; on request)"""
    if share_map == 1 and t > 0:
        return packed_generate_systematic(p, k, t, n, w2, w3, secrets, draws)
    return packed_generate(p, k, t, n, w2, w3, secrets, draws)


def packed_reconstruct(p, k, t, w2, w3, dimension, indices, shares):
    sh = np.ascontiguousarray(shares, dtype=np.int64)
    assert sh.ndim == 2 and sh.shape[0] == len(indices)
    idx = (C.c_size_t * len(indices))(*indices)
    out = np.empty(dimension, dtype=np.int64)
    st = lib().sdao_packed_reconstruct(C.c_int64(p), k, t, C.c_int64(w2), C.c_int64(w3), C.c_size_t(dimension), idx,
                                       C.c_size_t(len(indices)), sh.ctypes.data_as(I64P), C.c_size_t(sh.shape[1]),
                                       out.ctypes.data_as(I64P))
    if st != 0:
        raise ValueError(st)
    return out


def chacha_expand(seed_words, q, count):
    s, sp = _i64(seed_words)
    out = np.empty(count, dtype=np.int64)
    lib().sdao_chacha_expand(sp, C.c_size_t(s.size), C.c_int64(q), C.c_size_t(count), out.ctypes.data_as(I64P))
    return out


def chacha_combine(seeds, q, dimension, mode=0):
    s = np.ascontiguousarray(seeds, dtype=np.int64)
    if s.size == 0:
        return np.zeros(dimension, dtype=np.int64)
    assert s.ndim == 2
    out = np.empty(dimension, dtype=np.int64)
    lib().sdao_chacha_combine(s.ctypes.data_as(I64P), C.c_size_t(s.shape[0]), C.c_size_t(s.shape[1]), C.c_int64(q),
                              C.c_size_t(dimension), out.ctypes.data_as(I64P), mode)
    return out


def addsub(a, b, q, subtract=False, mode=0):
    a, ap = _i64(a)
    b, bp = _i64(b)
    assert a.size == b.size
    out = np.empty_like(a)
    lib().sdao_addsub(ap, bp, C.c_size_t(a.size), C.c_int64(q), int(subtract), out.ctypes.data_as(I64P), mode)
    return out


def drbg_fill(key: bytes, stream, batches, T, modulus, rounds=20):
    assert len(key) == 32
    out = np.empty(batches * T, dtype=np.int64)
    kb = (C.c_uint8 * 32)(*key)
    lib().sdao_drbg_fill(kb, rounds, C.c_uint64(stream), C.c_size_t(batches), C.c_uint32(T), C.c_int64(modulus),
                         out.ctypes.data_as(I64P))
    return out


def drbg_call_key(master: bytes, call_index: int) -> bytes:
    """key of the call_index-th CSPRNG-drawing call of a handle whose master key is `master`"""
    assert len(master) == 32
    out = (C.c_uint8 * 32)()
    lib().sdao_drbg_call_key((C.c_uint8 * 32)(*master), C.c_uint64(call_index), out)
    return bytes(out)


def fill_synthetic(participants, length, first_participant, seed, modulus):
    out = np.empty((participants, length), dtype=np.int64)
    lib().sdao_fill_synthetic(out.ctypes.data_as(I64P), C.c_size_t(participants), C.c_size_t(length), C.c_size_t(length),
                              C.c_uint64(first_participant), C.c_uint64(seed), C.c_int64(modulus))
    return out


def chacha_block(state16, rounds=20):
    st = np.ascontiguousarray(state16, dtype=np.uint32)
    out = np.empty(16, dtype=np.uint32)
    U32P = C.POINTER(C.c_uint32)
    lib().sdao_chacha_block(st.ctypes.data_as(U32P), rounds, out.ctypes.data_as(U32P))
    return out


def baseline_pass(packed, modulus, n, k, t, w2, w3, participants, length, first_participant, seed, key: bytes):
    kk = k if packed else 1
    B = (length + kk - 1) // kk
    sums = np.empty((n, B), dtype=np.int64)
    kb = (C.c_uint8 * 32)(*key)
    done = lib().sdao_baseline_pass(int(packed), C.c_int64(modulus), n, k, t, C.c_int64(w2), C.c_int64(w3),
                                    C.c_size_t(participants), C.c_size_t(length), C.c_uint64(first_participant),
                                    C.c_uint64(seed), kb, sums.ctypes.data_as(I64P))
    return done, sums


def varint_encode(values) -> bytes:
    v, vp = _i64(values)
    out = np.empty(v.size * 10 + 1, dtype=np.uint8)
    lib().sdao_varint_encode.restype = C.c_size_t
    n = lib().sdao_varint_encode(vp, C.c_size_t(v.size), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out[:n].tobytes()


def varint_decode(raw: bytes, cap=None):
    b = np.frombuffer(raw, dtype=np.uint8)
    cap = len(raw) if cap is None else cap
    out = np.empty(max(cap, 1), dtype=np.int64)
    lib().sdao_varint_decode.restype = C.c_size_t
    n = lib().sdao_varint_decode(b.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(b.size), out.ctypes.data_as(I64P),
                                 C.c_size_t(cap))
    return out[:n].copy()
