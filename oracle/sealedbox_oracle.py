"""CPU ORACLE for libsodium sealed boxes: what the reference calls through sodiumoxide 0.0.14 in
client/src/crypto/encryption/sodium.rs:43 (`sealedbox::seal`) and :78 (`sealedbox::open`).
TEST INFRASTRUCTURE ONLY - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/.

libsodium is a third-party dependency absent from /root/reference (linked through the un-vendored crate
`sodiumoxide = "0.0.14"`, client/Cargo.toml) and from this image, so its PUBLISHED algorithms are restated here:

    crypto_box_seal(m, pk):  (epk, esk) <- fresh X25519 key pair;  nonce = BLAKE2b-24(epk || pk);
                             c = epk || crypto_box_easy(m, nonce, pk, esk)                       [48 bytes longer than m]
    crypto_box_easy        = crypto_box_curve25519xsalsa20poly1305: k = HSalsa20(X25519(sk, pk), 0^16);
                             secretbox_xsalsa20poly1305(m, nonce, k) = tag16 || (m xor stream[32:]),
                             stream = XSalsa20(k, nonce), tag = Poly1305(stream[0:32])(ciphertext)

PARITY PINNING: each primitive is checked against its published vectors in tests/test_oracle_sealedbox.py -
X25519: RFC 7748 section 5.2 and 6.1; Poly1305: RFC 8439 section 2.5.2; BLAKE2b: RFC 7693 appendix A and Python's
hashlib.blake2b; HSalsa20 / XSalsa20 / crypto_box: the worked example of "Cryptography in NaCl" (Bernstein, sections
5-10: firstkey, secondkey, the first stream bytes and the boxed packet) [vectors recalled from the paper / libsodium's
test suite - the box ciphertext and tag reproduce, which no wrong restatement would].  The sealed-box COMPOSITION
(epk || box, nonce = BLAKE2b-24(epk || pk)) has no reference-generated fixture ("parity unpinned" by anything inside
/root/reference: sealing is randomised); it follows libsodium's documented construction.

Byte work in numpy (Salsa20 is vectorised over blocks); X25519 and Poly1305 in Python integers.
"""
from __future__ import annotations

import hashlib
from typing import Tuple

import numpy as np

MASK32 = 0xFFFFFFFF
SIGMA = (0x61707865, 0x3320646E, 0x79622D32, 0x6B206574)
P25519 = 2**255 - 19
SEAL_BYTES = 48


# ---- X25519 (RFC 7748 section 5) ----------------------------------------------------------------------
def x25519(k: bytes, u: bytes) -> bytes:
    assert len(k) == 32 and len(u) == 32
    kb = bytearray(k)
    kb[0] &= 248; kb[31] &= 127; kb[31] |= 64                                  # decodeScalar25519
    kn = int.from_bytes(kb, "little")
    x1 = int.from_bytes(u, "little") & ((1 << 255) - 1)                          # decodeUCoordinate masks bit 255
    x2, z2, x3, z3, swap = 1, 0, x1, 1, 0
    p, a24 = P25519, 121665
    for t in range(254, -1, -1):
        kt = (kn >> t) & 1
        swap ^= kt
        if swap:
            x2, x3, z2, z3 = x3, x2, z3, z2
        swap = kt
        A = (x2 + z2) % p; AA = A * A % p
        B = (x2 - z2) % p; BB = B * B % p
        E = (AA - BB) % p
        Cc = (x3 + z3) % p; D = (x3 - z3) % p
        DA = D * A % p; CB = Cc * B % p
        x3 = (DA + CB) % p; x3 = x3 * x3 % p
        z3 = (DA - CB) % p; z3 = x1 * z3 * z3 % p
        x2 = AA * BB % p
        z2 = E * (AA + a24 * E) % p
    if swap:
        x2, x3, z2, z3 = x3, x2, z3, z2
    return (x2 * pow(z2, p - 2, p) % p).to_bytes(32, "little")


BASEPOINT = (9).to_bytes(32, "little")


def x25519_base(k: bytes) -> bytes:
    return x25519(k, BASEPOINT)


# ---- Salsa20 family (Bernstein, "The Salsa20 family of stream ciphers"; "Extending the Salsa20 nonce") --------
def _rotl(x, n):
    return ((x << np.uint32(n)) | (x >> np.uint32(32 - n)))


def _salsa_rounds(x):
    """20 rounds on a list of 16 uint32 numpy arrays (vectorised over blocks), in place"""
    def qr(a, b, c, d):
        x[b] ^= _rotl(x[a] + x[d], 7)
        x[c] ^= _rotl(x[b] + x[a], 9)
        x[d] ^= _rotl(x[c] + x[b], 13)
        x[a] ^= _rotl(x[d] + x[c], 18)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(5, 9, 13, 1); qr(10, 14, 2, 6); qr(15, 3, 7, 11)      # column round
        qr(0, 1, 2, 3); qr(5, 6, 7, 4); qr(10, 11, 8, 9); qr(15, 12, 13, 14)      # row round
    return x


def _words(b: bytes):
    return [int.from_bytes(b[i:i + 4], "little") for i in range(0, len(b), 4)]


def hsalsa20(key: bytes, inp: bytes) -> bytes:
    assert len(key) == 32 and len(inp) == 16
    k, n = _words(key), _words(inp)
    st = [SIGMA[0], k[0], k[1], k[2], k[3], SIGMA[1], n[0], n[1], n[2], n[3], SIGMA[2], k[4], k[5], k[6], k[7], SIGMA[3]]
    with np.errstate(over="ignore"):
        x = _salsa_rounds([np.array([w], dtype=np.uint32) for w in st])
    return b"".join(int(x[i][0]).to_bytes(4, "little") for i in (0, 5, 10, 15, 6, 7, 8, 9))


def salsa20_stream(key: bytes, nonce8: bytes, nbytes: int, counter0: int = 0) -> bytes:
    assert len(key) == 32 and len(nonce8) == 8
    nblocks = (nbytes + 63) // 64
    if nblocks == 0:
        return b""
    k, n = _words(key), _words(nonce8)
    ctr = np.arange(counter0, counter0 + nblocks, dtype=np.uint64)
    const = lambda w: np.full(nblocks, w, dtype=np.uint32)
    st = [const(SIGMA[0]), const(k[0]), const(k[1]), const(k[2]), const(k[3]), const(SIGMA[1]), const(n[0]), const(n[1]),
          (ctr & np.uint64(MASK32)).astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32), const(SIGMA[2]),
          const(k[4]), const(k[5]), const(k[6]), const(k[7]), const(SIGMA[3])]
    with np.errstate(over="ignore"):
        x = _salsa_rounds([a.copy() for a in st])
        out = np.stack([x[i] + st[i] for i in range(16)], axis=1)                  # [blocks][16] little-endian words
    return out.astype("<u4").tobytes()[:nbytes]


def xsalsa20_stream(key: bytes, nonce24: bytes, nbytes: int) -> bytes:
    assert len(nonce24) == 24
    return salsa20_stream(hsalsa20(key, nonce24[:16]), nonce24[16:], nbytes)


# ---- Poly1305 (RFC 8439 section 2.5) -------------------------------------------------------------------------
def poly1305(key32: bytes, msg: bytes) -> bytes:
    assert len(key32) == 32
    r = int.from_bytes(key32[:16], "little") & 0x0FFFFFFC0FFFFFFC0FFFFFFC0FFFFFFF
    s = int.from_bytes(key32[16:], "little")
    p = (1 << 130) - 5
    h = 0
    for i in range(0, len(msg), 16):
        blk = msg[i:i + 16]
        h = (h + int.from_bytes(blk, "little") + (1 << (8 * len(blk)))) * r % p
    return ((h + s) & ((1 << 128) - 1)).to_bytes(16, "little")


# ---- BLAKE2b (RFC 7693), unkeyed, one-shot -----------------------------------------------------------------------
_B2_IV = (0x6A09E667F3BCC908, 0xBB67AE8584CAA73B, 0x3C6EF372FE94F82B, 0xA54FF53A5F1D36F1,
          0x510E527FADE682D1, 0x9B05688C2B3E6C1F, 0x1F83D9ABFB41BD6B, 0x5BE0CD19137E2179)
_B2_SIGMA = ((0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15), (14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3),
             (11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4), (7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8),
             (9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13), (2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9),
             (12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11), (13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10),
             (6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5), (10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0),
             (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15), (14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3))
_M64 = (1 << 64) - 1


def _rotr64(x, n):
    return ((x >> n) | (x << (64 - n))) & _M64


def _b2_compress(h, block: bytes, t: int, last: bool):
    m = [int.from_bytes(block[8 * i:8 * i + 8], "little") for i in range(16)]
    v = list(h) + list(_B2_IV)
    v[12] ^= t & _M64
    v[13] ^= t >> 64
    if last:
        v[14] ^= _M64

    def g(a, b, c, d, x, y):
        v[a] = (v[a] + v[b] + x) & _M64; v[d] = _rotr64(v[d] ^ v[a], 32)
        v[c] = (v[c] + v[d]) & _M64; v[b] = _rotr64(v[b] ^ v[c], 24)
        v[a] = (v[a] + v[b] + y) & _M64; v[d] = _rotr64(v[d] ^ v[a], 16)
        v[c] = (v[c] + v[d]) & _M64; v[b] = _rotr64(v[b] ^ v[c], 63)
    for r in range(12):
        s = _B2_SIGMA[r]
        g(0, 4, 8, 12, m[s[0]], m[s[1]]); g(1, 5, 9, 13, m[s[2]], m[s[3]])
        g(2, 6, 10, 14, m[s[4]], m[s[5]]); g(3, 7, 11, 15, m[s[6]], m[s[7]])
        g(0, 5, 10, 15, m[s[8]], m[s[9]]); g(1, 6, 11, 12, m[s[10]], m[s[11]])
        g(2, 7, 8, 13, m[s[12]], m[s[13]]); g(3, 4, 9, 14, m[s[14]], m[s[15]])
    return [h[i] ^ v[i] ^ v[i + 8] for i in range(8)]


def blake2b(data: bytes, outlen: int) -> bytes:
    h = list(_B2_IV)
    h[0] ^= 0x01010000 ^ outlen
    n = len(data)
    off = 0
    while n - off > 128:
        h = _b2_compress(h, data[off:off + 128], off + 128, False)
        off += 128
    h = _b2_compress(h, data[off:].ljust(128, b"\0"), n, True)
    return b"".join(x.to_bytes(8, "little") for x in h)[:outlen]


# ---- secretbox / box / sealed box -------------------------------------------------------------------------------
def secretbox(m: bytes, nonce24: bytes, key: bytes) -> bytes:
    stream = xsalsa20_stream(key, nonce24, 32 + len(m))
    c = (np.frombuffer(m, dtype=np.uint8) ^ np.frombuffer(stream[32:], dtype=np.uint8)).tobytes()
    return poly1305(stream[:32], c) + c


def secretbox_open(boxed: bytes, nonce24: bytes, key: bytes) -> bytes:
    if len(boxed) < 16:
        raise ValueError("Sodium decryption failure")
    stream = xsalsa20_stream(key, nonce24, 32 + len(boxed) - 16)
    tag, c = boxed[:16], boxed[16:]
    if poly1305(stream[:32], c) != tag:
        raise ValueError("Sodium decryption failure")
    return (np.frombuffer(c, dtype=np.uint8) ^ np.frombuffer(stream[32:], dtype=np.uint8)).tobytes()


def box_beforenm(pk: bytes, sk: bytes) -> bytes:
    shared = x25519(sk, pk)
    if shared == bytes(32):
        # libsodium's crypto_scalarmult_curve25519 returns -1 for an all-zero result (a small-order point), and
        # crypto_box_beforenm / crypto_box_seal_open fail with it
        raise ValueError("Sodium decryption failure")
    return hsalsa20(shared, bytes(16))


def box(m: bytes, nonce24: bytes, pk: bytes, sk: bytes) -> bytes:
    return secretbox(m, nonce24, box_beforenm(pk, sk))


def box_open(boxed: bytes, nonce24: bytes, pk: bytes, sk: bytes) -> bytes:
    return secretbox_open(boxed, nonce24, box_beforenm(pk, sk))


def seal_nonce(epk: bytes, pk: bytes) -> bytes:
    return blake2b(epk + pk, 24)


def seal(m: bytes, pk: bytes, esk: bytes) -> bytes:
    """sealedbox::seal (sodium.rs:43) with the ephemeral secret key injected (libsodium draws it from randombytes)"""
    epk = x25519_base(esk)
    return epk + box(m, seal_nonce(epk, pk), pk, esk)


def seal_open(c: bytes, pk: bytes, sk: bytes) -> bytes:
    """sealedbox::open (sodium.rs:78); failure -> "Sodium decryption failure" (sodium.rs:80)"""
    if len(c) < SEAL_BYTES:
        raise ValueError("Sodium decryption failure")
    epk = c[:32]
    return box_open(c[32:], seal_nonce(epk, pk), epk, sk)
