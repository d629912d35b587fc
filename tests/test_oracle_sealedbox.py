"""Pins oracle/sealedbox_oracle.py (the checker for SURVEY.md 8f rank 4) against the PUBLISHED vectors of every
primitive a libsodium sealed box is made of, and against the committed sealed-box fixtures.  CPU only."""
import hashlib

import pytest

from conftest import load_golden
from oracle import sealedbox_oracle as so

h = bytes.fromhex
G = load_golden("sealedbox.json")


def test_x25519_rfc7748():
    for v in G["kats"]["x25519"]:
        assert so.x25519(h(v["scalar"]), h(v["u"])).hex() == v["out"], v["source"]
    for v in G["kats"]["x25519_base"]:
        assert so.x25519_base(h(v["scalar"])).hex() == v["out"]
    # RFC 7748 section 5.2, iterated: k, u = X25519(k, u), k   (1 and 1000 iterations)
    k = u = h("0900000000000000000000000000000000000000000000000000000000000000")
    k, u = so.x25519(k, u), k
    assert k.hex() == "422c8e7a6227d7bca1350b3e2bb7279f7897b87bb6854b783c60e80311ae3079"
    for _ in range(999):
        k, u = so.x25519(k, u), k
    assert k.hex() == "684cf59ba83309552800ef566f2f4d3c1c3887c49360e3875f2eb94d99532c51"


def test_salsa_family_nacl_paper():
    for v in G["kats"]["hsalsa20"]:
        assert so.hsalsa20(h(v["key"]), h(v["in"])).hex() == v["out"], v["source"]
    for v in G["kats"]["xsalsa20_stream"]:
        assert so.xsalsa20_stream(h(v["key"]), h(v["nonce"]), 32).hex() == v["first32"]
    # the stream is seekable by block: any window equals the same window of a longer stream
    key, nonce = h(G["kats"]["secretbox"][0]["key"]), h(G["kats"]["secretbox"][0]["nonce"])
    long = so.xsalsa20_stream(key, nonce, 1000)
    sub = so.hsalsa20(key, nonce[:16])
    assert so.salsa20_stream(sub, nonce[16:], 64 * 3, counter0=5) == long[320:512]


def test_poly1305_rfc8439_and_blake2b():
    for v in G["kats"]["poly1305"]:
        assert so.poly1305(h(v["key"]), h(v["msg"])).hex() == v["tag"]
    # RFC 7693 appendix A: BLAKE2b-512("abc")
    assert so.blake2b(b"abc", 64).hex() == ("ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d1"
                                            "7d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923")
    for data in (b"", b"a", bytes(range(64)), bytes(range(128)), bytes(200), bytes(range(256)) * 3):
        for outlen in (24, 32, 64):
            assert so.blake2b(data, outlen) == hashlib.blake2b(data, digest_size=outlen).digest()


def test_secretbox_and_box_nacl_paper():
    v = G["kats"]["secretbox"][0]
    assert so.secretbox(h(v["m"]), h(v["nonce"]), h(v["key"])).hex() == v["c"]
    assert so.secretbox_open(h(v["c"]), h(v["nonce"]), h(v["key"])) == h(v["m"])
    alice_sk = h("77076d0a7318a57d3c16c17251b26645df4c2f87ebc0992ab177fba51db92c2a")
    bob_pk = h("de9edb7d7b7dc1b4d35b61c2ece435373f8343c85b78674dadfc7e146f882b4f")
    assert so.box(h(v["m"]), h(v["nonce"]), bob_pk, alice_sk).hex() == v["c"]          # crypto_box = beforenm + secretbox
    bad = bytearray(h(v["c"])); bad[20] ^= 1
    with pytest.raises(ValueError, match="Sodium decryption failure"):
        so.secretbox_open(bytes(bad), h(v["nonce"]), h(v["key"]))


def test_sealed_fixtures_roundtrip_and_tamper():
    for v in G["sealed"]["vectors"]:
        pk, sk, esk, m, c = (h(v[k]) for k in ("pk", "sk", "esk", "m", "c"))
        assert so.seal(m, pk, esk) == c and so.seal_open(c, pk, sk) == m
        assert len(c) == len(m) + so.SEAL_BYTES and c[:32] == so.x25519_base(esk)
        for pos in {0, 31, 32, 47, len(c) - 1}:
            bad = bytearray(c); bad[pos] ^= 0x80
            with pytest.raises(ValueError, match="Sodium decryption failure"):
                so.seal_open(bytes(bad), pk, sk)
    with pytest.raises(ValueError):
        so.seal_open(bytes(47), pk, sk)
    assert G["cross_checks"]["openssl_x25519_random_pairs"] >= 0
