"""Host unit test of sda_amd/csrc/sbox_primitives.hpp - the header the sealed-box kernels are built from - compiled
with g++ and compared with the published vectors (tests/golden/sealedbox.json) and the oracle on random operands.
(The kernels themselves are covered by tests/test_sealedbox_gpu.py; this catches arithmetic slips without a GPU.)"""
import hashlib
import os
import random
import subprocess

import pytest

from conftest import load_golden
from oracle import sealedbox_oracle as so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("sbox") / "sbox_primitives_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Wno-unknown-pragmas", "-Werror",
                           os.path.join(ROOT, "tests", "cpp", "sbox_primitives_test.cpp"), "-o", out])
    return out


def _run(exe, cmds):
    out = subprocess.run([exe], input="\n".join(cmds) + "\n", capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return out.stdout.split()


def test_header_primitives_match_published_vectors_and_the_oracle(exe):
    g = load_golden("sealedbox.json")["kats"]
    rng = random.Random(1)
    rb = lambda n: bytes(rng.randrange(256) for _ in range(n))
    cmds, want = [], []
    for v in g["x25519"]:
        cmds.append(f"x25519 {v['scalar']} {v['u']}"); want.append(v["out"])
    for v in g["x25519_base"]:
        cmds.append(f"x25519 {v['scalar']} {(9).to_bytes(32, 'little').hex()}"); want.append(v["out"])
    for v in g["hsalsa20"]:
        cmds.append(f"hsalsa {v['key']} {v['in']}"); want.append(v["out"])
    for v in g["poly1305"]:
        cmds.append(f"poly {v['key']} {v['msg']}"); want.append(v["tag"])
    for _ in range(12):
        k, u = rb(32), rb(32)
        cmds.append(f"x25519 {k.hex()} {u.hex()}"); want.append(so.x25519(k, u).hex())
    for u in (bytes(32), (1).to_bytes(32, "little"), (2**255 - 19).to_bytes(32, "little"), b"\xff" * 32,
              (2**255 - 20).to_bytes(32, "little")):                       # edge u-coordinates, incl. non-canonical ones
        k = rb(32)
        cmds.append(f"x25519 {k.hex()} {u.hex()}"); want.append(so.x25519(k, u).hex())
    for ctr in (0, 1, 2**32 - 1, 2**32, 2**40 + 7):
        k, n = rb(32), rb(8)
        cmds.append(f"salsa {k.hex()} {n.hex()} {ctr}"); want.append(so.salsa20_stream(k, n, 64, ctr).hex())
    for _ in range(5):
        e, p = rb(32), rb(32)
        cmds.append(f"nonce {e.hex()} {p.hex()}"); want.append(hashlib.blake2b(e + p, digest_size=24).digest().hex())
    for n in (0, 1, 15, 16, 17, 31, 32, 33, 100, 1000):
        k, m = rb(32), rb(n)
        cmds.append(f"poly {k.hex()} {m.hex() or '-'}"); want.append(so.poly1305(k, m).hex())
    cmds.append(f"poly {'ff' * 32} {'ff' * 64}"); want.append(so.poly1305(b"\xff" * 32, b"\xff" * 64).hex())
    got = _run(exe, cmds)
    assert len(got) == len(want)
    for c, a, b in zip(cmds, got, want):
        assert a == b, c[:40]
