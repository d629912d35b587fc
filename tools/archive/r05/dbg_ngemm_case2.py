# the failing parametrisation of tests/test_ngemm_gpu.py::test_narrow_limb_gemm_vs_oracle_and_transform, with the mismatches located
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from sda_amd import crypto
from sda_amd.device import DeviceBuffer
from oracle import coracle
from test_ngemm_gpu import _root, KEY, TSS_P1
p, k, t, n, dim = TSS_P1, 70, 57, 242, 70 * 260
w2, w3 = _root(p, k + t + 1), _root(p, n + 1)
rng = np.random.default_rng(k * 1000 + n)
secrets = rng.integers(-(1 << 62), 1 << 62, size=dim, dtype=np.int64)
P = 3
sec2 = rng.integers(0, p, size=(P, dim), dtype=np.int64)
gen = crypto.ShareGenerator(crypto.PackedShamir(k, n, t, p, w2, w3))
B = gen.batch_count(dim)
if "norand" not in sys.argv:
    rand = np.random.default_rng(7).integers(-(1 << 62), 1 << 62, size=B * t, dtype=np.int64)
    got = gen.generate(secrets, rand)
    print("injected ok:", np.array_equal(got, coracle.packed_generate(p, k, t, n, w2, w3, secrets, rand)))
gen.set_drbg_key(KEY)
d_sec = DeviceBuffer.from_numpy(sec2)
Bs = (B + 15) // 16 * 16 + 16
first = (1 << 33) + 9
for share_map in (gen.SHARE_MAP_SYSTEMATIC, gen.SHARE_MAP_TSS_NODES):
    gen.set_csprng_share_map(share_map)
    for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
        d_out = DeviceBuffer(P * n * Bs).zero()
        gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_out.ptr, n * Bs, Bs, first_participant=first)
        o = d_out.to_numpy().reshape(P, n, Bs)
        for q in range(P):
            w = coracle.packed_generate_csprng(p, k, t, n, w2, w3, sec2[q], coracle.drbg_fill(KEY, first + q, B, t, p), share_map)
            bad = np.argwhere(o[q, :, :B] != w)
            print("map", share_map, "rep", rep, "participant", q, "mismatches", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:24])
            if len(bad) and "show" in sys.argv:
                r, c = bad[0]
                print("   first bad (row %d, col %d): got %d want %d" % (r, c, o[q, r, c], w[r, c]))
                hits = np.argwhere(w == o[q, r, c])
                print("   the value got appears in the expected shares at", hits[:6].tolist())
                print("   got around:", o[q, r, c - 2:c + 3].tolist(), " want:", w[r, c - 2:c + 3].tolist())
