# where do the limb GEMM's shares differ from the restatement?  (debugging aid)  usage: python tools/dbg_ngemm_case.py k t n dim [P]
import sys, numpy as np
sys.path.insert(0, '.')
from sda_amd import crypto
from sda_amd.device import DeviceBuffer
from oracle import coracle
sys.path.insert(0, 'tests')
from test_ngemm_gpu import _root, KEY, TSS_P1
k, t, n, dim = (int(x) for x in sys.argv[1:5])
P = int(sys.argv[5]) if len(sys.argv) > 5 else 3
FIRST = int(sys.argv[6]) if len(sys.argv) > 6 else 5
p = TSS_P1
w2, w3 = _root(p, k + t + 1), _root(p, n + 1)
gen = crypto.ShareGenerator(crypto.PackedShamir(k, n, t, p, w2, w3))
gen.set_drbg_key(KEY)
B = gen.batch_count(dim)
Bs = (B + 15) // 16 * 16 + 16
rng = np.random.default_rng(k * 1000 + n)
sec = rng.integers(0, p, size=(P, dim), dtype=np.int64)
d_sec = DeviceBuffer.from_numpy(sec)
for rep in range(int(sys.argv[7]) if len(sys.argv) > 7 else 3):
    d_out = DeviceBuffer(P * n * Bs).zero()
    gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_out.ptr, n * Bs, Bs, first_participant=FIRST)
    o = d_out.to_numpy().reshape(P, n, Bs)
    for q in range(P):
        w = coracle.packed_generate_csprng(p, k, t, n, w2, w3, sec[q], coracle.drbg_fill(KEY, FIRST + q, B, t, p), gen.csprng_share_map())
        bad = np.argwhere(o[q, :, :B] != w)
        print("rep", rep, "participant", q, "mismatches", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:24])
