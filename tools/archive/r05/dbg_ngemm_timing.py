# phase timing of the narrow limb GEMM (library built with -DNG_TIMING: the first two shares of every workgroup's first batch
# column hold the cycles of the staging phase and of the row-tile phase)
import sys, numpy as np
sys.path.insert(0, '.')
from sda_amd import crypto
from sda_amd.device import DeviceBuffer
p, k, t, n, w2, w3 = 746497, 100, 155, 728, 95660, 610121
dim, P = 1048576, 200
sch = crypto.PackedShamir(k, n, t, p, w2, w3)
gen = crypto.ShareGenerator(sch)
gen.set_drbg_key(bytes(32))
B = gen.batch_count(dim)
Bs = (B + 15) // 16 * 16
rng = np.random.default_rng(1)
sec = rng.integers(0, p, size=(P, dim), dtype=np.int64)
d_sec = DeviceBuffer.from_numpy(sec)
d_out = DeviceBuffer(n * P * Bs).zero()
import time
from sda_amd.device import synchronize
for rep in range(4):
    synchronize(); t0 = time.perf_counter()
    gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_out.ptr, Bs, P * Bs, first_participant=0)
    synchronize(); wall = time.perf_counter() - t0
print("launch wall time %.3f ms (%d participants)" % (wall * 1e3, P))
o = d_out.to_numpy().reshape(n, P, Bs)
WGB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
row0 = o[0][:, 0:B:WGB]
st, gm = row0, o[0][:, 1:B:WGB]
for j, nm in ((2, "load pass (wave 0)"), (3, "load + draw passes"), (4, "wait at the barrier"), (5, "direct-row pass")):
    x = o[0][:, j:B:WGB]
    print("  %-22s mean %.0f median %.0f" % (nm, x.mean(), np.median(x)))
print("workgroups", st.size, "staging cycles mean %.0f median %.0f  | row-tile phase mean %.0f median %.0f (s_memtime ticks)" % (st.mean(), np.median(st), gm.mean(), np.median(gm)))
w0, w1 = o[0][:, 6:B:WGB].astype(np.int64), o[0][:, 7:B:WGB].astype(np.int64)
span = (w1.max() - w0.min()) * 10e-9
busy = (w1 - w0).sum() * 10e-9 / 256
print("device span %.3f ms, workgroup time summed / 256 CUs %.3f ms (%.1f %% of the span), cycles per workgroup / its duration = %.2f GHz" % (
    span * 1e3, busy * 1e3, 100 * busy / span, float((st + gm).mean()) / float((w1 - w0).mean() * 10)))
for w in range(8):
    v = [np.median(o[0][:, 8 + 4 * w + j:B:WGB]) for j in range(4)]
    print("  wave %d: load pass %.0f  load + draw %.0f  barrier wait %.0f  direct rows %.0f" % (w, *v))
