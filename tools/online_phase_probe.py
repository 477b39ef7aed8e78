"""The server's online phase in process (what host/BatchedFHEPSIServer.hpp times as OnlineComputation): the query staged
ciphertext by ciphertext, then run_staged + wait.  For rocprofv3 timelines:  python tools/online_phase_probe.py [nq] [reps]"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from nested_hashing_psi_amd import pie

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
N, L, t, K, E, b, B = 16384, 4, 4296540161, 2, 14, 14, 9898
rng = np.random.default_rng(1)
cc = pie.PieContext(N, L, t)


def limbs(shape):
    out = np.zeros(shape + (L, N), dtype=np.uint64)
    for i in range(L):
        out[..., i, :] = rng.integers(0, int(cc.q[i]), shape + (N,), dtype=np.uint64)
    return out


cc.load_relin_key(limbs((L, 2)))
op = pie.BatchedFHEHIPPIE(cc, slots=rng.integers(0, 1000, (K, b, E, B), dtype=np.int64), mask_slots=rng.integers(1, 1000, (b, B), dtype=np.int64))
op.setQueryBatch(nq)
if len(sys.argv) > 3:
    cc.set_run_streams(int(sys.argv[3]))
bufs = [op.hostBuffers(query=q) for q in range(nq)]
idx, minus = limbs((K, E, 2)), limbs((2,))
for bi, bm, _ in bufs:
    bi[...] = idx
    bm[...] = minus
ts = []
for rep in range(reps):
    for q in range(nq):
        op.stageMinus(bufs[q][1], query=q)
        for h in range(K):
            for j in range(E):
                op.stageIndexCiphertext(h, j, bufs[q][0][h, j], query=q)
                time.sleep(0.0002)     # the next message arrives
    t0 = time.perf_counter()
    op.runStaged(bufs[0][2])
    t1 = time.perf_counter()
    op.waitHost()
    t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
print("nq %d: run_staged returns after / results in host memory after (us):" % nq, " ".join("%.0f/%.0f" % x for x in ts))
cc.close()
