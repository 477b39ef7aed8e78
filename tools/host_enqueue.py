"""Host-side cost of enqueueing run() (no sync) vs the GPU time of the same K runs, per stream count.
Usage on the GPU box: python tools/host_enqueue.py [queues]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from nested_hashing_psi_amd import pie


def limbs(rng, moduli, prefix, N):
    out = np.zeros(tuple(prefix) + (len(moduli), N), dtype=np.uint64)
    for i, m in enumerate(moduli):
        out[..., i, :] = rng.integers(0, int(m), tuple(prefix) + (N,), dtype=np.uint64)
    return out


cfg = bench.CONFIGS["C3"]
cc = pie.PieContext(cfg["N"], cfg["L"], cfg["t"])
rng = np.random.default_rng(1)
K, E, b, L, N = cfg["K"], cfg["E"], cfg["b"], cfg["L"], cfg["N"]
op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=limbs(rng, cc.q, (K, b, E), N), preCalcRandomMask=limbs(rng, cc.q, (b,), N))
cc.load_relin_key(limbs(rng, cc.q, (L, 2), N))
op.setIndex(limbs(rng, cc.q, (K, E, 2), N))
op.setMinusCompareElement(limbs(rng, cc.q, (2,), N))
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cc.set_run_streams(nq)
for _ in range(20):
    op.run(sync=False)
op.sync()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    op.run(sync=False)
t1 = time.perf_counter()
op.sync()
t2 = time.perf_counter()
print("streams=%s host enqueue %.1f us/run, total %.1f us/run" % (nq or "default", (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
