"""Soak of query batches: thousands of runs of a batch of three on fixed inputs, every run's [nq][b] result list compared with
the one-query-per-run() results of the same queries (bit-identical expected), batch size and queue count switched on the way.
GPU box: python tools/soak_batch.py [runs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from nested_hashing_psi_amd import pie

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = bench.CONFIGS["C3"]
N, L, t, K, E, b = cfg["N"], cfg["L"], cfg["t"], cfg["K"], cfg["E"], cfg["b"]
cc = pie.PieContext(N, L, t)
rng = np.random.default_rng(6)


def limbs(prefix):
    out = np.zeros(tuple(prefix) + (L, N), dtype=np.uint64)
    for i, m in enumerate(cc.q):
        out[..., i, :] = rng.integers(0, int(m), tuple(prefix) + (N,), dtype=np.uint64)
    return out


cc.load_relin_key(limbs((L, 2)))
op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=limbs((K, b, E)), preCalcRandomMask=limbs((b,)))
queries = [(limbs((K, E, 2)), limbs((2,))) for _ in range(4)]
ref = []
for idx, minus in queries:
    op.setMinusCompareElement(minus)
    op.setIndex(idx)
    op.run()
    ref.append(op.getResultList().copy())
bad = 0
nq = 0
for r in range(runs):
    if r % 211 == 0:
        nq = 3 if nq != 3 else 4
        op.setQueryBatch(nq)
        for q in range(nq):
            op.setMinusCompareElement(queries[q][1], query=q)
            op.setIndex(queries[q][0], query=q)
        cc.set_run_streams((r // 211) % 3)     # library default, one queue, two queues
    op.run(sync=False)
    if r % 13 == 0:
        got = op.getResultList()
        for q in range(nq):
            if not (got[q] == ref[q]).all():
                bad += 1
                print("run %d: query %d of a batch of %d differs" % (r, q, nq))
    if r % 500 == 0:
        print("run", r, "bad", bad, flush=True)
got = op.getResultList()
for q in range(nq):
    bad += int(not (got[q] == ref[q]).all())
print("soak of %d batched runs: %s" % (runs, "ok" if not bad else "%d MISMATCHES" % bad))
sys.exit(1 if bad else 0)
