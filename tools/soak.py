"""Soak of the two-queue run(): thousands of pipelined runs on fixed inputs, results compared with the first run's
(bit-identical expected), interleaved with input changes.  GPU box: python tools/soak.py [runs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from nested_hashing_psi_amd import pie

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cfg = bench.CONFIGS["C3"]
N, L, t, K, E, b = cfg["N"], cfg["L"], cfg["t"], cfg["K"], cfg["E"], cfg["b"]
cc = pie.PieContext(N, L, t)
rng = np.random.default_rng(5)


def limbs(prefix):
    out = np.zeros(tuple(prefix) + (L, N), dtype=np.uint64)
    for i, m in enumerate(cc.q):
        out[..., i, :] = rng.integers(0, int(m), tuple(prefix) + (N,), dtype=np.uint64)
    return out


cc.load_relin_key(limbs((L, 2)))
op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=limbs((K, b, E)), preCalcRandomMask=limbs((b,)))
idx = [limbs((K, E, 2)), limbs((K, E, 2))]
op.setMinusCompareElement(limbs((2,)))
ref = []
for i in range(2):
    op.setIndex(idx[i])
    op.run()
    ref.append(op.getResultList().copy())
assert not (ref[0] == ref[1]).all()
bad = 0
cur = 1
for r in range(runs):
    if r % 97 == 0:
        cur ^= 1
        op.setIndex(idx[cur])
    op.run(sync=False)
    if r % 211 == 210:
        if not (op.getResultList() == ref[cur]).all():
            bad += 1
            print("mismatch at run", r)
print("soak: %d runs, %d mismatches" % (runs, bad))
sys.exit(1 if bad else 0)
