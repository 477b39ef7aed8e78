"""stage A time per bin layer for different layer counts (different layers-per-thread choices). GPU box:
python tools/stage_a_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from nested_hashing_psi_amd import pie

cfg = bench.CONFIGS["C3"]
N, L, t, K, E = cfg["N"], int(os.environ.get("SWEEP_L", cfg["L"])), cfg["t"], cfg["K"], cfg["E"]
cc = pie.PieContext(N, L, t)
cc.set_run_streams(1)
rng = np.random.default_rng(1)


def limbs(prefix):
    out = np.zeros(tuple(prefix) + (L, N), dtype=np.uint64)
    for i, m in enumerate(cc.q):
        out[..., i, :] = rng.integers(0, int(m), tuple(prefix) + (N,), dtype=np.uint64)
    return out


cc.load_relin_key(limbs((L, 2)))
idx, minus = limbs((K, E, 2)), limbs((2,))
for b in (4, 5, 6, 7, 8, 12, 14, 16, 21):
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=limbs((K, b, E)), preCalcRandomMask=limbs((b,)))
    op.setIndex(idx)
    op.setMinusCompareElement(minus)
    cc.set_profiling(True)
    tot = 0.0
    for _ in range(10):
        op.run()
        tot += cc.profile()["stage_a_mac"]["ms"]
    cc.set_profiling(False)
    print("L=%d b=%2d stage_a %.1f us  %.2f us/bin  %.3f us/bin/limb" % (L, b, tot / 10 * 1e3, tot / 10 * 1e3 / b, tot / 10 * 1e3 / b / L))
