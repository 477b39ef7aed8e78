"""Soak of the host-memory path: hundreds of staged runs (query pieces staged one ciphertext at a time, run_staged, results in
page-locked host memory), one query per run() and a batch of three with a key per client, every result list compared with the
device-resident path's (piehip_run + piehip_get_results) for the same inputs; queries alternate so that a stale download shows.
GPU box: python tools/soak_host.py [runs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from nested_hashing_psi_amd import pie

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = bench.CONFIGS["C3"]
N, L, t, K, E, b = cfg["N"], cfg["L"], cfg["t"], cfg["K"], cfg["E"], cfg["b"]
cc = pie.PieContext(N, L, t)
rng = np.random.default_rng(11)


def limbs(prefix):
    out = np.zeros(tuple(prefix) + (L, N), dtype=np.uint64)
    for i, m in enumerate(cc.q):
        out[..., i, :] = rng.integers(0, int(m), tuple(prefix) + (N,), dtype=np.uint64)
    return out


keys = [limbs((L, 2)) for _ in range(3)]
cc.load_relin_key(keys[0])
op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=limbs((K, b, E)), preCalcRandomMask=limbs((b,)))
queries = [(limbs((K, E, 2)), limbs((2,))) for _ in range(4)]
bad = 0
for nq in (1, 3):
    op.setQueryBatch(nq)
    if nq > 1:
        for i in range(nq):
            cc.load_relin_key(keys[i], query=i)
    # expected result lists through the device-resident path
    want = {}
    for first in range(4):
        for i in range(nq):
            idx, minus = queries[(first + i) % 4]
            op.setIndex(idx, query=i)
            op.setMinusCompareElement(minus, query=i)
        op.run()
        got = op.getResultList()
        want[first] = np.stack([np.array(got[i]) for i in range(nq)], axis=1) if nq > 1 else np.array(got)[:, None]
    bufs = [op.hostBuffers(query=i) for i in range(nq)]
    pr = bufs[0][2]
    for r in range(runs):
        first = r % 4
        for i in range(nq):
            idx, minus = queries[(first + i) % 4]
            bufs[i][0][...] = idx
            bufs[i][1][...] = minus
            op.stageMinus(bufs[i][1], query=i)
            for h in range(K):
                for j in range(E):
                    op.stageIndexCiphertext(h, j, bufs[i][0][h, j], query=i)
        op.runStaged(pr)
        op.waitHost()
        res = pr.reshape(want[first].shape)
        if not (res == want[first]).all():
            bad += 1
            print("nq %d run %d: mismatch" % (nq, r))
        if r % 100 == 99:
            print("nq %d run %d bad %d" % (nq, r + 1, bad), flush=True)
print("soak of the host-memory path: %s" % ("ok" if not bad else "%d mismatches" % bad))
cc.close()
sys.exit(1 if bad else 0)
