"""Soak of the query slots: P slots on one database, every round each slot gets one of a few fixed queries (host or device
inputs, alternating), all slots run several times back to back without waiting, and every slot's result list is compared
with the reference result of its query (computed one query at a time beforehand).  GPU box: python tools/soak_slots.py [rounds] [P]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from nested_hashing_psi_amd import pie

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
P = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = bench.CONFIGS["C3"]
N, L, t, K, E, b = cfg["N"], cfg["L"], cfg["t"], cfg["K"], cfg["E"], 6
cc = pie.PieContext(N, L, t, stream=torch.cuda.Stream().cuda_stream)
rng = np.random.default_rng(5)


def limbs(prefix):
    out = np.zeros(tuple(prefix) + (L, N), dtype=np.uint64)
    for i, m in enumerate(cc.q):
        out[..., i, :] = rng.integers(0, int(m), tuple(prefix) + (N,), dtype=np.uint64)
    return out


cc.load_relin_key(limbs((L, 2)))
op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=limbs((K, b, E)), preCalcRandomMask=limbs((b,)))
NQ = 4
queries = [(limbs((K, E, 2)), limbs((2,))) for _ in range(NQ)]
ref = []
for idx, minus in queries:
    op.setMinusCompareElement(minus)
    op.setIndex(idx)
    op.run()
    ref.append(op.getResultList().copy())
dev_q = [(torch.from_numpy(i.view(np.int64)).cuda(), torch.from_numpy(m.view(np.int64)).cuda()) for i, m in queries]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(P - 1)]
it = iter(streams)
pipe = pie.QueryPipeline(op, P, lambda: pie.PieContext(N, L, t, stream=next(it).cuda_stream))
bufs = [s.hostBuffers() for s in pipe.slots]
bad = 0
for r in range(rounds):
    pick = rng.integers(0, NQ, P)
    mode = r % 3
    for s, q, (pi, pm, pr) in zip(pipe.slots, pick, bufs):
        if mode == 0:      # device-resident inputs
            s.setIndexDevice(dev_q[q][0].data_ptr())
            s.setMinusCompareElementDevice(dev_q[q][1].data_ptr())
        elif mode == 1:    # host inputs through the separate calls
            s.setMinusCompareElement(queries[q][1])
            s.setIndex(queries[q][0])
        else:              # host inputs through the asynchronous one-call path
            pi[...] = queries[q][0]
            pm[...] = queries[q][1]
    if mode == 2:
        for s, (pi, pm, pr) in zip(pipe.slots, bufs):
            s.runHostAsync(pi, pm, pr)
        for s, q, (pi, pm, pr) in zip(pipe.slots, pick, bufs):
            s.waitHost()
            if not (pr == ref[q]).all():
                bad += 1
    else:
        for _ in range(1 + r % 4):
            pipe.run_all()
        pipe.sync()
        for s, q in zip(pipe.slots, pick):
            if not (s.getResultList() == ref[q]).all():
                bad += 1
    if r % 50 == 0:
        print("round %d: %d mismatches so far" % (r, bad), flush=True)
print("soak: %d rounds x %d slots, %d mismatches" % (rounds, P, bad))
pipe.close()
cc.close()
sys.exit(1 if bad else 0)
