"""A stream of host-memory queries over query slots, one query or a batch per run() -- the `ref_timer` stream legs of bench.py on
their own, for traces: python tools/host_batch_stream.py [batch] [slots] [rounds]"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from nested_hashing_psi_amd import pie

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nslots = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
one_call = len(sys.argv) > 4 and sys.argv[4] == "async"   # piehip_run_host_async instead of the piecewise calls (batch 1)
N, L, t, K, E, b, B = 16384, 4, 4296540161, 2, 14, 14, 9898
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)


def limbs(cc, shape):
    out = np.zeros(shape + (L, N), dtype=np.uint64)
    for i in range(L):
        out[..., i, :] = rng.integers(0, int(cc.q[i]), shape + (N,), dtype=np.uint64)
    return out


ccs = [pie.PieContext(N, L, t, stream=torch.cuda.Stream(dev).cuda_stream) for _ in range(nslots)]
ccs[0].load_relin_key(limbs(ccs[0], (L, 2)))
slots = rng.integers(0, 1000, (K, b, E, B), dtype=np.int64)
ops = [pie.BatchedFHEHIPPIE(ccs[0], slots=slots, mask_slots=rng.integers(1, 1000, (b, B), dtype=np.int64))]
ops += [pie.BatchedFHEHIPPIE(c, attachTo=ops[0]) for c in ccs[1:]]
idx, minus = limbs(ccs[0], (K, E, 2)), limbs(ccs[0], (2,))
bufs = []
for o in ops:
    o.cc.set_run_streams(1)
    o.setQueryBatch(batch)
    qb = [o.hostBuffers(query=q) for q in range(batch)]
    for bi, bm, _ in qb:
        bi[...] = idx
        bm[...] = minus
    bufs.append(qb)


def stream(nb):
    for i in range(nb + nslots):
        o, qb = ops[i % nslots], bufs[i % nslots]
        if i >= nslots:
            o.waitHost()
        if i < nb and one_call:
            o.runHostAsync(qb[0][0], qb[0][1], qb[0][2])
        elif i < nb:
            for q in range(batch):
                o.stageMinus(qb[q][1], query=q)
            for h in range(K):
                for q in range(batch):
                    o.stageIndexRow(h, qb[q][0][h], query=q)
            o.runStaged(qb[0][2])


stream(2 * nslots)
torch.cuda.synchronize()
t0 = time.perf_counter()
stream(rounds * nslots)
dt = time.perf_counter() - t0
print("batch %d, %d slots: %.3f ms per query (%.1f k ct/s)" % (batch, nslots, dt / (rounds * nslots * batch) * 1e3, b * rounds * nslots * batch / dt / 1e3))
for c in reversed(ccs):
    c.close()
