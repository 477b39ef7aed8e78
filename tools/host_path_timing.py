"""Where the time of a host-memory query goes (run on the GPU box): upload only, run only, download only, and the pipelined call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from nested_hashing_psi_amd import pie

cfg = bench.CONFIGS["C3"]
N, L, t, K, E, b = cfg["N"], cfg["L"], cfg["t"], cfg["K"], cfg["E"], cfg["b"]
cc = pie.PieContext(N, L, t)
rng = np.random.default_rng(1)
def limbs(prefix):
    out = np.zeros(tuple(prefix) + (L, N), dtype=np.uint64)
    for i, m in enumerate(cc.q):
        out[..., i, :] = rng.integers(0, int(m), tuple(prefix) + (N,), dtype=np.uint64)
    return out
op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=limbs((K, b, E)), preCalcRandomMask=limbs((b,)))
cc.load_relin_key(limbs((L, 2)))
idx, minus = limbs((K, E, 2)), limbs((2,))
pi, pm, pr = op.hostBuffers()
pi[...] = idx
pm[...] = minus
def med(f, n=15):
    ts = []
    for _ in range(n + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return sorted(ts[2:])[n // 2] * 1e3
tp = torch.from_numpy(pi.view(np.int64))
dev = torch.empty_like(tp, device="cuda")
res_dev = torch.empty((b, 2, L, N), dtype=torch.int64, device="cuda")
rp = torch.from_numpy(pr.view(np.int64))
print("torch H2D 28 MiB from the pinned staging array: %.3f ms" % med(lambda: (dev.copy_(tp, non_blocking=True), torch.cuda.synchronize())))
print("torch D2H 14 MiB into the pinned staging array: %.3f ms" % med(lambda: (rp.copy_(res_dev, non_blocking=True), torch.cuda.synchronize())))
op.setIndex(idx); op.setMinusCompareElement(minus)
print("run() alone, inputs resident: %.3f ms" % med(lambda: op.run(sync=True)))
print("setIndex(pageable) alone: %.3f ms" % med(lambda: op.setIndex(idx)))
print("setIndex(pinned) alone: %.3f ms" % med(lambda: op.setIndex(pi)))
print("getResultList alone: %.3f ms" % med(lambda: op.getResultList()))
print("runHost pinned, results in pinned: %.3f ms" % med(lambda: op.runHost(pi, pm, pr)))
print("runHost pageable: %.3f ms" % med(lambda: op.runHost(idx, minus)))
import ctypes as C
from nested_hashing_psi_amd._lib import lib, u64p
print("piehip_run_host pinned, no results: %.3f ms" % med(lambda: lib().piehip_run_host(cc._h, pi.ctypes.data_as(u64p), pm.ctypes.data_as(u64p), None)))
cc.close()
