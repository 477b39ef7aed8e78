"""Where does a stream of host-memory queries over the query slots spend its time?  (run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from nested_hashing_psi_amd import pie

cfg = dict(bench.CONFIGS["C3"])
N, L, t, K, E, b = cfg["N"], cfg["L"], cfg["t"], cfg["K"], cfg["E"], cfg["b"]
dev = torch.device("cuda", 0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ops = []
for i in range(P):
    st = torch.cuda.Stream(dev)
    cc = pie.PieContext(N, L, t, stream=st.cuda_stream)
    gen = torch.Generator(device=dev); gen.manual_seed(1 + i)
    if i == 0:
        evk = bench.uniform_limbs(torch, (L, 2), cc.q, N, dev, gen)
        idx = bench.uniform_limbs(torch, (K, E, 2), cc.q, N, dev, gen)
        minus = bench.uniform_limbs(torch, (2,), cc.q, N, dev, gen)
        torch.cuda.synchronize()
        cc.load_relin_key(evk.cpu().numpy().view(np.uint64))
        op = bench.synthetic_operator(pie, cc, cfg, b, np.random.default_rng(3), (idx, minus))
    else:
        op = pie.BatchedFHEHIPPIE(cc, attachTo=ops[0][0])
    cc.set_run_streams(int(os.environ.get("QUEUES", "1")))
    ops.append((op, cc, st))
idx_h = idx.cpu().numpy().view(np.uint64); minus_h = minus.cpu().numpy().view(np.uint64)
bufs = [o[0].hostBuffers() for o in ops]
for bi, bm, br in bufs:
    bi[...] = idx_h; bm[...] = minus_h
for rep in range(3):
    ts, tw = [], []
    nq = 20 * P
    t00 = time.perf_counter()
    for i in range(nq + P):
        o, (bi, bm, br) = ops[i % P][0], bufs[i % P]
        if i >= P:
            t0 = time.perf_counter(); o.waitHost(); tw.append(time.perf_counter() - t0)
        if i < nq:
            t0 = time.perf_counter(); o.runHostAsync(bi, bm, br); ts.append(time.perf_counter() - t0)
    tot = time.perf_counter() - t00
    print("P=%d: %.3f ms per query; submit call median %.3f ms, wait call median %.3f ms" % (P, tot / nq * 1e3, 1e3 * sorted(ts)[len(ts) // 2], 1e3 * sorted(tw)[len(tw) // 2]))
# pieces on one slot
o, (bi, bm, br) = ops[0][0], bufs[0]
def med(f, n=15):
    r = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); r.append(time.perf_counter() - t0)
    return 1e3 * sorted(r)[n // 2]
print("one slot: runHost (sync) %.3f ms" % med(lambda: o.runHost(bi, bm, br)))
d = torch.empty(idx_h.size, dtype=torch.int64, device=dev)
hp = torch.from_numpy(bi.view(np.int64).reshape(-1))
print("H2D 29 MiB from the pinned staging: %.3f ms" % med(lambda: (d.copy_(hp, non_blocking=True), torch.cuda.synchronize())))
for op_, cc_, st_ in reversed(ops):
    cc_.close()
