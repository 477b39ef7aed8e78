"""Several queries in flight: P handles (own workspace, own copy of the packed database) on P streams, run() round-robin.
   python tools/pipeline_probe.py [E b_local k e]      (C3 ring; default: the C3 shape)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from nested_hashing_psi_amd import pie

cfg = dict(bench.CONFIGS["C3"])
if len(sys.argv) > 4:
    cfg.update(E=int(sys.argv[1]), b=int(sys.argv[2]), k=int(sys.argv[3]), e=int(sys.argv[4]))
N, L, t, K, E, b = cfg["N"], cfg["L"], cfg["t"], cfg["K"], cfg["E"], cfg["b"]
dev = torch.device("cuda", 0)
ops, ccs = [], []
PS = [2, 3, 4, 6]
for i in range(max(PS)):
    stream = torch.cuda.Stream(dev)
    cc = pie.PieContext(N, L, t, stream=stream.cuda_stream)
    gen = torch.Generator(device=dev); gen.manual_seed(1 + i)
    evk = bench.uniform_limbs(torch, (L, 2), cc.q, N, dev, gen)
    idx = bench.uniform_limbs(torch, (K, E, 2), cc.q, N, dev, gen)
    minus = bench.uniform_limbs(torch, (2,), cc.q, N, dev, gen)
    torch.cuda.synchronize()
    if i == 0:
        cc.load_relin_key(evk.cpu().numpy().view(np.uint64))
    if i == 0:
        op = bench.synthetic_operator(pie, cc, cfg, b, np.random.default_rng(3), (idx, minus))
    else:
        op = pie.BatchedFHEHIPPIE(cc, attachTo=ops[0][0])
        op.setIndexDevice(idx.data_ptr())
        op.setMinusCompareElementDevice(minus.data_ptr())
    ops.append((op, idx, minus, stream))
    ccs.append(cc)
sync = lambda: torch.cuda.synchronize(dev)
GRAPH = os.environ.get("GRAPH", "0") == "1"
for q in (1, 2):
    for cc in ccs:
        cc.set_run_streams(q)
        cc.set_graph(GRAPH)
    print("queues per run %d: one query at a time %.1f us per run()" % (q, 1e3 * bench.time_runs(ops[0][0], 200, 20, sync)))
    for P in PS:
        for _ in range(10):
            for o in ops[:P]:
                o[0].run(sync=False)
        sync()
        t0 = time.perf_counter()
        for _ in range(100):
            for o in ops[:P]:
                o[0].run(sync=False)
        sync()
        print("   %d in flight: %.1f us per run()" % (P, (time.perf_counter() - t0) / (100 * P) * 1e6))
for cc in reversed(ccs):
    cc.close()
