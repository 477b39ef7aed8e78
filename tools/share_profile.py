"""Per-kernel time of one rank's share of a sharded server (run on the GPU box):
   python tools/share_profile.py E b_local [k e]      (C3 ring; K = 2)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
if "PIEHIP_AB_LIB" in os.environ:   # A/B of two builds inside one gpurun call (boxes differ by several percent)
    import nested_hashing_psi_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["PIEHIP_AB_LIB"])
from nested_hashing_psi_amd import pie

E, bl = int(sys.argv[1]), int(sys.argv[2])
k, e = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2, 4949)
cfg = dict(bench.CONFIGS["C3"], E=E, b=bl, k=k, e=e)
N, L, t, K = cfg["N"], cfg["L"], cfg["t"], cfg["K"]
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
cc = pie.PieContext(N, L, t, stream=stream.cuda_stream)
gen = torch.Generator(device=dev); gen.manual_seed(1)
evk = bench.uniform_limbs(torch, (L, 2), cc.q, N, dev, gen)
idx = bench.uniform_limbs(torch, (K, E, 2), cc.q, N, dev, gen)
minus = bench.uniform_limbs(torch, (2,), cc.q, N, dev, gen)
torch.cuda.synchronize()
cc.load_relin_key(evk.cpu().numpy().view(np.uint64))
op = bench.synthetic_operator(pie, cc, cfg, bl, np.random.default_rng(3), (idx, minus))
sync = lambda: torch.cuda.synchronize(dev)
for streams in (2, 1):
    cc.set_run_streams(streams)
    print("queues=%s: %.1f us per run()" % (streams, 1e3 * bench.time_runs(op, 100, 10, sync)))
cc.set_run_streams(1)
cc.set_profiling(True)
agg = {}
for _ in range(20):
    op.run(sync=True)
    for name, rec in cc.profile().items():
        a = agg.setdefault(name, [0, 0.0])
        a[0] += rec["launches"]; a[1] += rec["ms"]
cc.set_profiling(False)
print("serial, HIP events: " + ", ".join("%s %.1f us (%d)" % (n, 1e3 * v[1] / 20, v[0] // 20) for n, v in agg.items()),
      "| sum %.1f us" % (1e3 * sum(v[1] for v in agg.values()) / 20))
cc.close()
