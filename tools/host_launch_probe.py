"""Is a rank's small share bound by the host's launch path?  (C3, n bin layers, one queue per run(), s query slots)
  enqueue   host time of one piehip_run call (asynchronous: returns when the launches are queued)
  1 thread  ms per run() with one host thread taking the slots round-robin (what bench.py times)
  s threads ms per run() with one host thread per slot (ctypes drops the GIL inside the call)
python tools/host_launch_probe.py [layers] [slots]"""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
import torch
import bench
from nested_hashing_psi_amd import pie

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nslots = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = bench.CONFIGS["C3"]
N, L, t = cfg["N"], cfg["L"], cfg["t"]
device = torch.device("cuda:0")
gen = torch.Generator(device=device); gen.manual_seed(1)
stream = torch.cuda.Stream(device)
cc = pie.PieContext(N, L, t, device=0, stream=stream.cuda_stream)
evk = bench.uniform_limbs(torch, (L, 2), cc.q, N, device, gen)
idx = bench.uniform_limbs(torch, (cfg["K"], cfg["E"], 2), cc.q, N, device, gen)
minus = bench.uniform_limbs(torch, (2,), cc.q, N, device, gen)
cc.load_relin_key(evk.cpu().numpy().view(np.uint64))
op = bench.synthetic_operator(pie, cc, cfg, layers, np.random.default_rng(1), (idx, minus))
cc.set_run_streams(1)
more = bench.make_query_slots(torch, pie, cc, op, (N, L, t, cfg["K"], cfg["E"]), nslots, device, 0, gen, 1)
ops = [op] + [m[1] for m in more]
sync = lambda: torch.cuda.synchronize(device)
for _ in range(50):
    for o in ops: o.run(sync=False)
sync()
# enqueue cost
ts = []
for _ in range(20):
    sync()
    t0 = time.perf_counter()
    for o in ops: o.run(sync=False)
    ts.append((time.perf_counter() - t0) / len(ops))
    sync()
print("enqueue per run(): %.1f us (host, nothing queued ahead)" % (1e6 * sorted(ts)[len(ts) // 2]))
R = 400
def one_thread():
    sync(); t0 = time.perf_counter()
    for i in range(R * len(ops)): ops[i % len(ops)].run(sync=False)
    sync(); return (time.perf_counter() - t0) / (R * len(ops))
def many_threads():
    def work(o):
        for _ in range(R): o.run(sync=False)
    th = [threading.Thread(target=work, args=(o,)) for o in ops]
    sync(); t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    sync(); return (time.perf_counter() - t0) / (R * len(ops))
for name, f in (("1 thread", one_thread), ("%d threads" % len(ops), many_threads)):
    v = sorted(f() for _ in range(7))
    print("%-10s %.1f us per run() (min %.1f max %.1f)" % (name, 1e6 * v[3], 1e6 * v[0], 1e6 * v[-1]))
