"""Break-down of the e2e server-online phase (GPU box): python tools/e2e_timing.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from nested_hashing_psi_amd import pie
from nested_hashing_psi_amd.client import BatchedFHEPSIClient

cfg = bench.CONFIGS["C3"]
t, k, e, K, E, b = cfg["t"], cfg["k"], cfg["e"], cfg["K"], cfg["E"], cfg["b"]
cc = pie.PieContext(cfg["N"], cfg["L"], t)
rng = np.random.default_rng(1)
items = np.unique(rng.integers(1, t, cfg["S"] + cfg["C"] + 8192, dtype=np.uint64))
rng.shuffle(items)
server, clientset = items[:cfg["S"]].copy(), items[cfg["S"] - 500:cfg["S"] + 524].copy()
for rep in range(3):
    T = [time.perf_counter()]
    cl = BatchedFHEPSIClient(cc, k, e, K, E, b)
    evk = cl.runSetUpPhase()
    T.append(time.perf_counter())
    cc.load_relin_key(evk)
    T.append(time.perf_counter())
    srv = pie.BatchedFHEHIPPIE(cc, serverSet=server, hashParams=dict(k=k, e=e, K=K, b=b, E=E))
    T.append(time.perf_counter())
    minus_ct, idx_ct = cl.runOfflinePhase(clientset)
    T.append(time.perf_counter())
    srv.setMinusCompareElement(minus_ct)
    T.append(time.perf_counter())
    srv.setIndex(idx_ct)
    T.append(time.perf_counter())
    srv.run()
    T.append(time.perf_counter())
    res = srv.getResultList()
    T.append(time.perf_counter())
    names = ["client setup", "load_relin_key", "server offline", "client offline", "setMinus", "setIndex", "run", "getResultList"]
    print("rep", rep, {n: round((T[i + 1] - T[i]) * 1e3, 2) for i, n in enumerate(names)})
