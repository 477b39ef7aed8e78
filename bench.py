#!/usr/bin/env python3
"""bench.py -- server-side batched-FHE PIE throughput on MI355X.

One "step" = one BatchedFHEHIPPIE::run() (reference BatchedFHEHIPPIE.cpp:88-129; what the server
times at src/Server/FHE/BatchedFHEPSIServer.cpp:98-106) over one batch of synthetic queries (three
by default: `queries_per_step`; `one_query_at_a_time` in the same line is the reference's one-query
run()), inputs already resident in HBM.  Metric: result ciphertexts per second, whole job.

  python bench.py --gpus 1 --steps K --warmup W            (N=1: config C3 of BASELINE.json)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (config C4)

Multi-GPU (BASELINE config C4): bin layers are independent, so the b = 14 bin layers of the C3 database are
split over the ranks (strong scaling, the default for N > 1; speed-up is capped at b / ceil(b / N)); the only collective
is the RCCL gather of the result ciphertexts to rank 0.  --scaling weak gives every rank its own b layers instead.
At N = 1 the line also carries `projected_strong_scaling`: the time of one rank's share ceil(b / G) of the bin layers,
measured on this GPU, for G = 2, 4, 8 and for the balanced parameter rows SURVEY 8e names.

Synthetic data: the arithmetic is data-independent, so index/minus ciphertexts and the
relinearisation key are uniform residues (what real ones are indistinguishable from) and the
database is uniform slot values packed on the device.  Correctness is the tests' job, not this
script's: the timed region's exact shape (C3, three queries per run() on one handle, one and two
queues) is pinned to the oracle by tests/test_gpu_fullsize.py::test_c3_headline_batch_of_three.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: N, L, t, k, e, K, E, b, |S|, |C|   
    "C1": dict(N=4096, L=2, t=65537, k=3, e=110, K=2, E=7, b=7, S=1 << 12, C=1 << 8),
    "C2": dict(N=8192, L=3, t=4296540161, k=3, e=443, K=2, E=12, b=12, S=1 << 16, C=1 << 10),
    "C3": dict(N=16384, L=4, t=4296540161, k=2, e=4949, K=2, E=14, b=14, S=1 << 20, C=1 << 10),
    "C5": dict(N=32768, L=6, t=4296540161, k=2, e=13004, K=3, E=30, b=30, S=1 << 24, C=1 << 12),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8 TB/s


def uniform_limbs(torch, shape_prefix, moduli, N, device, gen):
    """uniform residues [*shape_prefix][len(moduli)][N] as int64 (bit pattern of the uint64 limbs)"""
    out = torch.empty(tuple(shape_prefix) + (len(moduli), N), dtype=torch.int64, device=device)
    for i, m in enumerate(moduli):
        out[..., i, :] = torch.randint(0, int(m), tuple(shape_prefix) + (N,), dtype=torch.int64, device=device, generator=gen)
    return out


def cpu_baseline(cfg, seconds_target=12.0, max_threads=16):
    """The oracle (a CPU restatement of the reference path -- OpenFHE itself is not available)
    timed on this host, one core, on a bounded sample of the same workload."""
    from oracle import binding as ob
    build_flags = ob.prefer_native()   # -O3 -march=native on the host that is timed, else the shipped build
    N, L, t, K, E, b = cfg["N"], cfg["L"], cfg["t"], cfg["K"], cfg["E"], cfg["b"]
    o = ob.Oracle(N, L, t)
    rng = np.random.default_rng(1)

    def rl(shape):
        out = np.zeros(shape + (L, N), dtype=np.uint64)
        for i in range(L):
            out[..., i, :] = rng.integers(0, int(o.q[i]), shape + (N,), dtype=np.uint64)
        return out

    idx, minus, evk = rl((K, E, 2)), rl((2,)), rl((L, 2))
    # time one bin layer first, then as many bin layers as fit the target (bin layers are independent
    # and cost the same, BatchedFHEHIPPIE.cpp:91)
    db1, m1 = rl((K, 1, E)), rl((1,))
    t0 = time.perf_counter()
    o.pie_run(idx, minus, db1, m1, evk)
    per_bin = time.perf_counter() - t0
    nb = int(max(1, min(64 * b, seconds_target / per_bin)))
    done, t0 = 0, time.perf_counter()
    while done < nb:
        o.pie_run(idx, minus, db1, m1, evk)
        done += 1
    dt = time.perf_counter() - t0
    out = {"value": done / dt, "unit": "ciphertexts/s", "cores": 1, "kind": "port", "build": build_flags, "host": host_cpu_info(),
           "sample": "%d bin layers of the %s workload (K=%d, E=%d; 1 ct x ct + %d ct x pt MACs each), %.1f s; "
                     "CPU restatement in C (oracle/), OpenFHE not available" % (done, cfg["name"], K, E, K * E, dt)}
    # (b) of SURVEY 8d: the same work on every core this process may use, one bin layer per task (bin layers are
    # independent; the C call releases the GIL)
    import concurrent.futures
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, max_threads)  # the GPU box grants a one-GPU job 16 CPUs' worth of time whatever the affinity mask says (see "host")
    if cores > 1:
        ntask = int(max(cores, min(64 * b, 0.5 * seconds_target / per_bin * cores)))
        t0 = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(max_workers=cores) as pool:
            list(pool.map(lambda _: o.pie_run(idx, minus, db1, m1, evk), range(ntask)))
        dtp = time.perf_counter() - t0
        out["all_cores"] = {"value": ntask / dtp, "unit": "ciphertexts/s", "cores": cores,
                            "sample": "%d bin layers on %d threads, %.1f s" % (ntask, cores, dtp)}
    # the same end-to-end PSI as e2e_psi(), every phase on one host core (encode of the K*b*E+b plaintexts
    # extrapolated from a sample of 16 to keep this leg bounded)
    k, e, nS, nC = cfg["k"], cfg["e"], cfg["S"], cfg["C"]
    items = np.unique(rng.integers(1, t, nS + nC + 8192, dtype=np.uint64))
    rng.shuffle(items)
    server, ninter = items[:nS].copy(), nC // 2 + 1
    clientset = np.concatenate([server[:ninter], items[nS:nS + nC - ninter]])
    t0 = time.perf_counter()
    sk = o.keygen(11)
    evk2 = o.relin_keygen(sk, 12)
    t1 = time.perf_counter()
    tab = ob.Tabulation(987654321, k + K)
    tbl = ob.hct_build(tab, server, k, e, K, b, E, evict_seed=1)
    ob.hct_shuffle_bins(tbl, 2)
    slots = ob.pack_db(tbl)
    ta = time.perf_counter()
    for i in range(16):
        o.encode_eval(slots[0, 0, i % E])
    enc_each = (time.perf_counter() - ta) / 16
    t2 = time.perf_counter() + enc_each * (K * b * E + b - 16)
    ctab = ob.client_build(tab, clientset, k, e)
    index, minus_v = ob.client_vectors(tab, ctab, K, E)
    tb_ = time.perf_counter()
    for h in range(K):
        for j in range(E):
            o.encrypt_slots(sk, index[h, j], 100 + h * E + j)
    o.encrypt_slots(sk, minus_v, 99)
    client_off = time.perf_counter() - tb_
    run_s = b * per_bin
    ct = rl((2,))
    tc = time.perf_counter()
    o.decrypt_slots(sk, ct, k * e)
    dec_s = (time.perf_counter() - tc) * b
    out["e2e_psi_cpu_s"] = {"setup_s": t1 - t0, "server_offline_s": (t2 - t1), "client_offline_s": client_off,
                            "server_online_s": run_s, "client_online_s": dec_s,
                            "total_s": (t1 - t0) + (t2 - t1) + client_off + run_s + dec_s}
    return out


def host_cpu_info():
    """what the CPU legs could use on this host: logical CPUs, this process's affinity mask, the cgroup CPU quota"""
    info = {"host_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cgroup_quota_cpus": None, "model": None}
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:      # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()
            info["cgroup_quota_cpus"] = None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        try:
            q_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            info["cgroup_quota_cpus"] = None if q_ < 0 else q_ / p_
        except (OSError, ValueError):
            pass
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                info["model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return info


def e2e_psi(cfg, pie, cc, device_sync):
    """End-to-end PSI wall-clock (the second half of BASELINE.json's metric), in process, no TCP: what the
    reference client times as Setup + Offline + Online (src/Client/PSIClient.hpp:87-116).  Server offline phase
    (nested hashing + packing) and the client's BFV operations run on the device, ciphertexts cross PCIe as
    host arrays exactly once each way.  Sets: distinct uniform non-zero items < t, |I| = |C|/2 + 1
    (Parameters1.txt column 3), hash seed 987654321 (CLI.cpp:67)."""
    from nested_hashing_psi_amd.client import BatchedFHEPSIClient
    t, k, e, K, E, b = cfg["t"], cfg["k"], cfg["e"], cfg["K"], cfg["E"], cfg["b"]
    nS, nC = cfg["S"], cfg["C"]
    rng = np.random.default_rng(123456789)
    items = np.unique(rng.integers(1, t, nS + nC + 8192, dtype=np.uint64))
    rng.shuffle(items)
    server = items[:nS].copy()
    ninter = nC // 2 + 1
    clientset = np.concatenate([server[:ninter], items[nS:nS + nC - ninter]])
    rng.shuffle(clientset)
    out = {}
    t0 = time.perf_counter()
    cl = BatchedFHEPSIClient(cc, k, e, K, E, b)
    evk = cl.runSetUpPhase()
    cc.load_relin_key(evk)
    cc.reserve(nS, k, e, K, b, E)     # sizes are known at set-up (HashTableParameter): the offline phase allocates nothing
    device_sync()
    t1 = time.perf_counter()
    srv = pie.BatchedFHEHIPPIE(cc, serverSet=server, hashParams=dict(k=k, e=e, K=K, b=b, E=E))
    device_sync()
    t2 = time.perf_counter()
    minus_ct, idx_ct = cl.runOfflinePhase(clientset)
    device_sync()
    t3 = time.perf_counter()
    srv.setMinusCompareElement(minus_ct)
    srv.setIndex(idx_ct)
    srv.run()
    res = srv.getResultList()
    t4 = time.perf_counter()
    found = cl.extractIntersection(res)
    t5 = time.perf_counter()
    ok = sorted(int(v) for v in found) == sorted(int(v) for v in server[:ninter])
    out.update(setup_s=t1 - t0, server_offline_s=t2 - t1, client_offline_s=t3 - t2, server_online_s=t4 - t3,
               client_online_s=t5 - t4, total_s=t5 - t0, intersection_size=int(len(found)), intersection_correct=bool(ok),
               note="in-process, no TCP; includes PCIe transfers of keys, %d input and %d result ciphertexts; server offline = "
                    "device hashing + packing of |S|=%d items" % (K * E + 1, b, nS))
    return out


def alg_bytes_run(cfg):
    """algorithmic bytes of one run() by the unfused limb-pass formulas of SURVEY.md 8d (reference schedule)"""
    N, L, K, E, b = cfg["N"], cfg["L"], cfg["K"], cfg["E"], cfg["b"]
    W, M = 8 * N, 2 * L + 1
    stage_a = (b * K * E * L + K * E * 2 * L + 2 * L + b * K * 2 * L) * W
    per_mul = (2 * (L * L + 22 * L + 7) + (4 * L + 4 * M) + (4 * M + 3 * M) + (3 * M + 3 * L) + (3 * L * L + 2 * L)) * W
    stage_c = b * 5 * L * W
    return stage_a + b * (K - 1) * per_mul + stage_c


def pmc_traffic(config):
    """HBM bytes per NTT launch from the committed rocprofv3 PMC passes of this same command
    (profiles/latest_pmc.json, written by tools/pmc_summary.py; FETCH_SIZE x2 + WRITE_SIZE, gfx950
    corrections of MI355X_MICROARCH.md).  None if no profile of this config is committed."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
        if d.get("config") == config:
            return d["ntt"]["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


# parameter rows for the projected strong scaling (SURVEY 8e "Limits"): the headline row (cap 7x at 8 GPUs), a row with a
# bin count that 8 divides, and the reference's own k = 3 row of this set size (Parameters1.txt:59)
SCALING_ROWS = {
    "C3 b=14 E=14": dict(k=2, e=4949, K=2, E=14, b=14),
    "b=16 E=13": dict(k=2, e=4949, K=2, E=13, b=16),
    "k=3 e=443 b=40 E=40 (Parameters1.txt:59)": dict(k=3, e=443, K=2, E=40, b=40),
}


def synthetic_operator(pie, cc, cfg, b_local, rng, device_inputs):
    """an operator over b_local bin layers of synthetic database content (uniform slot values, packed on the device),
    inputs already resident"""
    t, K, E, B = cfg["t"], cfg["K"], cfg["E"], cfg["k"] * cfg["e"]
    slots = rng.integers(0, t, (K, b_local, E, B), dtype=np.int64)
    slots[slots > t // 2] -= t
    mask_slots = rng.integers(1, t, (b_local, B), dtype=np.int64)
    mask_slots[mask_slots > t // 2] -= t
    op = pie.BatchedFHEHIPPIE(cc, slots=slots, mask_slots=mask_slots)  # packed + NTT'd on the device
    idx, minus = device_inputs
    op.setIndexDevice(idx.data_ptr())
    op.setMinusCompareElementDevice(minus.data_ptr())
    return op


WARM_SECONDS = 0.3   # warm-up of the timed region by TIME: clocks, caches, lazily created queues (whatever --warmup says)
REPEATS = 15         # timed blocks of exactly --steps steps each; the median block is reported


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


def timed_blocks(step, finish, steps, warmup, repeats=REPEATS, warm_seconds=WARM_SECONDS):
    """ms per step: at least `warmup` steps and `warm_seconds` of warm-up, then `repeats` blocks of exactly `steps` steps, each
    bracketed by finish() (drain + device synchronisation); returns (median, min, max) over the blocks"""
    t_end = time.perf_counter() + warm_seconds
    done = 0
    while done < warmup or time.perf_counter() < t_end:
        for _ in range(max(1, warmup)):
            step()
        done += max(1, warmup)
        finish()
    ts = []
    for _ in range(repeats):
        finish()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        finish()
        ts.append((time.perf_counter() - t0) / steps * 1e3)
    return median(ts), min(ts), max(ts)


def time_runs(op, steps, warmup, sync, repeats=7, warm_seconds=0.1):
    return timed_blocks(lambda: op.run(sync=False), sync, steps, warmup, repeats, warm_seconds)[0]


def make_query_slots(torch, pie, cc, op, shape, depth, device, local_rank, gen, run_streams):
    """depth - 1 further query slots on op's database: (context, operator, stream, idx, minus) each with its own synthetic
    query resident in HBM (piehip_attach_database: the key and the packed database are shared by reference)"""
    N, L, t, K, E = shape
    slots = []
    for _ in range(1, depth):
        st = torch.cuda.Stream(device)
        c = pie.PieContext(N, L, t, device=local_rank, stream=st.cuda_stream)
        idx = uniform_limbs(torch, (K, E, 2), cc.q, N, device, gen)
        minus = uniform_limbs(torch, (2,), cc.q, N, device, gen)
        torch.cuda.synchronize(device)
        o = pie.BatchedFHEHIPPIE(c, attachTo=op)
        o.setIndexDevice(idx.data_ptr())
        o.setMinusCompareElementDevice(minus.data_ptr())
        c.set_run_streams(run_streams)
        slots.append((c, o, st, idx, minus))
    return slots


def time_slots(ops, steps, warmup, sync, repeats=7, warm_seconds=0.1):
    """ms per run() with the operators taking the steps round-robin (len(ops) queries in flight)"""
    n = [0]

    def step():
        ops[n[0] % len(ops)].run(sync=False)
        n[0] += 1
    return timed_blocks(step, sync, steps, warmup, repeats, warm_seconds)[0]


PROJECTION_IN_FLIGHT = 3
# The default timed region: three queries per run() on one handle (a query batch: piehip_set_query_batch).  Until round 3 the
# default was three query slots in flight with one query each; the batch serves the same three clients 5-6 % faster (stage A
# streams the database once for the three, the transform launches are three times the size).  Both are measured; see main().
DEFAULT_BATCH = 3


def projected_strong_scaling(torch, pie, cfg, device, local_rank, gen, steps, warmup):
    """SURVEY 8e without an 8-GPU box: a rank of G owns ceil(b / G) bin layers (the slowest rank sets the time), so
    speed-up(G) = t(b) / t(ceil(b / G)), every t measured here on one GPU with the same kernels, queues and launch path.
    Leaves out the result gather (b * 2LW bytes in total, each slice over its own xGMI link) and the broadcast of the query."""
    out = {"note": "t(n) = ms per run() over n bin layers on this one GPU; speedup(G) = t(b) / t(ceil(b/G)); excludes the RCCL "
                   "gather of b*2LW result bytes and the query broadcast; cap = b / ceil(b/G).  Two tables: one query at a time "
                   "(latency: the better of eager launches and a replayed hipGraph), and %d queries in flight on query slots "
                   "(throughput: a rank's small share leaves most of the chip idle, further queries fill it); and a third, "
                   "ms per QUERY with %d queries per run() (query batches) on one handle or on all %d slots, whichever is faster"
                   % (PROJECTION_IN_FLIGHT, DEFAULT_BATCH, PROJECTION_IN_FLIGHT),
           "queries_in_flight": PROJECTION_IN_FLIGHT, "queries_per_run_batched": DEFAULT_BATCH, "rows": {}}
    N, L, t = cfg["N"], cfg["L"], cfg["t"]
    sync = lambda: torch.cuda.synchronize(device)

    def upload_ms(words):   # the query crossing PCIe from page-locked host memory: what every rank of a sharded server pays per query
        h = torch.zeros(words, dtype=torch.int64).pin_memory()
        d = torch.empty(words, dtype=torch.int64, device=device)
        ts = []
        for _ in range(12):
            sync()
            t0 = time.perf_counter()
            d.copy_(h, non_blocking=True)
            sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        return median(ts[2:])
    for name, row in SCALING_ROWS.items():
        c2 = dict(cfg, **row)
        b = c2["b"]
        up = upload_ms((c2["K"] * c2["E"] * 2 + 2) * L * N)
        shares = sorted({-(-b // G) for G in (1, 2, 4, 8)}, reverse=True)
        tms, tfl, graph_better, tbq, tbq_slots = {}, {}, {}, {}, {}
        for n in shares:
            stream = torch.cuda.Stream(device)
            cc = pie.PieContext(N, L, t, device=local_rank, stream=stream.cuda_stream)
            evk = uniform_limbs(torch, (L, 2), cc.q, N, device, gen)
            idx = uniform_limbs(torch, (c2["K"], c2["E"], 2), cc.q, N, device, gen)
            minus = uniform_limbs(torch, (2,), cc.q, N, device, gen)
            cc.load_relin_key(evk.cpu().numpy().view(np.uint64))
            op = synthetic_operator(pie, cc, c2, n, np.random.default_rng(n), (idx, minus))
            te = time_runs(op, steps, warmup, sync)
            cc.set_graph(True)     # a rank evaluating few bin layers is bound by the launch path: take the better of the two
            tg = time_runs(op, steps, warmup, sync)
            cc.set_graph(False)
            tms[n] = min(te, tg)
            graph_better[n] = tg < te
            cc.set_run_streams(1)
            more = make_query_slots(torch, pie, cc, op, (N, L, t, c2["K"], c2["E"]), PROJECTION_IN_FLIGHT, device, local_rank, gen, 1)
            tfl[n] = time_slots([op] + [m[1] for m in more], max(60, steps), max(12, warmup), sync)
            # query batches: DEFAULT_BATCH queries per run() on every slot; one handle alone (two queues when it has eight or more
            # layers) or all slots in flight, per query
            keep = []
            for o_ in [op] + [m[1] for m in more]:
                o_.setQueryBatch(DEFAULT_BATCH)
                for q_ in range(1, DEFAULT_BATCH):
                    iq = uniform_limbs(torch, (c2["K"], c2["E"], 2), cc.q, N, device, gen)
                    mq = uniform_limbs(torch, (2,), cc.q, N, device, gen)
                    keep.append((iq, mq))
                    o_.setIndexDevice(iq.data_ptr(), query=q_)
                    o_.setMinusCompareElementDevice(mq.data_ptr(), query=q_)
            sync()
            t_all = time_slots([op] + [m[1] for m in more], max(40, steps), max(9, warmup), sync) / DEFAULT_BATCH
            cc.set_run_streams(0)
            t_one = time_runs(op, max(40, steps), max(9, warmup), sync) / DEFAULT_BATCH
            tbq[n], tbq_slots[n] = min(t_all, t_one), (PROJECTION_IN_FLIGHT if t_all <= t_one else 1)
            del keep
            for m in reversed(more):
                m[0].close()
            cc.close()
            del op, cc, idx, minus, evk, more
        out["rows"][name] = {"b": b, "E": c2["E"], "ms_per_run": {str(n): tms[n] for n in shares},
                             "hipgraph_faster": {str(n): bool(v) for n, v in graph_better.items()},
                             "speedup": {str(G): tms[b] / tms[-(-b // G)] for G in (2, 4, 8)},
                             "ms_per_run_in_flight": {str(n): tfl[n] for n in shares},
                             "speedup_in_flight": {str(G): tfl[b] / tfl[-(-b // G)] for G in (2, 4, 8)},
                             "ms_per_query_batched": {str(n): tbq[n] for n in shares},
                             "batched_runs_in_flight": {str(n): tbq_slots[n] for n in shares},
                             "speedup_batched": {str(G): tbq[b] / tbq[-(-b // G)] for G in (2, 4, 8)},
                             "cap": {str(G): b / -(-b // G) for G in (2, 4, 8)},
                             # "with query upload": every rank also receives the (K E + 1) query ciphertexts over its own PCIe link
                             # (measured on this GPU, page-locked source).  serial = upload + share, nothing overlapped (one query
                             # at a time); overlapped = max(upload, share): uploads of the next queries hidden behind the evaluation
                             "query_upload_ms": up,
                             "speedup_with_upload_serial": {str(G): (tms[b] + up) / (tms[-(-b // G)] + up) for G in (2, 4, 8)},
                             "speedup_in_flight_with_upload_overlapped": {str(G): max(tfl[b], up) / max(tfl[-(-b // G)], up) for G in (2, 4, 8)}}
    return out


def reference_timer(torch, op, idx, minus, b, iters, device, more_ops=(), run_streams=0, batch=1, keep_gc=False):
    """The reference's own timer placement (BatchedFHEPSIServer.cpp:98-106): setMinusCompareElement + setIndex + run, with the
    query in HOST memory as the deserialised ciphertexts are -- so these figures include the PCIe upload that `value` leaves
    out.  Three ways across the boundary, medians of `iters` queries each:
      separate calls   piehip_set_minus + piehip_set_index + piehip_run + sync (pageable numpy arrays), then getResultList
      run_host         the same work as one pipelined call (piehip_run_host: row-wise upload under stage A, per-group download),
                       from pageable arrays and from the library's page-locked staging arrays (what a deserialiser would fill)"""
    idx_h = idx.cpu().numpy().view(np.uint64)
    minus_h = minus.cpu().numpy().view(np.uint64)
    # CPython's cyclic garbage collector is switched off inside these legs (a full collection of a process that has torch and numpy
    # loaded takes ~40 ms, and it fires -- deterministically, by allocation count -- inside one piehip_run_host_async call of the
    # first stream leg: that one pause was r04's "slow leg"; --keep-gc leaves the collector alone)
    import gc
    gc_was_on = gc.isenabled() and not keep_gc
    if gc_was_on:
        gc.collect()
        gc.disable()

    def med(f, n):
        ts = []
        for _ in range(n + 2):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[2:])
        return ts[len(ts) // 2]

    def separate(with_results):
        op.setMinusCompareElement(minus_h)
        op.setIndex(idx_h)
        op.run(sync=True)
        if with_results:
            op.getResultList()

    sep = med(lambda: separate(False), iters)
    sep_res = med(lambda: separate(True), iters)
    res_pageable = np.zeros((b,) + minus_h.shape, dtype=np.uint64)
    host_pageable = med(lambda: op.runHost(idx_h, minus_h, res_pageable), iters)
    pi, pm, pr = op.hostBuffers()
    pi[...] = idx_h
    pm[...] = minus_h
    host_pinned = med(lambda: op.runHost(pi, pm, pr), iters)
    # Every stream leg also reports how long its staging sequences waited (on the host) for their turn on the PCIe link
    # (piehip_upload_turn_wait): a leg that runs far from the link bound names its wait -- the staging thread of slot B polls a
    # page-locked word until slot A's query has left host memory, and that poll is the one place of this path where the host blocks.

    def turn_waits(ops_):
        return [o_.cc.upload_turn_wait() for o_ in ops_]

    def turn_wait_delta(ops_, before, nqueries):
        after = turn_waits(ops_)
        tot = sum(a_[1] - b_[1] for a_, b_ in zip(after, before))
        cnt = sum(a_[2] - b_[2] for a_, b_ in zip(after, before))
        return {"ms_per_query": tot / nqueries, "waits": cnt, "ms_per_wait": (tot / cnt if cnt else 0.0)}

    # Every pass of a stream leg is instrumented: three events per run() stamped by the DEVICE in stream order
    # (piehip_set_host_path_timing) give the time of its uploads (first staged piece -> last upload done) and of the rest (evaluation
    # + the result list's way down); the host's time to issue each run(), its wait for that run()'s results and its wait for its
    # turn on the link are recorded beside them, medians and maxima -- one stall names its side.  A leg is one warm-up
    # pass and TWO timed passes; the leg's figure is the faster timed pass, and all three passes are in the JSON line
    # (`stream_passes`) -- a pass that ran far from the link bound then shows which side was slow.  (r04: one leg of four sat at
    # 1.5-2.7 ms per query in some processes.  r05's passes show it is TRANSIENT -- the pass behind the slow one runs at the bound --
    # and that the host is not waiting for its turn during it: DESIGN.md section 4.)
    stream_passes = {"run_host_async_stream": {}, "staged_batch_stream": {}}
    collect = [None]
    issue_ms = {}

    def run_leg(ops_, one_pass, nqueries, queries_per_run):
        for o_ in ops_:
            o_.cc.set_host_path_timing(True)
        passes_ = []
        for _ in range(3):
            torch.cuda.synchronize(device)
            collect[0] = []
            w0_ = turn_waits(ops_)
            t0_ = time.perf_counter()
            one_pass()
            wall = time.perf_counter() - t0_
            got, collect[0] = collect[0], None
            med_ = lambda k_: sorted(g_[k_] for g_ in got)[len(got) // 2]
            passes_.append({"ms_per_query": wall * 1e3 / nqueries, "upload_ms_per_run": med_(0), "evaluate_and_download_ms_per_run": med_(1),
                            "upload_ms_per_run_max": max(g_[0] for g_ in got), "evaluate_and_download_ms_per_run_max": max(g_[1] for g_ in got),
                            "host_wait_for_results_ms_per_run": med_(2), "host_wait_for_results_ms_per_run_max": max(g_[2] for g_ in got),
                            "host_issue_ms_per_run": med_(3), "host_issue_ms_per_run_max": max(g_[3] for g_ in got),
                            "upload_turn_wait": turn_wait_delta(ops_, w0_, nqueries),
                            "queries_per_run": queries_per_run})
        for o_ in ops_:
            o_.cc.set_host_path_timing(False)
        return min(p_["ms_per_query"] for p_ in passes_[1:]) * 1e-3, passes_

    # a stream of queries from host memory over the query slots (piehip_run_host_async / _wait, page-locked staging per slot):
    # slot B's 29 MiB cross PCIe while slot A evaluates, so a query costs its upload, not upload + run + download
    pipelined = {}
    if more_ops:
        for nslots in (2, 3):
            if nslots > 1 + len(more_ops):
                break
            allops = [op] + [m[0] for m in more_ops[:nslots - 1]]   # query slots, one queue per run() as in the timed region
            bufs = [o.hostBuffers() for o in allops]
            for (bi, bm, br) in bufs:
                bi[...] = idx_h
                bm[...] = minus_h
            nq = max(8, iters) * len(allops)

            def stream_queries():
                for i in range(nq + len(allops)):
                    o, (bi, bm, br) = allops[i % len(allops)], bufs[i % len(allops)]
                    if i >= len(allops):
                        t_w = time.perf_counter()
                        o.waitHost()           # results of this slot's previous query are in host memory
                        if collect[0] is not None:
                            collect[0].append(o.cc.host_path_times() + ((time.perf_counter() - t_w) * 1e3, issue_ms.pop(id(o), 0.0)))
                    if i < nq:
                        t_i = time.perf_counter()
                        o.runHostAsync(bi, bm, br)
                        issue_ms[id(o)] = (time.perf_counter() - t_i) * 1e3

            pipelined[nslots], stream_passes["run_host_async_stream"][str(nslots)] = run_leg(allops, stream_queries, nq, 1)
    # The same stream with BATCHES of queries (the default timed region's mode, reached through the host-memory boundary the
    # reference's server uses): every slot takes `batch` queries per run(), each query staged piece by piece from its own
    # page-locked arrays (piehip_stage_*_q: minus elements first, then the index matrices row by row across the batch), the
    # result lists [b][batch] come back per queue group.  While one slot evaluates and downloads, the next slot's queries cross
    # PCIe.
    batched = {}
    if more_ops and batch > 1:
        for nslots in (2, 3):
            if nslots > 1 + len(more_ops):
                break
            allops = [op] + [m[0] for m in more_ops[:nslots - 1]]
            for o in allops:
                o.setQueryBatch(batch)
            bufs = []
            for o in allops:
                qb = [o.hostBuffers(query=q_) for q_ in range(batch)]
                for bi, bm, _ in qb:
                    bi[...] = idx_h
                    bm[...] = minus_h
                bufs.append(qb)
            K_ = idx_h.shape[0]
            nbatches = max(6, iters // 2) * nslots

            def stream_batches():
                for i in range(nbatches + nslots):
                    o, qb = allops[i % nslots], bufs[i % nslots]
                    if i >= nslots:
                        t_w = time.perf_counter()
                        o.waitHost()
                        if collect[0] is not None:
                            collect[0].append(o.cc.host_path_times() + ((time.perf_counter() - t_w) * 1e3, issue_ms.pop(id(o), 0.0)))
                    if i < nbatches:
                        t_i = time.perf_counter()
                        for q_ in range(batch):
                            o.stageMinus(qb[q_][1], query=q_)
                        for h_ in range(K_):
                            for q_ in range(batch):
                                o.stageIndexRow(h_, qb[q_][0][h_], query=q_)
                        o.runStaged(qb[0][2])
                        issue_ms[id(o)] = (time.perf_counter() - t_i) * 1e3

            batched[nslots], stream_passes["staged_batch_stream"][str(nslots)] = run_leg(allops, stream_batches, nbatches * batch, batch)
            for o in allops:
                o.setQueryBatch(1)
    # leave the operators as the timed region expects them: inputs resident
    op.setIndexDevice(idx.data_ptr())
    op.setMinusCompareElementDevice(minus.data_ptr())
    for o_, i_, m_ in more_ops:
        o_.setIndexDevice(i_.data_ptr())
        o_.setMinusCompareElementDevice(m_.data_ptr())
    if gc_was_on:
        gc.enable()
    mib = (idx_h.nbytes + minus_h.nbytes) / 2**20
    out = {}
    if pipelined:
        best1 = min(pipelined, key=pipelined.get)
        out = {"run_host_async_stream_ms_per_query": {str(n_): v_ * 1e3 for n_, v_ in pipelined.items()}, "run_host_async_slots": best1,
               "value_run_host_async_stream": b / pipelined[best1]}
    if batched:
        best = min(batched, key=batched.get)
        out.update({"staged_batch_stream_ms_per_query": {str(n_): v_ * 1e3 for n_, v_ in batched.items()}, "staged_batch_queries_per_run": batch,
                    "staged_batch_slots": best, "value_staged_batch_stream": b / batched[best]})
    return {**out, "unit": "ms", "iters": iters, "stream_passes": stream_passes, "python_gc_during_legs": bool(keep_gc),
            "stream_figure": "the faster of two timed passes per leg (after a warm-up pass); every pass is listed in stream_passes",
            "separate_calls_ms": sep * 1e3, "separate_calls_with_results_ms": sep_res * 1e3,
            "run_host_pageable_with_results_ms": host_pageable * 1e3, "run_host_pinned_with_results_ms": host_pinned * 1e3,
            "value_separate_calls": b / sep, "value_run_host_pinned_with_results": b / host_pinned,
            "what": "median host wall per query of %d ciphertexts (%.1f MiB) in host memory; 'with results' ends when the %d result "
                    "ciphertexts (%.1f MiB) are back in host memory" % (idx_h.shape[0] * idx_h.shape[1] + 1, mib, b, b * minus_h.nbytes / 2**20)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1: strong (default; BASELINE config C4: the b bin layers split over the ranks) or weak (b layers per rank)")
    ap.add_argument("--bins-per-rank", type=int, default=0,
                    help="one GPU: evaluate only n bin layers per run() -- the share of one rank of ceil(b/n) GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the multi-core leg of the CPU baseline")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end PSI wall-clock leg")
    ap.add_argument("--no-projection", action="store_true", help="skip the projected strong-scaling legs")
    ap.add_argument("--keep-gc", action="store_true", help="leave CPython's cyclic garbage collector on during the host-memory legs")
    ap.add_argument("--transform-slots", type=int, default=None,
                    help="cap on the persistent transform grids (workgroups); default: every slot on one GPU, 32 fewer (16 CUs left to RCCL) for N > 1")
    ap.add_argument("--no-ref-timer", action="store_true", help="skip the host-inputs (reference timer placement) leg")
    ap.add_argument("--timed-only", action="store_true",
                    help="only the timed region and the per-kernel passes: no further legs (for profiler runs, whose per-kernel "
                         "averages should see one launch shape per kernel)")
    ap.add_argument("--graph", action="store_true", help="run() as one captured hipGraph (piehip_set_graph)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="queries in flight: query slots (own stream + workspace, one shared database: piehip_attach_database) that "
                         "run() round-robin.  0 = default (3: one hardware queue each on a runtime with four), 1 = one query at a time")
    ap.add_argument("--batch", type=int, default=0,
                    help="queries per run() (piehip_set_query_batch): a step is one run() over that many queries, each with its own "
                         "inputs and results; stage A reads the database once per batch.  0 = default")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams run() spreads the bin layers over (0 = library default, 1 = serial: every kernel alone on the GPU)")
    ap.add_argument("--collective", default="auto", choices=["auto", "gather", "all_gather"],
                    help="result collection across ranks: to rank 0 (what a server needs) or to every rank; auto times both "
                         "once before the warm-up and keeps the faster")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="rehearsal of the N > 1 code path on a one-GPU box: every rank uses cuda:0 and the results are gathered "
                         "over gloo (host-staged).  The timing means nothing; the line is marked")
    ap.add_argument("--force-collective", action="store_true",
                    help="rehearsal: run the RCCL gather path even with one rank (launch under torch.distributed.run)")
    ap.add_argument("--repeats", type=int, default=REPEATS, help="timed blocks of --steps steps; the median block is reported")
    ap.add_argument("--warm-seconds", type=float, default=WARM_SECONDS, help="minimum warm-up time before the timed blocks")
    ap.add_argument("--query-dist", default="auto", choices=["auto", "broadcast", "scatter_gather", "none"],
                    help="N > 1: how the per-query inputs travel from rank 0 to the ranks inside every step (auto: time both, keep "
                         "the faster); none: every rank already holds them")
    ap.add_argument("--query-source", default="hbm", choices=["hbm", "host"],
                    help="N > 1: rank 0's copy of the query is resident in HBM (as at N = 1) or in page-locked host memory")
    args = ap.parse_args()
    if args.timed_only:
        args.no_cpu_baseline = args.no_e2e = args.no_projection = args.no_ref_timer = True

    # stdout carries exactly one JSON line: libraries that print banners to fd 1 (RCCL's version block) go to stderr
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from nested_hashing_psi_amd import pie, shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the PIE hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or (args.force_collective and "RANK" in os.environ)
    if use_dist:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)  # nccl == RCCL on ROCm

    cfg = dict(CONFIGS[args.config], name=args.config)
    N, L, t, K, E, b = cfg["N"], cfg["L"], cfg["t"], cfg["K"], cfg["E"], cfg["b"]
    B = cfg["k"] * cfg["e"]
    scaling = args.scaling or ("strong" if world > 1 else "weak")   # one GPU: the two coincide
    if scaling == "strong" and world > 1:
        lo, hi = shard.bin_slice(b, rank, world)
        b_local = hi - lo
        b_total = b
    else:
        b_local = b
        b_total = b * world
    if args.bins_per_rank and world == 1:
        b_local = b_total = min(b, args.bins_per_rank)

    # the context's stream: a non-default torch stream, so that torch / RCCL work queued under it and the library's own
    # queues order against each other through the same stream (the null stream does not order against non-blocking ones)
    stream = torch.cuda.Stream(device)
    cc = pie.PieContext(N, L, t, device=local_rank, stream=stream.cuda_stream)
    gen = torch.Generator(device=device)
    gen.manual_seed(123456789)                 # the query and the key are the same on every rank
    rng = np.random.default_rng(987654321 + rank)
    # relinearisation key and per-query inputs: resident in HBM before the timed region
    evk = uniform_limbs(torch, (L, 2), cc.q, N, device, gen)
    idx = uniform_limbs(torch, (K, E, 2), cc.q, N, device, gen)
    minus = uniform_limbs(torch, (2,), cc.q, N, device, gen)
    torch.cuda.synchronize(device)
    cc.load_relin_key(evk.cpu().numpy().view(np.uint64))
    op = synthetic_operator(pie, cc, cfg, b_local, rng, (idx, minus)) if b_local > 0 else None
    # Several queries at once, two ways.
    # Queries in flight (--in-flight): further query slots (a context with its own stream and run() workspace, reading slot 0's
    # key and database by reference) take the steps round-robin; one query's ct x pt stage is HBM-bound while another's
    # transforms are ALU-bound, and a small share of bin layers leaves most of the chip idle.  With slots, one queue per run()
    # is the faster setting; three slots, because the HIP runtime multiplexes streams onto four hardware queues and streams that
    # share one serialise (profiles/r03/queries_in_flight.txt).
    # Queries per run() (--batch).  A server with several clients waiting evaluates their queries together: stage A streams the packed
    # database (3/4 of its traffic) once for the batch, and every later launch carries `batch` times the ciphertexts.
    # Defaults: a batch of three on one handle (one GPU); three slots with one query each where batches do not apply (N > 1:
    # the gather's buffers are per query; --graph; a rank's share of a few bin layers).
    batching = not (args.graph or args.bins_per_rank)
    if args.batch:
        batch = args.batch if not args.graph else 1
        in_flight = args.in_flight or (3 if use_dist else 1)
    elif args.in_flight or not batching:
        batch, in_flight = 1, args.in_flight or 3
    else:
        # a rank's share of the bin layers is small: batches of three on three slots (projected_strong_scaling measures both)
        batch, in_flight = DEFAULT_BATCH, (3 if use_dist else 1)
    if args.graph:
        in_flight = 1
    run_streams = args.streams or (1 if in_flight > 1 else 0)
    cc.set_run_streams(run_streams)
    cc.set_graph(args.graph)
    slots = [(cc, op, stream, idx, minus)]
    # (one GPU: two further slots exist in any case -- the legs after the timed region use them -- but only the first
    # `in_flight` take steps)
    n_slots = max(in_flight, 3) if (batching and not use_dist and op is not None) else in_flight
    if n_slots > 1 and op is not None:
        slots += make_query_slots(torch, pie, cc, op, (N, L, t, K, E), n_slots, device, local_rank, gen, args.streams or 1)
    elif n_slots > 1:   # a rank without bin layers still takes part in every slot's collective
        slots += [(None, None, torch.cuda.Stream(device), None, None) for _ in range(1, n_slots)]
    extra_inputs = []
    if batch > 1:
        for c_, o_, st_, i_, m_ in slots[:in_flight]:
            if o_ is None:
                continue
            o_.setQueryBatch(batch)
            for q_ in range(1, batch):
                iq = uniform_limbs(torch, (K, E, 2), cc.q, N, device, gen)
                mq = uniform_limbs(torch, (2,), cc.q, N, device, gen)
                extra_inputs.append((iq, mq))
                o_.setIndexDevice(iq.data_ptr(), query=q_)
                o_.setMinusCompareElementDevice(mq.data_ptr(), query=q_)
        torch.cuda.synchronize(device)
    # N > 1: leave CUs to RCCL's kernels.  The transforms are persistent grids on every workgroup slot of the device (two per CU, with
    # the CU's whole register file) for 60-100 us at a time; the broadcast of the next query and the gather of the previous results
    # need CUs during exactly those launches.  Default: 16 CUs (32 slots) free on every rank; --transform-slots 0 fills the device.
    tcap, tdev = cc.transform_slots()
    if args.transform_slots is not None:
        tcap = args.transform_slots
    elif world > 1 and not args.rehearse_on_one_gpu:
        tcap = max(2, tdev - 32)
    for s_ in slots:
        if s_[0] is not None:
            s_[0].set_transform_slots(tcap)
    ct_words = 2 * L * N
    rg = None
    rgs = []
    if use_dist:
        if args.collective == "auto":
            # both move the same payload; which one RCCL runs faster over this node's xGMI topology is measured, not assumed
            times = {}
            for kind in ("gather", "all_gather"):
                try:
                    probe = shard.ResultGather(None, b_total, b_local, ct_words * batch, device, stream, kind=kind)
                    for rep in range(6):
                        if rep == 1:
                            torch.cuda.synchronize(device)
                            dist.barrier()
                            t_ = time.perf_counter()
                        probe.step()
                    probe.drain()
                    torch.cuda.synchronize(device)
                    tt_ = torch.tensor([time.perf_counter() - t_], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else device)
                    dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                    times[kind] = float(tt_.item())
                    del probe
                except (RuntimeError, NotImplementedError) as exc:   # every rank fails alike
                    sys.stderr.write("%s unavailable: %s\n" % (kind, exc))
            # the gather moves 1/world of the all-gather's bytes (less interference with run()): keep it unless clearly slower
            if "gather" in times and ("all_gather" not in times or times["gather"] <= 1.15 * times["all_gather"]):
                args.collective = "gather"
            else:
                args.collective = "all_gather"
            if rank == 0:
                sys.stderr.write("collective timing (5 rounds, s): %s -> %s\n" % (times, args.collective))
        # one double-buffered gather per query slot (slots without bin layers still take part in the collective); every slot has
        # a communicator of its own per direction (shard.slot_groups: collectives of one group run in issue order on one stream,
        # so slots that shared the default group would serialise against each other)
        q_words1 = (K * E * 2 + 2) * L * N   # one query: index matrix, then minus element
        groups = shard.slot_groups(len(slots))
        rgs = [shard.ResultGather(s_[1], b_total, b_local, ct_words * batch, device, s_[2], kind=args.collective, batch=batch,
                                  query_words=q_words1, group=groups[i_][1]) for i_, s_ in enumerate(slots)]
        rg = rgs[0]
    # Per-query input distribution (N > 1): the query ((K E + 1) ciphertexts) is resident in rank 0's HBM and reaches every rank
    # through the slot's QueryBroadcast inside every step -- the second collective of the sharded server, after
    # which the operator is pointed at the received copy.  --query-source host puts rank 0's copy in page-locked host memory
    # instead (the reference server's situation; PCIe upload inside the step).
    q_words, q_split = batch * (K * E * 2 + 2) * L * N, K * E * 2 * L * N
    qdist_kind, qdist_times = None, {}
    if use_dist and args.query_dist != "none":
        # rank 0's copy is the one that counts: the `batch` queries of a step one after the other
        gq = torch.Generator(device=device)
        gq.manual_seed(24680)
        flat_q = torch.cat([idx.reshape(-1), minus.reshape(-1)] +
                           [uniform_limbs(torch, shp, cc.q, N, device, gq).reshape(-1) for _ in range(1, batch) for shp in ((K, E, 2), (2,))])
        flat_h = flat_q.cpu().pin_memory() if args.query_source == "host" else None
        kinds = ["broadcast", "scatter_gather"] if args.query_dist == "auto" else [args.query_dist]
        if args.rehearse_on_one_gpu:
            kinds = kinds[:1]
        times = {}
        for kind in kinds:
            try:
                probe = shard.QueryBroadcast(q_words, device, src=0, kind=kind)
                probe.set_query_device(flat_q)
                for rep in range(6):
                    if rep == 1:
                        torch.cuda.synchronize(device)
                        dist.barrier()
                        t_ = time.perf_counter()
                    probe.step(rep & 1)
                torch.cuda.synchronize(device)
                tt_ = torch.tensor([time.perf_counter() - t_], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else device)
                dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                times[kind] = float(tt_.item()) / 5
                del probe
            except (RuntimeError, NotImplementedError, ValueError) as exc:   # every rank fails alike
                sys.stderr.write("query distribution %s unavailable: %s\n" % (kind, exc))
        if times:
            qdist_kind = min(times, key=times.get)
            if rank == 0:
                sys.stderr.write("query distribution timing (s per %.1f MiB query): %s -> %s\n" % (q_words * 8 / 2**20, times, qdist_kind))
            qdist_times = times
    nstep = [0]

    def step():
        i_ = nstep[0] % in_flight
        nstep[0] += 1
        if rgs:
            rgs[i_].step()      # query distribution, run() into a gather buffer, gather of the results, double-buffered
        elif op is not None:
            slots[i_][1].run(sync=False)

    def finish():               # both sides of every timed block: collectives drained, device idle, ranks together
        for r_ in rgs:
            r_.drain()
        torch.cuda.synchronize(device)
        if dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    def agree(x, how):          # a decision every rank must take alike (the ranks issue the same sequence of collectives)
        if not dist:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else device)
        dist.all_reduce(tt, op=how)
        return float(tt.item())

    # Warm-up: at least --warmup steps and at least WARM_SECONDS of them (a 20-step block is 6 ms of GPU time; clocks, caches and
    # the lazily created queues of every slot need longer than that), then REPEATS timed blocks of exactly --steps steps.
    def run_blocks():
        t_w = time.perf_counter()
        while True:
            for _ in range(max(args.warmup, in_flight)):
                step()
            finish()
            if agree(time.perf_counter() - t_w, dist.ReduceOp.MIN if dist else None) >= args.warm_seconds:
                break
        out_ = []
        for _ in range(max(1, args.repeats)):
            finish()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            finish()
            out_.append(agree(time.perf_counter() - t0, dist.ReduceOp.MAX if dist else None) / args.steps * 1e3)
        return out_

    # The timed region.  N = 1: the queries are resident in HBM when a step starts.  N > 1: they are resident in RANK 0's HBM (the
    # rank that holds the client's socket, BatchedFHEPSIServer.cpp:94-99) and every step distributes its queries to all ranks -- the
    # line's `value`; the steps are first timed with the queries already on every rank (`queries_resident_on_every_rank`).
    # (CPython's cyclic garbage collector stays off from here to the end of the measurements: a full collection of this process takes
    # ~40 ms -- 65 steps of the timed region -- and fires by allocation count wherever the Python harness happens to be;
    # profiles/r05/host_stream_slow_leg_is_python_gc.txt.  The product is the C library; --keep-gc leaves the collector alone.)
    if not args.keep_gc:
        import gc
        gc.collect()
        gc.disable()
    blocks = run_blocks()
    ms_per_step = median(blocks)
    dist_blocks = None
    if rgs and qdist_kind:
        for i_, rg_ in enumerate(rgs):
            qb = shard.QueryBroadcast(q_words, device, src=0, kind=qdist_kind, group=groups[i_][0])
            if flat_h is not None:
                qb.set_query_host(flat_h)
            else:
                qb.set_query_device(flat_q)
            rg_.query, rg_.query_split = qb, q_split
        dist_blocks = run_blocks()

    # per-kernel times of the same run(), HIP events on the launch stream (separate, untimed passes).  These passes are
    # serial (one stream): with the default two queues a kernel shares the chip with the other queue's kernels and its
    # duration says nothing about the kernel itself.  `bench.py --streams 1` runs the timed region the same way.
    roofline = None
    kernels = {}
    if op is not None and rank == 0:
        cc.set_run_streams(1)
        # warm, then every recorded pass runs straight behind an unrecorded one (the device does not idle in between);
        # per kernel class the MEDIAN pass is reported
        t_end = time.perf_counter() + 0.15
        while time.perf_counter() < t_end:
            for _ in range(10):
                op.run(sync=False)
            torch.cuda.synchronize(device)
        passes = []
        for _ in range(max(3, args.profile_steps)):
            cc.set_profiling(False)
            op.run(sync=False)
            cc.set_profiling(True)
            op.run(sync=True)
            passes.append(cc.profile())
        cc.set_profiling(False)
        cc.set_run_streams(run_streams)
        passes = passes[2:] if len(passes) > 4 else passes
        agg = {}
        for name in passes[0]:
            recs = sorted((p_[name] for p_ in passes if name in p_), key=lambda r_: r_["ms"])
            agg[name] = dict(recs[len(recs) // 2])
        pair = agg.pop("event_pair", None)   # the bracket itself: two events back to back, once per queue group of a pass
        pair_us = 1e3 * pair["ms"] / pair["launches"] if pair and pair["launches"] else None
        for name, a in agg.items():
            kernels[name] = dict(launches_per_step=a["launches"], us_per_step=1e3 * a["ms"],
                                 alg_GBps=a["alg_bytes"] / (a["ms"] * 1e-3) / 1e9 if a["ms"] > 0 else None)
        ntt_ms = sum(agg[n]["ms"] for n in ("ntt_fwd", "ntt_inv") if n in agg)
        ntt_bytes = sum(agg[n]["alg_bytes"] for n in ("ntt_fwd", "ntt_inv") if n in agg)
        ntt_launch = sum(agg[n]["launches"] for n in ("ntt_fwd", "ntt_inv") if n in agg)
        if ntt_ms > 0:
            ach = ntt_bytes / (ntt_ms * 1e-3) / 1e9
            roofline = {"kernel": "ntt (forward+inverse, LDS-resident limb)", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(args.config),
                        "traffic_source": "HBM bytes per launch from the committed rocprofv3 PMC passes of this command (profiles/latest_pmc.json: "
                                          "FETCH_SIZE x 2 + WRITE_SIZE), not measured in this run",
                        "avg_launch_us": 1e3 * ntt_ms / ntt_launch, "alg_bytes_per_launch": ntt_bytes / ntt_launch,
                        "launches_per_step": ntt_launch,
                        # what the bracket reads with no launch inside it; rocprofv3's kernel durations (profiles/) do not contain it
                        "event_pair_us": pair_us,
                        "frac_net_of_event_pair": (ntt_bytes / ((ntt_ms - 1e-3 * pair_us * ntt_launch) * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                   if pair_us is not None and ntt_ms > 1e-3 * pair_us * ntt_launch else None),
                        "measured": "HIP events around every launch, median of %d serial passes of run() after a warm-up (one stream; the "
                                    "timed region uses %s)" % (len(passes), "%d queue(s) per run(), %d quer%s per run(), %d run() in flight"
                                                              % (run_streams or 2, batch, "y" if batch == 1 else "ies", in_flight))}

    if batch > 1:   # the legs below take one query per run()
        for c_, o_, st_, i_, m_ in slots[:in_flight]:
            if o_ is not None:
                o_.setQueryBatch(1)
    resident_blocks = None
    if dist_blocks:
        # N > 1: the headline is the step WITH the per-query distribution from rank 0 -- what the shipped sharded servers do inside
        # their online timer (host/ShardedBatchedFHEPSIServer.hpp evaluateStagedQuery: piehip_rccl_broadcast_query, run, gather);
        # the steps with the queries already resident on every rank are reported beside it (`queries_resident_on_every_rank`).
        # (r04 had the two the other way round: ADVICE r04.)
        resident_blocks, blocks = blocks, dist_blocks
        ms_per_step = median(blocks)
    if rank == 0:
        value = batch * b_total / (ms_per_step * 1e-3)
        cname = "C4 (C3's bin layers over %d GPUs)" % world if (world > 1 and scaling == "strong" and args.config == "C3") else args.config
        line = {
            "metric": "server PIE ciphertexts/sec", "value": value, "unit": "ciphertexts/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling if world > 1 else "n/a",
            "repeats": len(blocks), "ms_per_step_min": min(blocks), "ms_per_step_max": max(blocks),
            "timing": "median of %d blocks of exactly %d steps, each bracketed by barrier + synchronize, after >= %d steps and >= %.2f s "
                      "of warm-up" % (len(blocks), args.steps, args.warmup, args.warm_seconds),
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "python_gc": "left on (--keep-gc)" if args.keep_gc else "off during the measurements (a harness pause, not the product's: see bench.py)",
            "config": {"workload": "%s: BatchedFHEHIPPIE::run(), N=%d, %d RNS primes (60-bit), t=%d, |S|=2^%d |C|=2^%d, k=%d e=%d (B=%d slots), "
                                   "K=%d E=%d, b=%d bin layers in all, %d on rank 0; %d quer%s per step: %d ct x pt MACs + %d ct x ct (HPS + BV relin) + %d mask mults per step"
                                   % (cname, N, L, t, cfg["S"].bit_length() - 1, cfg["C"].bit_length() - 1, cfg["k"], cfg["e"], B, K, E,
                                      b_total, b_local, batch, "y" if batch == 1 else "ies", batch * b_total * K * E, batch * b_total * (K - 1),
                                      batch * b_total),
                       "result_ciphertexts_per_step": batch * b_total, "queries_per_step": batch, "parallelism": "bins%d" % world,
                       "collective": ("%s: inside every step %s of its queries from rank 0 (%s memory) to every rank and %s of the results to "
                                      "rank 0, one communicator per query slot and direction%s"
                                      % ("gloo (rehearsal)" if args.rehearse_on_one_gpu else "rccl", qdist_kind or "NO distribution",
                                         args.query_source, args.collective,
                                         "; the same steps with the queries resident on every rank: queries_resident_on_every_rank" if dist_blocks else
                                         " (queries resident on every rank)")) if use_dist else "none",
                       "rccl_ranks": (dist.get_world_size() if dist else 1), "backend": (dist.get_backend() if dist else "none"),
                       "transform_slots": {"cap": tcap, "device": tdev},
                       "query_distribution_s": qdist_times,
                       "queries_in_flight": in_flight},
            "mac_per_s": batch * b_total * K * E / (ms_per_step * 1e-3), "mul_per_s": batch * b_total * (K - 1) / (ms_per_step * 1e-3),
            "ms_per_query": ms_per_step / batch, "queries_per_step": batch,
            "queries_in_flight": in_flight, "run_streams": run_streams or 2, "hipgraph": bool(args.graph),
            **({"rehearsal": "all ranks on one GPU, gloo gather: not a measurement"} if args.rehearse_on_one_gpu else {}),
            # whole run(): algorithmic bytes of the REFERENCE's unfused schedule over the measured time.  Not HBM
            # utilisation: this build's schedule moves fewer bytes than the formula counts (fused stage A, 95 instead of 111
            # limb transforms per multiplication), so the fraction says how far the run is from the 8 TB/s bound of that schedule
            "run_roofline": {"alg_bytes_per_run_per_gpu": alg_bytes_run(dict(cfg, b=b_local)),
                             "ref_schedule_GBps_per_gpu": batch * alg_bytes_run(dict(cfg, b=b_local)) / (ms_per_step * 1e-3) / 1e9,
                             "ref_schedule_bytes_over_time_frac": batch * alg_bytes_run(dict(cfg, b=b_local)) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "bytes of the reference's unfused schedule / measured time / 8 TB/s; not measured HBM traffic"},
            "roofline": roofline, "kernels": kernels,
        }
        if resident_blocks:
            rms = median(resident_blocks)
            line["query_distribution"] = {"kind": qdist_kind, "source": args.query_source, "MiB_per_step": q_words * 8 / 2**20}
            line["queries_resident_on_every_rank"] = {
                "ms_per_step": rms, "value": batch * b_total / (rms * 1e-3), "ms_per_step_min": min(resident_blocks),
                "ms_per_step_max": max(resident_blocks),
                "what": "the same steps WITHOUT the per-query distribution: every step's %d quer%s already resident in every rank's HBM "
                        "when the step starts.  No shipped server works like that (the C++ sharded server broadcasts inside its online "
                        "timer); `value` above has the distribution inside every step" % (batch, "y" if batch == 1 else "ies")}
        if world == 1 and op is not None and (in_flight > 1 or batch > 1) and not args.bins_per_rank and not args.timed_only:
            # the same steps with one query at a time (one slot, one query per run(), the library's default of two queues)
            cc.set_run_streams(args.streams)
            one_ms = time_runs(op, args.steps, args.warmup, lambda: torch.cuda.synchronize(device), repeats=args.repeats, warm_seconds=0.15)
            line["one_query_at_a_time"] = {"ms_per_step": one_ms, "value": b_total / (one_ms * 1e-3), "run_streams": args.streams or 2}
            if len(slots) >= 3 and not (in_flight == 3 and batch == 1):
                # ... and with three query slots in flight, one query per run() each, one queue per run(): the default timed
                # region of rounds 2 and 3 before query batches
                cc.set_run_streams(1)
                tri_ms = time_slots([s_[1] for s_ in slots[:3]], args.steps, args.warmup, lambda: torch.cuda.synchronize(device),
                                    repeats=args.repeats, warm_seconds=0.15)
                line["three_query_slots_in_flight"] = {"ms_per_step": tri_ms, "value": b_total / (tri_ms * 1e-3), "run_streams": 1,
                                                       "queries_per_step": 1}
            cc.set_run_streams(run_streams)
        if world == 1 and op is not None and not args.no_ref_timer:
            if len(slots) > 1:
                cc.set_run_streams(args.streams or 1)   # the pipelined leg keeps every slot on one queue per run()
            rt = reference_timer(torch, op, idx, minus, b_local, 15, device, [(s_[1], s_[3], s_[4]) for s_ in slots[1:]], run_streams,
                                 batch=DEFAULT_BATCH, keep_gc=args.keep_gc)
            cc.set_run_streams(run_streams)
            line["ref_timer"] = rt
            # reference timer placement, query in host memory, result list back in host memory when the timer stops: one query
            # (latency), and a stream of queries over the query slots (throughput)
            line["value_ref_timer"] = rt["value_run_host_pinned_with_results"]
            # a stream of host-memory queries: one query per run() over three slots, or batches of three per run() (the better one)
            streams_ = {k_: rt[k_] for k_ in ("value_run_host_async_stream", "value_staged_batch_stream") if k_ in rt}
            if streams_:
                best_ = max(streams_, key=streams_.get)
                line["value_ref_timer_stream"] = streams_[best_]
                line["value_ref_timer_stream_mode"] = "batches of %d queries per run()" % DEFAULT_BATCH if best_ == "value_staged_batch_stream" else "one query per run()"
        if world == 1 and not args.no_projection and args.config == "C3" and not args.bins_per_rank:
            line["projected_strong_scaling"] = projected_strong_scaling(torch, pie, cfg, device, local_rank, gen, max(20, args.steps // 4),
                                                                        max(5, args.warmup // 2))
        if not args.no_e2e and world == 1:
            # a context of its own: the timed one owns the key and database its query slots are attached to (it refuses reloads)
            cc_e2e = pie.PieContext(N, L, t, device=local_rank, stream=torch.cuda.Stream(device).cuda_stream)
            line["e2e_psi"] = e2e_psi(cfg, pie, cc_e2e, lambda: torch.cuda.synchronize(device))
            cc_e2e.close()
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, max_threads=max(1, args.cpu_threads))
            line["speedup_vs_cpu_1core"] = value / line["cpu_baseline"]["value"]
            if "all_cores" in line["cpu_baseline"]:
                line["speedup_vs_cpu_all_cores"] = value / line["cpu_baseline"]["all_cores"]["value"]
            if "e2e_psi" in line:
                line["e2e_speedup_vs_cpu_1core"] = line["cpu_baseline"]["e2e_psi_cpu_s"]["total_s"] / line["e2e_psi"]["total_s"]
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(line))
        sys.stdout.flush()
        os.dup2(2, 1)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    for s_ in reversed(slots):
        if s_[0] is not None:
            s_[0].close()


if __name__ == "__main__":
    main()
