"""BASELINE config C4 on the hardware available to the tests: the bin layers of one database split over several ranks,
every rank evaluating its slice with the real library on the GPU (not the oracle), the query known to rank 0 only and
distributed inside every step (shard.QueryBroadcast), results gathered to rank 0 through shard.ResultGather -- the
double-buffered distribute / run_into / join / gather sequence bench.py times.  The small-ring tests compare, bit for bit,
with one handle evaluating all bin layers; the C4 tests at the end run BASELINE's real shape (N=16384, 4 primes, |S|=2^20,
K=2, E=14, b=14) with every rank building its slice from the raw server set (piehip_build_db_bins) and compare with the
ORACLE's run() on all 14 bin layers.  One MI355X is visible to the tests, so the ranks share it and the
collective runs over `gloo` (RCCL refuses two ranks on one device); a one-rank RCCL group covers the device-tensor path.
Consecutive queries differ, so a gather that read a buffer too early or a run that overwrote it too soon shows up as a
mismatch."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, backend, b, nq, q, nslots=1, batch=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    try:
        device = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from nested_hashing_psi_amd import pie, shard
        N, L, t, K, E = 4096, 2, 65537, 2, 5
        rng = np.random.default_rng(42)  # the same database, key and queries on every rank

        def rl(shape, moduli):
            out = np.zeros(shape + (L, N), dtype=np.uint64)
            for i in range(L):
                out[..., i, :] = rng.integers(0, int(moduli[i]), shape + (N,), dtype=np.uint64)
            return out

        stream = torch.cuda.Stream(device)
        cc = pie.PieContext(N, L, t, device=0, stream=stream.cuda_stream)
        db, masks, evk = rl((K, b, E), cc.q), rl((b,), cc.q), rl((L, 2), cc.q)
        # nq steps of `batch` queries each (batch > 1: query batches, piehip_set_query_batch -- bench.py's N > 1 path)
        queries = [[(rl((K, E, 2), cc.q), rl((2,), cc.q)) for _ in range(batch)] for _ in range(nq)]
        if rank != 0:
            queries = [None] * nq     # only rank 0 (the rank that talks to the client) knows the queries
        cc.load_relin_key(evk)
        lo, hi = shard.bin_slice(b, rank, world)
        op = None
        if hi > lo:
            op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=np.ascontiguousarray(db[:, lo:hi]), preCalcRandomMask=np.ascontiguousarray(masks[lo:hi]))
        ct_words = 2 * L * N
        # query slots (bench.py's N > 1 path): every slot has its own context, stream, inputs and double-buffered gather; the
        # queries go round the slots, so the slots' collectives interleave in the same order on every rank
        slots = []
        for s_ in range(nslots):
            st_ = stream if s_ == 0 else torch.cuda.Stream(device)
            cc_ = cc if s_ == 0 else pie.PieContext(N, L, t, device=0, stream=st_.cuda_stream)
            op_ = op if s_ == 0 else (pie.BatchedFHEHIPPIE(cc_, attachTo=op) if op is not None else None)
            q_split = K * E * 2 * L * N
            q_words = q_split + 2 * L * N
            if op_ is not None and batch > 1:
                op_.setQueryBatch(batch)
            qb = shard.QueryBroadcast(batch * q_words, device, src=0, kind="broadcast")
            slots.append(dict(cc=cc_, op=op_, stream=st_, qb=qb,
                              rg=shard.ResultGather(op_, b, hi - lo, batch * ct_words, device, st_, kind="gather", query=qb,
                                                    query_split=q_split, batch=batch, query_words=q_words)))
        got, keep = [], []
        for i, query in enumerate(queries):
            sl = slots[i % nslots]
            if rank == 0:   # the query in page-locked host memory, one array per query (the upload is asynchronous)
                flat = torch.from_numpy(np.concatenate([part.reshape(-1) for one in query for part in one]).view(np.int64)).pin_memory()
                keep.append(flat)
                sl["qb"].set_query_host(flat)
            got.append((sl, sl["rg"].step()))
        for sl in slots:
            sl["rg"].drain()
        torch.cuda.synchronize(device)
        ok = True
        detail = ""
        if rank == 0:
            # the last two queries of every slot are still in its two buffer sets: compare them with an unsharded handle
            cc1 = pie.PieContext(N, L, t, device=0)
            cc1.load_relin_key(evk)
            full = pie.BatchedFHEHIPPIE(cc1, vectorizedHCT=db, preCalcRandomMask=masks)
            for i in range(max(0, nq - 2 * nslots), nq):
                rows = got[i][0]["rg"].rows(got[i][1]).cpu().numpy()      # [b][batch * ct_words]: a bin layer's results, query by query
                if rows.shape != (b, batch * ct_words):
                    ok = False
                    detail += "step %d: shape %s; " % (i, rows.shape)
                    continue
                for j, (idx, minus) in enumerate(queries[i]):
                    full.setMinusCompareElement(minus)
                    full.setIndex(idx)
                    full.run()
                    want = full.getResultList().reshape(b, ct_words).view(np.int64)
                    if not (rows[:, j * ct_words:(j + 1) * ct_words] == want).all():
                        ok = False
                        detail += "step %d query %d differs; " % (i, j)
            cc1.close()
        for sl in reversed(slots):
            sl["cc"].close()
        q.put((rank, ok, detail))
    except Exception as e:  # report instead of hanging the parent on q.get
        import traceback
        q.put((rank, False, repr(e) + traceback.format_exc()))
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


def _run(world, backend, b, nq=5, nslots=1, batch=1):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, b, nq, q, nslots, batch)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    assert all(ok for _, ok, _ in res), res


@pytest.mark.parametrize("world,b", [(2, 7), (3, 14), (4, 14)])
def test_sharded_ranks_on_one_gpu_match_unsharded(world, b):
    """uneven splits included: b = 14 over 4 ranks is 3 + 4 + 3 + 4, b = 7 over 2 is 3 + 4"""
    _run(world, "gloo", b)


def test_sharded_ranks_with_query_slots():
    """two ranks x three query slots each (bench.py's N > 1 configuration): nine different queries go round the slots, every
    slot double-buffers its own gather, and the last two queries of every slot equal the unsharded evaluation"""
    _run(2, "gloo", 7, nq=9, nslots=3)


def test_sharded_ranks_with_query_batches():
    """bench.py's default N > 1 step: three queries per run() (piehip_set_query_batch) on three query slots per rank; the
    distributed array holds the batch's queries one after the other, a gathered row the batch's results of one bin layer.
    Every query against the unsharded, one-query-per-run evaluation (gloo; b = 14 over four ranks is 3 + 4 + 3 + 4)."""
    _run(2, "gloo", 7, nq=7, nslots=3, batch=3)
    _run(4, "gloo", 14, nq=4, nslots=1, batch=2)


def test_one_rank_rccl_device_gather():
    """the RCCL form of the same sequence (device tensors, asynchronous gather on RCCL's stream), also over three slots"""
    _run(1, "nccl", 6)
    _run(1, "nccl", 6, nq=7, nslots=3)
    _run(1, "nccl", 6, nq=5, nslots=2, batch=3)


# ---- C4 at its real shape, against the oracle ------------------------------------------------------------------------------
T32 = 4296540161
C3 = dict(N=16384, L=4, t=T32, k=2, e=4949, K=2, E=14, b=14, nS=1 << 20, nC=1 << 10)
SEEDS = dict(hash_seed=987654321, evict_seed=1, shuffle_seed=2, mask_seed=3)


def _c3_sets(seed=123456789):
    rng = np.random.default_rng(seed)
    items = np.unique(rng.integers(1, C3["t"], C3["nS"] + 2 * C3["nC"] + 8192, dtype=np.uint64))
    rng.shuffle(items)
    server = items[:C3["nS"]].copy()
    ninter = C3["nC"] // 2 + 1
    clients = []
    for q_ in range(2):     # two different client sets: their intersections with the server set differ
        fresh = items[C3["nS"] + q_ * C3["nC"]: C3["nS"] + (q_ + 1) * C3["nC"] - ninter]
        inter = server[q_ * 4096: q_ * 4096 + ninter]
        c = np.concatenate([inter, fresh])
        rng.shuffle(c)
        clients.append((c, inter))
    return server, clients


def _c3_oracle_side(ob, server, clients):
    """what only the client / the checker know: keys, the encrypted queries, the oracle's packed database"""
    N, L, t, k, e, K, E, b = (C3[x] for x in ("N", "L", "t", "k", "e", "K", "E", "b"))
    o = ob.Oracle(N, L, t)
    tab = ob.Tabulation(SEEDS["hash_seed"], k + K)
    tbl = ob.hct_build(tab, server, k, e, K, b, E, evict_seed=SEEDS["evict_seed"])
    ob.hct_shuffle_bins(tbl, SEEDS["shuffle_seed"])
    slots = ob.pack_db(tbl)
    mask_slots = ob.masks(t, b, k * e, SEEDS["mask_seed"])
    sk = o.keygen(11)
    evk = o.relin_keygen(sk, 12)
    queries = []
    for qi, (client, inter) in enumerate(clients):
        ctab = ob.client_build(tab, client, k, e, evict_seed=4 + qi)
        index, minus_v = ob.client_vectors(tab, ctab, K, E)
        idx = np.stack([o.encrypt_slots(sk, index[h, j], 1000 * qi + 100 + h * E + j) for h in range(K) for j in range(E)]).reshape(K, E, 2, L, N)
        minus = o.encrypt_slots(sk, minus_v, 1000 * qi + 99)
        queries.append(dict(idx=idx, minus=minus, ctab=ctab, inter=inter))
    return o, tbl, slots, mask_slots, sk, evk, queries


def _c3_oracle_results(ob, o, slots, mask_slots, evk, query, threads=8):
    """the oracle's run() on all b bin layers (bin layers are independent: one task each; the C call releases the GIL)"""
    import concurrent.futures
    L, N, K, E, b = C3["L"], C3["N"], C3["K"], C3["E"], C3["b"]

    def layer(bn):
        db = np.stack([o.encode_eval(slots[h, bn, j]) for h in range(K) for j in range(E)]).reshape(K, 1, E, L, N)
        return o.pie_run(query["idx"], query["minus"], db, o.encode_eval(mask_slots[bn])[None], evk)[0]
    with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as pool:
        return np.stack(list(pool.map(layer, range(b))))


def _c3_check(ob, o, sk, query, rows, want):
    """rows == the oracle's ciphertexts, and they decrypt (positive budget) to exactly the 513-item intersection"""
    k, e, b = C3["k"], C3["e"], C3["b"]
    assert rows.shape == want.shape and (rows == want).all(), "gathered result differs from the oracle"
    dec = []
    for bn in range(b):
        d, bud = o.decrypt_slots(sk, rows[bn], k * e)
        assert bud > 0
        dec.append(d)
    found = ob.client_scan(query["ctab"], np.stack(dec))
    assert len(found) == C3["nC"] // 2 + 1
    assert sorted(int(v) for v in found) == sorted(int(v) for v in query["inter"])


def _c4_worker(rank, world, port, q, nslots, batch=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    try:
        device = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from nested_hashing_psi_amd import pie, shard
        N, L, t, k, e, K, E, b = (C3[x] for x in ("N", "L", "t", "k", "e", "K", "E", "b"))
        server, clients = _c3_sets()          # the server set is every rank's; the queries are rank 0's
        ct_words = 2 * L * N
        side = None
        evk_t = torch.zeros((L, 2, L, N), dtype=torch.int64)
        if rank == 0:
            from oracle import binding as ob
            ob.build()
            side = _c3_oracle_side(ob, server, clients)
            evk_t = torch.from_numpy(side[5].view(np.int64)).clone()
        dist.broadcast(evk_t, src=0)          # the EvalMult key reaches the server once per session (BatchedFHEPSIServer.cpp:49)
        stream = torch.cuda.Stream(device)
        cc = pie.PieContext(N, L, t, device=0, stream=stream.cuda_stream)
        cc.load_relin_key(evk_t.numpy().view(np.uint64))
        lo, hi = shard.bin_slice(b, rank, world)
        # the offline phase on every rank: the whole table is hashed and shuffled, this rank packs its bin layers only
        op = pie.BatchedFHEHIPPIE(cc, serverSet=server, hashParams=dict(k=k, e=e, K=K, b=b, E=E, **SEEDS), binSlice=(lo, hi))
        ok, detail = True, ""
        if rank == 0 and not (op.hashTable() == side[1]).all():
            ok, detail = False, "hash table differs from the oracle's; "
        q_split = K * E * 2 * L * N
        q_words = q_split + ct_words          # one query in the distributed array: index matrix, then minus element
        slots = []
        for s_ in range(nslots):
            st_ = stream if s_ == 0 else torch.cuda.Stream(device)
            cc_ = cc if s_ == 0 else pie.PieContext(N, L, t, device=0, stream=st_.cuda_stream)
            op_ = op if s_ == 0 else pie.BatchedFHEHIPPIE(cc_, attachTo=op)
            if batch > 1:
                op_.setQueryBatch(batch)      # a step of this slot evaluates `batch` queries (bench.py's N > 1 default: 3 on 3 slots)
            qb = shard.QueryBroadcast(batch * q_words, device, src=0, kind="broadcast")
            slots.append(dict(cc=cc_, qb=qb, rg=shard.ResultGather(op_, b, hi - lo, batch * ct_words, device, st_, kind="gather", query=qb,
                                                                   query_split=q_split, batch=batch, query_words=q_words)))
        nq = 2 * nslots + 1                   # every slot uses both of its buffer sets; consecutive queries differ
        got, keep = [], []
        for i in range(nq):
            sl = slots[i % nslots]
            if rank == 0:
                parts = []
                for j in range(batch):        # the queries of a step alternate between the two clients, starting with i
                    qy = side[6][(i + j) % 2]
                    parts += [qy["idx"].reshape(-1), qy["minus"].reshape(-1)]
                flat = torch.from_numpy(np.concatenate(parts).view(np.int64)).pin_memory()
                keep.append(flat)
                sl["qb"].set_query_host(flat)
            got.append((sl, sl["rg"].step()))
        for sl in slots:
            sl["rg"].drain()
        torch.cuda.synchronize(device)
        if rank == 0:
            from oracle import binding as ob
            o, tbl, oslots, mask_slots, sk, evk, queries = side
            want = [_c3_oracle_results(ob, o, oslots, mask_slots, evk, qy) for qy in queries]
            for i in range(max(0, nq - 2 * nslots), nq):    # the last two steps of every slot are still in its buffer sets
                allrows = got[i][0]["rg"].rows(got[i][1]).cpu().numpy().view(np.uint64).reshape(b, batch, 2, L, N)
                for j in range(batch):
                    try:
                        _c3_check(ob, o, sk, queries[(i + j) % 2], np.ascontiguousarray(allrows[:, j]), want[(i + j) % 2])
                    except AssertionError as exc:
                        ok = False
                        detail += "step %d query %d: %s; " % (i, j, exc)
        for sl in reversed(slots):
            sl["cc"].close()
        q.put((rank, ok, detail))
    except Exception as e_:  # report instead of hanging the parent on q.get
        import traceback
        q.put((rank, False, repr(e_) + traceback.format_exc()))
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


@pytest.mark.parametrize("world,batch", [(2, 1), (4, 1), (5, 1), (4, 3)])
def test_c4_real_shape_against_the_oracle(world, batch):
    """BASELINE config C4 (C3's 14 bin layers over `world` ranks: 7+7, 3+4+3+4, 2+3+3+3+3), three query slots per rank, real
    secret-key encrypted queries that only rank 0 holds, every rank's database slice built by piehip_build_db_bins from the
    raw server set: rank 0's gathered rows equal the oracle's run() on all 14 layers and decrypt to the 513-item
    intersection.  The ranks share the one GPU of the test box and talk over gloo; the box admits six processes on one card
    (five ranks + this test runner; 8 slices: test_c4_eight_bin_slices_in_one_process).  batch = 3: every step of a slot is a
    batch of three queries (piehip_set_query_batch), the N > 1 step of bench.py; each query of each batch against the oracle."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c4_worker, args=(r, world, port, q, 3, batch)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    assert all(ok for _, ok, _ in res), res


def test_c4_eight_bin_slices_in_one_process(ob, pie_mod):
    """the 8-GPU partition of C3 (14 layers -> 1,2,2,2,1,2,2,2) as eight handles of one process, each built from the raw server
    set with its bin slice; the concatenated result lists equal the oracle's run() and decrypt to the intersection; the
    hash table every handle holds equals the oracle's"""
    from nested_hashing_psi_amd import shard
    pie = pie_mod
    N, L, t, k, e, K, E, b = (C3[x] for x in ("N", "L", "t", "k", "e", "K", "E", "b"))
    server, clients = _c3_sets()
    o, tbl, oslots, mask_slots, sk, evk, queries = _c3_oracle_side(ob, server, clients[:1])
    rows = np.zeros((b, 2, L, N), dtype=np.uint64)
    sizes = []
    for r in range(8):
        lo, hi = shard.bin_slice(b, r, 8)
        sizes.append(hi - lo)
        cc = pie.PieContext(N, L, t)
        cc.load_relin_key(evk)
        op = pie.BatchedFHEHIPPIE(cc, serverSet=server, hashParams=dict(k=k, e=e, K=K, b=b, E=E, **SEEDS), binSlice=(lo, hi))
        if r in (0, 7):
            assert (op.hashTable() == tbl).all()
        op.setMinusCompareElement(queries[0]["minus"])
        op.setIndex(queries[0]["idx"])
        op.run()
        rows[lo:hi] = op.getResultList()
        cc.close()
    assert sorted(sizes) == [1, 1, 2, 2, 2, 2, 2, 2]
    _c3_check(ob, o, sk, queries[0], rows, _c3_oracle_results(ob, o, oslots, mask_slots, evk, queries[0]))


# ---- RCCL behind the C ABI (piehip_rccl.cpp): what a C++ server calls -------------------------------------------------------
def test_native_rccl_gather_one_rank(ob, pie_mod):
    """piehip_rccl_unique_id / _init / _broadcast_query / piehip_gather_results_host with one rank (RCCL refuses two ranks on the one
    device of the test box): the staged query reaches the input buffers, run() evaluates it, the gathered list in page-locked host
    memory equals the oracle's run(); also for a batch of three queries, and the call-order errors."""
    pie = pie_mod
    N, L, t, K, E, b = 4096, 2, 65537, 2, 4, 5
    o = ob.Oracle(N, L, t)
    rng = np.random.default_rng(99)

    def rl(shape):
        out = np.zeros(shape + (L, N), dtype=np.uint64)
        for i in range(L):
            out[..., i, :] = rng.integers(0, int(o.q[i]), shape + (N,), dtype=np.uint64)
        return out

    db, masks, evk = rl((K, b, E)), rl((b,)), rl((L, 2))
    cc = pie.PieContext(N, L, t)
    cc.load_relin_key(evk)
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    with pytest.raises(RuntimeError, match="communicator"):
        op.gatherResultsHost(b)
    assert pie.rccl_bin_slice(b, 1, 0) == (0, b)
    cc.rccl_init(pie.rccl_unique_id(), 1, 0)
    with pytest.raises(RuntimeError, match="already"):
        cc.rccl_init(pie.rccl_unique_id(), 1, 0)
    with pytest.raises(RuntimeError, match="no staged query"):
        op.broadcastQuery(0)
    for nq in (1, 3):
        op.setQueryBatch(nq)
        queries = [(rl((K, E, 2)), rl((2,))) for _ in range(nq)]
        bufs = [op.hostBuffers(query=i) for i in range(nq)]
        for i, (idx, minus) in enumerate(queries):
            bufs[i][0][...] = idx
            bufs[i][1][...] = minus
            op.stageMinus(bufs[i][1], query=i)
            if i == nq - 1:
                with pytest.raises(RuntimeError, match="not staged"):
                    op.broadcastQuery(0)
            for h in range(K):
                op.stageIndexRow(h, bufs[i][0][h], query=i)
        op.broadcastQuery(0)
        op.run(sync=False)
        with pytest.raises(ValueError, match="slice"):
            op.gatherResultsHost(b + 1)
        got = op.gatherResultsHost(b, root=0)
        op.sync()
        for i, (idx, minus) in enumerate(queries):
            want = o.pie_run(idx, minus, db, masks, evk)
            assert ((got if nq == 1 else got[:, i]) == want).all(), "query %d of %d" % (i, nq)
    cc.rccl_destroy()
    cc.close()


@pytest.mark.parametrize("shape", ["small", "C3"])
def test_cpp_server_over_rccl_one_rank(pie_mod, tmp_path, shape):
    """host/ShardedBatchedFHEPSIServer.hpp as its own process -- C++ over the C ABI only, no torch: context, key and seeds through its
    set-up, the database built for the rank's bin layers, the query staged under the receive loop, piehip_rccl_broadcast_query +
    piehip_run + piehip_gather_results_host inside the reference's timer.  One rank (one GPU here); this process is the client."""
    import os
    import socket
    import struct
    import subprocess
    from nested_hashing_psi_amd.client import BatchedFHEPSIClient
    pie = pie_mod
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "nested_hashing_psi_amd")
    exe = str(tmp_path / "sharded_server_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(root, "tests", "sharded_server_main.cpp"),
                           "-L" + libdir, "-lpiehip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    rng = np.random.default_rng(4242)
    if shape == "small":
        N, L, t, k, e, K, E, b, nS, nC, ninter = 8192, 3, T32, 3, 40, 2, 8, 7, 2000, 64, 33
    else:
        N, L, t, k, e, K, E, b, nS, nC, ninter = C3["N"], C3["L"], T32, C3["k"], C3["e"], C3["K"], C3["E"], C3["b"], C3["nS"], C3["nC"], 513
    items = np.unique(rng.integers(1, t, nS + nC + 8192, dtype=np.uint64))
    rng.shuffle(items)
    server = items[:nS].copy()
    clientset = np.concatenate([server[:ninter], items[nS:nS + nC - ninter]])
    rng.shuffle(clientset)
    setfile = tmp_path / "server_set.bin"
    server.astype(np.uint64).tofile(setfile)
    a, bsock = socket.socketpair()
    proc = subprocess.Popen([exe, "0", "1", "0", str(bsock.fileno()), "-", str(setfile), str(k), str(e), str(K), str(E), str(b)],
                            pass_fds=(bsock.fileno(),), stdout=subprocess.PIPE)
    bsock.close()

    def send(payload):
        a.sendall(struct.pack("i", len(payload)) + payload)

    def recv():
        hdr = b""
        while len(hdr) < 4:
            hdr += a.recv(4 - len(hdr))
        n, = struct.unpack("i", hdr)
        buf = bytearray()
        while len(buf) < n:
            buf += a.recv(min(1 << 20, n - len(buf)))
        return bytes(buf)

    def ct_msg(ct):
        return struct.pack("IIIIQ", 0x48454950, 1, L, N, 0) + np.ascontiguousarray(ct, dtype=np.uint64).tobytes()

    cc = pie.PieContext(N, L, t)
    cl = BatchedFHEPSIClient(cc, k, e, K, E, b)
    evk = cl.runSetUpPhase()
    moduli = np.zeros(15, dtype=np.uint64)
    moduli[:2 * L + 1] = cc.moduli[:2 * L + 1]
    send(struct.pack("IIQ", N, L, t) + moduli.tobytes())
    send(b"")
    send(np.ascontiguousarray(evk, dtype=np.uint64).tobytes())
    assert recv() == b""
    minus_ct, idx_ct = cl.runOfflinePhase(clientset)
    assert recv() == b""
    send(ct_msg(minus_ct))
    for h in range(K):
        for j in range(E):
            send(ct_msg(idx_ct[h, j]))
    res = []
    for _ in range(b):
        m = recv()
        assert struct.unpack("IIIIQ", m[:24]) == (0x48454950, 1, L, N, 0)
        res.append(np.frombuffer(m[24:], dtype=np.uint64).reshape(2, L, N))
    out, _ = proc.communicate(timeout=180)
    assert proc.returncode == 0
    online_us = int([ln for ln in out.decode().splitlines() if ln.startswith("OnlineComputation,")][0].split(",")[1])
    print("C++ server over RCCL (one rank), %s shape: OnlineComputation %d us" % (shape, online_us))
    found = cl.extractIntersection(np.stack(res))
    assert sorted(int(v) for v in found) == sorted(int(v) for v in server[:ninter])
    a.close()
    cc.close()
