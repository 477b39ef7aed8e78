"""N > 1 path on CPU: two `gloo` ranks each evaluate their slice of bin layers (with the oracle standing
in for the GPU) and gather; the result must equal the unsharded run bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, b, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nested_hashing_psi_amd import shard
        from oracle import binding as ob
        N, L, t, K, E = 1024, 2, 65537, 2, 3
        o = ob.Oracle(N, L, t)
        rng = np.random.default_rng(42)  # same inputs on every rank

        def rl(shape):
            out = np.zeros(shape + (L, N), dtype=np.uint64)
            for i in range(L):
                out[..., i, :] = rng.integers(0, int(o.q[i]), shape + (N,), dtype=np.uint64)
            return out

        idx, minus, db, masks, evk = rl((K, E, 2)), rl((2,)), rl((K, b, E)), rl((b,)), rl((L, 2))
        lo, hi = shard.bin_slice(b, rank, world)
        # this rank holds only its slice of the database (as a GPU rank would)
        local = o.pie_run(idx, minus, np.ascontiguousarray(db[:, lo:hi]), np.ascontiguousarray(masks[lo:hi]), evk)
        local_t = torch.from_numpy(local.reshape(hi - lo, 2 * L * N).view(np.int64))
        full = shard.gather_bins(local_t, b, world)
        want = o.pie_run(idx, minus, db, masks, evk).reshape(b, 2 * L * N).view(np.int64)
        ok = bool((full.numpy() == want).all()) and full.shape[0] == b
        # the server's form: results to rank 0 only, asynchronously (what bench.py does over RCCL)
        out0, work = shard.gather_bins_to(local_t, b, world, dst=0, async_op=True)
        if work is not None:
            work.wait()
        if rank == 0:
            bmax = shard.max_bins(b, world)
            rows = []
            for r in range(world):
                rlo, rhi = shard.bin_slice(b, r, world)
                rows.append(out0[r * bmax: r * bmax + (rhi - rlo)])
            ok = ok and bool((torch.cat(rows).numpy() == want).all())
        else:
            ok = ok and out0 is None
        q.put((rank, ok, (lo, hi)))
    except Exception as e:  # report instead of hanging the parent on q.get
        q.put((rank, False, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,b", [(2, 4), (2, 5), (2, 1), (4, 14), (4, 3)])
def test_gather_matches_unsharded(world, b):
    """two ranks, and four ranks with uneven slices (b = 14 -> 3 + 4 + 3 + 4 bin layers; b = 3 leaves one rank empty)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, b, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    slices = sorted(s for _, _, s in res)
    assert slices[0][0] == 0 and slices[-1][1] == b and all(slices[i][1] == slices[i + 1][0] for i in range(world - 1))
    assert max(hi - lo for lo, hi in slices) - min(hi - lo for lo, hi in slices) <= 1


def _worker_queries(rank, world, port, b, q, kind="broadcast"):
    """the sharded server's step on CPU: rank 0 alone knows the query; QueryBroadcast hands it to every rank, every rank
    evaluates its bin layers (the oracle stands in for the GPU), the results are gathered to rank 0"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nested_hashing_psi_amd import shard
        from oracle import binding as ob
        N, L, t, K, E = 1024, 2, 65537, 2, 3
        o = ob.Oracle(N, L, t)
        rng = np.random.default_rng(7)   # database and key: the same on every rank (loaded once per session)

        def rl(r_, shape):
            out = np.zeros(shape + (L, N), dtype=np.uint64)
            for i in range(L):
                out[..., i, :] = r_.integers(0, int(o.q[i]), shape + (N,), dtype=np.uint64)
            return out

        db, masks, evk = rl(rng, (K, b, E)), rl(rng, (b,)), rl(rng, (L, 2))
        lo, hi = shard.bin_slice(b, rank, world)
        seeds = shard.shared_seeds()      # drawn on rank 0, the same on every rank (what binSlice databases must be built with)
        split = K * E * 2 * L * N
        qb = shard.QueryBroadcast(split + 2 * L * N, "cpu", src=0, kind=kind)
        qrng = np.random.default_rng(1000 + rank)   # rank 0's stream is the only one that is used
        ok = True
        for i in range(3):
            want = None
            if rank == 0:
                idx, minus = rl(qrng, (K, E, 2)), rl(qrng, (2,))
                qb.set_query_host(torch.from_numpy(np.concatenate([idx.reshape(-1), minus.reshape(-1)]).view(np.int64)))
                want = o.pie_run(idx, minus, db, masks, evk).reshape(b, 2 * L * N).view(np.int64)
            s = i & 1
            qb.step(s)
            flat = qb.ready(s).numpy().view(np.uint64)
            idx_r = flat[:split].reshape(K, E, 2, L, N)
            minus_r = flat[split:split + 2 * L * N].reshape(2, L, N)
            local = np.zeros((0, 2, L, N), dtype=np.uint64)
            if hi > lo:
                local = o.pie_run(idx_r, minus_r, np.ascontiguousarray(db[:, lo:hi]), np.ascontiguousarray(masks[lo:hi]), evk)
            local_t = torch.from_numpy(local.reshape(hi - lo, 2 * L * N).view(np.int64))
            out0, work = shard.gather_bins_to(local_t, b, world, dst=0, async_op=True)
            if work is not None:
                work.wait()
            if rank == 0:
                bmax = shard.max_bins(b, world)
                rows = [out0[r * bmax: r * bmax + (shard.bin_slice(b, r, world)[1] - shard.bin_slice(b, r, world)[0])] for r in range(world)]
                ok = ok and bool((torch.cat(rows).numpy() == want).all())
        q.put((rank, ok, seeds))
    except Exception as e:  # report instead of hanging the parent on q.get
        import traceback
        q.put((rank, False, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,b,kind", [(2, 5, "broadcast"), (4, 6, "broadcast"), (3, 5, "scatter_gather"), (4, 6, "scatter_gather")])
def test_query_distribution_then_gather(world, b, kind):
    """three consecutive, different queries that only rank 0 knows; both ways of distributing them (a plain broadcast; a scatter
    of 1/world to every rank followed by an all-gather -- world = 3 does not divide the array: padded chunks)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_queries, args=(r, world, port, b, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert len({tuple(sd) for _, _, sd in res}) == 1 and len(res[0][2]) == 3, res   # one set of (evict, shuffle, mask) seeds


def test_bin_slices_partition():
    from nested_hashing_psi_amd import shard
    for b in (1, 7, 14, 30, 40):
        for world in (1, 2, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = shard.bin_slice(b, r, world)
                cover += list(range(lo, hi))
                assert hi - lo <= shard.max_bins(b, world)
            assert cover == list(range(b))


def _worker_slot_groups(rank, world, port, b, nslots, q):
    """bench.py's N > 1 wiring on CPU: every query slot owns a communicator per direction (shard.slot_groups); rank 0 alone knows
    the queries.  The slots' collectives are then issued in an order that one shared communicator could not serve --
    every slot's distribution first, the gathers in REVERSE slot order, and the distribution of slot s's NEXT query before slot
    s's gather has been waited for -- and every gathered result must still equal the unsharded evaluation."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nested_hashing_psi_amd import shard
        from oracle import binding as ob
        N, L, t, K, E = 1024, 2, 65537, 2, 3
        o = ob.Oracle(N, L, t)
        rng = np.random.default_rng(11)

        def rl(r_, shape):
            out = np.zeros(shape + (L, N), dtype=np.uint64)
            for i in range(L):
                out[..., i, :] = r_.integers(0, int(o.q[i]), shape + (N,), dtype=np.uint64)
            return out

        db, masks, evk = rl(rng, (K, b, E)), rl(rng, (b,)), rl(rng, (L, 2))
        lo, hi = shard.bin_slice(b, rank, world)
        groups = shard.slot_groups(nslots)
        flat_groups = [g for pair in groups for g in pair]
        distinct = len({id(g) for g in flat_groups}) == 2 * nslots and all(g is not dist.group.WORLD for g in flat_groups)
        split = K * E * 2 * L * N
        qbs = [shard.QueryBroadcast(split + 2 * L * N, "cpu", src=0, kind="broadcast", group=groups[s][0]) for s in range(nslots)]
        uses_own = all(qbs[s].group is groups[s][0] for s in range(nslots))
        qrng = np.random.default_rng(2000 + rank)
        ok = distinct and uses_own
        bmax = shard.max_bins(b, world)
        for rnd in range(2):
            wants, works, outs = [None] * nslots, [None] * nslots, [None] * nslots
            for s in range(nslots):                       # all distributions of the round first
                if rank == 0:
                    idx, minus = rl(qrng, (K, E, 2)), rl(qrng, (2,))
                    qbs[s].set_query_host(torch.from_numpy(np.concatenate([idx.reshape(-1), minus.reshape(-1)]).view(np.int64)))
                    wants[s] = o.pie_run(idx, minus, db, masks, evk).reshape(b, 2 * L * N).view(np.int64)
                qbs[s].step(rnd & 1)
            for s in reversed(range(nslots)):             # gathers in reverse slot order, asynchronously, on the slot's gather group
                flat = qbs[s].ready(rnd & 1).numpy().view(np.uint64)
                idx_r, minus_r = flat[:split].reshape(K, E, 2, L, N), flat[split:split + 2 * L * N].reshape(2, L, N)
                local = np.zeros((0, 2, L, N), dtype=np.uint64)
                if hi > lo:
                    local = o.pie_run(idx_r, minus_r, np.ascontiguousarray(db[:, lo:hi]), np.ascontiguousarray(masks[lo:hi]), evk)
                local_t = torch.from_numpy(local.reshape(hi - lo, 2 * L * N).view(np.int64))
                outs[s], works[s] = shard.gather_bins_to(local_t, b, world, dst=0, group=groups[s][1], async_op=True)
            for s in range(nslots):
                if works[s] is not None:
                    works[s].wait()
                if rank == 0:
                    rows = [outs[s][r * bmax: r * bmax + (shard.bin_slice(b, r, world)[1] - shard.bin_slice(b, r, world)[0])] for r in range(world)]
                    ok = ok and bool((torch.cat(rows).numpy() == wants[s]).all())
        q.put((rank, ok, None))
    except Exception as e:  # report instead of hanging the parent on q.get
        import traceback
        q.put((rank, False, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,b,nslots", [(2, 5, 3), (3, 4, 2)])
def test_query_slots_own_their_communicators(world, b, nslots):
    """the slots' collectives may not share one communicator (ProcessGroupNCCL runs a group's collectives in
    issue order on one stream, so slot s + 1's distribution would wait for slot s's run).  shard.slot_groups gives every slot one
    group per direction; bench.py passes them to every QueryBroadcast / ResultGather it builds."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_slot_groups, args=(r, world, port, b, nslots, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    # bench.py's wiring: every ResultGather and QueryBroadcast of the timed loop is given a slot group
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    calls = [n for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute) and n.func.attr in ("ResultGather", "QueryBroadcast")]
    timed = [c for c in calls if any(k.arg == "group" for k in c.keywords)]
    probes = [c for c in calls if not any(k.arg == "group" for k in c.keywords)]
    assert len(timed) >= 2 and {c.func.attr for c in timed} == {"ResultGather", "QueryBroadcast"}
    # the only constructions without a group are the one-off probes that pick the collective kind before the warm-up
    assert all(any(isinstance(t, ast.Name) and t.id == "probe" for p_ in ast.walk(tree) if isinstance(p_, ast.Assign) and p_.value is c
                   for t in p_.targets) for c in probes)


def test_cpp_sharded_server_set_up_protocol_without_a_gpu(tmp_path):
    """host/ShardedBatchedFHEPSIServer.hpp's session set-up over its side sockets, executed on the CPU box (the worker
    side of the protocol had never run).  Rank 0 of two (tests/sharded_server_main.cpp) receives context, public key and EvalMult
    key from this process playing the client, and must forward unique id, context, table secrets and key to its worker -- this
    process again, reading the side socket -- with the channel's framing; a worker process receives the same four messages from
    this process playing rank 0.  Both then reach the first call that needs a device and end there with the library's 'no HIP
    device' error (the product has no CPU path): everything before it is host logic.  The unique id comes from the test-only RCCL
    stand-in (tests/fake_rccl), preloaded so that the 0.5 GB library is not mapped for 128 random bytes.  Malformed set-up
    messages are refused by size.  With a GPU present the same processes are run to completion by tests/test_rccl_ranks.py."""
    import socket
    import struct
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/test_rccl_ranks.py runs the whole protocol")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "nested_hashing_psi_amd")
    fake = str(tmp_path / "librccl.so.1")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", fake,
                           os.path.join(root, "tests", "fake_rccl", "fake_rccl.cpp"), "-Wl,-soname,librccl.so.1", "-L/opt/rocm/lib",
                           "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-lpthread"])
    exe = str(tmp_path / "sharded_server_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(root, "tests", "sharded_server_main.cpp"), "-L" + libdir,
                           "-lpiehip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, LD_PRELOAD=fake, PIEHIP_TEST_SEEDS="11,22,33")
    setfile = tmp_path / "set.bin"
    np.arange(1, 500, dtype=np.uint64).tofile(setfile)
    tail = [str(setfile), "3", "40", "2", "8", "7"]
    N, L, t = 4096, 2, 65537
    from nested_hashing_psi_amd import pie
    q, p = pie.default_moduli(N, L)
    moduli = np.zeros(15, dtype=np.uint64)
    moduli[:L], moduli[L:2 * L + 1] = q, p
    ctx = struct.pack("IIQ", N, L, t) + moduli.tobytes()
    rng = np.random.default_rng(1)
    evk = rng.integers(0, int(q[-1]), (L, 2, L, N), dtype=np.uint64)

    def send(s, payload):
        s.sendall(struct.pack("i", len(payload)) + payload)

    def recv(s):
        hdr = b""
        while len(hdr) < 4:
            c = s.recv(4 - len(hdr))
            assert c, "channel closed"
            hdr += c
        n, = struct.unpack("i", hdr)
        buf = bytearray()
        while len(buf) < n:
            c = s.recv(min(1 << 20, n - len(buf)))
            assert c, "channel closed"
            buf += c
        return bytes(buf)

    # ---- rank 0: the client's messages in, four set-up messages out to the worker
    client, c_srv = socket.socketpair()
    side, s_srv = socket.socketpair()
    proc = subprocess.Popen([exe, "0", "2", "0", str(c_srv.fileno()), str(s_srv.fileno())] + tail, pass_fds=(c_srv.fileno(), s_srv.fileno()),
                            env=env, stderr=subprocess.PIPE)
    c_srv.close()
    s_srv.close()
    send(client, ctx)
    send(client, b"")
    send(client, evk.tobytes())
    uid = recv(side)
    assert len(uid) == 128 and any(uid[:16]) and not any(uid[16:])     # the stand-in's id: 16 random bytes
    assert recv(side) == ctx
    assert recv(side) == struct.pack("QQQ", 11, 22, 33)                # evict, shuffle, mask: the same table on every rank
    assert recv(side) == evk.tobytes()
    _, err = proc.communicate(timeout=60)
    assert proc.returncode == 1 and b"no HIP device" in err, err
    client.close()
    side.close()
    # ---- a worker: the same four messages in
    for bad in (None, "id", "seeds", "key"):
        side, s_srv = socket.socketpair()
        proc = subprocess.Popen([exe, "1", "2", "0", "-1", str(s_srv.fileno())] + tail, pass_fds=(s_srv.fileno(),), env=env, stderr=subprocess.PIPE)
        s_srv.close()
        try:
            send(side, uid[:-1] if bad == "id" else uid)
            send(side, ctx)
            send(side, struct.pack("QQ", 1, 2) if bad == "seeds" else struct.pack("QQQ", 11, 22, 33))
            send(side, evk.tobytes()[8:] if bad == "key" else evk.tobytes())
        except (BrokenPipeError, ConnectionResetError):
            assert bad is not None      # the worker hung up on a malformed message
        _, err = proc.communicate(timeout=60)
        assert proc.returncode == 1
        want = {None: b"no HIP device", "id": b"unique id message size", "seeds": b"seed message size", "key": b"EvalMult key message size"}[bad]
        assert want in err, (bad, err)
        side.close()
