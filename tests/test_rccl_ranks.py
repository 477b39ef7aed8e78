"""The native collectives (csrc/piehip_rccl.cpp) and the C++ server of one process per GPU (host/ShardedBatchedFHEPSIServer.hpp) with
MORE THAN ONE RANK -- on the one GPU of the test box.

RCCL refuses two ranks on one device, so until round 5 the root's per-peer receives, a worker's send, the row arithmetic of a query
batch, uneven bin slices, a root other than rank 0 and the worker side of the server's set-up protocol had never executed anywhere.
libpiehip binds RCCL by name at run time; every process started here preloads the TEST-ONLY stand-in of tests/fake_rccl (built by
this file into a temporary directory; the same twelve entry points, stream-ordered transfers over Unix sockets, ncclCommAbort that
releases a blocked stream) -- so the multi-rank LOGIC of the product runs exactly as it would over xGMI, and every result is
compared with the ORACLE.  What this cannot show: RCCL itself (topology, bandwidth, CU occupancy next to the transforms); that
remains unmeasured until an N > 1 run on a multi-GPU node.

  test_native_collectives_many_ranks   piehip_rccl_init / _broadcast_query / piehip_run / piehip_gather_results_host / piehip_rccl_wait
                                       with 2, 4 and 5 processes, nq in {1, 3}, b = 14 over 4 and 5 ranks (3+4+3+4, 2+3+3+3+3), root != 0
  test_a_rank_that_never_joins         a rank leaves before the gather: the root's wait ends by time-out (silent exit) or at once
                                       (piehip_rccl_abort) with PIEHIP_EHIP -- nobody hangs; piehip_rccl_agree carries one rank's "no" to all
  test_cpp_server_many_ranks           tests/sharded_server_main.cpp (reference: BatchedFHEPSIServer.cpp:75-152) as 2, 4 and 5 processes:
                                       set-up forwarded over the side sockets, every rank builds its slice, query broadcast, gather;
                                       this process is the client; result ciphertexts bit-exact vs the oracle's run(); a rank whose
                                       offline phase fails ends the session on every rank
"""
import os
import socket
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "nested_hashing_psi_amd")
T16, T32 = 65537, 4296540161


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    """the stand-in library and the two native programs, compiled once per test session into a temporary directory"""
    d = tmp_path_factory.mktemp("rccl_ranks")
    fake = str(d / "librccl.so.1")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", fake,
                           os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp"), "-Wl,-soname,librccl.so.1", "-L/opt/rocm/lib",
                           "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-lpthread"])
    exes = {}
    for name in ("rccl_ranks_main", "sharded_server_main"):
        exes[name] = str(d / name)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exes[name], os.path.join(ROOT, "tests", name + ".cpp"), "-L" + LIBDIR,
                               "-lpiehip", "-Wl,-rpath," + LIBDIR, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, LD_PRELOAD=fake, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return exes, env


def _rand_limbs(rng, q, shape, N):
    out = np.zeros(shape + (len(q), N), dtype=np.uint64)
    for i, qi in enumerate(q):
        out[..., i, :] = rng.integers(0, int(qi), shape + (N,), dtype=np.uint64)
    return out


def _write_case(ob, d, N, L, t, K, E, b, nq, seed):
    o = ob.Oracle(N, L, t)
    rng = np.random.default_rng(seed)
    q = o.moduli[:L]
    db, masks, evk = _rand_limbs(rng, q, (K, b, E), N), _rand_limbs(rng, q, (b,), N), _rand_limbs(rng, q, (L, 2), N)
    queries = [(_rand_limbs(rng, q, (K, E, 2), N), _rand_limbs(rng, q, (2,), N)) for _ in range(nq)]
    db.tofile(d / "db.bin")
    masks.tofile(d / "masks.bin")
    evk.tofile(d / "evk.bin")
    for i, (idx, minus) in enumerate(queries):
        idx.tofile(d / ("idx%d.bin" % i))
        minus.tofile(d / ("minus%d.bin" % i))
    want = [o.pie_run(idx, minus, db, masks, evk) for idx, minus in queries]
    return want


def _launch(exe, env, G, root, N, L, t, K, E, b, nq, d, mode=""):
    procs = []
    for r in range(G):
        args = [exe, str(r), str(G), str(root), str(N), str(L), str(t), str(K), str(E), str(b), str(nq), str(d)] + ([mode] if mode else [])
        procs.append(subprocess.Popen(args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for pp in procs:
                pp.kill()
            raise
        outs.append((p.returncode, so.decode(), se.decode()))
    return outs


@pytest.mark.parametrize("G,root,N,L,t,K,E,b,nq", [
    (2, 0, 4096, 2, T16, 2, 4, 5, 1),       # 2 + 3 layers
    (2, 1, 4096, 2, T16, 2, 4, 5, 3),       # the root is not rank 0; a batch of three: rows [bin layer][query]
    (4, 2, 4096, 2, T16, 2, 3, 14, 3),      # b = 14 over four ranks: 3 + 4 + 3 + 4, root in the middle
    (5, 4, 4096, 2, T16, 3, 3, 14, 1),      # ... over five: 2 + 3 + 3 + 3 + 3, the last rank is the root; K = 3
    (4, 0, 16384, 4, T32, 2, 3, 14, 3),     # the headline ring (folded 2^13 slices, two queues on the larger slices)
    (3, 1, 8192, 3, T32, 2, 4, 4, 2),       # one slice per limb; 1 + 1 + 2 layers, a batch of two
])
def test_native_collectives_many_ranks(ob, built, tmp_path, G, root, N, L, t, K, E, b, nq):
    exes, env = built
    want = _write_case(ob, tmp_path, N, L, t, K, E, b, nq, 1000 * G + b + nq)
    outs = _launch(exes["rccl_ranks_main"], env, G, root, N, L, t, K, E, b, nq, tmp_path)
    for r, (rc, so, se) in enumerate(outs):
        assert rc == 0, "rank %d: %s %s" % (r, so, se)
    for rnd in range(2):
        got = np.fromfile(tmp_path / ("out%d.bin" % rnd), dtype=np.uint64).reshape(b, nq, 2, L, N)
        for i in range(nq):
            # round 1 staged query (i + 1) % nq in place i
            assert (got[:, i] == want[(i + rnd) % nq]).all(), "round %d, query %d of the batch" % (rnd, i)


@pytest.mark.parametrize("mode", ["skip", "abort", "agree"])
def test_a_rank_that_never_joins(ob, built, tmp_path, mode):
    """A collective completes when every rank has queued its side.  Rank 2 of 3 leaves before the second gather -- silently ("skip"):
    the root's piehip_rccl_wait gives up after its bound (4 s in the test program) and aborts the communicator; or after
    piehip_rccl_abort ("abort"): the root's wait ends at once with the communicator's error.  Either way the root returns
    PIEHIP_EHIP, refuses further collectives (no communicator) and exits -- nobody hangs.  "agree": one rank's no reaches all."""
    exes, env = built
    G, root, N, L, t, K, E, b, nq = 3, 0, 4096, 2, T16, 2, 3, 6, 1
    want = _write_case(ob, tmp_path, N, L, t, K, E, b, nq, 77)
    outs = _launch(exes["rccl_ranks_main"], env, G, root, N, L, t, K, E, b, nq, tmp_path, mode)
    got = np.fromfile(tmp_path / "out0.bin", dtype=np.uint64).reshape(b, nq, 2, L, N)
    assert (got[:, 0] == want[0]).all()     # the round before the failure is complete and right
    if mode == "agree":
        assert [o[0] for o in outs] == [5, 5, 5], outs
        assert all("agree -> 0" in o[1] for o in outs)
        return
    assert outs[2][0] == 7
    rc, so, se = outs[0]
    assert rc == 6, (so, se)
    assert "communicator aborted" in so and "gather after the abort -> -2" in so    # PIEHIP_ESTATE: no communicator
    ms = float(so.split("wait ended after ")[1].split(" ms")[0])
    if mode == "skip":
        assert "timed out" in so and 3500 < ms < 8000
    else:
        assert "reports" in so and ms < 3500
    assert outs[1][0] in (0, 6)     # the other worker sent its rows; its own wait either completed or saw the abort


def _client_session(pie, ob, a, cl, cc, N, L, t, K, E, b, clientset):
    def send(payload):
        a.sendall(struct.pack("i", len(payload)) + payload)

    def recv():
        hdr = b""
        while len(hdr) < 4:
            chunk = a.recv(4 - len(hdr))
            if not chunk:
                raise ConnectionError("server closed the channel")
            hdr += chunk
        n, = struct.unpack("i", hdr)
        buf = bytearray()
        while len(buf) < n:
            chunk = a.recv(min(1 << 20, n - len(buf)))
            if not chunk:
                raise ConnectionError("server closed the channel")
            buf += chunk
        return bytes(buf)

    def ct_msg(ct):
        return struct.pack("IIIIQ", 0x48454950, 1, L, N, 0) + np.ascontiguousarray(ct, dtype=np.uint64).tobytes()

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
    return evk, minus_ct, idx_ct, np.stack(res)


def _start_servers(exe, env, G, setfile, k, e, K, E, b, extra_env=None):
    a, bsock = socket.socketpair()
    sides = [socket.socketpair() for _ in range(G - 1)]
    env = dict(env, PIEHIP_TEST_SEEDS="1,2,3", PIEHIP_TEST_TIMEOUT_MS="20000", **(extra_env or {}))
    procs = []
    side0 = ",".join(str(s[0].fileno()) for s in sides) or "-"
    tail = [str(setfile), str(k), str(e), str(K), str(E), str(b)]
    procs.append(subprocess.Popen([exe, "0", str(G), "0", str(bsock.fileno()), side0] + tail,
                                  pass_fds=[bsock.fileno()] + [s[0].fileno() for s in sides], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    for r in range(1, G):
        procs.append(subprocess.Popen([exe, str(r), str(G), "0", "-1", str(sides[r - 1][1].fileno())] + tail,
                                      pass_fds=[sides[r - 1][1].fileno()], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    bsock.close()
    for s in sides:
        s[0].close()
        s[1].close()
    return a, procs


@pytest.mark.parametrize("G,shape", [(2, "small"), (4, "small"), (5, "C3"), (2, "C3")])
def test_cpp_server_many_ranks(ob, pie_mod, built, tmp_path, G, shape):
    from nested_hashing_psi_amd.client import BatchedFHEPSIClient
    pie = pie_mod
    exes, env = built
    rng = np.random.default_rng(100 + G)
    if shape == "small":
        N, L, t, k, e, K, E, b, nS, nC, ninter = 8192, 3, T32, 3, 40, 2, 8, 7, 2000, 64, 33
    else:
        N, L, t, k, e, K, E, b, nS, nC, ninter = 16384, 4, T32, 2, 4949, 2, 14, 14, 1 << 20, 1 << 10, 513
    items = np.unique(rng.integers(1, t, nS + nC + 8192, dtype=np.uint64))
    rng.shuffle(items)
    server = items[:nS].copy()
    clientset = np.concatenate([server[:ninter], items[nS:nS + nC - ninter]])
    rng.shuffle(clientset)
    setfile = tmp_path / "server_set.bin"
    server.astype(np.uint64).tofile(setfile)
    a, procs = _start_servers(exes["sharded_server_main"], env, G, setfile, k, e, K, E, b)
    cc = pie.PieContext(N, L, t)
    cl = BatchedFHEPSIClient(cc, k, e, K, E, b)
    evk, minus_ct, idx_ct, res = _client_session(pie, ob, a, cl, cc, N, L, t, K, E, b, clientset)
    for r, p in enumerate(procs):
        so, se = p.communicate(timeout=180)
        assert p.returncode == 0, "server rank %d: %s" % (r, se.decode())
        if r == 0:
            online_us = int([ln for ln in so.decode().splitlines() if ln.startswith("OnlineComputation,")][0].split(",")[1])
            print("C++ server, %d ranks on one GPU over the stand-in transport, %s shape: OnlineComputation %d us" % (G, shape, online_us))
    found = cl.extractIntersection(res)
    assert sorted(int(v) for v in found) == sorted(int(v) for v in server[:ninter])
    # ciphertext bits vs the oracle: the same table (hash seed of the server's default, the fixed test secrets 1, 2, 3)
    o = ob.Oracle(N, L, t)
    tab = ob.Tabulation(987654321, k + K)
    tbl = ob.hct_build(tab, server, k, e, K, b, E, evict_seed=1)
    ob.hct_shuffle_bins(tbl, 2)
    slots = ob.pack_db(tbl)
    mask_slots = ob.masks(t, b, k * e, 3)
    idx = np.ascontiguousarray(idx_ct, dtype=np.uint64).reshape(K, E, 2, L, N)
    minus = np.ascontiguousarray(minus_ct, dtype=np.uint64).reshape(2, L, N)
    evk = np.ascontiguousarray(evk, dtype=np.uint64).reshape(L, 2, L, N)
    import concurrent.futures

    def layer_ok(bn):
        db = np.stack([o.encode_eval(slots[h, bn, j]) for h in range(K) for j in range(E)]).reshape(K, 1, E, L, N)
        return bool((res[bn] == o.pie_run(idx, minus, db, o.encode_eval(mask_slots[bn])[None], evk)[0]).all())
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as pool:
        ok = list(pool.map(layer_ok, range(b)))
    assert all(ok), "bin layers %s differ from the oracle" % [i for i, v in enumerate(ok) if not v]
    a.close()
    cc.close()


def test_cpp_server_rank_fails_offline(ob, pie_mod, built, tmp_path):
    """rank 2 of 3 cannot build its slice: the ranks agree on that at the end of the offline phase (piehip_rccl_agree) and every
    process ends with an error -- the client sees its channel close; nobody waits in a collective for the failed rank."""
    from nested_hashing_psi_amd.client import BatchedFHEPSIClient
    pie = pie_mod
    exes, env = built
    N, L, t, k, e, K, E, b, nS = 8192, 3, T32, 3, 40, 2, 8, 7, 2000
    rng = np.random.default_rng(5)
    server = np.unique(rng.integers(1, t, nS + 100, dtype=np.uint64))[:nS]
    setfile = tmp_path / "server_set.bin"
    server.astype(np.uint64).tofile(setfile)
    a, procs = _start_servers(exes["sharded_server_main"], env, 3, setfile, k, e, K, E, b, {"PIEHIP_TEST_FAIL_OFFLINE_RANK": "2"})
    cc = pie.PieContext(N, L, t)
    cl = BatchedFHEPSIClient(cc, k, e, K, E, b)
    with pytest.raises((ConnectionError, AssertionError, BrokenPipeError)):
        _client_session(pie, ob, a, cl, cc, N, L, t, K, E, b, server[:64].copy())
    errs = []
    for p in procs:
        so, se = p.communicate(timeout=60)
        assert p.returncode == 1
        errs.append(se.decode())
    assert "offline phase failed on this rank" in errs[2]
    assert all("could not build its slice" in x for x in errs[:2]), errs
    a.close()
    cc.close()
