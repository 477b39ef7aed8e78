"""CPU-side checks of the drop-in boundary: libpiehip.so loads without a GPU and exports every
symbol include/piehip.h declares; parameter generation agrees with the oracle; no silent CPU path."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from nested_hashing_psi_amd import build
    return build()


def test_library_exports_every_declared_symbol(built):
    import ctypes
    from nested_hashing_psi_amd._lib import SYMBOLS
    hdr = open(os.path.join(ROOT, "include", "piehip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(piehip_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations found"
    lib = ctypes.CDLL(built)
    for name in declared:
        assert hasattr(lib, name), "libpiehip.so does not export %s" % name
    assert declared == set(SYMBOLS), "python binding and header disagree: %s" % (declared ^ set(SYMBOLS))


def test_default_moduli_match_oracle(built, ob):
    from nested_hashing_psi_amd import pie
    for N, L in [(4096, 2), (8192, 3), (16384, 4), (32768, 6)]:
        q, p = pie.default_moduli(N, L)
        oq, op_ = ob.default_moduli(N, L)
        assert (q == oq).all() and (p == op_).all()
    with pytest.raises(ValueError):
        pie.default_moduli(1000, 2)


def test_no_cpu_fallback(built):
    """without a GPU the product refuses to create a context instead of computing on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nested_hashing_psi_amd import pie
    with pytest.raises(RuntimeError, match="no HIP device"):
        pie.PieContext(4096, 2, 65537)


def test_product_does_not_touch_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/"""
    pkg = os.path.join(ROOT, "nested_hashing_psi_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                src = open(os.path.join(dirpath, f)).read()
                for line in src.splitlines():
                    code = line.split("//")[0].split("#")[0] if not f.endswith(".py") else line.split("#")[0]
                    assert "pie_oracle" not in code and "libpieoracle" not in code and "from oracle" not in code \
                        and "import oracle" not in code, "%s references the oracle: %s" % (f, line)


def test_cpp_facade_compiles_and_links(built, tmp_path):
    """nested_hashing_psi_amd/host/BatchedFHEHIPPIE.hpp (the reference class shape over the C ABI) builds
    with g++ against libpiehip.so; without a GPU it must fail loudly at context creation (rc 77)."""
    import subprocess
    exe = str(tmp_path / "facade_check")
    libdir = os.path.dirname(built)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "facade_check.cpp"),
                           "-L" + libdir, "-lpiehip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    rc = subprocess.call([exe])
    import torch
    assert rc == (0 if torch.cuda.is_available() else 77)


def test_tabulation_hash_matches_oracle(built, ob):
    """product-side TabulationHashing (std::mt19937 + uniform_int_distribution, as the reference) == oracle restatement"""
    from nested_hashing_psi_amd import pie
    rng = np.random.default_rng(3)
    x = rng.integers(0, 1 << 63, 500, dtype=np.uint64)
    x[0], x[1] = 0, np.uint64((1 << 64) - 1)
    for seed in (987654321, 12223222, 342797434736):
        tab = ob.Tabulation(seed, 4)
        for hf in range(4):
            got = pie.tabulation_hash(seed, 4, hf, x)
            assert [int(v) for v in got] == [tab.hash(int(v), hf) for v in x]


def test_client_cuckoo_table_matches_oracle(built, ob):
    """piehip_client_cuckoo_table (host side, no device) == the oracle's client table, on loads that force eviction walks,
    with duplicates, and the failure when the set cannot be placed"""
    import ctypes as C
    from nested_hashing_psi_amd._lib import lib, u64p
    from tests.test_oracle_pie import distinct_items
    rng = np.random.default_rng(9)
    for (k, e, n, K) in ((2, 20, 12, 2), (2, 64, 90, 2), (3, 40, 100, 2), (2, 4949, 1024, 2)):
        items = distinct_items(rng, 4296540161, n)
        items = np.concatenate([items, items[:3]])                 # duplicates are skipped
        tab = ob.Tabulation(987654321, k + K)
        want = ob.client_build(tab, items, k, e)
        got = np.zeros((k, e), dtype=np.uint64)
        rc = lib().piehip_client_cuckoo_table(987654321, k + K, k, e, items.ctypes.data_as(u64p), len(items), got.ctypes.data_as(u64p))
        assert rc == 0 and (got == want).all()
        assert np.count_nonzero(got) == n
    items = distinct_items(rng, 65537, 9)                          # 9 items into 2 x 4 positions
    got = np.zeros((2, 4), dtype=np.uint64)
    assert lib().piehip_client_cuckoo_table(1, 4, 2, 4, items.ctypes.data_as(u64p), len(items), got.ctypes.data_as(u64p)) == -5


def test_bench_algorithmic_bytes_follow_the_survey():
    """one run() of the reference schedule moves 253 + 14 * 54 + 35 = 1044 MiB at C3 (unfused limb passes)"""
    import bench
    cfg = bench.CONFIGS["C3"]
    assert bench.alg_bytes_run(cfg) == 1044 * 2**20
    N, L, K, E, b = cfg["N"], cfg["L"], cfg["K"], cfg["E"], cfg["b"]
    W = 8 * N
    assert (b * K * E * L + K * E * 2 * L + 2 * L + b * K * 2 * L) * W == 253 * 2**20      # stage A
    assert bench.alg_bytes_run(dict(cfg, b=1)) - bench.alg_bytes_run(dict(cfg, b=0)) > 54 * 2**20


def test_wire_framing_round_trip(tmp_path):
    """host/WireFraming.hpp: the reference channel's size-prefixed messages and empty phase-barrier message
    (BatchedFHEPSIServer.cpp:26,118,134,150; PSIServer.hpp:46-49) over a socket pair, online-phase message order"""
    import subprocess
    exe = str(tmp_path / "wire_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "wire_check.cpp")])
    assert subprocess.call([exe]) == 0


def test_cpp_server_harness_compiles(built, tmp_path):
    """host/BatchedFHEPSIServer.hpp (the caller of the hot path, reference phase order) builds warning-free against the C ABI;
    it runs in tests/test_gpu_parity.py::test_two_process_psi_over_the_wire"""
    import subprocess
    libdir = os.path.join(ROOT, "nested_hashing_psi_amd")
    exe = str(tmp_path / "server_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "server_main.cpp"),
                           "-L" + libdir, "-lpiehip", "-Wl,-rpath," + libdir])
    assert subprocess.call([exe]) == 2  # usage error: no arguments


def test_cpp_sharded_server_compiles(built, tmp_path):
    """host/ShardedBatchedFHEPSIServer.hpp -- one process per GPU, the query distribution and the final gather over RCCL behind the
    C ABI (piehip_rccl_*, piehip_gather_results*) -- builds warning-free against the library alone: no RCCL headers, no torch.
    It runs with one rank over RCCL in tests/test_sharding_gpu.py::test_cpp_server_over_rccl_one_rank and with 2-5 ranks over the test-only
    stand-in transport in tests/test_rccl_ranks.py."""
    import subprocess
    libdir = os.path.join(ROOT, "nested_hashing_psi_amd")
    exe = str(tmp_path / "sharded_server_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "sharded_server_main.cpp"),
                           "-L" + libdir, "-lpiehip", "-Wl,-rpath," + libdir])
    assert subprocess.call([exe]) == 2  # usage error: no arguments
    # the library does not LINK against RCCL (bound at run time: a one-GPU deployment never loads it)
    needed = subprocess.check_output(["readelf", "-d", os.path.join(libdir, "libpiehip.so")]).decode()
    assert "rccl" not in needed


def test_rccl_stand_in_is_test_infrastructure_only(built, tmp_path):
    """tests/fake_rccl (the transport that lets several ranks share the one GPU of the test box, tests/test_rccl_ranks.py) and the
    native multi-rank program compile here; the PACKAGE ships no RCCL of any kind -- the stand-in exists only as source under tests/
    and as a library in a test's temporary directory -- and nothing in the product names it."""
    import subprocess
    libdir = os.path.join(ROOT, "nested_hashing_psi_amd")
    fake = str(tmp_path / "librccl.so.1")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", fake,
                           os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp"), "-Wl,-soname,librccl.so.1", "-L/opt/rocm/lib",
                           "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-lpthread"])
    syms = subprocess.check_output(["nm", "-D", "--defined-only", fake]).decode()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommAbort", "ncclCommGetAsyncError", "ncclGroupStart",
                 "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclBroadcast", "ncclAllReduce", "ncclGetErrorString"):
        assert (" T " + name) in syms, name
    exe = str(tmp_path / "rccl_ranks_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "rccl_ranks_main.cpp"),
                           "-L" + libdir, "-lpiehip", "-Wl,-rpath," + libdir])
    assert subprocess.call([exe]) == 2
    for dirpath, _, files in os.walk(libdir):
        if os.sep + "build" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            assert "rccl" not in f.lower() or f == "piehip_rccl.cpp", "the package ships %s" % os.path.join(dirpath, f)
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", ".inc")) or f == "Makefile":
                assert "fake_rccl" not in open(os.path.join(dirpath, f), errors="replace").read(), os.path.join(dirpath, f)


def test_rccl_entry_points_check_their_arguments(built):
    """call-order / argument errors of the sharded entry points need neither a GPU nor RCCL"""
    import ctypes as C
    from nested_hashing_psi_amd import _lib
    L = _lib.lib()
    lo, hi = C.c_uint32(), C.c_uint32()
    assert L.piehip_rccl_bin_slice(14, 8, 3, C.byref(lo), C.byref(hi)) == 0 and (lo.value, hi.value) == (5, 7)
    cover = []
    for r in range(8):
        assert L.piehip_rccl_bin_slice(14, 8, r, C.byref(lo), C.byref(hi)) == 0
        cover += list(range(lo.value, hi.value))
    assert cover == list(range(14))
    from nested_hashing_psi_amd import shard
    assert all(shard.bin_slice(14, r, 8) == tuple(_slice(L, 14, 8, r)) for r in range(8))   # the same partition as the Python harness
    assert L.piehip_rccl_bin_slice(14, 8, 8, C.byref(lo), C.byref(hi)) == -1
    assert L.piehip_gather_results(None, 14, 0, None) == -1 and L.piehip_rccl_broadcast_query(None, 0) == -1
    ok = C.c_int()
    assert L.piehip_rccl_wait(None, 10) == -1 and L.piehip_rccl_abort(None) == -1 and L.piehip_rccl_agree(None, 1, C.byref(ok), 10) == -1


def _slice(L, b, G, r):
    import ctypes as C
    lo, hi = C.c_uint32(), C.c_uint32()
    assert L.piehip_rccl_bin_slice(b, G, r, C.byref(lo), C.byref(hi)) == 0
    return lo.value, hi.value


def test_client_parameter_selection_follows_the_reference():
    """BatchedFHEPSIClient.cpp:22-57: plaintext modulus by bit size, depth by inner table size, ring 16384"""
    import pytest
    from nested_hashing_psi_amd.client import select_parameters
    assert select_parameters(32, 14) == dict(N=16384, t=4296540161, depth=3, L=4)
    assert select_parameters(16, 499)["t"] == 65537 and select_parameters(16, 500)["depth"] == 5
    assert select_parameters(48, 5000)["depth"] == 10 and select_parameters(40, 1)["t"] == 1099579260929
    with pytest.raises(ValueError):
        select_parameters(24, 10)


def test_sharded_database_needs_explicit_seeds():
    """a bin-slice shard built with seeds of its own would be a slice of a different table than its peers': the
    mirror refuses missing evict / shuffle / mask seeds before anything reaches the library"""
    import types

    import numpy as np
    import pytest
    from nested_hashing_psi_amd import pie
    cc = types.SimpleNamespace(_h=None, L=2, N=64)
    items = np.arange(1, 40, dtype=np.uint64)
    base = dict(k=2, e=4, K=2, b=4, E=4)
    for missing in ("evict_seed", "shuffle_seed", "mask_seed"):
        hp = dict(base, evict_seed=1, shuffle_seed=2, mask_seed=3)
        del hp[missing]
        with pytest.raises(ValueError, match="identical on every shard"):
            pie.BatchedFHEHIPPIE(cc, serverSet=items, hashParams=hp, binSlice=(0, 2))
    with pytest.raises(ValueError, match="identical on every shard"):
        pie.BatchedFHEHIPPIE(cc, hashTable=np.zeros((2, 4, 2, 4, 4), dtype=np.uint64), shuffle_seed=5, binSlice=(0, 2))
