"""GPU parity: libpiehip.so (through its C ABI) against the CPU oracle, bit for bit.

Everything here is integer work, so the bar is exact equality of every limb word.  The oracle is
test infrastructure (oracle/); the product path under test is nested_hashing_psi_amd -> libpiehip.so.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T16 = 65537
T32 = 4296540161


@pytest.fixture(scope="module")
def pie():
    from nested_hashing_psi_amd import pie as p
    return p


def rand_limbs(rng, moduli, shape_prefix, N):
    """uniform residues: array [*shape_prefix][len(moduli)][N]"""
    out = np.zeros(tuple(shape_prefix) + (len(moduli), N), dtype=np.uint64)
    for i, m in enumerate(moduli):
        out[..., i, :] = rng.integers(0, int(m), tuple(shape_prefix) + (N,), dtype=np.uint64)
    return out


@pytest.mark.parametrize("N,L,t", [(64, 2, T16), (1024, 3, T32), (4096, 2, T16), (8192, 3, T32), (16384, 4, T32), (32768, 6, T32)])
def test_tables_match_oracle(ob, pie, N, L, t):
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    assert (cc.moduli == o.moduli).all()
    for mi in range(2 * L + 2):
        assert cc.psi(mi) == o.psi(mi)
    for mi in (0, 2 * L, 2 * L + 1):
        f1, i1 = cc.twiddles(mi)
        f2, i2 = o.twiddles(mi)
        assert (f1[1:] == f2[1:]).all() and (i1[1:] == i2[1:]).all()
    assert (cc.slot_positions() == o.slot_positions()).all()
    cc.close()


@pytest.mark.parametrize("N,L,t", [(64, 2, T16), (256, 1, T16), (1024, 3, T32), (4096, 2, T16), (8192, 3, T32), (16384, 4, T32),
                                   (32768, 6, T32)])
def test_ntt_bit_exact(ob, pie, N, L, t):
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(N + L)
    M = 2 * L + 1
    # a batch over QP (moduli cycle through all 2L+1), ragged batch count
    x = rand_limbs(rng, o.moduli[:M], (3,), N)
    x[0, 0, :] = 0                      # all-zero limb
    x[0, 1, :] = o.moduli[1] - np.uint64(1)  # all q-1
    f = cc.ntt(x, 0, M)
    want = np.stack([np.stack([o.ntt(mi, x[k, mi]) for mi in range(M)]) for k in range(3)])
    assert (f == want).all()
    back = cc.ntt(f, 0, M, inverse=True)
    assert (back == x).all()
    # Q-only batch and the plaintext modulus
    y = rand_limbs(rng, o.moduli[:L], (5,), N)
    assert (cc.ntt(y, 0, L) == np.stack([np.stack([o.ntt(mi, y[k, mi]) for mi in range(L)]) for k in range(5)])).all()
    z = rng.integers(0, t, (2, N), dtype=np.uint64)
    assert (cc.ntt(z, M, 1, inverse=True) == np.stack([o.intt(M, z[k]) for k in range(2)])).all()
    cc.close()


@pytest.mark.parametrize("N,L,t", [(64, 1, T16), (256, 2, T16), (1024, 3, T32), (4096, 4, T32), (2048, 6, T32)])
def test_base_conversions_bit_exact(ob, pie, N, L, t):
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(17 * N + L)
    M = 2 * L + 1
    xq = rand_limbs(rng, o.moduli[:L], (4,), N)
    xq[0, :, 0] = 0
    xq[0, :, 1] = o.q - np.uint64(1)
    got = cc.base_convert(0, xq)
    assert (got == np.stack([o.expand_q_to_qp(xq[k]) for k in range(4)])).all()
    got = cc.base_convert(1, xq)
    assert (got == np.stack([o.scale_pq_expand(xq[k]) for k in range(4)])).all()
    xqp = rand_limbs(rng, o.moduli[:M], (6,), N)
    got = cc.base_convert(2, xqp)
    assert (got == np.stack([o.scale_round_tp(xqp[k]) for k in range(6)])).all()
    cc.close()


@pytest.mark.parametrize("N,L,t", [(64, 2, T16), (4096, 2, T16), (8192, 3, T32), (16384, 4, T32), (32768, 6, T32)])
def test_eval_ops_bit_exact(ob, pie, N, L, t):
    """EvalAdd / EvalMult(ct,pt) / EvalMult(ct,ct) with and without relinearisation, batched"""
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(N)
    sk = o.keygen(1)
    evk = o.relin_keygen(sk, 2)
    cc.load_relin_key(evk)
    nct = 3 if N <= 8192 else 2
    lim = 150
    xs = [rng.integers(-lim, lim, min(N, 64)) for _ in range(nct)]
    ys = [rng.integers(-lim, lim, min(N, 64)) for _ in range(nct)]
    cx = np.stack([o.encrypt_slots(sk, v, 10 + i) for i, v in enumerate(xs)])
    cy = np.stack([o.encrypt_slots(sk, v, 20 + i) for i, v in enumerate(ys)])
    pt = o.encode_eval(ys[0])
    assert (cc.EvalAdd(cx[0], cy[0]) == o.add(cx[0], cy[0])).all()
    assert (cc.EvalMultPlain(cx[0], pt) == o.mul_plain(cx[0], pt)).all()
    t3 = cc.EvalMult(cx, cy, relin=False)
    assert (t3 == np.stack([o.mul_tensor(cx[i], cy[i]) for i in range(nct)])).all()
    r2 = cc.EvalMult(cx, cy, relin=True)
    assert (r2 == np.stack([o.mul(cx[i], cy[i], evk) for i in range(nct)])).all()
    dec, budget = o.decrypt_slots(sk, r2[1], len(xs[1]))
    assert budget > 0 and (dec == xs[1] * ys[1]).all()
    cc.close()


@pytest.mark.parametrize("N,L,t", [(64, 2, T16), (4096, 2, T16), (16384, 4, T32)])
def test_encode_bit_exact(ob, pie, N, L, t):
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(5)
    for B in (1, 7, N // 2, N):
        s = rng.integers(-(t // 2), t // 2, (3, B), dtype=np.int64)
        s[0, 0] = 0
        got = cc.MakePackedPlaintext(s)
        assert (got == np.stack([o.encode_eval(s[k]) for k in range(3)])).all()
    with pytest.raises(ValueError):
        cc.MakePackedPlaintext(np.array([t], dtype=np.int64))
    cc.close()


def test_automorphism_kat1(ob, pie):
    """tests/TestOpenFHE.cpp:36,62-65: rotate [1..12] by +-1, +-2 (A9; not on the batched path)"""
    N, L, t = 4096, 2, T16
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    sk = o.keygen(3)
    v1 = np.arange(1, 13)
    c1 = o.encrypt_slots(sk, v1, 1)
    for r in (1, 2, -1, -2):
        g = o.rot_index(r)
        rk = o.rot_keygen(sk, g, 50 + r)
        got = cc.EvalAutomorphism(c1, g, rk)
        assert (got == o.automorph(c1, g, rk)).all()
        dec, budget = o.decrypt_slots(sk, got, 12)
        full = np.zeros(N // 2, dtype=np.int64)
        full[:12] = v1
        assert budget > 0 and (dec == np.roll(full, -r)[:12]).all()
    with pytest.raises(ValueError):
        cc.EvalAutomorphism(c1, 4, rk)
    cc.close()


def _query(ob, o, rng, nS, nC, k, e, K, E, b, hash_seed=987654321):
    from tests.test_oracle_pie import distinct_items
    universe = distinct_items(rng, o.t, nS + nC)
    server = universe[:nS].copy()
    ninter = nC // 2 + 1
    client = np.concatenate([server[:ninter], universe[nS:nS + nC - ninter]])
    rng.shuffle(client)
    tab = ob.Tabulation(hash_seed, k + K)
    tbl = ob.hct_build(tab, server, k, e, K, b, E, evict_seed=1)
    ob.hct_shuffle_bins(tbl, 2)
    slots = ob.pack_db(tbl)
    mk = ob.masks(o.t, b, k * e, 3)
    ctab = ob.client_build(tab, client, k, e, evict_seed=4)
    index, minus = ob.client_vectors(tab, ctab, K, E)
    return dict(server=server, client=client, inter=server[:ninter], slots=slots, mask_slots=mk, ctab=ctab, index=index,
                minus=minus)


@pytest.mark.parametrize("N,L,t,nS,nC,k,e,K,E,b", [
    (4096, 2, T16, 100, 1, 2, 1, 2, 10, 20),      # KAT-0 shape (TestBatchedFHEPIE.cpp:89-94)
    (4096, 2, T16, 300, 16, 2, 12, 2, 6, 6),
    (2048, 3, T32, 500, 24, 3, 10, 2, 8, 5),
    (1024, 4, T32, 200, 10, 2, 12, 3, 6, 4),      # K = 3: chained ct x ct
    (8192, 3, T32, 2000, 64, 3, 40, 2, 8, 7),
    (1024, 2, T16, 12, 4, 2, 3, 2, 5, 1),         # a single bin layer
])
def test_run_bit_exact_and_semantics(ob, pie, N, L, t, nS, nC, k, e, K, E, b):
    """BatchedFHEHIPPIE::run() on the GPU == the oracle's restated run(), and the decrypted result
    is the intersection (PSIClient.hpp:142-164)"""
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(N + nS)
    sk = o.keygen(11)
    evk = o.relin_keygen(sk, 12)
    d = _query(ob, o, rng, nS, nC, k, e, K, E, b)
    B = k * e
    db = np.stack([o.encode_eval(d["slots"][h, bn, j]) for h in range(K) for bn in range(b) for j in range(E)]).reshape(K, b, E, L, N)
    masks = np.stack([o.encode_eval(d["mask_slots"][bn]) for bn in range(b)])
    idx = np.stack([o.encrypt_slots(sk, d["index"][h, j], 100 + h * E + j) for h in range(K) for j in range(E)]).reshape(K, E, 2, L, N)
    minus = o.encrypt_slots(sk, d["minus"], 99)
    cc.load_relin_key(evk)
    # (a) database handed over as EVALUATION limbs
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    op.setMinusCompareElement(minus)
    op.setIndex(idx)
    op.run()
    got = op.getResultList()
    want = o.pie_run(idx, minus, db, masks, evk)
    assert (got == want).all()
    # (b) database handed over as raw slot values, encoded on the device
    op2 = pie.BatchedFHEHIPPIE(cc, slots=d["slots"], mask_slots=d["mask_slots"])
    op2.setMinusCompareElement(minus)
    op2.setIndex(idx)
    op2.run()
    assert (op2.getResultList() == want).all()
    # run() is repeatable (same inputs -> same ciphertexts), as the server calls it once per query
    op2.run()
    assert (op2.getResultList() == want).all()
    dec = np.stack([o.decrypt_slots(sk, got[bn], B)[0] for bn in range(b)])
    found = ob.client_scan(d["ctab"], dec)
    assert sorted(int(v) for v in found) == sorted(int(v) for v in d["inter"])
    cc.close()


def test_error_behaviour(ob, pie):
    """argument checks of the reference: BatchedFHEHIPPIE.cpp:13-21 (invalid_argument), call order"""
    cc = pie.PieContext(1024, 2, T16)
    with pytest.raises(ValueError, match="stash"):
        pie.BatchedFHEHIPPIE(cc, slots=np.zeros((2, 1, 1, 4), dtype=np.int64), mask_slots=np.ones((1, 4), dtype=np.int64), serverStashSize=1)
    with pytest.raises(ValueError, match="combined tables"):
        pie.BatchedFHEHIPPIE(cc, slots=np.zeros((2, 1, 1, 4), dtype=np.int64), mask_slots=np.ones((1, 4), dtype=np.int64), cuckooMultiTables=False)
    op = pie.BatchedFHEHIPPIE(cc, slots=np.zeros((2, 1, 1, 4), dtype=np.int64), mask_slots=np.ones((1, 4), dtype=np.int64))
    with pytest.raises(RuntimeError):
        op.run()  # no key, no inputs
    with pytest.raises(ValueError):
        pie.PieContext(1000, 2, T16)  # N not a power of two
    with pytest.raises(ValueError):
        pie.PieContext(1024, 2, 65539)  # t not 1 mod 2N / not prime
    cc.close()


def test_cpp_facade_runs_reference_call_order(tmp_path):
    """the C++ facade (reference class shape) drives setMinusCompareElement / setIndex / run / getResultList"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "nested_hashing_psi_amd")
    exe = str(tmp_path / "facade_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(root, "tests", "facade_check.cpp"),
                           "-L" + libdir, "-lpiehip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    assert subprocess.call([exe]) == 0


def test_several_handles_and_sharded_facade_in_one_process(tmp_path):
    """tests/sharded_check.cpp: handles of different rings and moduli side by side; host/ShardedBatchedFHEHIPPIE.hpp over three
    handles equals the single-handle operator bit for bit"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "nested_hashing_psi_amd")
    exe = str(tmp_path / "sharded_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(root, "tests", "sharded_check.cpp"),
                           "-L" + libdir, "-lpiehip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    assert subprocess.call([exe]) == 0


@pytest.mark.parametrize("N,L,t,nS,k,e,K,E,b", [
    (4096, 2, T16, 300, 2, 12, 2, 6, 6),
    (2048, 3, T32, 5000, 3, 100, 2, 10, 6),
    (16384, 4, T32, 1 << 17, 2, 1000, 2, 14, 8),
    (1024, 2, T16, 40, 2, 3, 3, 5, 2),
    (1024, 2, T16, 600, 2, 4, 2, 4, 70),       # more bin layers than lanes: the column scan takes two steps
    (1024, 2, T16, 3000, 2, 2, 3, 50, 60),     # 72 KiB per inner table: the one-thread-per-table kernel
])
def test_offline_phase_on_device(ob, pie, N, L, t, nS, k, e, K, E, b):
    """piehip_build_db == oracle ph_hct_build + ph_hct_shuffle_bins + ph_pack_db + ph_masks + encode, bit for bit:
    the hash table itself and the resulting run() output"""
    from tests.test_oracle_pie import distinct_items
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(nS)
    server = distinct_items(rng, t, nS)
    seeds = dict(hash_seed=987654321, evict_seed=11, shuffle_seed=22, mask_seed=33)
    op = pie.BatchedFHEHIPPIE(cc, serverSet=server, hashParams=dict(k=k, e=e, K=K, b=b, E=E, **seeds))
    tab = ob.Tabulation(seeds["hash_seed"], k + K)
    tbl = ob.hct_build(tab, server, k, e, K, b, E, evict_seed=seeds["evict_seed"])
    ob.hct_shuffle_bins(tbl, seeds["shuffle_seed"])
    assert (op.hashTable() == tbl).all()
    # every server item sits in each of the k outer tables exactly once
    assert int((tbl != 0).sum()) == k * nS
    slots = ob.pack_db(tbl)
    mask_slots = ob.masks(t, b, k * e, seeds["mask_seed"])
    sk = o.keygen(1)
    evk = o.relin_keygen(sk, 2)
    cc.load_relin_key(evk)
    B = k * e
    rng2 = np.random.default_rng(5)
    idx = np.stack([o.encrypt_slots(sk, rng2.integers(0, 2, B), 10 + i) for i in range(K * E)]).reshape(K, E, 2, L, N)
    minus = o.encrypt_slots(sk, -rng2.integers(1, 1000, B), 9)
    op.setMinusCompareElement(minus)
    op.setIndex(idx)
    op.run()
    got = op.getResultList()
    op2 = pie.BatchedFHEHIPPIE(cc, slots=slots, mask_slots=mask_slots)
    op2.setMinusCompareElement(minus)
    op2.setIndex(idx)
    op2.run()
    assert (got == op2.getResultList()).all()
    # constructor-only path: a table built elsewhere (here: by the oracle, unshuffled) handed to the device
    tbl0 = ob.hct_build(tab, server, k, e, K, b, E, evict_seed=seeds["evict_seed"])
    op3 = pie.BatchedFHEHIPPIE(cc, hashTable=tbl0, shuffle_seed=seeds["shuffle_seed"], mask_seed=seeds["mask_seed"])
    assert (op3.hashTable() == tbl).all()
    op3.setMinusCompareElement(minus)
    op3.setIndex(idx)
    op3.run()
    assert (got == op3.getResultList()).all()
    cc.close()


def test_offline_phase_errors(ob, pie):
    cc = pie.PieContext(1024, 2, T16)
    items = np.arange(1, 400, dtype=np.uint64)
    with pytest.raises(RuntimeError, match="Cuckoo"):   # 399 items cannot fit 2*2 tables of 2*1*3 cells
        pie.BatchedFHEHIPPIE(cc, serverSet=items, hashParams=dict(k=2, e=2, K=2, b=1, E=3))
    with pytest.raises(ValueError, match="plaintext modulus"):
        pie.BatchedFHEHIPPIE(cc, serverSet=np.array([T16 + 5], dtype=np.uint64), hashParams=dict(k=2, e=2, K=2, b=2, E=3))
    with pytest.raises(ValueError, match="ring dimension"):
        pie.BatchedFHEHIPPIE(cc, serverSet=items, hashParams=dict(k=2, e=600, K=2, b=2, E=3))
    cc.close()


@pytest.mark.parametrize("N,L,t", [(1024, 2, T16), (4096, 2, T16), (16384, 4, T32)])
def test_client_harness_matches_oracle(ob, pie, N, L, t):
    """piehip_client_* (keygen, relin keygen, secret-key encryption, decryption) == the oracle, bit for bit"""
    from nested_hashing_psi_amd.client import BatchedFHEPSIClient
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    k, e, K, E, b = 2, 20, 2, 5, 3
    cl = BatchedFHEPSIClient(cc, k, e, K, E, b)
    evk = cl.runSetUpPhase(keySeed=11, evalKeySeed=12)
    sk = o.keygen(11)
    assert (cl.sk == sk).all()
    assert (evk == o.relin_keygen(sk, 12)).all()
    rng = np.random.default_rng(4)
    from tests.test_oracle_pie import distinct_items
    client = distinct_items(rng, t, 12)
    minus_ct, idx_ct = cl.runOfflinePhase(client, encSeedBase=100)
    tab = ob.Tabulation(987654321, k + K)
    ctab = ob.client_build(tab, client, k, e)
    assert (cl.clientTable == ctab).all()
    index, minus = ob.client_vectors(tab, ctab, K, E)
    assert (cl.plainIndex == index).all() and (cl.plainMinus == minus).all()
    assert (minus_ct == o.encrypt_slots(sk, minus, 99)).all()
    for h in range(K):
        for j in range(E):
            assert (idx_ct[h, j] == o.encrypt_slots(sk, index[h, j], 100 + h * E + j)).all()
    # decryption, including a product (noisy) ciphertext
    prod = o.mul(minus_ct, idx_ct[0, 0], evk)
    got = cl.decrypt(np.stack([minus_ct, prod]))
    assert (got[0] == o.decrypt_slots(sk, minus_ct, k * e)[0]).all()
    assert (got[1] == o.decrypt_slots(sk, prod, k * e)[0]).all()
    cc.close()


def test_end_to_end_psi_all_on_device(ob, pie):
    """server offline phase, client harness and the hot path, no oracle in the loop: the computed
    intersection must be a permutation of the true one (PSIClient.hpp:142-164)"""
    from nested_hashing_psi_amd.client import BatchedFHEPSIClient
    from tests.test_oracle_pie import distinct_items
    N, L, t = 8192, 3, T32
    k, e, K, E, b = 3, 443, 2, 12, 12          # Parameters1.txt:53 (config C2)
    rng = np.random.default_rng(17)
    items = distinct_items(rng, t, (1 << 16) + 1024)
    server = items[: 1 << 16]
    inter = server[:513]
    clientset = np.concatenate([inter, items[1 << 16: (1 << 16) + 511]])
    rng.shuffle(clientset)
    cc = pie.PieContext(N, L, t)
    cl = BatchedFHEPSIClient(cc, k, e, K, E, b)
    cc.load_relin_key(cl.runSetUpPhase())
    srv = pie.BatchedFHEHIPPIE(cc, serverSet=server, hashParams=dict(k=k, e=e, K=K, b=b, E=E))
    minus_ct, idx_ct = cl.runOfflinePhase(clientset)
    srv.setMinusCompareElement(minus_ct)
    srv.setIndex(idx_ct)
    srv.run()
    found = cl.extractIntersection(srv.getResultList())
    assert sorted(int(v) for v in found) == sorted(int(v) for v in inter)
    cc.close()


# ---- rotation-based sibling operator (FHEHIPPIE) ----------------------------------------------
@pytest.mark.parametrize("N,L,t,K,E", [(2048, 3, T16, 3, 12), (4096, 3, T32, 2, 10), (16384, 4, T32, 2, 5)])
def test_fhepie_bit_exact_and_semantics(ob, pie, N, L, t, K, E):
    """piehip_fhepie_run == the oracle's restatement of FHEHIPPIE::run (FHEHIPPIE.cpp:61-77), bit for bit; one
    zero slot among the K results when the element is in the table (tests/TestFHEPIE.cpp:125-137)"""
    from tests.test_oracle_pie import fhepie_case
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    sk = o.keygen(1)
    c = fhepie_case(ob, o, sk, t, K, E, 10 * E, True)
    for r in c["keys"]:
        assert cc.rotation_galois(r) == o.rot_index(r)
    cc.load_rotation_keys(c["keys"])
    op = pie.FHEHIPPIE(cc, c["tbl"], perm_seed=False, masks=c["masks"])
    assert (op.slots[0] == c["slots"]).all()
    op.setIndex(c["idx"])
    op.run()
    got = op.getResultList()
    want = ob.fhe_pie_run(o, c["idx"], c["slots"], c["masks"], c["keys"])
    assert (got == want).all()
    zeros = sum(int((o.decrypt_slots(sk, got[hf], E)[0] == 0).sum()) for hf in range(K))
    assert zeros == 1
    cc.close()


def test_fhepie_collection_with_client_harness(ob, pie):
    """a collection of operators (one per client slot, PIECollection.hpp) in one batch, keys and ciphertexts from the
    product's client harness, bin and result permutations on: exactly the present elements produce a zero slot"""
    from nested_hashing_psi_amd.client import BatchedFHEPSIClient
    from tests.test_oracle_pie import distinct_items
    N, L, t, K, E, npie = 4096, 3, T32, 2, 10, 4
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    cl = BatchedFHEPSIClient(cc, 1, 1, K, E, E)
    cl.runSetUpPhase(keySeed=21, evalKeySeed=22)
    keys = cl.rotationKeyGen(E, seedBase=50)
    sk = o.keygen(21)
    R = int(np.ceil(np.log2(E)))
    rots = [1 << r for r in range(R)] + [-i for i in range(1, E)]
    for i, r in enumerate(rots):  # the harness' keys are the oracle's keys
        assert (keys[r] == o.rot_keygen(sk, o.rot_index(r), 50 + i)).all()
    cc.load_rotation_keys(keys)
    rng = np.random.default_rng(77)
    tab = ob.Tabulation(987654321, K + 1)
    tables, index, present = [], [], [True, False, True, False]
    for i in range(npie):
        items = distinct_items(rng, t, 61)
        tables.append(ob.hct_build(tab, items[:60], 1, 1, K, E, E, evict_seed=3 + i)[0, 0])
        x = int(items[7 + i]) if present[i] else int(items[60])
        index.append(ob.fhe_pie_index_vectors(tab, x, 1, K, E))
    index = np.stack(index)
    idx = cl._encrypt(index.reshape(npie * K, E + 1), 300 + np.arange(npie * K)).reshape(npie, K, 2, L, N)
    op = pie.FHEHIPPIE(cc, np.stack(tables), perm_seed=5, mask_seed=6)
    op.setIndex(idx)
    op.run()
    res = op.getResultList()
    dec = cl.decrypt(res.reshape(npie * K, 2, L, N), nslots=E).reshape(npie, K, E)
    for i in range(npie):
        assert int((dec[i] == 0).sum()) == (1 if present[i] else 0)
    # the oracle agrees bit for bit on one operator of the batch (undo the result permutation)
    want = ob.fhe_pie_run(o, idx[2], op.slots[2], op.masks[2], keys)
    assert (res[2][op.permutationVector[2]] == want).all()
    cc.close()


def test_fhepie_error_behaviour(ob, pie):
    cc = pie.PieContext(1024, 2, T16)
    with pytest.raises(ValueError, match="cuckoo bin"):       # FHEHIPPIE.cpp:13-16
        pie.FHEHIPPIE(cc, np.ones((2, 4, 5), dtype=np.uint64))
    with pytest.raises(ValueError, match="stash"):            # FHEHIPPIE.cpp:17-20
        pie.FHEHIPPIE(cc, np.ones((2, 4, 4), dtype=np.uint64), stashSize=1)
    op = pie.FHEHIPPIE(cc, np.ones((2, 4, 4), dtype=np.uint64))
    op.setIndex(np.zeros((2, 2, 2, 1024), dtype=np.uint64))
    with pytest.raises(RuntimeError, match="key for rotation"):
        op.run()
    with pytest.raises(ValueError):
        op.setIndex(np.zeros((3, 2, 2, 1024), dtype=np.uint64))
    cc.close()


# ---- stream order of run() (bin layers on the handle's own queues, lazy joins) -------------------------------
@pytest.mark.parametrize("streams", [0, 1, 3])
def test_run_pipelining_and_result_buffers(ob, pie, streams):
    """back-to-back run() calls without a host wait, a change of inputs between runs, and piehip_run_into a caller-owned
    buffer all give the oracle's ciphertexts, whatever the number of queues"""
    import torch
    N, L, t, nS, nC, k, e, K, E, b = 8192, 3, T32, 1500, 40, 3, 30, 2, 8, 7
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    cc.set_run_streams(streams)
    rng = np.random.default_rng(99)
    sk = o.keygen(11)
    evk = o.relin_keygen(sk, 12)
    d = _query(ob, o, rng, nS, nC, k, e, K, E, b)
    db = np.stack([o.encode_eval(d["slots"][h, bn, j]) for h in range(K) for bn in range(b) for j in range(E)]).reshape(K, b, E, L, N)
    masks = np.stack([o.encode_eval(d["mask_slots"][bn]) for bn in range(b)])
    idx = np.stack([o.encrypt_slots(sk, d["index"][h, j], 100 + h * E + j) for h in range(K) for j in range(E)]).reshape(K, E, 2, L, N)
    minus = o.encrypt_slots(sk, d["minus"], 99)
    idx2 = np.ascontiguousarray(idx[:, ::-1])          # a different query: index rows permuted
    cc.load_relin_key(evk)
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    op.setMinusCompareElement(minus)
    op.setIndex(idx)
    want = o.pie_run(idx, minus, db, masks, evk)
    want2 = o.pie_run(idx2, minus, db, masks, evk)
    for _ in range(5):
        op.run(sync=False)                              # pipelined
    assert (op.getResultList() == want).all()
    op.run(sync=False)
    op.setIndex(idx2)                                   # joins, uploads on the handle's stream, marks the inputs dirty
    op.run(sync=False)
    op.run(sync=False)
    assert (op.getResultList() == want2).all()
    # caller-owned result buffers, alternating, read back through the handle's stream order (join + torch copy)
    bufs = [torch.zeros((b, 2, L, N), dtype=torch.int64, device="cuda") for _ in range(2)]
    op.setIndex(idx)
    for i in range(4):
        op.run(sync=False, into=bufs[i & 1].data_ptr())
    op.sync()
    for bf in bufs:
        assert (bf.cpu().numpy().view(np.uint64) == want).all()
    cc.close()


# ---- caller-supplied moduli: the paths that the default 60-bit chain never takes ---------------------------------
@pytest.mark.parametrize("streams", [0, 1])
def test_run_as_captured_graph(ob, pie, streams):
    """piehip_set_graph: the replayed graph computes what the eager launches compute; it is re-captured when the inputs, the
    result buffer or the queue count change, and survives a database reload of another shape"""
    N, L, t, K, E, b = 4096, 2, T16, 2, 4, 5
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(77 + streams)
    evk = rand_limbs(rng, cc.q, (L, 2), N)
    cc.load_relin_key(evk)
    cc.set_run_streams(streams)
    cc.set_graph(True)
    for bb in (b, 3):
        db, masks = rand_limbs(rng, cc.q, (K, bb, E), N), rand_limbs(rng, cc.q, (bb,), N)
        op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
        for q in range(3):
            idx, minus = rand_limbs(rng, cc.q, (K, E, 2), N), rand_limbs(rng, cc.q, (2,), N)
            op.setMinusCompareElement(minus)
            op.setIndex(idx)
            want = o.pie_run(idx, minus, db, masks, evk)
            op.run()
            assert (op.getResultList() == want).all()
            op.run(sync=False)          # replay, back to back
            op.run()
            assert (op.getResultList() == want).all()
        cc.set_run_streams(1 - streams if streams else 1)   # another queue count: re-capture
        op.run()
        assert (op.getResultList() == want).all()
        cc.set_run_streams(streams)
    cc.set_graph(False)
    op.run()
    assert (op.getResultList() == want).all()
    cc.close()


@pytest.mark.parametrize("N,L,K,E,b,depth", [(4096, 2, 2, 4, 5, 2), (16384, 4, 2, 3, 9, 3)])
def test_query_slots_on_one_database(ob, pie, N, L, K, E, b, depth):
    """piehip_attach_database: further query slots (own context, stream, workspace) read the owner's key and database by
    reference; different queries run on the slots at the same time and each gets the oracle's result; a slot that is closed
    leaves the owner intact; loading a key into an attached slot is refused; a slot given its own database detaches"""
    import torch
    t = T16 if N == 4096 else T32
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(N + depth)
    db, masks, evk = rand_limbs(rng, cc.q, (K, b, E), N), rand_limbs(rng, cc.q, (b,), N), rand_limbs(rng, cc.q, (L, 2), N)
    cc.load_relin_key(evk)
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    streams = [torch.cuda.Stream() for _ in range(depth - 1)]
    it = iter(streams)
    pipe = pie.QueryPipeline(op, depth, lambda: pie.PieContext(N, L, t, stream=next(it).cuda_stream))
    for rnd in range(2):
        queries = [(rand_limbs(rng, cc.q, (K, E, 2), N), rand_limbs(rng, cc.q, (2,), N)) for _ in range(depth)]
        for s, (idx, minus) in zip(pipe.slots, queries):
            s.setMinusCompareElement(minus)
            s.setIndex(idx)
        for _ in range(3):  # several rounds in flight, all slots at once
            pipe.run_all()
        pipe.sync()
        for s, (idx, minus) in zip(pipe.slots, queries):
            assert (s.getResultList() == o.pie_run(idx, minus, db, masks, evk)).all()
    # the host-memory form of the same: each slot's query uploads, evaluates and downloads on its own queues
    # (piehip_run_host_async / _wait), page-locked staging, all slots queued before the first wait
    bufs = [s.hostBuffers() for s in pipe.slots]
    for rnd in range(2):
        queries = [(rand_limbs(rng, cc.q, (K, E, 2), N), rand_limbs(rng, cc.q, (2,), N)) for _ in range(depth)]
        for s, (pi, pm, pr), (idx, minus) in zip(pipe.slots, bufs, queries):
            pi[...] = idx
            pm[...] = minus
            pr[...] = 0
            s.runHostAsync(pi, pm, pr)
        for s, (pi, pm, pr), (idx, minus) in zip(pipe.slots, bufs, queries):
            s.waitHost()
            assert (pr == o.pie_run(idx, minus, db, masks, evk)).all()
    with pytest.raises(ValueError):
        pipe.slots[1].cc.load_relin_key(evk)
    with pytest.raises(RuntimeError):   # the owner's buffers may not move while slots are attached
        pie.BatchedFHEHIPPIE(cc, vectorizedHCT=rand_limbs(rng, cc.q, (K, 2, E), N), preCalcRandomMask=rand_limbs(rng, cc.q, (2,), N))
    with pytest.raises(RuntimeError):   # ... nor be rewritten in place (same shape): the slots read them on their own streams
        pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    with pytest.raises(RuntimeError):
        cc.load_relin_key(evk)
    with pytest.raises(RuntimeError):   # ... nor may the owner drop them by attaching itself elsewhere
        pie.BatchedFHEHIPPIE(cc, attachTo=pipe.slots[1])
    for s, (idx, minus) in zip(pipe.slots, queries):   # nothing was freed or overwritten by the refused calls
        s.setMinusCompareElement(minus)
        s.setIndex(idx)
        s.run()
        assert (s.getResultList() == o.pie_run(idx, minus, db, masks, evk)).all()
    # a slot with a database of its own: detached, the owner's buffers untouched
    db2, masks2 = rand_limbs(rng, cc.q, (K, 3, E), N), rand_limbs(rng, cc.q, (3,), N)
    c1 = pipe.slots[1].cc
    op2 = pie.BatchedFHEHIPPIE(c1, vectorizedHCT=db2, preCalcRandomMask=masks2)
    c1.load_relin_key(evk)
    idx, minus = queries[0]
    op2.setMinusCompareElement(minus)
    op2.setIndex(idx)
    op2.run()
    assert (op2.getResultList() == o.pie_run(idx, minus, db2, masks2, evk)).all()
    pipe.close()
    op.setMinusCompareElement(minus)
    op.setIndex(idx)
    op.run()
    assert (op.getResultList() == o.pie_run(idx, minus, db, masks, evk)).all()
    cc.close()


def test_staged_queries_from_several_host_threads(ob, pie):
    """One host thread per handle, as the C ABI allows (SURVEY 8b "Threading"): three query slots on one database, each driven by a
    thread of its own that stages queries piece by piece from page-locked memory and waits for the result lists, all at once.  The
    threads meet in the library's per-device upload order (piehip_host.cpp DeviceUploads: one staging sequence at a time takes the
    link, the others queue up behind it -- r05 replaced a process-wide mutex that was held while polling); every query of every
    thread equals the oracle's run(), the turn waits are recorded, nothing deadlocks."""
    import threading
    import torch
    N, L, t, K, E, b, depth, rounds = 8192, 3, T32, 2, 4, 6, 3, 6
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(31337)
    db, masks, evk = rand_limbs(rng, cc.q, (K, b, E), N), rand_limbs(rng, cc.q, (b,), N), rand_limbs(rng, cc.q, (L, 2), N)
    cc.load_relin_key(evk)
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    streams = [torch.cuda.Stream() for _ in range(depth - 1)]
    it = iter(streams)
    pipe = pie.QueryPipeline(op, depth, lambda: pie.PieContext(N, L, t, stream=next(it).cuda_stream))
    queries = [[(rand_limbs(rng, cc.q, (K, E, 2), N), rand_limbs(rng, cc.q, (2,), N)) for _ in range(rounds)] for _ in range(depth)]
    want = [[o.pie_run(idx, minus, db, masks, evk) for idx, minus in qs] for qs in queries]
    bufs = [s.hostBuffers() for s in pipe.slots]
    errors = []
    start = threading.Barrier(depth)

    def drive(i):
        try:
            s, (pi, pm, pr) = pipe.slots[i], bufs[i]
            start.wait()
            for r, (idx, minus) in enumerate(queries[i]):
                pi[...] = idx
                pm[...] = minus
                pr[...] = 0
                s.stageMinus(pm)
                for h in range(K):
                    for j in range(E):
                        s.stageIndexCiphertext(h, j, pi[h, j])
                s.runStaged(pr)
                s.waitHost()
                if not (pr == want[i][r]).all():
                    errors.append("thread %d, query %d differs from the oracle" % (i, r))
        except Exception as exc:   # noqa: BLE001 -- reported by the main thread
            errors.append("thread %d: %r" % (i, exc))

    threads = [threading.Thread(target=drive, args=(i,)) for i in range(depth)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in threads), "a staging thread is stuck"
    assert not errors, errors
    waits = [s.cc.upload_turn_wait() for s in pipe.slots]
    assert sum(w[2] for w in waits) > 0          # the threads did queue up behind each other's uploads ...
    assert max(w[0] for w in waits) < 1000.0     # ... for as long as an upload takes, not for the 5 s bound
    pipe.close()
    cc.close()


@pytest.mark.parametrize("N,L,K,E,b", [(4096, 2, 2, 5, 5), (16384, 4, 2, 3, 4), (8192, 3, 3, 4, 3)])
def test_run_host_pipelined_call_matches_separate_calls(ob, pie, N, L, K, E, b):
    """piehip_run_host (row-wise upload under stage A, per-group download) == setMinusCompareElement + setIndex + run +
    getResultList == the oracle; from pageable arrays and from the library's page-locked staging arrays; consecutive
    queries differ, and a plain run() afterwards still sees the right inputs"""
    t = T16 if N == 4096 else T32
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(N + b)
    db, masks, evk = rand_limbs(rng, cc.q, (K, b, E), N), rand_limbs(rng, cc.q, (b,), N), rand_limbs(rng, cc.q, (L, 2), N)
    cc.load_relin_key(evk)
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    pi, pm, pr = op.hostBuffers()
    for q in range(3):
        idx, minus = rand_limbs(rng, cc.q, (K, E, 2), N), rand_limbs(rng, cc.q, (2,), N)
        want = o.pie_run(idx, minus, db, masks, evk)
        if q == 1:   # the staging arrays
            pi[...] = idx
            pm[...] = minus
            got = op.runHost(pi, pm, pr)
        else:
            got = op.runHost(idx, minus)
        assert (got == want).all()
    op.run()   # inputs of the last run_host are still set
    assert (op.getResultList() == want).all()
    for streams in (1, 0):
        cc.set_run_streams(streams)
        idx, minus = rand_limbs(rng, cc.q, (K, E, 2), N), rand_limbs(rng, cc.q, (2,), N)
        assert (op.runHost(idx, minus) == o.pie_run(idx, minus, db, masks, evk)).all()
    cc.close()


@pytest.mark.parametrize("N,L,t,below,what", [
    (4096, 3, T32, (1 << 61) - 1, "61-bit primes: no lazy-residue NTT, no mad arithmetic"),
    (4096, 2, T16, 1 << 50, "50-bit primes: register-blocked NTT, 128-bit Barrett instead of the one-word form"),
    (16384, 2, T32, 1 << 58, "58-bit primes with the folded transforms"),
])
def test_caller_supplied_moduli(ob, pie, N, L, t, below, what):
    """piehip_create with the context's own moduli (a deployment hands over OpenFHE's): tables, NTT,
    base conversions, EvalMult and run() against the oracle built on the same moduli"""
    ch = ob.gen_primes(N, 2 * L + 1, below)
    q, p = ch[:L].copy(), ch[L:].copy()
    o = ob.Oracle(N, L, t, q, p)
    cc = pie.PieContext(N, L, t, q, p)
    assert (cc.moduli == o.moduli).all()
    rng = np.random.default_rng(N + L)
    M = 2 * L + 1
    x = rand_limbs(rng, o.moduli[:M], (2,), N)
    f = cc.ntt(x, 0, M)
    assert (f == np.stack([np.stack([o.ntt(mi, x[k, mi]) for mi in range(M)]) for k in range(2)])).all()
    assert (cc.ntt(f, 0, M, inverse=True) == x).all()
    xq = rand_limbs(rng, o.moduli[:L], (4,), N)
    assert (cc.base_convert(0, xq) == np.stack([o.expand_q_to_qp(v) for v in xq])).all()
    assert (cc.base_convert(1, xq) == np.stack([o.scale_pq_expand(v) for v in xq])).all()
    xqp = rand_limbs(rng, o.moduli[:M], (3,), N)
    assert (cc.base_convert(2, xqp) == np.stack([o.scale_round_tp(v) for v in xqp])).all()
    sk = o.keygen(3)
    evk = o.relin_keygen(sk, 4)
    cc.load_relin_key(evk)
    a = o.encrypt_slots(sk, [1, 2, 3, -4], 5)
    b = o.encrypt_slots(sk, [5, -6, 7, 8], 6)
    prod = cc.EvalMult(a, b)
    assert (prod == o.mul(a, b, evk)).all()
    assert list(o.decrypt_slots(sk, prod, 4)[0]) == [5, -12, 21, -32]
    # a small query through run()
    k, e, K, E, nb = 2, 6, 2, 5, 3
    d = _query(ob, o, rng, 60, 6, k, e, K, E, nb)
    db = np.stack([o.encode_eval(d["slots"][h, bn, j]) for h in range(K) for bn in range(nb) for j in range(E)]).reshape(K, nb, E, L, N)
    masks = np.stack([o.encode_eval(d["mask_slots"][bn]) for bn in range(nb)])
    idx = np.stack([o.encrypt_slots(sk, d["index"][h, j], 100 + h * E + j) for h in range(K) for j in range(E)]).reshape(K, E, 2, L, N)
    minus = o.encrypt_slots(sk, d["minus"], 99)
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    op.setMinusCompareElement(minus)
    op.setIndex(idx)
    op.run()
    assert (op.getResultList() == o.pie_run(idx, minus, db, masks, evk)).all()
    cc.close()


@pytest.mark.parametrize("N,L,t,K,E,b,nqs", [
    (4096, 2, T16, 2, 4, 5, (2, 3, 4, 5, 8)),      # every group size of stage A's batch kernel: 2, 3, 4, 3 + 2, 4 + 4; b = 5 = 4 + 1 / 2 + 2 + 1
    (16384, 4, T32, 2, 3, 14, (2, 3)),             # the headline ring, two queues (8 + 6 layers), folded transforms
    (2048, 3, T32, 3, 17, 4, (2, 7)),              # K = 3 (two chained products), E > 15: carry sweeps and a mid-sum reduction
    (4096, 2, T16, 1, 6, 3, (2, 4)),               # K = 1: stage A + mask multiply
    (8192, 3, T32, 2, 4, 7, (6,)),                 # 3 + 3 queries, b = 7: 4 + 3 / 2 + 2 + 2 + 1 layers per thread
    (16384, 4, T32, 2, 14, 1, (3,)),               # one bin layer: a rank's share of b = 14 over eight GPUs
    (16384, 4, T32, 2, 14, 2, (3,)),               # ... and two
    (8192, 3, T32, 3, 5, 4, (3,)),                 # K = 3 on a ring of the 16-coefficient transform (one slice per limb): X of the first product
                                                   # comes lane-ordered from stage A, X of the second is the first product itself
    (16384, 4, T32, 3, 3, 3, (2,)),                # ... with folded slices
    (32768, 6, T32, 3, 4, 3, (3, 2)),              # the 2^14-slice geometry (C5's ring, L = 6, K = 3): stage A writes operand X lane-ordered into the QP array
                                                   # and ntt16_kernel_t<14, ...> reads it there; the second product's X is the first product itself
    (32768, 6, T32, 2, 17, 2, (3,)),               # ... E past the carry sweep of the batched stage A on that ring
])
def test_query_batches(ob, pie, N, L, t, K, E, b, nqs):
    """piehip_set_query_batch: run() over nq queries at once.  Every query's ciphertexts equal the oracle's for that query alone
    (random limbs), whatever the batch size, the queue count and the order the inputs were set in (the host-memory path of a
    batch: test_staged_query_batches)."""
    import torch
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(N + 7 * K + b)
    q = cc.q
    db, masks, evk = rand_limbs(rng, q, (K, b, E), N), rand_limbs(rng, q, (b,), N), rand_limbs(rng, q, (L, 2), N)
    if K > 1:
        cc.load_relin_key(evk)
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    nmax = max(nqs)
    queries = [(rand_limbs(rng, q, (K, E, 2), N), rand_limbs(rng, q, (2,), N)) for _ in range(nmax)]
    want = [o.pie_run(idx, minus, db, masks, evk) for idx, minus in queries]
    for nq in nqs:
        op.setQueryBatch(nq)
        with pytest.raises(ValueError):
            op.setIndex(queries[0][0], query=nq)
        order = list(rng.permutation(nq))
        dev = []
        for i in order:
            idx, minus = queries[i]
            if i % 2:   # odd queries: inputs already resident in HBM
                di, dm = torch.from_numpy(idx.view(np.int64)).cuda(), torch.from_numpy(minus.view(np.int64)).cuda()
                dev += [di, dm]
                op.setIndexDevice(di.data_ptr(), query=int(i))
                op.setMinusCompareElementDevice(dm.data_ptr(), query=int(i))
            else:
                op.setMinusCompareElement(minus, query=int(i))
                op.setIndex(idx, query=int(i))
        torch.cuda.synchronize()
        for streams in (0, 1):
            cc.set_run_streams(streams)
            op.run()
            op.run()    # back-to-back runs of the same batch
            got = op.getResultList()
            assert got.shape == (nq, b, 2, L, N)
            for i in range(nq):
                assert (got[i] == want[i]).all(), "query %d of a batch of %d" % (i, nq)
    op.setQueryBatch(1)
    idx, minus = queries[1]
    assert (op.runHost(idx, minus) == want[1]).all()
    op.setQueryBatch(2)
    op.setIndex(queries[0][0])
    op.setMinusCompareElement(queries[0][1])
    op.setIndex(queries[1][0], query=1)
    op.setMinusCompareElement(queries[1][1], query=1)
    op.run()
    got = op.getResultList()
    assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
    op2 = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)   # same shape: the inputs of the old database are stale
    op2.setIndex(queries[0][0])
    op2.setMinusCompareElement(queries[0][1])
    with pytest.raises(RuntimeError, match="query of the batch"):
        op2.run()
    with pytest.raises(ValueError):
        op.setQueryBatch(9)
    cc.close()


def test_key_slots_follow_the_handles_key(ob, pie):
    """piehip_load_relin_key_q gives query q its client's EvalMult key; queries without one use the HANDLE's key -- also when that key
    is loaded or replaced AFTER the per-query slots exist (the slots held a copy taken once, a later
    piehip_load_relin_key left them stale and run() relinearised with the old key, silently)."""
    N, L, t, K, E, b = 4096, 2, T16, 2, 3, 3
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(2025)
    q = cc.q
    db, masks = rand_limbs(rng, q, (K, b, E), N), rand_limbs(rng, q, (b,), N)
    key_old, key_new, key_a = (rand_limbs(rng, q, (L, 2), N) for _ in range(3))
    queries = [(rand_limbs(rng, q, (K, E, 2), N), rand_limbs(rng, q, (2,), N)) for _ in range(2)]
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    op.setQueryBatch(2)
    for i, (idx, minus) in enumerate(queries):
        op.setIndex(idx, query=i)
        op.setMinusCompareElement(minus, query=i)
    # (1) a per-query key first, the handle's key afterwards: query 1 had NO key when the slots were made
    cc.load_relin_key(key_a, query=0)
    with pytest.raises(RuntimeError, match="key"):
        op.run()
    cc.load_relin_key(key_old)
    op.run()
    got = op.getResultList()
    assert (got[0] == o.pie_run(*queries[0], db, masks, key_a)).all() and (got[1] == o.pie_run(*queries[1], db, masks, key_old)).all()
    # (2) the handle's key replaced: query 1 follows it, query 0 keeps its own
    cc.load_relin_key(key_new)
    op.run()
    got = op.getResultList()
    assert (got[0] == o.pie_run(*queries[0], db, masks, key_a)).all() and (got[1] == o.pie_run(*queries[1], db, masks, key_new)).all()
    cc.close()


@pytest.mark.parametrize("N,L,t,K,E,b,nq", [(16384, 4, T32, 2, 3, 9, 2), (32768, 6, T32, 2, 2, 2, 1), (8192, 3, T32, 3, 3, 5, 3)])
def test_transform_slots_do_not_change_results(ob, pie, N, L, t, K, E, b, nq):
    """piehip_set_transform_slots caps the persistent transform grids (a sharded server leaves CUs to RCCL's kernels): every cap --
    one workgroup, an odd number, fewer than the launch has slices, more than the device has slots -- gives the oracle's bits, on
    one and two queues."""
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(N + b)
    q = cc.q
    db, masks, evk = rand_limbs(rng, q, (K, b, E), N), rand_limbs(rng, q, (b,), N), rand_limbs(rng, q, (L, 2), N)
    cc.load_relin_key(evk)
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    if nq > 1:
        op.setQueryBatch(nq)
    queries = [(rand_limbs(rng, q, (K, E, 2), N), rand_limbs(rng, q, (2,), N)) for _ in range(nq)]
    want = [o.pie_run(idx, minus, db, masks, evk) for idx, minus in queries]
    for i, (idx, minus) in enumerate(queries):
        op.setIndex(idx, query=i)
        op.setMinusCompareElement(minus, query=i)
    cap0, dev_slots = cc.transform_slots()
    assert cap0 == 0 and dev_slots >= 2
    for cap in (1, 7, 64, dev_slots - 32, 4 * dev_slots, 0):
        cc.set_transform_slots(cap)
        assert cc.transform_slots()[0] == cap
        for streams in (0, 1):
            cc.set_run_streams(streams)
            op.run()
            got = op.getResultList()
            got = got[None] if nq == 1 else got
            for i in range(nq):
                assert (got[i] == want[i]).all(), "cap %d, %d queue(s), query %d" % (cap, streams, i)
    cc.close()


def test_profile_read_lengths(ob, pie):
    """piehip_profile_read_n writes min(n, PIEHIP_NKERNELS) entries; piehip_profile_read is version 100's entry point and keeps that
    version's twelve (a caller built against the old header passes arrays of twelve: ADVICE r04); piehip_version says which"""
    import ctypes as C
    from nested_hashing_psi_amd._lib import lib, u32p, f64p, NKERNELS
    N, L, t, K, E, b = 4096, 2, T16, 2, 3, 3
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(5)
    q = cc.q
    cc.load_relin_key(rand_limbs(rng, q, (L, 2), N))
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=rand_limbs(rng, q, (K, b, E), N), preCalcRandomMask=rand_limbs(rng, q, (b,), N))
    op.setIndex(rand_limbs(rng, q, (K, E, 2), N))
    op.setMinusCompareElement(rand_limbs(rng, q, (2,), N))
    cc.set_run_streams(1)
    cc.set_profiling(True)
    op.run()
    assert lib().piehip_version() >= 101
    prof = cc.profile()
    assert prof["event_pair"]["launches"] >= 1 and prof["stage_a_mac"]["launches"] == 1
    for entry, n in ((lib().piehip_profile_read, 12), (None, 5), (None, NKERNELS + 7)):
        size = max(n, 12) + 4
        cnt = np.full(size, 0xDEAD, dtype=np.uint32)
        ms = np.full(size, -1.0)
        by = np.full(size, -1.0)
        if entry is not None:
            assert entry(cc._h, cnt.ctypes.data_as(u32p), ms.ctypes.data_as(f64p), by.ctypes.data_as(f64p)) == 0
        else:
            assert lib().piehip_profile_read_n(cc._h, n, cnt.ctypes.data_as(u32p), ms.ctypes.data_as(f64p), by.ctypes.data_as(f64p)) == 0
        filled = min(n, NKERNELS)
        assert (cnt[filled:] == 0xDEAD).all() and (ms[filled:] == -1.0).all() and (by[filled:] == -1.0).all()
        assert cnt[0] == 1 and (cnt[:filled] != 0xDEAD).all()
    cc.set_profiling(False)
    cc.close()


def test_batch_layer_groups_over_many_shapes(ob, pie):
    """Stage A of a query batch picks its bin layers per thread by the layer count (groups of up to four / three layers for two / three
    and four queries, one launch, the last group ragged: r05).  Every layer count 1 .. 17 with every batch size 2 .. 8 (i.e. every
    split into launches of two, three and four queries), on both queue counts: each query's result list equals the result of running
    that query ALONE (the single-query kernel, a different code path), and for a sample of the shapes the oracle's run()."""
    N, L, t, K = 2048, 2, T32, 2
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(20261005)
    q = cc.q
    evk = rand_limbs(rng, q, (L, 2), N)
    cc.load_relin_key(evk)
    for b in range(1, 18):
        E = (5, 9, 17)[b % 3]     # no carry sweep / one carry sweep / a mid-sum reduction of the column accumulators
        queries = [(rand_limbs(rng, q, (K, E, 2), N), rand_limbs(rng, q, (2,), N)) for _ in range(8)]
        db, masks = rand_limbs(rng, q, (K, b, E), N), rand_limbs(rng, q, (b,), N)
        op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
        alone = []
        for idx, minus in queries:
            op.setMinusCompareElement(minus)
            op.setIndex(idx)
            op.run()
            alone.append(op.getResultList().copy())
        if b in (1, 5, 7, 14, 17):
            for i in (0, 7):
                assert (alone[i] == o.pie_run(queries[i][0], queries[i][1], db, masks, evk)).all(), "b = %d, query %d alone vs the oracle" % (b, i)
        for nq in range(2, 9):
            op.setQueryBatch(nq)
            for i in range(nq):
                op.setMinusCompareElement(queries[i][1], query=i)
                op.setIndex(queries[i][0], query=i)
            for streams in (1, 0):
                cc.set_run_streams(streams)
                op.run()
                got = op.getResultList()
                for i in range(nq):
                    assert (got[i] == alone[i]).all(), "b = %d, batch of %d, query %d, %d queue(s)" % (b, nq, i, streams or 2)
        op.setQueryBatch(1)
    cc.close()


@pytest.mark.parametrize("E,b", [(1, 1), (1, 3), (15, 2), (16, 3), (40, 5)])
def test_run_shape_extremes(ob, pie, E, b):
    """one inner position, the last E of the carry-free accumulator (15), the first E of the 128-bit accumulator (16), a long
    reduction (40), odd bin-layer counts across the two queues -- random limbs, ciphertext bits vs the oracle"""
    N, L, t, K = 2048, 3, T32, 2
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(1000 * E + b)
    q = o.moduli[:L]
    idx, minus = rand_limbs(rng, q, (K, E, 2), N), rand_limbs(rng, q, (2,), N)
    db, masks, evk = rand_limbs(rng, q, (K, b, E), N), rand_limbs(rng, q, (b,), N), rand_limbs(rng, q, (L, 2), N)
    cc.load_relin_key(evk)
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    op.setMinusCompareElement(minus)
    op.setIndex(idx)
    op.run()
    assert (op.getResultList() == o.pie_run(idx, minus, db, masks, evk)).all()
    cc.close()


@pytest.mark.parametrize("N,L,t,E,b", [(4096, 2, T16, 6, 5), (16384, 4, T32, 4, 9)])
def test_run_with_one_inner_hash_function(ob, pie, N, L, t, E, b):
    """K = 1 (BatchedFHEHIPPIE.cpp:117-120,126): multipliedResult is the inner product itself, run() = stage A + mask multiply;
    no relinearisation key is needed.  Ciphertext bits vs the oracle, decrypted semantics, every entry point of run()."""
    K = 1
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(N + E)
    sk = o.keygen(5)
    B = 24
    # a table with one inner hash function: cell [bin][j][slot]; the client in slot s selects position pos[s] and asks for x[s]
    cells = rng.integers(1, t, (K, b, E, B), dtype=np.int64)
    pos = rng.integers(0, E, B)
    x = rng.integers(1, t, B, dtype=np.int64)
    hit = rng.random(B) < 0.5
    hit_bin = rng.integers(0, b, B)
    for s in np.nonzero(hit)[0]:
        cells[0, hit_bin[s], pos[s], s] = x[s]
    mask_slots = rng.integers(1, t, (b, B), dtype=np.int64)
    index = np.zeros((K, E, B), dtype=np.int64)
    index[0, pos, np.arange(B)] = 1
    idx = np.stack([o.encrypt_slots(sk, index[0, j], 40 + j) for j in range(E)]).reshape(K, E, 2, L, N)
    minus = o.encrypt_slots(sk, -x, 39)
    db = np.stack([o.encode_eval(cells[0, bn, j]) for bn in range(b) for j in range(E)]).reshape(K, b, E, L, N)
    masks = np.stack([o.encode_eval(mask_slots[bn]) for bn in range(b)])
    want = o.pie_run(idx, minus, db, masks, np.zeros((L, 2, L, N), dtype=np.uint64))
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)   # no key loaded
    op.setMinusCompareElement(minus)
    op.setIndex(idx)
    op.run()
    got = op.getResultList().copy()
    assert (got == want).all()
    for streams in (1, 0):
        cc.set_run_streams(streams)
        assert (op.runHost(idx, minus) == want).all()
    op2 = pie.BatchedFHEHIPPIE(cc, slots=cells, mask_slots=mask_slots)
    op2.setMinusCompareElement(minus)
    op2.setIndex(idx)
    op2.run()
    assert (op2.getResultList() == want).all()
    dec = np.stack([o.decrypt_slots(sk, got[bn], B)[0] for bn in range(b)])
    zero = (dec == 0)
    for s in range(B):
        expect = set(np.nonzero(cells[0, :, pos[s], s] == x[s])[0])
        assert set(np.nonzero(zero[:, s])[0]) == expect
    # the hashing entry points refuse one inner hash function as the reference's CuckooHashTable does (CuckooHashTable.cpp:39-42)
    with pytest.raises(ValueError, match="more than one hash function"):
        pie.BatchedFHEHIPPIE(cc, serverSet=np.arange(1, 50, dtype=np.uint64), hashParams=dict(k=2, e=8, K=1, b=4, E=4, evict_seed=1,
                                                                                            shuffle_seed=2, mask_seed=3))
    cc.close()


def test_staged_query_upload(ob, pie):
    """piehip_stage_minus / piehip_stage_index_row / piehip_run_staged: the pieces of a query uploaded as a server receives them
    (any order), evaluated when all have been staged; a missing piece is a call-order error; the one-call form still works"""
    N, L, t, K, E, b = 4096, 2, T16, 3, 4, 9
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(77)
    db, masks, evk = rand_limbs(rng, cc.q, (K, b, E), N), rand_limbs(rng, cc.q, (b,), N), rand_limbs(rng, cc.q, (L, 2), N)
    cc.load_relin_key(evk)
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    pi, pm, pr = op.hostBuffers()
    for order in ([0, 1, 2], [2, 0, 1]):
        idx, minus = rand_limbs(rng, cc.q, (K, E, 2), N), rand_limbs(rng, cc.q, (2,), N)
        pi[...] = idx
        pm[...] = minus
        pr[...] = 0
        op.stageIndexRow(order[0], pi[order[0]])
        op.stageMinus(pm)
        with pytest.raises(RuntimeError, match="not staged"):
            op.runStaged(pr)
        op.stageIndexRow(order[1], pi[order[1]])
        op.stageIndexRow(order[2], pi[order[2]])
        op.runStaged(pr)
        op.waitHost()
        assert (pr == o.pie_run(idx, minus, db, masks, evk)).all()
    with pytest.raises(RuntimeError, match="not staged"):   # nothing staged since the last run
        op.runStaged(pr)
    with pytest.raises(ValueError):
        op.stageIndexRow(K, pi[0])
    # ciphertext by ciphertext (one message of the reference's receive loop each: piehip_stage_index_ct_q), mixed with whole rows
    op.stageReset()
    idx, minus = rand_limbs(rng, cc.q, (K, E, 2), N), rand_limbs(rng, cc.q, (2,), N)
    pi[...] = idx
    pm[...] = minus
    op.stageMinus(pm)
    op.stageIndexRow(1, pi[1])
    for h in (2, 0):
        for j in rng.permutation(E):
            if h == 0 and j == E - 1:
                continue
            op.stageIndexCiphertext(h, int(j), pi[h, int(j)])
    with pytest.raises(RuntimeError, match="not staged"):   # ciphertext (0, E - 1) has not arrived
        op.runStaged(pr)
    with pytest.raises(ValueError):
        op.stageIndexCiphertext(0, E, pi[0, 0])
    op.stageIndexCiphertext(0, E - 1, pi[0, E - 1])
    op.runStaged(pr)
    op.waitHost()
    assert (pr == o.pie_run(idx, minus, db, masks, evk)).all()
    idx, minus = rand_limbs(rng, cc.q, (K, E, 2), N), rand_limbs(rng, cc.q, (2,), N)
    assert (op.runHost(idx, minus) == o.pie_run(idx, minus, db, masks, evk)).all()
    cc.close()


@pytest.mark.parametrize("N,L,t,K,E,b,nq", [
    (4096, 2, T16, 2, 4, 5, 3),        # one queue
    (16384, 4, T32, 2, 3, 9, 2),       # the headline ring, two queues (5 + 4 layers): results leave per queue group
    (2048, 3, T32, 3, 5, 4, 4),        # K = 3: three rows per query, two chained products
    (16384, 4, T32, 2, 14, 3, 3),      # E = 14 as at C3, batch of three
    (32768, 6, T32, 3, 3, 3, 3),       # C5's ring (2^14 slices, L = 6), K = 3, batch of three through the host-memory path
])
def test_staged_query_batches(ob, pie, N, L, t, K, E, b, nq):
    """The host-memory path of a query batch -- what a server with nq clients connected calls (BatchedFHEPSIServer.cpp:94-108 per
    client): every query has its own page-locked staging (piehip_host_buffers_q), its pieces are staged in whatever order the
    clients' messages arrive (piehip_stage_minus_q / piehip_stage_index_row_q), run_staged evaluates the batch and the result list
    [b][nq] comes back in host memory.  Every client has its OWN EvalMult key (piehip_load_relin_key_q): query q's results equal
    the oracle's run() of that query alone under key q.  Also: a missing piece is a call-order error, a piece staged twice is
    replaced, stage_reset drops a partial sequence, and piehip_run_host takes the batch in one call."""
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(1000 * nq + N + b)
    q = cc.q
    db, masks = rand_limbs(rng, q, (K, b, E), N), rand_limbs(rng, q, (b,), N)
    keys = [rand_limbs(rng, q, (L, 2), N) for _ in range(nq)]
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    op.setQueryBatch(nq)
    for i in range(nq):
        cc.load_relin_key(keys[i], query=i)
    with pytest.raises(ValueError):
        cc.load_relin_key(keys[0], query=nq)
    bufs = [op.hostBuffers(query=i) for i in range(nq)]
    pr = bufs[0][2]
    assert pr.shape == (b, nq, 2, L, N) and all(bf[2].ctypes.data == pr.ctypes.data for bf in bufs)
    with pytest.raises(ValueError):
        op.hostBuffers(query=nq)
    for rep in range(2):
        queries = [(rand_limbs(rng, q, (K, E, 2), N), rand_limbs(rng, q, (2,), N)) for _ in range(nq)]
        want = [o.pie_run(idx, minus, db, masks, keys[i]) for i, (idx, minus) in enumerate(queries)]
        for i, (idx, minus) in enumerate(queries):
            bufs[i][0][...] = idx
            bufs[i][1][...] = minus
        pr[...] = 0
        pieces = [(i, h) for i in range(nq) for h in range(-1, K)]     # h = -1: the minus element
        pieces = [pieces[j] for j in rng.permutation(len(pieces))]
        if rep == 1:
            # a sequence that is abandoned half way (a client hung up): the next piece starts a fresh one
            for i, h in pieces[:len(pieces) // 2]:
                op.stageMinus(bufs[i][1], query=i) if h < 0 else op.stageIndexRow(h, bufs[i][0][h], query=i)
            op.stageReset()
            with pytest.raises(RuntimeError, match="not staged"):
                op.runStaged(pr)
        for n_, (i, h) in enumerate(pieces):
            if n_ == len(pieces) - 1:
                with pytest.raises(RuntimeError, match="not staged"):
                    op.runStaged(pr)
                # a piece staged twice before the run: the later contents count (here: garbage first, then the query's)
                garbage = np.zeros_like(bufs[i][1]) if h < 0 else np.zeros_like(bufs[i][0][h])
                op.stageMinus(garbage, query=i) if h < 0 else op.stageIndexRow(h, garbage, query=i)
            op.stageMinus(bufs[i][1], query=i) if h < 0 else op.stageIndexRow(h, bufs[i][0][h], query=i)
        op.runStaged(pr)
        op.waitHost()
        for i in range(nq):
            assert (pr[:, i] == want[i]).all(), "query %d of the staged batch (round %d)" % (i, rep)
        # the same batch again by the plain entry points (inputs are still in HBM)
        op.run()
        got = op.getResultList()
        for i in range(nq):
            assert (got[i] == want[i]).all()
    # the one-call form over pageable arrays
    queries = [(rand_limbs(rng, q, (K, E, 2), N), rand_limbs(rng, q, (2,), N)) for _ in range(nq)]
    res = op.runHost(np.stack([x[0] for x in queries]), np.stack([x[1] for x in queries]))
    for i, (idx, minus) in enumerate(queries):
        assert (res[:, i] == o.pie_run(idx, minus, db, masks, keys[i])).all()
    # back to one query per run(): the context's own key again (none loaded here -> a call-order error, then fine)
    op.setQueryBatch(1)
    if K > 1:
        with pytest.raises(RuntimeError, match="key"):
            op.runHost(queries[0][0], queries[0][1])
    cc.load_relin_key(keys[0])
    assert (op.runHost(queries[0][0], queries[0][1]) == o.pie_run(queries[0][0], queries[0][1], db, masks, keys[0])).all()
    cc.close()


@pytest.mark.parametrize("N,L", [(8192, 3), (16384, 4), (32768, 6)])
def test_extreme_residues_through_run(ob, pie, N, L):
    """residues 0 and q - 1 in every array that enters run() (index matrix, minus element, database, masks, key): the boundary
    cases of the lazy ranges in the butterfly blocks, of the quotient estimates (shoup63 / divmod63) and of the column
    accumulators, on the three transform geometries (one 2^13 slice per limb; two folded 2^13; two folded 2^14)"""
    t, K, E, b = T32, 2, 3, 2
    o = ob.Oracle(N, L, t)
    cc = pie.PieContext(N, L, t)
    rng = np.random.default_rng(N)
    q = o.moduli[:L]

    def rl(shape):
        a = rand_limbs(rng, q, shape, N)
        for i in range(L):
            a[..., i, 0:64] = q[i] - np.uint64(1)
            a[..., i, 64:128] = 0
            a[..., i, N // 2:N // 2 + 32] = q[i] - np.uint64(1)     # the folded partner positions too
            a[..., i, N - 32:] = q[i] - np.uint64(1)
        return a
    idx, minus = rl((K, E, 2)), rl((2,))
    db, masks, evk = rl((K, b, E)), rl((b,)), rl((L, 2))
    cc.load_relin_key(evk)
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    op.setMinusCompareElement(minus)
    op.setIndex(idx)
    op.run()
    assert (op.getResultList() == o.pie_run(idx, minus, db, masks, evk)).all()
    # ... and with every residue at q - 1
    allmax = lambda shape: np.broadcast_to((q - np.uint64(1))[:, None], shape + (L, N)).copy()
    idx, minus, db, masks = allmax((K, E, 2)), allmax((2,)), allmax((K, b, E)), allmax((b,))
    op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=db, preCalcRandomMask=masks)
    op.setMinusCompareElement(minus)
    op.setIndex(idx)
    op.run()
    assert (op.getResultList() == o.pie_run(idx, minus, db, masks, evk)).all()
    cc.close()


# ---- the caller of the hot path as its own process, talking the reference's framing ---------------------------------------
@pytest.mark.parametrize("shape,nclients", [("small", 1), ("C3", 1), ("small", 3), ("C3", 3)])
def test_two_process_psi_over_the_wire(ob, pie, tmp_path, shape, nclients):
    """host/BatchedFHEPSIServer.hpp (C++, reference phase order PSIServer.hpp:66-87) in a child process behind a socket;
    this process plays the client with the product's harness.  The computed intersection equals the true one.  At the
    headline shape (C3: 29 query messages of 1 MiB, 14 result messages) the server's OnlineComputation -- the reference's
    timer, BatchedFHEPSIServer.cpp:98-106 -- is also bounded: the query's upload runs underneath the receive loop.
    nclients = 3: three clients on three channels, each with its own key pair, set and query; the server evaluates their
    queries as ONE batch (bench.py's default timed region, reached from the reference's call site) and every client
    recovers exactly its own intersection."""
    import os
    import socket
    import struct
    import subprocess
    from nested_hashing_psi_amd.client import BatchedFHEPSIClient
    from tests.test_oracle_pie import distinct_items
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "nested_hashing_psi_amd")
    exe = str(tmp_path / "server_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(root, "tests", "server_main.cpp"),
                           "-L" + libdir, "-lpiehip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    rng = np.random.default_rng(31337)
    if shape == "small":
        N, L, t = 8192, 3, T32
        k, e, K, E, b = 3, 40, 2, 8, 7
        nS, nC, ninter = 2000, 64, 33
        items = distinct_items(rng, t, nS + nclients * nC)
    else:
        N, L, t = 16384, 4, T32
        k, e, K, E, b = 2, 4949, 2, 14, 14
        nS, nC, ninter = 1 << 20, 1 << 10, 513
        items = np.unique(rng.integers(1, t, nS + nclients * nC + 8192, dtype=np.uint64))
        rng.shuffle(items)
    server = items[:nS].copy()
    clientsets, inters = [], []
    for c in range(nclients):   # client c shares items [c * 101, c * 101 + ninter) of the server set
        inter = server[c * 101:c * 101 + ninter]
        cs = np.concatenate([inter, items[nS + c * nC:nS + c * nC + nC - ninter]])
        rng.shuffle(cs)
        clientsets.append(cs)
        inters.append(inter)
    setfile = tmp_path / "server_set.bin"
    server.astype(np.uint64).tofile(setfile)
    pairs = [socket.socketpair() for _ in range(nclients)]
    socks = [p_[0] for p_ in pairs]
    fds = [p_[1].fileno() for p_ in pairs]
    proc = subprocess.Popen([exe, ",".join(str(f) for f in fds), str(setfile), str(k), str(e), str(K), str(E), str(b)],
                            pass_fds=tuple(fds), stdout=subprocess.PIPE)
    for p_ in pairs:
        p_[1].close()

    def send(a, payload):
        a.sendall(struct.pack("i", len(payload)) + payload)

    def recv(a):
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

    ccs = [pie.PieContext(N, L, t) for _ in range(nclients)]
    cls = [BatchedFHEPSIClient(ccs[c], k, e, K, E, b) for c in range(nclients)]
    moduli = np.zeros(15, dtype=np.uint64)
    moduli[:2 * L + 1] = ccs[0].moduli[:2 * L + 1]
    for c, a in enumerate(socks):
        evk = cls[c].runSetUpPhase(keySeed=11 + 10 * c, evalKeySeed=12 + 10 * c)     # every client its own key pair
        send(a, struct.pack("IIQ", N, L, t) + moduli.tobytes())          # context
        send(a, b"")                                                      # public key (unused by the operator)
        send(a, np.ascontiguousarray(evk, dtype=np.uint64).tobytes())     # EvalMult key
    for a in socks:
        assert recv(a) == b""                                             # server: setup phase over
    queries = [cls[c].runOfflinePhase(clientsets[c], encSeedBase=100 + 1000 * c) for c in range(nclients)]
    for a in socks:
        assert recv(a) == b""                                             # server: offline phase over
    for c, a in enumerate(socks):
        minus_ct, idx_ct = queries[c]
        send(a, ct_msg(minus_ct))
        for h in range(K):
            for j in range(E):
                send(a, ct_msg(idx_ct[h, j]))
    for c, a in enumerate(socks):
        res = []
        for _ in range(b):
            m = recv(a)
            assert struct.unpack("IIIIQ", m[:24]) == (0x48454950, 1, L, N, 0)
            res.append(np.frombuffer(m[24:], dtype=np.uint64).reshape(2, L, N))
        found = cls[c].extractIntersection(np.stack(res))
        assert sorted(int(v) for v in found) == sorted(int(v) for v in inters[c]), "client %d" % c
    out, _ = proc.communicate(timeout=120)
    assert proc.returncode == 0
    assert b"OnlineComputation," in out and b"OfflineComputation," in out
    online_us = int([ln for ln in out.decode().splitlines() if ln.startswith("OnlineComputation,")][0].split(",")[1])
    print("two-process PSI, %s shape, %d client(s): server OnlineComputation %d us" % (shape, nclients, online_us))
    if shape == "C3":
        # one client: run() 0.25 ms + 14 MiB of results over PCIe 0.27 ms; the 29 MiB upload overlaps the receive loop.
        # three clients: one batched run() 0.64 ms + 42 MiB of results
        assert online_us < (900 if nclients == 1 else 2200)
    for a in socks:
        a.close()
    for c_ in ccs:
        c_.close()
