"""Every BASELINE.json configuration at its full size on the GPU, against the oracle.

Each case is a real query end to end: the server set goes through the device's offline phase (nested hashing, Cuckoo
insertion, bin shuffle, gather, packed encoding: piehip_build_db), the client's one-hot index matrix and minus vector are
secret-key encrypted, run() evaluates them on the GPU, and the result must
  (a) equal the oracle's restated run() (oracle/pie_oracle.c, reference BatchedFHEHIPPIE.cpp:88-129) bit for bit on every
      bin layer of every configuration (the oracle needs 0.5 s per C5 layer: the layers are checked on a thread pool);
  (b) decrypt, with a positive noise budget in every result, to exactly the true intersection (reference check:
      src/Client/PSIClient.hpp:142-164);
and the device-built hash table must equal the oracle's (same seeds).  "Bit for bit" is relative to the in-tree oracle:
parity with OpenFHE itself is unpinned.

  C1  N=4096,  2 primes, t=65537,      |S|=2^12, |C|=2^8,  k=3 e=110   K=2 E=7  b=7    (derived)
  C2  N=8192,  3 primes, t=4296540161, |S|=2^16, |C|=2^10, k=3 e=443   K=2 E=12 b=12   (Parameters1.txt:53)
  C3  N=16384, 4 primes,               |S|=2^20, |C|=2^10, k=2 e=4949  K=2 E=14 b=14   (Parameters1.txt:11)
      + the same database with THREE different queries in one run() on one handle: bench.py's default timed region
      + the reference's alternative row for these set sizes, E=20 b=10 (Parameters1.txt:35)
  C5  N=32768, 6 primes,               |S|=2^24, |C|=2^12, k=2 e=13004 K=3 E=30 b=30   (Parameters1.txt:19 with -K 3)
      + C2 and C5 with three queries per run() (what bench.py --config C2 / C5 time by default)
  KAT-0 at the reference test's own parameters: tests/TestBatchedFHEPIE.cpp:14-41,89-94 (N=16384, 33-bit t).
C4 (C3's bin layers over several GPUs) is tests/test_sharding_gpu.py.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
T16 = 65537
T32 = 4296540161
SEEDS = dict(hash_seed=987654321, evict_seed=1, shuffle_seed=2, mask_seed=3)


def run_case(ob, pie, N, L, t, nS, nC, k, e, K, E, b, seed, compare_layers, streams=0, nq=1):
    """nq > 1: nq DIFFERENT clients (own item sets, own Cuckoo tables, own encryption randomness -- one key, as the operator
    holds one EvalMult key) are evaluated by ONE run() on one handle (setQueryBatch); every query's result list is checked
    as if it had been run alone."""
    o = ob.Oracle(N, L, t)
    rng = np.random.default_rng(seed)
    items = np.unique(rng.integers(1, t, nS + nq * nC + 4096, dtype=np.uint64))
    rng.shuffle(items)
    server = items[:nS].copy()
    ninter = nC // 2 + 1
    # oracle side of the offline phase
    tab = ob.Tabulation(SEEDS["hash_seed"], k + K)
    tbl = ob.hct_build(tab, server, k, e, K, b, E, evict_seed=SEEDS["evict_seed"])
    ob.hct_shuffle_bins(tbl, SEEDS["shuffle_seed"])
    slots = ob.pack_db(tbl)
    mask_slots = ob.masks(t, b, k * e, SEEDS["mask_seed"])
    # clients: query q shares items [q * 97, q * 97 + ninter) of the server set and brings nC - ninter of its own
    sk = o.keygen(11)
    evk = o.relin_keygen(sk, 12)
    queries = []
    for q in range(nq):
        inter = server[q * 97:q * 97 + ninter]
        client = np.concatenate([inter, items[nS + q * nC:nS + q * nC + nC - ninter]])
        rng.shuffle(client)
        ctab = ob.client_build(tab, client, k, e, evict_seed=4 + q)
        index, minus_v = ob.client_vectors(tab, ctab, K, E)
        idx = np.stack([o.encrypt_slots(sk, index[h, j], 100 + 1000 * q + h * E + j) for h in range(K) for j in range(E)]).reshape(K, E, 2, L, N)
        minus = o.encrypt_slots(sk, minus_v, 99 + 1000 * q)
        queries.append((inter, ctab, idx, minus))
    # device: whole offline phase from the raw server set, then the queries
    cc = pie.PieContext(N, L, t)
    cc.load_relin_key(evk)
    cc.set_run_streams(streams)
    op = pie.BatchedFHEHIPPIE(cc, serverSet=server, hashParams=dict(k=k, e=e, K=K, b=b, E=E, **SEEDS))
    assert (op.hashTable() == tbl).all()
    del tbl
    if nq > 1:
        op.setQueryBatch(nq)
    for q, (_, _, idx, minus) in enumerate(queries):
        op.setMinusCompareElement(minus, query=q)
        op.setIndex(idx, query=q)
    op.run()
    got = op.getResultList().copy()
    cc.close()
    if nq == 1:
        got = got[None]
    assert got.shape == (nq, b, 2, L, N)
    # (b) decrypted semantics on every bin layer of every query
    budgets = []
    for q, (inter, ctab, _, _) in enumerate(queries):
        dec = []
        for bn in range(b):
            d, bud = o.decrypt_slots(sk, got[q, bn], k * e)
            dec.append(d)
            budgets.append(bud)
        found = ob.client_scan(ctab, np.stack(dec))
        assert len(found) == ninter
        assert sorted(int(v) for v in found) == sorted(int(v) for v in inter)
    assert min(budgets) > 0
    # (a) ciphertext bits: bin layers are independent, one oracle task each (the C calls release the GIL)
    import concurrent.futures

    def layer_differs(bn):
        db = np.stack([o.encode_eval(slots[h, bn, j]) for h in range(K) for j in range(E)]).reshape(K, 1, E, L, N)
        masks = o.encode_eval(mask_slots[bn])[None]
        bad = []
        for q, (_, _, idx, minus) in enumerate(queries):
            want = o.pie_run(idx, minus, db, masks, evk)
            if not (got[q, bn] == want[0]).all():
                bad.append((q, bn))
        return bad
    with concurrent.futures.ThreadPoolExecutor(max_workers=12) as pool:
        bad = [x for r in pool.map(layer_differs, list(compare_layers)) for x in r]
    assert not bad, "(query, bin layer) pairs %s differ from the oracle" % bad
    return min(budgets)


def test_c1_plumbing_config_all_layers(ob, pie_mod):
    run_case(ob, pie_mod, 4096, 2, T16, 1 << 12, 1 << 8, 3, 110, 2, 7, 7, 5, range(7))


def test_c2_config_all_layers(ob, pie_mod):
    run_case(ob, pie_mod, 8192, 3, T32, 1 << 16, 1 << 10, 3, 443, 2, 12, 12, 7, range(12))


def test_c3_headline_config_all_layers(ob, pie_mod):
    run_case(ob, pie_mod, 16384, 4, T32, 1 << 20, 1 << 10, 2, 4949, 2, 14, 14, 123456789, range(14))


def test_c3_serial_queue_all_layers(ob, pie_mod):
    """the same query with run() on one queue (what bench.py --streams 1 and the per-kernel profile passes execute)"""
    run_case(ob, pie_mod, 16384, 4, T32, 1 << 20, 1 << 10, 2, 4949, 2, 14, 14, 123456789, range(14), streams=1)


@pytest.mark.parametrize("streams", [0, 1])
def test_c3_headline_batch_of_three(ob, pie_mod, streams):
    """bench.py's default timed region itself: C3 with setQueryBatch(3) on ONE handle (b = 14, E = 14 -> stage_a_mad_batch_kernel<2,3>
    over seven layer pairs; transform launches of 1 344 / 2 352 / 2 268 / 2 016 slices), three different real queries, on two queues
    (8 + 6 bin layers, the default) and on one (bench.py --streams 1, the per-kernel profile passes): every query's 14 result
    ciphertexts equal the oracle's run() of that query alone, and each decrypts to its own client's intersection."""
    run_case(ob, pie_mod, 16384, 4, T32, 1 << 20, 1 << 10, 2, 4949, 2, 14, 14, 20261005, range(14), streams=streams, nq=3)


def test_c3_alternative_row_e20_b10(ob, pie_mod):
    """The reference's second parameter row for |S| = 2^20, |C| = 2^10 (Performance-Evaluation/Parameters1.txt:35: maxPP 10,
    eachCuckooTableSize 20): E = 20 crosses the 15-term renormalisation of stage A's carry-free column accumulators at the real
    ring size; 10 bin layers on two queues (6 + 4)."""
    run_case(ob, pie_mod, 16384, 4, T32, 1 << 20, 1 << 10, 2, 4949, 2, 20, 10, 35, range(10))


def test_c3_alternative_row_batch_of_three(ob, pie_mod):
    """the same row as a batch of three queries per run() (E = 20 in the batched stage A: two accumulator sweeps per thread)"""
    run_case(ob, pie_mod, 16384, 4, T32, 1 << 20, 1 << 10, 2, 4949, 2, 20, 10, 36, range(10), nq=3)


def test_c5_full_size(ob, pie_mod):
    """|S| = 2^24 hashed on the device (e = 13004 positions, B = 26008 slots > 2^14), 2700 + 30 plaintexts encoded (4 GiB
    resident), E = 30 terms per inner product (the 128-bit stage-A kernel, not the carry-free one), K = 3 (two chained
    ct x ct), L = 6 base conversions on folded 2^14 slices, 30 bin layers split over the run queues."""
    b = 30
    budget = run_case(ob, pie_mod, 32768, 6, T32, 1 << 24, 1 << 12, 2, 13004, 3, 30, b, 2024, range(b))   # all 30 bin layers
    assert budget > 0


def test_c5_batch_of_three(ob, pie_mod):
    """What `bench.py --config C5` times by default: three queries per run() at C5's real shape (Parameters1.txt:19 with -K 3).  The
    batched stage A with E = 30 (two accumulator sweeps) writes operand X of the first product lane-ordered into the QP array,
    ntt16_kernel_t<14, ...> reads it there, the second product's X is the first product; L = 6, 30 bin layers over two queues.  Every
    query's 30 result ciphertexts equal the oracle's run() of that query alone."""
    b = 30
    budget = run_case(ob, pie_mod, 32768, 6, T32, 1 << 24, 1 << 12, 2, 13004, 3, 30, b, 2025, range(b), nq=3)
    assert budget > 0


def test_c2_batch_of_three(ob, pie_mod):
    """`bench.py --config C2`'s default timed region (Parameters1.txt:53): three queries per run() on the one-slice-per-limb ring"""
    run_case(ob, pie_mod, 8192, 3, T32, 1 << 16, 1 << 10, 3, 443, 2, 12, 12, 8, range(12), nq=3)


@pytest.mark.parametrize("N,L,t", [(4096, 2, T16), (16384, 4, T32)])
@pytest.mark.parametrize("equals", [True, False])
def test_kat0_reference_test_on_the_gpu(ob, pie_mod, N, L, t, equals):
    """tests/TestBatchedFHEPIE.cpp:54-149 on the GPU, at a 2-prime ring that fits t = 65537 and at the reference test's
    own parameters (:14-41: ring 16384, plaintext modulus 4296540161): k=2, e=1, K=2, E=10, b=20, 100 items; the index
    matrix is built by hand with BOTH slots selecting the element (:101-124).  "Test should output matches twice"
    (:73); none when the element is not in the set (:72-82).  Ciphertexts equal the oracle's bit for bit."""
    pie = pie_mod
    from tests.test_oracle_pie import distinct_items
    o = ob.Oracle(N, L, t)
    rng = np.random.default_rng(122333444455555 % (1 << 32))
    items = distinct_items(rng, t, 100)
    k, e, K, E, b = 2, 1, 2, 10, 20
    present = set(int(x) for x in items)
    elem = int(items[50]) if equals else next(v for v in range(1, t) if v not in present)
    sk = o.keygen(1)
    evk = o.relin_keygen(sk, 2)
    tab = ob.Tabulation(12223222, k + K)
    tbl = ob.hct_build(tab, items, k, e, K, b, E, evict_seed=5)
    idx = np.zeros((K, E, 2, L, N), dtype=np.uint64)
    for h in range(K):
        hi = tab.hash(elem, k + h) % E
        for j in range(E):
            idx[h, j] = o.encrypt_slots(sk, [1, 1] if j == hi else [0, 0], 10 + h * E + j)
    minus = o.encrypt_slots(sk, [-elem, -elem], 9)
    cc = pie.PieContext(N, L, t)
    cc.load_relin_key(evk)
    op = pie.BatchedFHEHIPPIE(cc, hashTable=tbl, shuffle_seed=6, mask_seed=7)   # the reference constructor on the device
    op.setMinusCompareElement(minus)
    op.setIndex(idx)
    op.run()
    got = op.getResultList().copy()
    cc.close()
    ob.hct_shuffle_bins(tbl, 6)
    slots = ob.pack_db(tbl)
    db = np.stack([o.encode_eval(slots[h, bn, j]) for h in range(K) for bn in range(b) for j in range(E)]).reshape(K, b, E, L, N)
    mk = ob.masks(t, b, 2, 7)
    masks = np.stack([o.encode_eval(mk[bn]) for bn in range(b)])
    assert (got == o.pie_run(idx, minus, db, masks, evk)).all()
    matches = 0
    for bn in range(b):
        dec, budget = o.decrypt_slots(sk, got[bn], 2)
        assert budget > 0
        matches += int((dec == 0).sum())
    assert matches == (2 if equals else 0)
