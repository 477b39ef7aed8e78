"""Decrypted-slot semantics of the oracle's restated hot path: the reference's own checks.

KAT-0: tests/TestBatchedFHEPIE.cpp:54-149  (k=2,e=1,K=2,E=10,b=20, 100 items, client item = item
       #50 replicated in both slots, "Test should output matches twice"; with
       clientElementEquals=false (:72-82) no match).
KAT-1: tests/TestOpenFHE.cpp:36-65  (slot-wise add / mult / rotations of 12-vectors).
KAT-2: src/Client/PSIClient.hpp:142-164 (computed intersection is a permutation of the true one).
"""
import numpy as np
import pytest

T16 = 65537
T32 = 4296540161


def distinct_items(rng, t, n):
    """n distinct non-zero items < t (0 is the empty-slot sentinel, CuckooHashTable.cpp:89-90)"""
    out = np.unique(rng.integers(1, t, 2 * n + 16, dtype=np.uint64))
    rng.shuffle(out)
    assert len(out) >= n
    return out[:n].copy()


def pie_inputs(ob, o, sk, items, client_items, k, e, K, E, b, hash_seed, seeds=(1, 2, 3, 4)):
    """server DB + masks (reference BatchedFHEHIPPIE ctor) and client ciphertexts
    (reference BatchedFHEPSIClient::runOfflinePhase) for one query"""
    tab = ob.Tabulation(hash_seed, k + K)
    tbl = ob.hct_build(tab, items, k, e, K, b, E, evict_seed=seeds[0])
    ob.hct_shuffle_bins(tbl, seeds[1])
    B = k * e
    slots = ob.pack_db(tbl)
    db = np.stack([o.encode_eval(slots[h, bn, j]) for h in range(K) for bn in range(b) for j in range(E)])
    db = db.reshape(K, b, E, o.L, o.N)
    mk = ob.masks(o.t, b, B, seeds[2])
    masks = np.stack([o.encode_eval(mk[bn]) for bn in range(b)])
    ctab = ob.client_build(tab, client_items, k, e, evict_seed=seeds[3])
    index, minus = ob.client_vectors(tab, ctab, K, E)
    idx = np.stack([o.encrypt_slots(sk, index[h, j], 100 + h * E + j) for h in range(K) for j in range(E)])
    idx = idx.reshape(K, E, 2, o.L, o.N)
    mct = o.encrypt_slots(sk, minus, 99)
    return dict(tab=tab, tbl=tbl, db=db, masks=masks, ctab=ctab, idx=idx, minus=mct, B=B)


@pytest.mark.parametrize("N,L,t,equals", [(4096, 2, T16, True), (4096, 2, T16, False), (2048, 4, T32, True)])
def test_kat0_reference_test_shape(ob, N, L, t, equals):
    o = ob.Oracle(N, L, t)
    rng = np.random.default_rng(122333444455555 % (1 << 32))
    items = distinct_items(rng, t, 100)
    k, e, K, E, b = 2, 1, 2, 10, 20
    elem = int(items[50]) if equals else int(next(v for v in range(1, t) if v not in set(int(x) for x in items)))
    sk = o.keygen(1)
    evk = o.relin_keygen(sk, 2)
    # the reference test builds the index matrix by hand with BOTH slots holding the element
    # (TestBatchedFHEPIE.cpp:101-124); the client Cuckoo table would place it in one slot only.
    tab = ob.Tabulation(12223222, k + K)
    tbl = ob.hct_build(tab, items, k, e, K, b, E, evict_seed=5)
    ob.hct_shuffle_bins(tbl, 6)
    slots = ob.pack_db(tbl)
    db = np.stack([o.encode_eval(slots[h, bn, j]) for h in range(K) for bn in range(b) for j in range(E)])
    db = db.reshape(K, b, E, L, N)
    mk = ob.masks(t, b, 2, 7)
    masks = np.stack([o.encode_eval(mk[bn]) for bn in range(b)])
    idx = np.zeros((K, E, 2, L, N), dtype=np.uint64)
    for h in range(K):
        hi = tab.hash(elem, k + h) % E
        for j in range(E):
            v = [1, 1] if j == hi else [0, 0]
            idx[h, j] = o.encrypt_slots(sk, v, 10 + h * E + j)
    minus = o.encrypt_slots(sk, [-elem, -elem], 9)
    res = o.pie_run(idx, minus, db, masks, evk)
    matches = 0
    for bn in range(b):
        dec, budget = o.decrypt_slots(sk, res[bn], 2)
        assert budget > 0
        matches += int((dec == 0).sum())
    assert matches == (2 if equals else 0)  # "Test should output matches twice"


def test_kat1_openfhe_smoke_vectors(ob):
    """tests/TestOpenFHE.cpp:36-65: add, mult and rotations of the three 12-vectors"""
    o = ob.Oracle(4096, 2, T16)
    sk = o.keygen(3)
    evk = o.relin_keygen(sk, 4)
    v1 = np.arange(1, 13)
    v2 = np.array([3, 2, 1, 4, 5, 6, 7, 8, 9, 10, 11, 12])
    v3 = np.array([1, 2, 5, 2, 5, 6, 7, 8, 9, 10, 11, 12])
    c1, c2, c3 = (o.encrypt_slots(sk, v, s) for v, s in ((v1, 1), (v2, 2), (v3, 3)))
    dec, _ = o.decrypt_slots(sk, o.add(o.add(c1, c2), c3), 12)
    assert (dec == v1 + v2 + v3).all()
    dec, budget = o.decrypt_slots(sk, o.mul(c1, c2, evk), 12)
    assert budget > 0 and (dec == v1 * v2).all()
    for r in (1, 2, -1, -2):
        g = o.rot_index(r)
        rk = o.rot_keygen(sk, g, 50 + r)
        dec, budget = o.decrypt_slots(sk, o.automorph(c1, g, rk), 12)
        full = np.zeros(4096 // 2, dtype=np.int64)
        full[:12] = v1
        assert budget > 0 and (dec == np.roll(full, -r)[:12]).all()


@pytest.mark.parametrize("N,L,t,nS,nC,k,e,K,E,b", [
    (4096, 2, T16, 300, 16, 2, 12, 2, 6, 6),
    (2048, 3, T32, 500, 24, 3, 10, 2, 8, 5),
    (1024, 4, T32, 200, 10, 2, 8, 3, 6, 4),  # K=3: two sequential ct x ct (config C5's depth)
])
def test_kat2_end_to_end_intersection(ob, N, L, t, nS, nC, k, e, K, E, b):
    o = ob.Oracle(N, L, t)
    rng = np.random.default_rng(123456789 % (1 << 32))
    universe = distinct_items(rng, t, nS + nC)
    server = universe[:nS].copy()
    ninter = nC // 2 + 1
    client = np.concatenate([server[:ninter], universe[nS:nS + nC - ninter]])
    rng.shuffle(client)
    sk = o.keygen(11)
    evk = o.relin_keygen(sk, 12)
    d = pie_inputs(ob, o, sk, server, client, k, e, K, E, b, hash_seed=987654321)
    res = o.pie_run(d["idx"], d["minus"], d["db"], d["masks"], evk)
    dec = np.stack([o.decrypt_slots(sk, res[bn], d["B"])[0] for bn in range(b)])
    budget = min(o.decrypt_slots(sk, res[bn], d["B"])[1] for bn in range(b))
    assert budget > 0
    got = ob.client_scan(d["ctab"], dec)
    assert sorted(int(x) for x in got) == sorted(int(x) for x in server[:ninter])  # "Set matches!"
    # sharded evaluation over bin layers reproduces the same ciphertexts
    part = o.pie_run(d["idx"], d["minus"], d["db"], d["masks"], evk, 0, b // 2)
    part2 = o.pie_run(d["idx"], d["minus"], d["db"], d["masks"], evk, b // 2, b)
    assert (part[: b // 2] == res[: b // 2]).all() and (part2[b // 2:] == res[b // 2:]).all()


def test_c1_baseline_config_at_full_size(ob):
    """BASELINE.json config 1 on the oracle alone (its "plumbing, no GPU" case): N=4096, 2 primes, t=65537 (a 33-bit t
    does not fit two primes), |S|=2^12, |C|=2^8, k=3, e=110, K=2, E=b=7 (derived per BASELINE.md
    section 3).  The intersection of 129 items comes back exactly, with noise budget to spare."""
    N, L, t = 4096, 2, T16
    nS, nC, k, e, K, E, b = 1 << 12, 1 << 8, 3, 110, 2, 7, 7
    o = ob.Oracle(N, L, t)
    rng = np.random.default_rng(5)
    universe = distinct_items(rng, t, nS + nC)
    server = universe[:nS].copy()
    ninter = nC // 2 + 1
    client = np.concatenate([server[:ninter], universe[nS:nS + nC - ninter]])
    rng.shuffle(client)
    sk = o.keygen(11)
    evk = o.relin_keygen(sk, 12)
    d = pie_inputs(ob, o, sk, server, client, k, e, K, E, b, hash_seed=987654321)
    res = o.pie_run(d["idx"], d["minus"], d["db"], d["masks"], evk)
    outs = [o.decrypt_slots(sk, res[bn], d["B"]) for bn in range(b)]
    assert min(bud for _, bud in outs) > 0
    got = ob.client_scan(d["ctab"], np.stack([dec for dec, _ in outs]))
    assert len(got) == ninter == 129
    assert sorted(int(x) for x in got) == sorted(int(x) for x in server[:ninter])


def test_tabulation_hash_is_std_mt19937(ob):
    """TabulationHashing.cpp:22-33: std::mt19937(seed) + uniform_int_distribution<uint64_t>.
    First outputs of mt19937(5489) are 3499211612, 581869302 (the C++ standard's check value
    is the 10000th = 4123659995); the 64-bit draw is high word first under libstdc++."""
    tab = ob.Tabulation(5489, 1)
    # x = 0 selects entry [i][0] of all 16 byte tables -> XOR of draws number 256*i
    import ctypes
    h0 = tab.hash(0, 0)
    # recompute with a tiny Python mt19937
    mt = [5489]
    for i in range(1, 624):
        mt.append((1812433253 * (mt[-1] ^ (mt[-1] >> 30)) + i) & 0xFFFFFFFF)
    out = []
    state = mt[:]
    def twist(s):
        for i in range(624):
            y = (s[i] & 0x80000000) | (s[(i + 1) % 624] & 0x7FFFFFFF)
            v = s[(i + 397) % 624] ^ (y >> 1)
            if y & 1:
                v ^= 0x9908B0DF
            s[i] = v
    def temper(y):
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y
    need = 2 * 256 * 16
    while len(out) < need:
        twist(state)
        out.extend(temper(y) for y in state)
    assert out[0] == 3499211612 and out[1] == 581869302
    draws = [(out[2 * i] << 32) | out[2 * i + 1] for i in range(256 * 16)]
    want = 0
    for i in range(16):
        want ^= draws[256 * i]
    assert h0 == want
    x = 0x0123456789ABCDEF
    want = 0
    for i in range(16):
        want ^= draws[256 * i + ((x >> (8 * i)) & 0xFF if i < 8 else 0)]
    assert tab.hash(x, 0) == want


# ---- KAT-3: rotation-based sibling operator ---------------------------------------------------------------
def fhepie_case(ob, o, sk, t, K, E, nitems, present, hash_seed=424242, key_seed=50):
    """tests/TestFHEPIE.cpp:52-123 at a reduced shape: flat blocked Cuckoo table [K][b=E][E], one client element,
    hand-built index vectors (one-hot at hash_hf(x) mod E, last slot -x), EvalSum + EvalRotate keys"""
    rng = np.random.default_rng(20240607)
    items = distinct_items(rng, t, nitems + 1)
    table_items, absent = items[:nitems], int(items[nitems])
    tab = ob.Tabulation(hash_seed, K + 1)
    # one sub-table of the nested structure: [K][b][E] with the inner hash ids 1..K (k = 1 outer function, e = 1)
    tbl = ob.hct_build(tab, table_items, 1, 1, K, E, E, evict_seed=3)[0, 0]
    elem = int(table_items[nitems // 2]) if present else absent
    index = ob.fhe_pie_index_vectors(tab, elem, 1, K, E)
    idx = np.stack([o.encrypt_slots(sk, index[hf], 70 + hf) for hf in range(K)])
    R = int(np.ceil(np.log2(E)))
    rots = [1 << r for r in range(R)] + [-i for i in range(1, E)]
    keys = {r: o.rot_keygen(sk, o.rot_index(r), key_seed + i) for i, r in enumerate(rots)}
    slots = np.ones((K, E, E + 1), dtype=np.int64)
    slots[:, :, :E] = tbl.astype(np.int64)
    masks = np.random.default_rng(6).integers(1, t, size=(K, E), dtype=np.int64)
    return dict(tbl=tbl, elem=elem, idx=idx, keys=keys, slots=slots, masks=masks, tab=tab)


@pytest.mark.parametrize("present", [True, False])
def test_kat3_rotation_based_pie(ob, present):
    """FHEHIPPIE::run (FHEHIPPIE.cpp:61-77): exactly one zero among the first b slots of the K results when the
    element is in the table ("Matches", TestFHEPIE.cpp:125-137), none otherwise"""
    N, L, t, K, E = 2048, 3, T16, 3, 12
    o = ob.Oracle(N, L, t)
    sk = o.keygen(1)
    c = fhepie_case(ob, o, sk, t, K, E, 150, present)
    res = ob.fhe_pie_run(o, c["idx"], c["slots"], c["masks"], c["keys"])
    zeros = 0
    for hf in range(K):
        dec, budget = o.decrypt_slots(sk, res[hf], E)
        assert budget > 0
        # slot i = mask_i * (T[hf][i][h(x)] - x)
        hi = c["tab"].hash(c["elem"], 1 + hf) % E
        want = [(int(c["masks"][hf, i]) * ((int(c["tbl"][hf, i, hi]) - c["elem"]) % t)) % t for i in range(E)]
        got = [int(v) % t for v in dec]
        assert got == want
        zeros += int((dec == 0).sum())
    assert zeros == (1 if present else 0)
