"""C oracle (tier ii, RNS-native) against exact big-integer mathematics (tier i, oracle/exact.py).

The reference holds no numeric vectors for this path (tests/TestBatchedFHEPIE.cpp:139-149 prints
"Matches"), so the oracle is pinned against the mathematical definition of each routine instead.
"""
import numpy as np
import pytest

from oracle import exact as ex

T16 = 65537
T32 = 4296540161  # 2^32 + 2^20 + 2^19 + 1, reference BatchedFHEPSIClient.cpp:29


def cols(arr):
    return [tuple(int(v) for v in arr[:, n]) for n in range(arr.shape[1])]


@pytest.mark.parametrize("N", [1024, 4096, 16384, 32768])
def test_prime_chain_and_roots(ob, N):
    L = 2
    q, p = ob.default_moduli(N, L)
    chain = ex.prime_chain(N, 2 * L + 1)
    assert [int(x) for x in q] + [int(x) for x in p] == chain
    for c in chain:
        assert c < (1 << 60) and c % (2 * N) == 1 and ex.is_prime(c)
    if N <= 4096:  # the O(N) minimal-root scan in pure Python
        o = ob.Oracle(N, L, T16)
        for mi in range(2 * L + 1):
            assert o.psi(mi) == ex.min_primitive_root(chain[mi], N)
        assert o.psi(2 * L + 1) == ex.min_primitive_root(T16, N)


@pytest.mark.parametrize("t", [65537, 4296540161, 1099579260929, 281474981953537])
def test_reference_plaintext_moduli_are_ntt_friendly(ob, t):
    # BatchedFHEPSIClient.cpp:23-38 -- all four must be prime and 1 mod 2N for N = 16384
    assert ex.is_prime(t) and ob.lib().po_is_prime(t) and t % (2 * 16384) == 1


@pytest.mark.parametrize("N,L", [(64, 2), (4096, 2), (16384, 4)])
def test_ntt_matches_evaluation_definition(ob, N, L):
    o = ob.Oracle(N, L, T16 if N <= 4096 else T32)
    rng = np.random.default_rng(N)
    logN = N.bit_length() - 1
    for mi in (0, L, 2 * L, 2 * L + 1):
        q = int(o.moduli[mi])
        a = rng.integers(0, q, N, dtype=np.uint64)
        f = o.ntt(mi, a)
        assert (o.intt(mi, f) == a).all()
        for p in ([0, 1, 2, N // 2, N - 1] if N > 64 else range(N)):
            assert int(f[p]) == ex.ntt_eval_point(a, o.psi(mi), q, p, logN)
    # edge inputs: zeros, all q-1, a delta
    q = int(o.moduli[0])
    z = np.zeros(N, dtype=np.uint64)
    assert (o.ntt(0, z) == 0).all()
    m1 = np.full(N, q - 1, dtype=np.uint64)
    assert (o.intt(0, o.ntt(0, m1)) == m1).all()
    d = z.copy()
    d[0] = 1
    assert (o.ntt(0, d) == 1).all()


def test_twiddle_tables(ob):
    N, L = 256, 2
    o = ob.Oracle(N, L, T16)
    fwd, inv = o.twiddles(0)
    q, psi = int(o.q[0]), o.psi(0)
    for k in range(1, N):
        w = pow(psi, ex.bitrev(k, 8), q)
        assert int(fwd[k]) == w and int(inv[k]) == pow(w, -1, q)


@pytest.mark.parametrize("N,L,t", [(64, 1, T16), (64, 2, T16), (256, 3, T32), (128, 4, T32), (64, 6, T32)])
def test_base_conversions_exact(ob, N, L, t):
    o = ob.Oracle(N, L, t)
    qs = [int(x) for x in o.q]
    ps = [int(x) for x in o.p]
    rng = np.random.default_rng(1000 * N + L)
    xq = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs])
    # force edge columns: 0, Q-1 (= -1), 1, and values a safe 2^-40 away from the +-Q/2 wrap.
    # (Inputs within ~2^-57 (relative) of a rounding tie are the documented domain where the
    # 60-bit fixed-point rounding term may differ from exact rounding -- as OpenFHE's float
    # formulation does; they are not generated here.)
    Q = ex.prod(qs)
    for n, val in enumerate([0, Q - 1, Q // 2 - (Q >> 40), Q // 2 + (Q >> 40), 1]):
        for i, q in enumerate(qs):
            xq[i, n] = val % q
    got = o.expand_q_to_qp(xq)
    want = ex.expand_q_to_qp(cols(xq), qs, ps)
    assert cols(got) == want
    got = o.scale_pq_expand(xq)
    want = ex.scale_pq_expand(cols(xq), qs, ps)
    assert cols(got) == want
    mods = qs + ps
    xqp = np.stack([rng.integers(0, m, N, dtype=np.uint64) for m in mods])
    QP = Q * ex.prod(ps)
    for n, val in enumerate([0, QP - 1, QP // 2 - (QP >> 40), QP // 2 + (QP >> 40), 1]):
        for i, m in enumerate(mods):
            xqp[i, n] = val % m
    got = o.scale_round_tp(xqp)
    want = ex.scale_round_tp(cols(xqp), qs, ps, t)
    assert cols(got) == want


def _coeff_ints(o, ct_eval, qs):
    """EVALUATION ct [c][L][N] -> per component list of CRT-reconstructed coefficients in [0,Q)"""
    out = []
    for comp in ct_eval:
        limbs = np.stack([o.intt(i, comp[i]) for i in range(len(qs))])
        out.append([ex.crt(col, qs) for col in cols(limbs)])
    return out


@pytest.mark.parametrize("N,L,t", [(64, 2, T16), (1024, 2, T16), (512, 3, T32), (4096, 2, T16)])
def test_tensor_product_exact(ob, N, L, t):
    """po_mul_tensor == round(t/P * (a (x) round(P/Q b)))  computed with big integers mod QP"""
    o = ob.Oracle(N, L, t)
    qs = [int(x) for x in o.q]
    ps = [int(x) for x in o.p]
    Q, P = ex.prod(qs), ex.prod(ps)
    QP = Q * P
    rng = np.random.default_rng(7 * N + L)
    sk = o.keygen(5)
    lim = 150 if t == T16 else 1000  # keep |x*y| < t/2
    x = rng.integers(-lim, lim, N // 2)
    y = rng.integers(-lim, lim, N // 2)
    cx, cy = o.encrypt_slots(sk, x, 1), o.encrypt_slots(sk, y, 2)
    got = o.mul_tensor(cx, cy)
    a = [[ex.centered(v, Q) % QP for v in comp] for comp in _coeff_ints(o, cx, qs)]
    b = [[ex.centered(ex.rnd_div(P * ex.centered(v, Q), Q), P) % QP for v in comp] for comp in _coeff_ints(o, cy, qs)]
    d0 = ex.negacyclic_mul_mod(a[0], b[0], QP)
    d1 = [(u + v) % QP for u, v in zip(ex.negacyclic_mul_mod(a[0], b[1], QP), ex.negacyclic_mul_mod(a[1], b[0], QP))]
    d2 = ex.negacyclic_mul_mod(a[1], b[1], QP)
    want = [[ex.rnd_div(t * ex.centered(v, QP), P) % Q for v in d] for d in (d0, d1, d2)]
    assert _coeff_ints(o, got, qs) == want
    # and it decrypts (3-component) to the slot-wise product
    dec, budget = o.decrypt_slots(sk, got, N // 2)
    assert (dec == x * y).all() and budget > 0


@pytest.mark.parametrize("N,L,t", [(64, 2, T16), (1024, 3, T32)])
def test_decrypt_and_relin_against_exact(ob, N, L, t):
    o = ob.Oracle(N, L, t)
    qs = [int(x) for x in o.q]
    rng = np.random.default_rng(3)
    sk = o.keygen(5)
    evk = o.relin_keygen(sk, 6)
    s_coeff = o.intt(0, sk[0])
    q0 = qs[0]
    s = [int(v) if int(v) <= 1 else int(v) - q0 for v in s_coeff]
    assert set(s) <= {-1, 0, 1}
    x = rng.integers(-150, 150, N)
    y = rng.integers(-150, 150, N)
    cx, cy = o.encrypt_slots(sk, x, 1), o.encrypt_slots(sk, y, 2)
    for ct in (cx, o.mul_tensor(cx, cy), o.mul(cx, cy, evk)):
        m_exact, worst = ex.decrypt_exact(_coeff_ints(o, ct, qs), s, qs, t)
        m_oracle, budget = o.decrypt(sk, ct)
        assert [int(v) for v in m_oracle] == m_exact
        assert worst < 0.5
        # the fixed-point budget estimate agrees with the exact noise to within a bit
        import math
        exact_budget = min(58, int(math.floor(-math.log2(2 * worst)))) if worst > 0 else 58
        assert abs(budget - exact_budget) <= 1
    dec, _ = o.decrypt_slots(sk, o.mul(cx, cy, evk), N)
    assert (dec == x * y).all()
