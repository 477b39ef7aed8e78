"""Exact big-integer tier of the oracle (Python ints, no RNS shortcuts).

TEST INFRASTRUCTURE ONLY.  This is "tier (i)" of SURVEY.md section 7 step 3: the mathematical
definition of every RNS routine in oracle/pie_oracle.c, evaluated with CRT reconstruction and
exact rational rounding.  tests/test_oracle_exact.py checks the C oracle (tier ii, the one the
GPU must match bit for bit) against it.  Rounding convention everywhere: round(x) = floor(x + 1/2).
"""
from functools import reduce


def prod(xs):
    return reduce(lambda a, b: a * b, xs, 1)


def is_prime(n):
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in small:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in small:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def prime_chain(N, count, below=1 << 60):
    """largest primes < below with q = 1 (mod 2N), descending"""
    out = []
    step = 2 * N
    c = ((below - 2) // step) * step + 1
    while len(out) < count:
        if is_prime(c):
            out.append(c)
        c -= step
    return out


def min_primitive_root(q, N):
    """smallest primitive 2N-th root of unity mod q"""
    e = (q - 1) // (2 * N)
    for g in range(2, 1000):
        x = pow(g, e, q)
        if pow(x, N, q) == q - 1:
            break
    else:
        raise ValueError("no root")
    x2 = x * x % q
    best = cur = x
    for _ in range(1, N):
        cur = cur * x2 % q
        best = min(best, cur)
    return best


def bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def rnd_div(num, den):
    """floor(num/den + 1/2) for den > 0"""
    return (2 * num + den) // (2 * den)


def crt(residues, moduli):
    """x in [0, prod(moduli))"""
    M = prod(moduli)
    x = 0
    for r, m in zip(residues, moduli):
        Mi = M // m
        x += int(r) * Mi * pow(Mi, -1, m)
    return x % M


def centered(x, M):
    """representative in [-M/2, M/2) (matches v = floor(sum y_i/q_i + 1/2) in the RNS code)"""
    x %= M
    return x - M if 2 * x >= M else x


def ntt_eval_point(a, psi, q, p, logN):
    """what EVALUATION position p must hold: a(psi^(2*bitrev(p)+1)) mod q"""
    x = pow(psi, 2 * bitrev(p, logN) + 1, q)
    s = 0
    for c in reversed(a):
        s = (s * x + int(c)) % q
    return s


def expand_q_to_qp(xq_cols, qs, ps):
    """per coefficient: centred lift of x mod Q, reduced into every p_j.  xq_cols: list of L-tuples."""
    Q = prod(qs)
    out = []
    for col in xq_cols:
        xh = centered(crt(col, qs), Q)
        out.append(tuple(int(c) for c in col) + tuple(xh % p for p in ps))
    return out


def scale_pq_expand(xq_cols, qs, ps):
    """per coefficient: x' = round(P xhat / Q) taken mod P, centred, then reduced into Q and P limbs"""
    Q, P = prod(qs), prod(ps)
    out = []
    for col in xq_cols:
        xh = centered(crt(col, qs), Q)
        xs = centered(rnd_div(P * xh, Q), P)
        out.append(tuple(xs % q for q in qs) + tuple(xs % p for p in ps))
    return out


def scale_round_tp(xqp_cols, qs, ps, t):
    """per coefficient: round(t dhat / P) mod every q_k, dhat the centred lift mod QP"""
    P = prod(ps)
    QP = prod(qs) * P
    mods = list(qs) + list(ps)
    out = []
    for col in xqp_cols:
        dh = centered(crt(col, mods), QP)
        r = rnd_div(t * dh, P)
        out.append(tuple(r % q for q in qs))
    return out


def negacyclic_mul_mod(a, b, M):
    """a*b mod (X^N+1, M) for coefficient lists in [0, M), by Kronecker substitution"""
    N = len(a)
    width = (2 * M.bit_length() + N.bit_length() + 8 + 7) // 8  # bytes per packed coefficient
    A = int.from_bytes(b"".join(int(x).to_bytes(width, "little") for x in a), "little")
    B = int.from_bytes(b"".join(int(x).to_bytes(width, "little") for x in b), "little")
    Cb = (A * B).to_bytes(width * 2 * N, "little")
    c = [int.from_bytes(Cb[i * width:(i + 1) * width], "little") for i in range(2 * N)]
    return [(c[i] - c[i + N]) % M for i in range(N)]


def decrypt_exact(c_polys, s, qs, t):
    """c_polys: list of coefficient-form polys mod Q (as CRT-reconstructed ints in [0,Q));
    s: signed ternary secret.  Returns (message coefficients mod t, max |t x/Q - round| as a float)."""
    Q = prod(qs)
    N = len(s)
    sm = [x % Q for x in s]
    acc = list(c_polys[0])
    spow = sm
    for k in range(1, len(c_polys)):
        term = negacyclic_mul_mod(c_polys[k], spow, Q)
        acc = [(x + y) % Q for x, y in zip(acc, term)]
        if k + 1 < len(c_polys):
            spow = negacyclic_mul_mod(spow, sm, Q)
    out, worst = [], 0.0
    for x in acc:
        xh = centered(x, Q)
        m = rnd_div(t * xh, Q)
        worst = max(worst, abs((t * xh - m * Q) / Q))
        out.append(m % t)
    assert len(out) == N
    return out, worst
