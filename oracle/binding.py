"""ctypes binding of the CPU oracle (oracle/libpieoracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  See oracle/pie_oracle.h for what the oracle
restates (reference BatchedFHEHIPPIE.cpp:9-129 and the OpenFHE BFV-RNS calls it makes) and for
its parity status ("parity unpinned" at ciphertext-bit level).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpieoracle.so")

u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
u32p = C.POINTER(C.c_uint32)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "libpieoracle.so"] + (["-B"] if force else []))
    return _LIB_PATH


BUILD_FLAGS = "-O3 (portable; shipped oracle/libpieoracle.so)"


def prefer_native(outdir=None):
    """bench.py's cpu_baseline leg: rebuild the oracle with -O3 -march=native on the host that is about to time it
     and load that instead of the portable library that travelled here.  Must be called before
    the first lib() call; falls back to the portable build (and says so) when no compiler is available.
    Returns the flags actually in use."""
    global _LIB_PATH, BUILD_FLAGS
    if _lib is not None:
        return BUILD_FLAGS
    import tempfile
    out = os.path.join(outdir or tempfile.mkdtemp(prefix="pieoracle_"), "libpieoracle_native.so")
    cmd = ["gcc", "-O3", "-march=native", "-std=c11", "-fPIC", "-fno-strict-aliasing", "-shared", "-o", out,
           os.path.join(_HERE, "pie_oracle.c"), os.path.join(_HERE, "pie_hashing.c")]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        C.CDLL(out)  # loadable on this host?
        _LIB_PATH = out
        BUILD_FLAGS = "gcc -O3 -march=native, compiled on the host it is timed on"
    except (OSError, subprocess.CalledProcessError):
        pass
    return BUILD_FLAGS


def _p(a, ty=u64p):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ty)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.po_create.restype = C.c_void_p
        L.po_create.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, u64p, u64p]
        L.po_destroy.argtypes = [C.c_void_p]
        L.po_moduli.argtypes = [C.c_void_p, u64p]
        L.po_psi.restype = C.c_uint64
        L.po_psi.argtypes = [C.c_void_p, C.c_uint32]
        L.po_twiddles.argtypes = [C.c_void_p, C.c_uint32, u64p, u64p]
        L.po_slot_positions.argtypes = [C.c_void_p, u32p]
        L.po_gen_primes.restype = C.c_int
        L.po_gen_primes.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, u64p]
        L.po_is_prime.restype = C.c_int
        L.po_is_prime.argtypes = [C.c_uint64]
        L.po_min_root.restype = C.c_uint64
        L.po_min_root.argtypes = [C.c_uint64, C.c_uint32]
        for f in ("po_ntt_fwd", "po_ntt_inv"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_uint32, u64p]
        L.po_encode.restype = C.c_int
        L.po_encode.argtypes = [C.c_void_p, i64p, C.c_uint32, u64p, u64p]
        L.po_decode.argtypes = [C.c_void_p, u64p, i64p, C.c_uint32]
        L.po_keygen.argtypes = [C.c_void_p, C.c_uint64, u64p]
        L.po_relin_keygen.argtypes = [C.c_void_p, u64p, C.c_uint64, u64p]
        L.po_rot_keygen.argtypes = [C.c_void_p, u64p, C.c_uint32, C.c_uint64, u64p]
        L.po_encrypt_sk.argtypes = [C.c_void_p, u64p, u64p, C.c_uint64, u64p]
        L.po_decrypt.restype = C.c_int
        L.po_decrypt.argtypes = [C.c_void_p, u64p, u64p, C.c_uint32, u64p]
        L.po_add.argtypes = [C.c_void_p, u64p, u64p, u64p]
        L.po_mul_plain.argtypes = [C.c_void_p, u64p, u64p, u64p]
        L.po_mul_tensor.argtypes = [C.c_void_p, u64p, u64p, u64p]
        L.po_relin.argtypes = [C.c_void_p, u64p, u64p, u64p]
        L.po_mul.argtypes = [C.c_void_p, u64p, u64p, u64p, u64p]
        L.po_automorph.argtypes = [C.c_void_p, u64p, C.c_uint32, u64p, u64p]
        L.po_rot_index.restype = C.c_uint32
        L.po_rot_index.argtypes = [C.c_void_p, C.c_int32]
        for f in ("po_expand_q_to_qp", "po_scale_pq_expand", "po_scale_round_tp"):
            getattr(L, f).argtypes = [C.c_void_p, u64p, u64p]
        L.po_pie_run.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, u64p, u64p, u64p, u64p, u64p, u64p,
                                 C.c_uint32, C.c_uint32]
        # hashing layer
        L.ph_tab_create.restype = C.c_void_p
        L.ph_tab_create.argtypes = [C.c_uint64, C.c_uint32]
        L.ph_tab_destroy.argtypes = [C.c_void_p]
        L.ph_tab_hash.restype = C.c_uint64
        L.ph_tab_hash.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.ph_hct_build.restype = C.c_int
        L.ph_hct_build.argtypes = [C.c_void_p, u64p, C.c_size_t] + [C.c_uint32] * 5 + [C.c_uint64, u64p]
        L.ph_hct_shuffle_bins.argtypes = [u64p] + [C.c_uint32] * 5 + [C.c_uint64]
        L.ph_pack_db.argtypes = [u64p] + [C.c_uint32] * 5 + [i64p]
        L.ph_masks.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, i64p]
        L.ph_client_build.restype = C.c_int
        L.ph_client_build.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, u64p]
        L.ph_client_vectors.argtypes = [C.c_void_p, u64p] + [C.c_uint32] * 4 + [i64p, i64p]
        L.ph_client_scan.restype = C.c_size_t
        L.ph_client_scan.argtypes = [u64p, C.c_uint32, C.c_uint32, C.c_uint32, i64p, u64p]
        _lib = L
    return _lib


def gen_primes(N, count, below=1 << 60):
    out = np.zeros(count, dtype=np.uint64)
    rc = lib().po_gen_primes(N, below, count, _p(out))
    if rc:
        raise ValueError("prime chain exhausted")
    return out


def default_moduli(N, L):
    """(q[L], p[L+1]): largest primes < 2^60 congruent 1 mod 2N, descending, P continuing after Q."""
    ch = gen_primes(N, 2 * L + 1)
    return ch[:L].copy(), ch[L:].copy()


class Oracle:
    """BFV-RNS context of the CPU oracle.  All limb arrays are numpy uint64, C-contiguous."""

    def __init__(self, N, L, t, q=None, p=None):
        self.N, self.L, self.t, self.M = N, L, int(t), 2 * L + 1
        qa = None if q is None else np.ascontiguousarray(q, dtype=np.uint64)
        pa = None if p is None else np.ascontiguousarray(p, dtype=np.uint64)
        self._h = lib().po_create(N, L, int(t), _p(qa), _p(pa))
        if not self._h:
            raise ValueError("po_create failed (bad N/L/t/moduli)")
        m = np.zeros(self.M + 1, dtype=np.uint64)
        lib().po_moduli(self._h, _p(m))
        self.q = m[:L].copy()
        self.p = m[L:self.M].copy()
        self.moduli = m

    def __del__(self):
        if getattr(self, "_h", None):
            lib().po_destroy(self._h)
            self._h = None

    # --- tables
    def psi(self, mi):
        return int(lib().po_psi(self._h, mi))

    def twiddles(self, mi):
        f = np.zeros(self.N, dtype=np.uint64)
        i = np.zeros(self.N, dtype=np.uint64)
        lib().po_twiddles(self._h, mi, _p(f), _p(i))
        return f, i

    def slot_positions(self):
        pos = np.zeros(self.N, dtype=np.uint32)
        lib().po_slot_positions(self._h, _p(pos, u32p))
        return pos

    # --- transforms
    def ntt(self, mi, a):
        a = np.array(a, dtype=np.uint64)
        lib().po_ntt_fwd(self._h, mi, _p(a))
        return a

    def intt(self, mi, a):
        a = np.array(a, dtype=np.uint64)
        lib().po_ntt_inv(self._h, mi, _p(a))
        return a

    # --- encoding
    def encode(self, slots):
        s = np.ascontiguousarray(slots, dtype=np.int64)
        coeff = np.zeros(self.N, dtype=np.uint64)
        ev = np.zeros((self.L, self.N), dtype=np.uint64)
        rc = lib().po_encode(self._h, _p(s, i64p), len(s), _p(coeff), _p(ev))
        if rc:
            raise ValueError("encode: too many slots or |value| >= t")
        return coeff, ev

    def encode_eval(self, slots):
        return self.encode(slots)[1]

    def decode(self, coeff, nslots):
        out = np.zeros(nslots, dtype=np.int64)
        lib().po_decode(self._h, _p(np.ascontiguousarray(coeff, dtype=np.uint64)), _p(out, i64p), nslots)
        return out

    # --- keys / enc / dec
    def keygen(self, seed):
        sk = np.zeros((self.L, self.N), dtype=np.uint64)
        lib().po_keygen(self._h, seed, _p(sk))
        return sk

    def relin_keygen(self, sk, seed):
        evk = np.zeros((self.L, 2, self.L, self.N), dtype=np.uint64)
        lib().po_relin_keygen(self._h, _p(sk), seed, _p(evk))
        return evk

    def rot_keygen(self, sk, g, seed):
        rk = np.zeros((self.L, 2, self.L, self.N), dtype=np.uint64)
        lib().po_rot_keygen(self._h, _p(sk), g, seed, _p(rk))
        return rk

    def rot_index(self, r):
        return int(lib().po_rot_index(self._h, r))

    def encrypt(self, sk, coeff_t, seed):
        ct = np.zeros((2, self.L, self.N), dtype=np.uint64)
        lib().po_encrypt_sk(self._h, _p(sk), _p(np.ascontiguousarray(coeff_t, dtype=np.uint64)), seed, _p(ct))
        return ct

    def encrypt_slots(self, sk, slots, seed):
        return self.encrypt(sk, self.encode(slots)[0], seed)

    def decrypt(self, sk, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        out = np.zeros(self.N, dtype=np.uint64)
        budget = lib().po_decrypt(self._h, _p(sk), _p(ct), ct.shape[0], _p(out))
        return out, budget

    def decrypt_slots(self, sk, ct, nslots):
        coeff, budget = self.decrypt(sk, ct)
        return self.decode(coeff, nslots), budget

    # --- homomorphic ops
    def add(self, x, y):
        out = np.zeros_like(x)
        lib().po_add(self._h, _p(x), _p(y), _p(out))
        return out

    def mul_plain(self, x, pt):
        out = np.zeros_like(x)
        lib().po_mul_plain(self._h, _p(x), _p(pt), _p(out))
        return out

    def mul_tensor(self, x, y):
        out = np.zeros((3, self.L, self.N), dtype=np.uint64)
        lib().po_mul_tensor(self._h, _p(x), _p(y), _p(out))
        return out

    def relin(self, ct3, evk):
        out = np.zeros((2, self.L, self.N), dtype=np.uint64)
        lib().po_relin(self._h, _p(ct3), _p(evk), _p(out))
        return out

    def mul(self, x, y, evk):
        out = np.zeros((2, self.L, self.N), dtype=np.uint64)
        lib().po_mul(self._h, _p(x), _p(y), _p(evk), _p(out))
        return out

    def automorph(self, x, g, rk):
        out = np.zeros((2, self.L, self.N), dtype=np.uint64)
        lib().po_automorph(self._h, _p(x), g, _p(rk), _p(out))
        return out

    def expand_q_to_qp(self, xq):
        out = np.zeros((self.M, self.N), dtype=np.uint64)
        lib().po_expand_q_to_qp(self._h, _p(np.ascontiguousarray(xq)), _p(out))
        return out

    def scale_pq_expand(self, xq):
        out = np.zeros((self.M, self.N), dtype=np.uint64)
        lib().po_scale_pq_expand(self._h, _p(np.ascontiguousarray(xq)), _p(out))
        return out

    def scale_round_tp(self, xqp):
        out = np.zeros((self.L, self.N), dtype=np.uint64)
        lib().po_scale_round_tp(self._h, _p(np.ascontiguousarray(xqp)), _p(out))
        return out

    # --- the hot path
    def pie_run(self, idx, minus, db, masks, evk, bin_begin=0, bin_end=None):
        K, E = idx.shape[0], idx.shape[1]
        b = db.shape[1]
        assert db.shape[0] == K and db.shape[2] == E and masks.shape[0] == b
        if bin_end is None:
            bin_end = b
        out = np.zeros((b, 2, self.L, self.N), dtype=np.uint64)
        lib().po_pie_run(self._h, K, b, E, _p(idx), _p(minus), _p(db), _p(masks), _p(evk), _p(out), bin_begin, bin_end)
        return out


class Tabulation:
    def __init__(self, seed, nfun):
        self._h = lib().ph_tab_create(seed, nfun)
        self.nfun = nfun

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ph_tab_destroy(self._h)
            self._h = None

    def hash(self, x, hf):
        return int(lib().ph_tab_hash(self._h, int(x), hf))


def hct_build(tab, items, k, e, K, b, E, evict_seed=1):
    items = np.ascontiguousarray(items, dtype=np.uint64)
    tbl = np.zeros((k, e, K, b, E), dtype=np.uint64)
    rc = lib().ph_hct_build(tab._h, _p(items), len(items), k, e, K, b, E, evict_seed, _p(tbl))
    if rc:
        raise RuntimeError("(Blocked) Cuckoo hashing error")
    return tbl


def hct_shuffle_bins(tbl, seed):
    k, e, K, b, E = tbl.shape
    lib().ph_hct_shuffle_bins(_p(tbl), k, e, K, b, E, seed)


def pack_db(tbl):
    k, e, K, b, E = tbl.shape
    slots = np.zeros((K, b, E, k * e), dtype=np.int64)
    lib().ph_pack_db(_p(tbl), k, e, K, b, E, _p(slots, i64p))
    return slots


def masks(t, b, B, seed):
    out = np.zeros((b, B), dtype=np.int64)
    lib().ph_masks(int(t), b, B, seed, _p(out, i64p))
    return out


def client_build(tab, items, k, e, evict_seed=2):
    items = np.ascontiguousarray(items, dtype=np.uint64)
    ctab = np.zeros((k, e), dtype=np.uint64)
    rc = lib().ph_client_build(tab._h, _p(items), len(items), k, e, evict_seed, _p(ctab))
    if rc:
        raise RuntimeError("(Blocked) Cuckoo hashing error")
    return ctab


def client_vectors(tab, ctab, K, E):
    k, e = ctab.shape
    index = np.zeros((K, E, k * e), dtype=np.int64)
    minus = np.zeros(k * e, dtype=np.int64)
    lib().ph_client_vectors(tab._h, _p(ctab), k, e, K, E, _p(index, i64p), _p(minus, i64p))
    return index, minus


def client_scan(ctab, decrypted):
    k, e = ctab.shape
    b = decrypted.shape[0]
    dec = np.ascontiguousarray(decrypted, dtype=np.int64)
    out = np.zeros(k * e, dtype=np.uint64)
    n = lib().ph_client_scan(_p(ctab), k, e, b, _p(dec, i64p), _p(out))
    return out[:n].copy()


def fhe_pie_run(o, idx, slots, masks, rot_keys):
    """FHEHIPPIE::run() for one operator (reference FHEHIPPIE.cpp:61-77), composed from the C primitives.

    idx [K][2][L][N] index ciphertexts, slots [K][b][E+1] packed rows (FHEHIPPIE.cpp:41-51), masks [K][b],
    rot_keys {rotation index: BV key}.  EvalInnerProduct(ct, pt, b) = EvalMult then ceil(log2 b) steps of
    ct += rotate(ct, 2^r); EvalMerge masks slot 0 of ciphertext i and rotates it by -i [OFHE-UNVERIFIED].
    Returns [K][2][L][N] in hash-function order (the caller applies permutationVector).
    """
    K, nb, _ = slots.shape
    R = 0
    while (1 << R) < nb:
        R += 1
    e0 = o.encode_eval(np.array([1], dtype=np.int64))
    out = []
    for hf in range(K):
        evals = []
        for bn in range(nb):
            ct = o.mul_plain(idx[hf], o.encode_eval(slots[hf, bn]))
            for r in range(R):
                ct = o.add(ct, o.automorph(ct, o.rot_index(1 << r), rot_keys[1 << r]))
            evals.append(ct)
        merged = o.mul_plain(evals[0], e0)
        for i in range(1, nb):
            merged = o.add(merged, o.automorph(o.mul_plain(evals[i], e0), o.rot_index(-i), rot_keys[-i]))
        out.append(o.mul_plain(merged, o.encode_eval(masks[hf])))
    return np.stack(out)


def fhe_pie_index_vectors(tab, x, k, K, E):
    """Index vectors of the non-batched client for one element (reference SimpleFHEPSIClient.cpp:124-154):
    one-hot at hash_{k+hf}(x) mod E, last slot -x; the dummy element 1 gets an all-zero one-hot part."""
    v = np.zeros((K, E + 1), dtype=np.int64)
    for hf in range(K):
        if x != 1:
            v[hf, tab.hash(x, k + hf) % E] = 1
        v[hf, E] = -int(x)
    return v
