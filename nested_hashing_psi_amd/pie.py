"""Host-side mirror of the reference operator interface, over the C ABI.

  PieContext        <-> lbcrypto::CryptoContext<DCRTPoly> as the server holds it
                        (reference src/Server/FHE/BatchedFHEPSIServer.hpp:21, set at .cpp:21-54)
  BatchedFHEHIPPIE  <-> class BatchedFHEHIPPIE, reference
                        src/Common/Crypto/PrivateIndexedEqualityCheck/BatchedFHEHIPPIE.hpp:18-49:
                        same method names (setIndex, setMinusCompareElement, run, getResultList),
                        same argument meaning, same error behaviour (ValueError where the
                        reference throws std::invalid_argument, RuntimeError for runtime_error).
Ciphertexts / plaintexts are numpy uint64 limb arrays in EVALUATION format, the layout of OpenFHE's
DCRTPoly towers.  Everything executes in libpiehip.so on the GPU; nothing here computes.
"""
import ctypes as C
import secrets

import numpy as np

from ._lib import NKERNELS, f64p, i32p, i64p, lib, u32p, u64p

EINVAL, ESTATE, EHIP, ENOMEM, EHASH = -1, -2, -3, -4, -5


def _check(rc):
    if rc == 0:
        return
    msg = lib().piehip_last_error().decode()
    if rc == EINVAL:
        raise ValueError(msg)          # reference: std::invalid_argument
    if rc == ENOMEM:
        raise MemoryError(msg)
    if rc == EHASH:
        raise RuntimeError(msg)        # reference: runtime_error("(Blocked) Cuckoo hashing error")
    raise RuntimeError(msg)            # reference: std::runtime_error / OpenFHE exceptions


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


def tabulation_hash(hash_seed, nfun, hf, x):
    """TabulationHashing::hashWithIndicator (reference TabulationHashing.cpp:45-54), host side"""
    xa, xp = _u64(np.atleast_1d(x))
    out = np.zeros_like(xa)
    _check(lib().piehip_tabulation_hash(hash_seed, nfun, hf, xp, len(xa), out.ctypes.data_as(u64p)))
    return out


def rccl_unique_id():
    """ncclGetUniqueId through the library's RCCL binding: 128 bytes for piehip_rccl_init on every rank"""
    buf = (C.c_ubyte * 128)()
    _check(lib().piehip_rccl_unique_id(buf))
    return bytes(buf)


def rccl_bin_slice(b_total, nranks, rank):
    lo, hi = C.c_uint32(), C.c_uint32()
    _check(lib().piehip_rccl_bin_slice(int(b_total), int(nranks), int(rank), C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def default_moduli(N, L):
    q = np.zeros(L, dtype=np.uint64)
    p = np.zeros(L + 1, dtype=np.uint64)
    _check(lib().piehip_default_moduli(N, L, q.ctypes.data_as(u64p), p.ctypes.data_as(u64p)))
    return q, p


class PieContext:
    """BFV-RNS evaluation context on one GPU (ring dimension N, L primes, plaintext modulus t)."""

    def __init__(self, N, L, t, q=None, p=None, device=0, stream=None):
        self.N, self.L, self.t, self.M = N, L, int(t), 2 * L + 1
        self._h = C.c_void_p()
        qa = pa = None
        if q is not None:
            qa_arr, qa = _u64(q)
            pa_arr, pa = _u64(p)
        _check(lib().piehip_create(C.byref(self._h), N, L, int(t), qa, pa, device, stream))
        m = np.zeros(self.M + 1, dtype=np.uint64)
        _check(lib().piehip_get_moduli(self._h, m.ctypes.data_as(u64p)))
        self.moduli = m
        self.q, self.p = m[:L].copy(), m[L:self.M].copy()

    def close(self):
        if getattr(self, "_h", None):
            try:
                rc = lib().piehip_destroy(self._h)
            except TypeError:  # interpreter shutdown: module globals already cleared
                rc = 0
            if rc:          # refused: query slots are still attached to this context's database (close them first)
                _check(rc)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except RuntimeError:
            pass

    # -- tables (for cross-checks)
    def psi(self, mi):
        v = C.c_uint64()
        _check(lib().piehip_get_root(self._h, mi, C.byref(v)))
        return int(v.value)

    def twiddles(self, mi):
        f = np.zeros(self.N, dtype=np.uint64)
        i = np.zeros(self.N, dtype=np.uint64)
        _check(lib().piehip_get_twiddles(self._h, mi, f.ctypes.data_as(u64p), i.ctypes.data_as(u64p)))
        return f, i

    def slot_positions(self):
        pos = np.zeros(self.N, dtype=np.uint32)
        _check(lib().piehip_get_slot_positions(self._h, pos.ctypes.data_as(u32p)))
        return pos

    # -- the OpenFHE primitives under run()
    def ntt(self, limbs, mod_base=0, mod_count=None, inverse=False):
        a = np.array(limbs, dtype=np.uint64, order="C")
        flat = a.reshape(-1, self.N)
        if mod_count is None:
            mod_count = self.L
        _check(lib().piehip_ntt(self._h, flat.ctypes.data_as(u64p), flat.shape[0], mod_base, mod_count, int(inverse)))
        return a

    def EvalAdd(self, x, y):
        xa, xp = _u64(x)
        ya, yp = _u64(y)
        out = np.zeros_like(xa)
        _check(lib().piehip_eval_add(self._h, xp, yp, out.ctypes.data_as(u64p)))
        return out

    def EvalMultPlain(self, x, pt):
        xa, xp = _u64(x)
        pa, pp = _u64(pt)
        out = np.zeros_like(xa)
        _check(lib().piehip_eval_mult_plain(self._h, xp, pp, out.ctypes.data_as(u64p)))
        return out

    def EvalMult(self, x, y, relin=True):
        xa, xp = _u64(x)
        ya, yp = _u64(y)
        batched = xa.ndim == 4
        nct = xa.shape[0] if batched else 1
        out = np.zeros((nct, 2 if relin else 3, self.L, self.N), dtype=np.uint64)
        _check(lib().piehip_eval_mult(self._h, xp, yp, nct, int(relin), out.ctypes.data_as(u64p)))
        return out if batched else out[0]

    def EvalAutomorphism(self, x, g, rk):
        xa, xp = _u64(x)
        ka, kp = _u64(rk)
        out = np.zeros_like(xa)
        _check(lib().piehip_eval_automorph(self._h, xp, g, kp, out.ctypes.data_as(u64p)))
        return out

    def MakePackedPlaintext(self, slots):
        s = np.ascontiguousarray(slots, dtype=np.int64)
        single = s.ndim == 1
        s2 = s.reshape(1, -1) if single else s
        out = np.zeros((s2.shape[0], self.L, self.N), dtype=np.uint64)
        _check(lib().piehip_encode(self._h, s2.ctypes.data_as(i64p), s2.shape[0], s2.shape[1], out.ctypes.data_as(u64p)))
        return out[0] if single else out

    def base_convert(self, which, polys):
        a, ap = _u64(polys)
        npoly = a.shape[0]
        out = np.zeros((npoly, self.L if which == 2 else self.M, self.N), dtype=np.uint64)
        _check(lib().piehip_base_convert(self._h, which, ap, npoly, out.ctypes.data_as(u64p)))
        return out

    def load_relin_key(self, evk, query=None):
        """InsertEvalMultKey.  query=q: the key of query q's client in a batch (piehip_load_relin_key_q; the queries of a batch are
        different clients', each with its own key -- BatchedFHEPSIServer.cpp:45-49); queries without one use the context's key."""
        a, ap = _u64(evk)
        assert a.shape == (self.L, 2, self.L, self.N)
        if query is None:
            _check(lib().piehip_load_relin_key(self._h, ap))
        else:
            _check(lib().piehip_load_relin_key_q(self._h, int(query), ap))

    # -- sharded server over RCCL behind the C ABI (piehip_rccl.cpp): what a C++ server calls; shard.py is the torch.distributed way
    def rccl_init(self, unique_id, nranks, rank):
        """join the communicator made from `unique_id` (rccl_unique_id() of one rank, handed to the others by the caller)"""
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        _check(lib().piehip_rccl_init(self._h, buf, int(nranks), int(rank)))

    def rccl_destroy(self):
        _check(lib().piehip_rccl_destroy(self._h))

    def rccl_wait(self, timeout_ms=30000):
        """piehip_sync with a time-out: a rank that never joins a collective aborts the communicator here instead of hanging the group"""
        _check(lib().piehip_rccl_wait(self._h, int(timeout_ms)))

    def rccl_abort(self):
        _check(lib().piehip_rccl_abort(self._h))

    def rccl_agree(self, ok, timeout_ms=30000):
        """True when every rank of the communicator passed a true `ok` (piehip_rccl_agree)"""
        out = C.c_int()
        _check(lib().piehip_rccl_agree(self._h, int(bool(ok)), C.byref(out), int(timeout_ms)))
        return bool(out.value)

    def rotation_galois(self, index):
        """Galois element 5^index mod 2N of the row rotation by `index` (EvalAtIndex convention: > 0 rotates left)"""
        g = C.c_uint32()
        _check(lib().piehip_rotation_galois(self._h, int(index), C.byref(g)))
        return int(g.value)

    def load_rotation_keys(self, keys):
        """keys: {rotation index: [L][2][L][N]} -- the EvalSum / EvalAtIndex key maps the server receives
        (reference SimpleFHEPSIServer.cpp receives them with the context)"""
        idx = np.array(sorted(keys), dtype=np.int32)
        ka = np.ascontiguousarray(np.stack([np.asarray(keys[int(i)], dtype=np.uint64) for i in idx]))
        assert ka.shape[1:] == (self.L, 2, self.L, self.N)
        _check(lib().piehip_load_rotation_keys(self._h, len(idx), idx.ctypes.data_as(i32p), ka.ctypes.data_as(u64p)))

    # -- measurement
    def bench_ntt(self, nlimbs, mod_count=None, inverse=False, iters=20, lane_order=False):
        ms = C.c_double()
        _check(lib().piehip_bench_ntt(self._h, nlimbs, mod_count or self.M, int(inverse) | (2 if lane_order else 0), iters, C.byref(ms)))
        return ms.value

    def reserve(self, n_items, k, e, K, b, E, binSlice=None):
        """allocate ahead of the offline phase what a database of this shape and its queries need (piehip_reserve)"""
        lo, hi = binSlice if binSlice is not None else (0, b)
        _check(lib().piehip_reserve(self._h, int(n_items), k, e, K, b, E, lo, hi))

    def set_run_streams(self, n):
        """run() spreads the bin layers over up to n HIP streams (0 = default, 1 = serial on the handle's stream)"""
        _check(lib().piehip_set_run_streams(self._h, int(n)))

    def set_transform_slots(self, n):
        """cap the persistent transform grids at n workgroups (0 = every slot of the device): leaves CUs to RCCL's kernels on a sharded
        server (piehip_set_transform_slots); results do not depend on it"""
        _check(lib().piehip_set_transform_slots(self._h, int(n)))

    def transform_slots(self):
        """(cap, slots of the device)"""
        n, d = C.c_uint32(), C.c_uint32()
        _check(lib().piehip_get_transform_slots(self._h, C.byref(n), C.byref(d)))
        return int(n.value), int(d.value)

    def upload_turn_wait(self):
        """how long this handle's staging sequences waited for their turn on the PCIe link: (last ms, total ms, waits)"""
        a, b, n = C.c_double(), C.c_double(), C.c_uint64()
        _check(lib().piehip_upload_turn_wait(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return float(a.value), float(b.value), int(n.value)

    def set_host_path_timing(self, on):
        _check(lib().piehip_set_host_path_timing(self._h, int(on)))

    def host_path_times(self):
        """(upload ms, evaluation + download ms) of the last host-memory query that ran with timing on"""
        a, b = C.c_double(), C.c_double()
        _check(lib().piehip_host_path_times(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def set_graph(self, on):
        """run() as one captured hipGraph (piehip_set_graph)"""
        _check(lib().piehip_set_graph(self._h, int(on)))

    def set_profiling(self, on):
        _check(lib().piehip_set_profiling(self._h, int(on)))

    def profile(self):
        n = np.zeros(NKERNELS, dtype=np.uint32)
        ms = np.zeros(NKERNELS, dtype=np.float64)
        by = np.zeros(NKERNELS, dtype=np.float64)
        _check(lib().piehip_profile_read_n(self._h, NKERNELS, n.ctypes.data_as(u32p), ms.ctypes.data_as(f64p), by.ctypes.data_as(f64p)))
        names = [lib().piehip_kernel_name(k).decode() for k in range(NKERNELS)]
        return {names[k]: dict(launches=int(n[k]), ms=float(ms[k]), alg_bytes=float(by[k])) for k in range(NKERNELS) if n[k]}


class BatchedFHEHIPPIE:
    """Mirror of the reference operator (BatchedFHEHIPPIE.hpp:18-49).

    The reference constructor takes (cryptoContext, publicKey, HierarchicalCuckooHashTable) and packs
    the table (BatchedFHEHIPPIE.cpp:9-86).  Here the packed database arrives either as EVALUATION
    limbs (vectorizedHCT [K][b][E][L][N], preCalcRandomMask [b][L][N]) or as raw slot values
    ([K][b][E][B], [b][B]) which the device encodes (MakePackedPlaintext, .cpp:68,81).
    stash_size / multi-table flags reproduce the argument checks at .cpp:13-21.
    """

    def __init__(self, cryptoContext, vectorizedHCT=None, preCalcRandomMask=None, slots=None, mask_slots=None,
                 serverStashSize=0, simpleMultiTables=True, cuckooMultiTables=True, serverSet=None, hashParams=None,
                 hashTable=None, shuffle_seed=None, mask_seed=None, binSlice=None, attachTo=None):
        # Seeds of the Cuckoo evictions, the bin-layer shuffle and the random masks: secret by default (OS CSPRNG), as the
        # reference draws them from std::random_device (BatchedFHEHIPPIE.cpp:25-26,72-82; CuckooHashTable.cpp:51-52).
        # The masks hide prod_h(item - x) of non-matching slots from the client.  Explicit seeds are for parity tests.
        def secret(v):
            return secrets.randbits(64) if v is None else int(v)
        if serverStashSize != 0:
            raise ValueError("Error, batched FHE PIE does not support a stash (yet).")
        if not simpleMultiTables or not cuckooMultiTables:
            raise ValueError("Error, batched FHE PIE currently does not support combined tables.")
        self.cc = cryptoContext
        h = cryptoContext._h
        # binSlice = (lo, hi): a sharded server's rank keeps only these bin layers of the table; the shards of one
        # database must be given the same seeds
        if binSlice is not None and serverSet is None and hashTable is None:
            raise ValueError("binSlice applies to databases built from a server set or a hash table")
        if binSlice is not None:
            # every shard builds (or shuffles) the whole table and keeps its slice: with seeds drawn per rank the ranks would cut
            # slices out of different tables -- items in two shards or in none, silent false negatives.  The seeds of a sharded
            # database are therefore the caller's to distribute (drawn once, e.g. on rank 0, and broadcast: shard.shared_seeds).
            hp_ = hashParams or {}
            given = [hp_.get("shuffle_seed", shuffle_seed), hp_.get("mask_seed", mask_seed)]
            if serverSet is not None:
                given.append(hp_.get("evict_seed"))
            if any(v is None for v in given):
                raise ValueError("binSlice needs explicit evict / shuffle / mask seeds, identical on every shard of the database")
        if attachTo is not None:
            # a further query slot on attachTo's database and key (piehip_attach_database): own context, stream and workspace
            self.K, self.b, self.E = attachTo.K, attachTo.b, attachTo.E
            _check(lib().piehip_attach_database(h, attachTo.cc._h))
            self._owner = attachTo  # keeps the database alive
        elif serverSet is not None:
            # the whole offline phase on the device: nested hashing (HierarchicalCuckooHashTable::insertAll) +
            # the reference constructor's shuffle / gather / encode.  hashParams: k, e, K, b, E, hash_seed,
            # evict_seed, shuffle_seed, mask_seed
            p = hashParams
            items, ip = _u64(serverSet)
            lo, hi = binSlice if binSlice is not None else (0, p["b"])
            self.K, self.b, self.E = p["K"], hi - lo, p["E"]
            _check(lib().piehip_build_db_bins(h, ip, len(items), p["k"], p["e"], p["K"], p["b"], p["E"], p.get("hash_seed", 987654321),
                                              secret(p.get("evict_seed")), secret(p.get("shuffle_seed", shuffle_seed)),
                                              secret(p.get("mask_seed", mask_seed)), lo, hi))
            self._tbl_shape = (p["k"], p["e"], p["K"], p["b"], p["E"])
        elif hashTable is not None:
            # the reference constructor proper: hct.hierarchicalCuckooTable [k][e][K][b][E] -> shuffle, gather, encode
            tbl, tp = _u64(hashTable)
            k, e, self.K, ball, self.E = tbl.shape
            lo, hi = binSlice if binSlice is not None else (0, ball)
            self.b = hi - lo
            _check(lib().piehip_load_db_table_bins(h, tp, k, e, self.K, ball, self.E, secret(shuffle_seed), secret(mask_seed), lo, hi))
            self._tbl_shape = tbl.shape
        elif vectorizedHCT is not None:
            db, dbp = _u64(vectorizedHCT)
            mk, mkp = _u64(preCalcRandomMask)
            self.K, self.b, self.E = db.shape[0], db.shape[1], db.shape[2]
            if db.shape[3:] != (self.cc.L, self.cc.N) or mk.shape != (self.b, self.cc.L, self.cc.N):
                raise ValueError("database shape does not match the crypto context")
            _check(lib().piehip_load_db(h, self.K, self.b, self.E, dbp, mkp))
        else:
            s = np.ascontiguousarray(slots, dtype=np.int64)
            m = np.ascontiguousarray(mask_slots, dtype=np.int64)
            self.K, self.b, self.E, B = s.shape
            if m.shape != (self.b, B):
                raise ValueError("mask shape does not match the database")
            if np.abs(s).max(initial=0) >= self.cc.t or np.abs(m).max(initial=0) >= self.cc.t:
                raise ValueError("slot value out of range for the plaintext modulus")
            _check(lib().piehip_load_db_slots(h, self.K, self.b, self.E, B, s.ctypes.data_as(i64p), m.ctypes.data_as(i64p)))
        self._keep = []
        self._results = np.empty((self.b, 2, self.cc.L, self.cc.N), dtype=np.uint64)  # touched now, in the offline phase
        self._results.fill(0)

    def hashTable(self):
        """hierarchicalCuckooTable after the bin shuffle, [k][e][K][b][E] (only after a serverSet build)"""
        out = np.zeros(self._tbl_shape, dtype=np.uint64)
        _check(lib().piehip_get_hash_table(self.cc._h, out.ctypes.data_as(u64p)))
        return out

    def setQueryBatch(self, nq):
        """run() evaluates nq queries at once against the database (piehip_set_query_batch): setIndex / setMinusCompareElement take
        query=q, getResultList returns [nq][b] ciphertexts.  The reference operator has one query per run()."""
        _check(lib().piehip_set_query_batch(self.cc._h, int(nq)))
        self._results = None

    @property
    def nq(self):
        """queries per run() of the handle (piehip_get_query_batch)"""
        n = C.c_uint32()
        _check(lib().piehip_get_query_batch(self.cc._h, C.byref(n)))
        return n.value

    def setIndex(self, indexMatrix, query=0):
        a, ap = _u64(indexMatrix)
        if a.shape != (self.K, self.E, 2, self.cc.L, self.cc.N):
            raise ValueError("index matrix must be [K][E] ciphertexts")
        _check(lib().piehip_set_index_q(self.cc._h, query, ap))

    def setMinusCompareElement(self, minusCompareElement, query=0):
        a, ap = _u64(minusCompareElement)
        if a.shape != (2, self.cc.L, self.cc.N):
            raise ValueError("minus element must be one ciphertext")
        _check(lib().piehip_set_minus_q(self.cc._h, query, ap))

    def setIndexDevice(self, ptr, query=0):
        _check(lib().piehip_set_index_device_q(self.cc._h, query, ptr))

    def setMinusCompareElementDevice(self, ptr, query=0):
        _check(lib().piehip_set_minus_device_q(self.cc._h, query, ptr))

    def run(self, sync=True, into=None):
        """into: device address of a caller-owned result buffer [b][2][L][N] (piehip_run_into; [b][nq][2][L][N] for a query batch)"""
        if into is None:
            _check(lib().piehip_run(self.cc._h))
        else:
            _check(lib().piehip_run_into(self.cc._h, into))
        if sync:
            _check(lib().piehip_sync(self.cc._h))

    def _res_shape(self):
        nq = self.nq
        return (self.b, 2, self.cc.L, self.cc.N) if nq == 1 else (self.b, nq, 2, self.cc.L, self.cc.N)

    def hostBuffers(self, query=0):
        """page-locked numpy views (index matrix [K][E][2][L][N] and minus element [2][L][N] of query `query` of the batch, and the
        operator's result list [b][2][L][N] -- [b][nq][2][L][N] for a batch, the same array for every query) owned by the
        library: a deserialiser that writes the towers straight into them saves the staging copy of the upload"""
        pi, pm, pr = u64p(), u64p(), u64p()
        _check(lib().piehip_host_buffers_q(self.cc._h, int(query), C.byref(pi), C.byref(pm), C.byref(pr)))
        L, N = self.cc.L, self.cc.N
        mk = lambda ptr, shape: np.ctypeslib.as_array(ptr, shape=shape)
        return mk(pi, (self.K, self.E, 2, L, N)), mk(pm, (2, L, N)), mk(pr, self._res_shape())

    def _in_shapes(self):
        nq = self.nq
        pre = () if nq == 1 else (nq,)
        return pre + (self.K, self.E, 2, self.cc.L, self.cc.N), pre + (2, self.cc.L, self.cc.N)

    def runHost(self, indexMatrix, minusCompareElement, results=None):
        """setMinusCompareElement + setIndex + run + getResultList in one pipelined call (piehip_run_host): the query is in host
        memory, row h of the index matrix uploads while stage A of row h - 1 runs, results download per queue group.
        A batch of nq queries: indexMatrix [nq][K][E] ciphertexts, minusCompareElement [nq], results [b][nq]."""
        a, ap = _u64(indexMatrix)
        m, mp = _u64(minusCompareElement)
        if (a.shape, m.shape) != self._in_shapes():
            raise ValueError("index matrix must be [K][E] ciphertexts, the minus element one ciphertext (per query of the batch)")
        if results is None:
            if getattr(self, "_results", None) is None or self._results.shape != self._res_shape():
                self._results = np.zeros(self._res_shape(), dtype=np.uint64)
            results = self._results
        _check(lib().piehip_run_host(self.cc._h, ap, mp, results.ctypes.data_as(u64p)))
        return results

    def runHostAsync(self, indexMatrix, minusCompareElement, results):
        """queue runHost's uploads, evaluation and downloads and return (piehip_run_host_async); the three arrays -- page-locked
        ones from hostBuffers() for a call that never waits -- must stay untouched until waitHost()"""
        if indexMatrix.dtype != np.uint64 or not indexMatrix.flags.c_contiguous or minusCompareElement.dtype != np.uint64 \
                or not minusCompareElement.flags.c_contiguous or results.dtype != np.uint64 or not results.flags.c_contiguous:
            raise ValueError("runHostAsync needs contiguous uint64 arrays (no temporary copies may be taken)")
        if (indexMatrix.shape, minusCompareElement.shape) != self._in_shapes() or results.shape != self._res_shape():
            raise ValueError("index matrix must be [K][E] ciphertexts, the minus element one ciphertext, results [b] ciphertexts")
        _check(lib().piehip_run_host_async(self.cc._h, indexMatrix.ctypes.data_as(u64p), minusCompareElement.ctypes.data_as(u64p),
                                           results.ctypes.data_as(u64p)))

    def stageMinus(self, minusCompareElement, query=0):
        """start the upload of the minus element of query `query` (piehip_stage_minus_q); the array must stay untouched until waitHost()"""
        if minusCompareElement.dtype != np.uint64 or not minusCompareElement.flags.c_contiguous or minusCompareElement.shape != (2, self.cc.L, self.cc.N):
            raise ValueError("the minus element is one contiguous uint64 ciphertext")
        _check(lib().piehip_stage_minus_q(self.cc._h, int(query), minusCompareElement.ctypes.data_as(u64p)))

    def stageIndexRow(self, row, rowCiphertexts, query=0):
        """start the upload of row `row` of query `query`'s index matrix, [E][2][L][N] (piehip_stage_index_row_q)"""
        if rowCiphertexts.dtype != np.uint64 or not rowCiphertexts.flags.c_contiguous or rowCiphertexts.shape != (self.E, 2, self.cc.L, self.cc.N):
            raise ValueError("an index matrix row is E contiguous uint64 ciphertexts")
        _check(lib().piehip_stage_index_row_q(self.cc._h, int(query), int(row), rowCiphertexts.ctypes.data_as(u64p)))

    def stageIndexCiphertext(self, row, j, ciphertext, query=0):
        """start the upload of ciphertext (row, j) of query `query`'s index matrix, [2][L][N] (piehip_stage_index_ct_q)"""
        if ciphertext.dtype != np.uint64 or not ciphertext.flags.c_contiguous or ciphertext.shape != (2, self.cc.L, self.cc.N):
            raise ValueError("an index matrix entry is one contiguous uint64 ciphertext")
        _check(lib().piehip_stage_index_ct_q(self.cc._h, int(query), int(row), int(j), ciphertext.ctypes.data_as(u64p)))

    def stageReset(self):
        """drop a partial staging sequence (piehip_stage_reset)"""
        _check(lib().piehip_stage_reset(self.cc._h))

    def runStaged(self, results):
        """evaluate the staged queries and queue the download of the result list (piehip_run_staged); waitHost() completes it.
        results: [b][2][L][N], or [b][nq][2][L][N] for a batch (the library's row order: the nq results of a bin layer adjacent)"""
        if results.dtype != np.uint64 or not results.flags.c_contiguous or results.shape != self._res_shape():
            raise ValueError("results must be [b] ([b][nq] for a batch) contiguous uint64 ciphertexts")
        _check(lib().piehip_run_staged(self.cc._h, results.ctypes.data_as(u64p)))

    def waitHost(self):
        """block until the results of the last runHostAsync are complete in host memory"""
        _check(lib().piehip_run_host_wait(self.cc._h))

    def join(self):
        """order the context's stream behind the runs queued so far (no host wait)"""
        _check(lib().piehip_join(self.cc._h))

    def sync(self):
        _check(lib().piehip_sync(self.cc._h))

    def getResultList(self):
        """the b result ciphertexts.  As in the reference (BatchedFHEHIPPIE.hpp:35-38 returns a reference to the member
        vector) the array belongs to the operator and is overwritten by the next call; copy it to keep it.  (A fresh
        14 MiB numpy array per query costs ~25 ms of first-touch page faults under the device-to-host copy.)"""
        nq = self.nq
        want = (self.b, 2, self.cc.L, self.cc.N) if nq == 1 else (self.b, nq, 2, self.cc.L, self.cc.N)
        if getattr(self, "_results", None) is not None and self._results.shape != want:
            self._results = None
        if getattr(self, "_results", None) is None:
            self._results = np.empty((self.b, 2, self.cc.L, self.cc.N) if nq == 1 else (self.b, nq, 2, self.cc.L, self.cc.N), dtype=np.uint64)
            self._results.fill(0)
        out = self._results
        _check(lib().piehip_get_results(self.cc._h, out.ctypes.data_as(u64p)))
        # a query batch: the library's rows are [bin layer][query]; hand back [query][bin layer] (a view)
        return out if nq == 1 else out.transpose(1, 0, 2, 3, 4)

    def broadcastQuery(self, root=0):
        """the query staged on rank `root` (stageMinus / stageIndexRow) reaches every rank's input buffers (piehip_rccl_broadcast_query)"""
        _check(lib().piehip_rccl_broadcast_query(self.cc._h, int(root)))

    def gatherResultsHost(self, b_total, root=0):
        """after run() on every rank: the b_total result ciphertexts in bin order on rank `root`, as a page-locked numpy view owned by
        the library (None on the other ranks); complete after sync()  (piehip_gather_results_host)"""
        p = u64p()
        _check(lib().piehip_gather_results_host(self.cc._h, int(b_total), int(root), C.byref(p)))
        if not p:
            return None
        nq = self.nq
        shape = (b_total, 2, self.cc.L, self.cc.N) if nq == 1 else (b_total, nq, 2, self.cc.L, self.cc.N)
        return np.ctypeslib.as_array(p, shape=shape)

    def copyResultsToDevice(self, ptr):
        _check(lib().piehip_copy_results_device(self.cc._h, ptr))

    def resultsDevicePtr(self):
        p = C.c_void_p()
        _check(lib().piehip_results_device(self.cc._h, C.byref(p)))
        return p.value


class QueryPipeline:
    """Several queries in flight on one database: `depth` query slots, each a context with its own stream and run() workspace,
    all reading slot 0's key and packed database (piehip_attach_database).  submit() hands the next query to the next slot
    and returns it; slot.run(sync=False) calls of different slots overlap on the GPU.  Serving-side addition: the reference
    operator evaluates one query at a time."""

    def __init__(self, op, depth, make_context):
        """op: the operator that owns the database; make_context(): a fresh PieContext with the same parameters on its own
        stream (called depth - 1 times)"""
        self.slots = [op] + [BatchedFHEHIPPIE(make_context(), attachTo=op) for _ in range(depth - 1)]
        self._next = 0

    def submit(self, d_idx, d_minus, d_results=None):
        """enqueue one query whose inputs live in HBM (device addresses); returns the slot that evaluates it"""
        s = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        s.setIndexDevice(d_idx)
        s.setMinusCompareElementDevice(d_minus)
        s.run(sync=False, into=d_results)
        return s

    def run_all(self):
        """one more run() on every slot with the inputs it already has (benchmarks)"""
        for s in self.slots:
            s.run(sync=False)

    def sync(self):
        for s in self.slots:
            s.sync()

    def close(self):
        for s in self.slots[1:]:
            s.cc.close()


class FHEHIPPIE:
    """Mirror of the rotation-based operator (reference FHEHIPPIE.hpp:18-49, FHEHIPPIE.cpp), batched over `npie`
    operators (the reference holds one per client slot in an FHEHIPPIECollection, PIECollection.hpp).

    cuckooTable: [npie][K][b][E] (or [K][b][E]) table cells, b == E; rotation keys must be loaded into the context
    (PieContext.load_rotation_keys).  The constructor hides the bin order with one permutation per operator
    (permVec2, FHEHIPPIE.cpp:28,50) and draws the masks (FHEHIPPIE.cpp:52); getResultList applies the result
    permutation (permutationVector, FHEHIPPIE.hpp:30-34, FHEHIPPIE.cpp:76).  Both come from a seeded numpy
    generator here (the reference uses std::random_device); pass perm_seed=None to keep the natural order.
    """

    def __init__(self, cryptoContext, cuckooTable, stashSize=0, perm_seed=5, mask_seed=6, masks=None):
        if stashSize != 0:
            raise ValueError("Error, FHE PIE does not support a stash (yet).")
        self.cc = cryptoContext
        tbl = np.asarray(cuckooTable, dtype=np.uint64)
        self._single = tbl.ndim == 3
        if self._single:
            tbl = tbl[None]
        self.npie, self.K, self.b, self.E = tbl.shape
        if self.b != self.E:
            raise ValueError("Error, for FHE PIE the size of a cuckoo bin has to be equal than the number of bins per hash function.")
        rng = None if perm_seed is False else np.random.default_rng(secrets.randbits(128) if perm_seed is None else perm_seed)
        self.permVec2 = np.stack([rng.permutation(self.b) if rng is not None else np.arange(self.b) for _ in range(self.npie)])
        self.permutationVector = np.stack([rng.permutation(self.K) if rng is not None else np.arange(self.K) for _ in range(self.npie)])
        slots = np.ones((self.npie, self.K, self.b, self.E + 1), dtype=np.int64)  # last slot: exponent of the "minus client" element
        for i in range(self.npie):
            slots[i, :, self.permVec2[i], :self.E] = tbl[i].astype(np.int64).transpose(1, 0, 2)
        if masks is None:
            mrng = np.random.default_rng(secrets.randbits(128) if mask_seed is None else mask_seed)
            masks = mrng.integers(1, self.cc.t, size=(self.npie, self.K, self.b), dtype=np.int64)
        self.masks = np.ascontiguousarray(masks, dtype=np.int64).reshape(self.npie, self.K, self.b)
        self.slots = np.ascontiguousarray(slots)
        _check(lib().piehip_fhepie_load_table(self.cc._h, self.npie, self.K, self.b, self.E, self.slots.ctypes.data_as(i64p),
                                              self.masks.ctypes.data_as(i64p)))

    def setIndex(self, indexMatrix):
        a, ap = _u64(indexMatrix)
        if self._single and a.ndim == 4:
            a = a[None]
        if a.shape != (self.npie, self.K, 2, self.cc.L, self.cc.N):
            raise ValueError("index matrix must be one ciphertext per hash function")
        a, ap = _u64(a)
        _check(lib().piehip_fhepie_set_index(self.cc._h, ap))

    def run(self):
        _check(lib().piehip_fhepie_run(self.cc._h))

    def getResultList(self):
        out = np.zeros((self.npie, self.K, 2, self.cc.L, self.cc.N), dtype=np.uint64)
        _check(lib().piehip_fhepie_get_results(self.cc._h, out.ctypes.data_as(u64p)))
        shuffled = np.empty_like(out)
        for i in range(self.npie):
            shuffled[i, self.permutationVector[i]] = out[i]
        return shuffled[0] if self._single else shuffled
