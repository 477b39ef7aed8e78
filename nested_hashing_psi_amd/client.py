"""Client-side harness: the role of the reference's BatchedFHEPSIClient
(src/Client/FHE/BatchedFHEPSIClient.cpp), enough to drive the server hot path with real inputs, read
its outputs and measure the end-to-end PSI wall-clock.  Harness, not the product:
the product is the server path.  Hashing of the (small) client set runs on the host; BFV key
generation, encryption and decryption run on the device through the C ABI (piehip_client_*).
"""
import ctypes as C

import numpy as np

from ._lib import i64p, lib, u64p
from .pie import _check, _u64, tabulation_hash


def _out(shape, dtype):
    """output buffer for a device-to-host copy, pages touched here: a calloc'ed (np.zeros) array is faulted in page by
    page inside the copy, ~25 ms for the 30 MiB of one query's ciphertexts"""
    a = np.empty(shape, dtype=dtype)
    a.fill(0)
    return a


PLAINTEXT_MODULI = {16: 65537, 32: 4296540161, 40: 1099579260929, 48: 281474981953537}


def select_parameters(bitSize, eachCuckooTableSize):
    """The reference client's scheme parameters (BatchedFHEPSIClient.cpp:22-57,72-78): plaintext modulus by item bit size
    (ValueError as the reference's invalid_argument otherwise), multiplicative depth 3 / 5 / 10 by inner table size, ring
    dimension 16384.  The number of 60-bit primes OpenFHE derives from the depth is not in the reference tree
    [OFHE-UNVERIFIED]: depth 3 with the 33-bit modulus is the 4-prime chain BASELINE.json names; deeper settings take one
    prime per level plus one."""
    if bitSize not in PLAINTEXT_MODULI:
        raise ValueError("Error: FHE can only support bit sizes 16 or 32.")
    depth = 3 if eachCuckooTableSize < 500 else (5 if eachCuckooTableSize < 5000 else 10)
    return dict(N=16384, t=PLAINTEXT_MODULI[bitSize], depth=depth, L=depth + 1)


class BatchedFHEPSIClient:
    """Mirror of BatchedFHEPSIClient (BatchedFHEPSIClient.cpp:14-193): runSetUpPhase, runOfflinePhase,
    the online-phase result extraction.  No TCP: the caller moves the arrays."""

    def __init__(self, cryptoContext, numberOfSimpleHashFunctions, eachSimpleTableSize, numberOfCuckooHashFunctions,
                 eachCuckooTableSize, maxItemsPerPosition, hashSeed=987654321):
        self.cc = cryptoContext
        self.k, self.e = numberOfSimpleHashFunctions, eachSimpleTableSize
        self.K, self.E, self.b = numberOfCuckooHashFunctions, eachCuckooTableSize, maxItemsPerPosition
        self.hashSeed = hashSeed
        self.B = self.k * self.e

    # -- setup (BatchedFHEPSIClient.cpp:88-91): KeyGen + EvalMultKeyGen
    def runSetUpPhase(self, keySeed=11, evalKeySeed=12):
        cc = self.cc
        self.sk = _out((cc.L, cc.N), np.uint64)
        _check(lib().piehip_client_keygen(cc._h, keySeed, self.sk.ctypes.data_as(u64p)))
        self.evalMultKey = _out((cc.L, 2, cc.L, cc.N), np.uint64)
        _check(lib().piehip_client_relin_keygen(cc._h, self.sk.ctypes.data_as(u64p), evalKeySeed,
                                                self.evalMultKey.ctypes.data_as(u64p)))
        return self.evalMultKey

    # -- EvalSumKeyGen + EvalRotateKeyGen of the rotation-based sibling (SimpleFHEPSIClient.cpp:80-89): keys for the
    #    rotations 2^r (r < ceil(log2 b)) and -1 .. -(b-1) that FHEHIPPIE::run needs
    def rotationKeyGen(self, nbins, seedBase=50):
        cc = self.cc
        R = 0
        while (1 << R) < nbins:
            R += 1
        rots = [1 << r for r in range(R)] + [-i for i in range(1, nbins)]
        keys = {}
        for i, r in enumerate(rots):
            rk = _out((cc.L, 2, cc.L, cc.N), np.uint64)
            _check(lib().piehip_client_rot_keygen(cc._h, self.sk.ctypes.data_as(u64p), r, seedBase + i, rk.ctypes.data_as(u64p)))
            keys[r] = rk
        return keys

    # -- client Cuckoo table: CuckooHashTable(hash, e, k, startingHashId 0, stash 0, multi, 1 layer)
    #    (BatchedFHEPSIClient.cpp:97-99, insertAll at :109; insert at CuckooHashTable.cpp:72-114)
    def _hash_client_set(self, items):
        items = np.ascontiguousarray(items, dtype=np.uint64)
        table = np.zeros((self.k, self.e), dtype=np.uint64)
        _check(lib().piehip_client_cuckoo_table(self.hashSeed, self.k + self.K, self.k, self.e, items.ctypes.data_as(u64p), len(items),
                                                table.ctypes.data_as(u64p)))
        return table

    # -- offline (BatchedFHEPSIClient.cpp:107-169): index matrix + minus vector, secret-key encrypted
    def runOfflinePhase(self, clientSet, encSeedBase=100):
        self.clientTable = self._hash_client_set(clientSet)
        k, e, K, E, B = self.k, self.e, self.K, self.E, self.B
        flat = self.clientTable.reshape(-1)
        index = np.zeros((K, E, B), dtype=np.int64)
        minus = np.ones(B, dtype=np.int64)                         # dummy slot: +1, all-zero index column (:128-131)
        occ = np.nonzero(flat)[0]
        if len(occ):
            minus[occ] = -flat[occ].astype(np.int64)
            for hf in range(K):
                hi = tabulation_hash(self.hashSeed, k + K, k + hf, flat[occ]) % np.uint64(E)
                index[hf, hi.astype(np.int64), occ] = 1
        self.plainIndex, self.plainMinus = index, minus
        vecs = np.concatenate([minus.reshape(1, B), index.reshape(K * E, B)])
        seeds = np.array([encSeedBase - 1] + [encSeedBase + i for i in range(K * E)], dtype=np.uint64)
        cts = self._encrypt(vecs, seeds)
        self.encryptedMinusElements = cts[0]
        self.batchedEncryptedIndexMatrix = cts[1:].reshape(K, E, 2, self.cc.L, self.cc.N)
        return self.encryptedMinusElements, self.batchedEncryptedIndexMatrix

    def _encrypt(self, vecs, seeds):
        cc = self.cc
        v = np.ascontiguousarray(vecs, dtype=np.int64)
        out = _out((v.shape[0], 2, cc.L, cc.N), np.uint64)
        s = np.ascontiguousarray(seeds, dtype=np.uint64)
        _check(lib().piehip_client_encrypt(cc._h, self.sk.ctypes.data_as(u64p), v.ctypes.data_as(i64p), v.shape[0], v.shape[1],
                                           s.ctypes.data_as(u64p), out.ctypes.data_as(u64p)))
        return out

    def decrypt(self, cts, nslots=None):
        cc = self.cc
        a, ap = _u64(cts)
        n = a.shape[0]
        nslots = nslots or self.B
        out = _out((n, nslots), np.int64)
        _check(lib().piehip_client_decrypt(cc._h, self.sk.ctypes.data_as(u64p), ap, n, nslots, out.ctypes.data_as(i64p)))
        return out

    # -- online (BatchedFHEPSIClient.cpp:176-192): zero slot in any of the b results <=> item in the intersection
    def extractIntersection(self, resultList):
        self.batchedDecryptedResult = self.decrypt(resultList)
        flat = self.clientTable.reshape(-1)
        hit = (self.batchedDecryptedResult == 0).any(axis=0)
        return flat[hit].copy()
