import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # single HIP runtime
from oracle import binding as ob
lib = C.CDLL(os.environ["PIEHIP_LIB"])
u64p = C.POINTER(C.c_uint64)
lib.piehip_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_uint64, u64p, u64p, C.c_int, C.c_void_p]
lib.piehip_ntt.argtypes = [C.c_void_p, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
lib.piehip_last_error.restype = C.c_char_p
for N in (4096, 8192, 16384):
    L, t = 2, 65537
    o = ob.Oracle(N, L, t)
    h = C.c_void_p()
    rc = lib.piehip_create(C.byref(h), N, L, t, None, None, 0, None)
    assert rc == 0, lib.piehip_last_error()
    rng = np.random.default_rng(1)
    for inv in (0, 1):
        for nl in (1, 300):
            x = np.stack([rng.integers(0, int(o.q[0]), N, dtype=np.uint64) for _ in range(nl)])
            f = x.copy()
            rc = lib.piehip_ntt(h, f.ctypes.data_as(u64p), nl, 0, 1, inv)
            assert rc == 0
            w = np.stack([(o.intt if inv else o.ntt)(0, x[k]) for k in range(nl)])
            print(N, "inv" if inv else "fwd", "nl", nl, "bad", int((f != w).sum()))
N, L, t = 4096, 2, 65537
o = ob.Oracle(N, L, t)
h = C.c_void_p(); lib.piehip_create(C.byref(h), N, L, t, None, None, 0, None)
x = np.zeros((1, N), dtype=np.uint64); x[0, 1] = 1   # X -> evaluations psi^(2br(p)+1)
f = x.copy(); lib.piehip_ntt(h, f.ctypes.data_as(u64p), 1, 0, 1, 0)
w = o.ntt(0, x[0])
print("gpu", f[0, :6]); print("cpu", w[:6])
x = np.zeros((1, N), dtype=np.uint64); x[0, 0] = 5
f = x.copy(); lib.piehip_ntt(h, f.ctypes.data_as(u64p), 1, 0, 1, 0)
print("const gpu", f[0, :4], "n!=5:", int((f[0] != 5).sum()))
lib.piehip_get_twiddles.argtypes = [C.c_void_p, C.c_uint32, u64p, u64p]
a = np.zeros(N, dtype=np.uint64); b = np.zeros(N, dtype=np.uint64)
lib.piehip_get_twiddles(h, 0, a.ctypes.data_as(u64p), b.ctypes.data_as(u64p))
fo, io = o.twiddles(0)
print("tw equal", bool((a[1:] == fo[1:]).all()), a[1:4], fo[1:4])
