"""Sweep the NTT kernel over batch sizes: ms per launch, us per limb per CU-slot, algorithmic GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nested_hashing_psi_amd import pie
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
L = {4096: 2, 8192: 3, 16384: 4, 32768: 6}[N]
t = 65537 if N == 4096 else 4296540161
cc = pie.PieContext(N, L, t)
for inv in (False, True):
    for nl in (64, 112, 224, 256, 378, 504, 512, 1024, 2048, 4096):
        ms = cc.bench_ntt(nl, iters=10, inverse=inv)
        print("N=%d %s nlimbs=%5d  %8.1f us/launch  %6.2f us/limb*256  %7.1f GB/s alg" % (
            N, "inv" if inv else "fwd", nl, ms * 1e3, ms * 1e3 / nl * 256, 16.0 * N * nl / (ms * 1e-3) / 1e9))
cc.close()
