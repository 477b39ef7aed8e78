import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nested_hashing_psi_amd import pie
N = int(sys.argv[1]); nl = int(sys.argv[2]); inv = len(sys.argv) > 3 and sys.argv[3] == "inv"
L = {4096: 2, 8192: 3, 16384: 4, 32768: 6}[N]
cc = pie.PieContext(N, L, 65537 if N == 4096 else 4296540161)
ms = cc.bench_ntt(nl, iters=10, inverse=inv)
print("N=%d %s nlimbs=%d %.1f us/launch %.2f us/limb*256 %.1f GB/s" % (N, "inv" if inv else "fwd", nl, ms * 1e3, ms * 1e3 / nl * 256, 16.0 * N * nl / (ms * 1e-3) / 1e9))
