import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nested_hashing_psi_amd import pie
from oracle import binding as ob
N = int(sys.argv[1]); L = 2; t = 65537
o = ob.Oracle(N, L, t); cc = pie.PieContext(N, L, t)
rng = np.random.default_rng(1)
for inv in (False, True):
    for nl in (1, 2, 5, 300, 1200):
        x = np.stack([rng.integers(0, int(o.q[0]), N, dtype=np.uint64) for _ in range(nl)])
        f = cc.ntt(x, 0, 1, inverse=inv)
        w = np.stack([(o.intt if inv else o.ntt)(0, x[k]) for k in range(nl)])
        bad = np.argwhere(f != w)
        print("inv" if inv else "fwd", "nl", nl, "bad", len(bad), "limbs", sorted(set(bad[:, 0]))[:10], "pos", bad[:8, 1] if len(bad) else "")
