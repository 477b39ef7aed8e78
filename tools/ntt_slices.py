"""What would the five NTT launches of a C3 run() cost as 2^13 slices (outer stage folded into the
neighbouring kernels)?  Times the 2^13 kernel at the slice counts 2*{112,504,378,112,224}."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nested_hashing_psi_amd import pie
cc = pie.PieContext(8192, 3, 4296540161)
tot = 0
for nl, inv in ((224, True), (1008, False), (756, True), (224, False), (448, False)):
    ms = cc.bench_ntt(nl, iters=20, inverse=inv)
    tot += ms * 1e3
    print("slices=%5d %s %7.1f us" % (nl, "inv" if inv else "fwd", ms * 1e3))
print("total %.1f us" % tot)
cc = pie.PieContext(16384, 4, 4296540161)
tot = 0
for nl, inv in ((112, True), (504, False), (378, True), (112, False), (224, False)):
    ms = cc.bench_ntt(nl, iters=20, inverse=inv)
    tot += ms * 1e3
    print("limbs=%5d %s %7.1f us" % (nl, "inv" if inv else "fwd", ms * 1e3))
print("total %.1f us" % tot)
