"""per-kernel sums of rocprofv3 --pmc passes: python tools/pmc_kernel_table.py dir [dir ...] -> one row per kernel, one column per counter
(value per LAUNCH, averaged over the launches of the pass)"""
import collections
import csv
import glob
import re
import sys

tab = collections.defaultdict(dict)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        launches = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void piehip::", "").replace("piehip::", "")
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
        for k in per:
            for c, v in per[k].items():
                tab[k][c] = v / max(1, len(launches[k]))
cols = sorted({c for k in tab for c in tab[k]})
print("kernel".ljust(46) + " ".join(c[-26:].rjust(27) for c in cols))
for k in sorted(tab):
    if not any(s in k for s in ("stage_a", "ntt16", "expand", "tensor", "scale_round", "relin")):
        continue
    print(k[:45].ljust(46) + " ".join(("%.4g" % tab[k].get(c, float("nan"))).rjust(27) for c in cols))
