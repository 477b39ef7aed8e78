"""Digest of a rocprofv3 SQ counter pass (tools/profile_round.sh): per kernel, VALU activity and wait fractions per wave cycle.
usage: sq_summary.py <sq_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r["Kernel_Name"].replace("void ", "").split("(")[0].replace("piehip::", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
out = []
for k, c in sorted(agg.items()):
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if not wc or "piehip" not in "".join(r["Kernel_Name"] for r in rows[:1]) and False:
        continue
    if not wc:
        continue
    out.append(dict(kernel=k, waves=c.get("SQ_WAVES", 0),
                    valu_active_per_wave_cycle=round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
                    wait_any_per_wave_cycle=round(c.get("SQ_WAIT_ANY", 0) / wc, 3),
                    wait_inst_per_wave_cycle=round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                    lds_active_per_wave_cycle=round(c.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3)))
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(len(out), "kernels")
