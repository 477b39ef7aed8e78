"""Per-launch medians of the timed region's kernels from a rocprofv3 kernel trace (tools/profile_round.sh: serial/serial_kernel_trace.csv).
A step launches the inverse transform twice (the accumulators, then the tensor result): the kernel statistics merge them; here a
kernel name that occurs k times per step is split by its position in the step.
   python tools/per_launch_medians.py gpurun_out/prof_r04/serial/serial_kernel_trace.csv > profiles/r04/c3_serial_per_launch_medians.txt"""
import csv
import re
import statistics
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n)).replace("piehip::", "").replace("ntt16::", "")
# the step's launch sequence: everything between two launches of the stage A kernel
marks = [i for i, r in enumerate(rows) if "stage_a_mad" in r["Kernel_Name"]]
steps = [rows[a:b] for a, b in zip(marks, marks[1:])]
length = statistics.mode(len(s) for s in steps)
steps = [s for s in steps if len(s) == length]
print("per-launch medians from the serial trace (rocprofv3 --kernel-trace of bench.py --timed-only --streams 1), %d steps of %d launches;"
      % (len(steps), length))
print("launches in their order in a step; grid = workgroups (the transforms are persistent: slices = see DESIGN.md section 4)")
def wgs(r):
    n = 1
    for a in "XYZ":
        n *= int(r["Grid_Size_" + a]) // max(1, int(r["Workgroup_Size_" + a]))
    return n


total = 0.0
for pos in range(length):
    d = sorted((int(s[pos]["End_Timestamp"]) - int(s[pos]["Start_Timestamp"])) / 1e3 for s in steps)
    med = statistics.median(d)
    total += med
    print("%-46s grid %6d  n=%5d median %6.1f us  p10 %6.1f  p90 %6.1f" % (short(steps[0][pos]["Kernel_Name"])[:46],
          wgs(steps[0][pos]), len(d), med, d[len(d) // 10], d[9 * len(d) // 10]))
print("sum of medians %.1f us per step (three queries)" % total)
