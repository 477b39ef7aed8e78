#!/bin/bash
# tools/pmc_passes.sh <outdir> <counter group> [<counter group> ...]: one rocprofv3 --pmc pass of the serial timed region per group
# (separate passes: the counter blocks have few slots each); digests with tools/pmc_kernel_table.py.  Every pass under its own
# time limit (a TA_* group aborted inside rocprofv3 and then sat there: r05) and with a progress line for the box's watchdog.
ROOT=$(pwd); OUT=$ROOT/$1; shift
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --output-format csv --pmc $grp -d $OUT/g$i -o g$i -- python3 $ROOT/bench.py --timed-only --streams 1 --steps 5 --warmup 1 --profile-steps 1 > /dev/null 2> $OUT/g$i.log || { echo "group $i failed or timed out: $grp"; exit 1; }
  echo "group $i done: $grp"
done
