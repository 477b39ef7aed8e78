#!/bin/bash
# Profiles of the headline workload for profiles/<round>/ (run on the GPU box from the repository root):
#   1. rocprofv3 --output-format csv --kernel-trace --stats of `python3 bench.py --streams 1` (serial: the default timed region -- three
#      queries per run() -- on one queue, every kernel alone on the GPU; these are the durations bench.py's roofline uses), of the
#      default command (two queues per run(), kernels overlap) and of `--streams 1 --in-flight 1` (serial, ONE query per run():
#      the launches of rounds 1-2 and of the first half of round 3)
#   2. HBM traffic: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the serial command
#   3. one SQ pass (VALU activity / waits)
# Outputs land in gpurun_out/prof_<round>/; tools/pmc_summary.py digests them.
set -e
ROUND=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$ROUND
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="$ROOT/bench.py --timed-only"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/serial -o serial -- python3 $BENCH --streams 1 > $OUT/bench_serial.json 2> $OUT/serial.log
echo "serial trace done"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/serial1 -o serial1 -- python3 $BENCH --streams 1 --in-flight 1 > $OUT/bench_serial1.json 2> $OUT/serial1.log
echo "serial (one query per run) trace done"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/default -o default -- python3 $BENCH > $OUT/bench_default.json 2> $OUT/default.log
echo "default trace done"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python3 $BENCH --streams 1 --steps 5 --warmup 1 --profile-steps 1 > /dev/null 2> $OUT/fetch.log
echo "fetch pass done"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o write -- python3 $BENCH --streams 1 --steps 5 --warmup 1 --profile-steps 1 > /dev/null 2> $OUT/write.log
echo "write pass done"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS -d $OUT/sq -o sq -- python3 $BENCH --streams 1 --steps 5 --warmup 1 --profile-steps 1 > /dev/null 2> $OUT/sq.log || echo "sq pass failed (counter set not available)"
cd $ROOT
find $OUT -name "*.csv" | head -40
