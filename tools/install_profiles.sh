#!/bin/bash
# copies the digests of tools/profile_round.sh's output (gpurun_out/prof_<round>/) into profiles/<round>/ and refreshes
# profiles/latest_pmc.json (read by bench.py for roofline.traffic)
set -e
ROUND=${1:-r01}
P=gpurun_out/prof_$ROUND
D=profiles/$ROUND
mkdir -p $D
python tools/pmc_summary.py $P/serial/serial_kernel_stats.csv $P/fetch/fetch_counter_collection.csv $P/write/write_counter_collection.csv $P/summary.json
cp $P/serial/serial_kernel_stats.csv $D/c3_serial_kernel_stats.csv
cp $P/serial/serial_domain_stats.csv $D/c3_serial_domain_stats.csv
cp $P/serial1/serial1_kernel_stats.csv $D/c3_serial_one_query_kernel_stats.csv
cp $P/bench_serial1.json $D/bench_c3_serial_one_query_under_rocprof.json
cp $P/default/default_kernel_stats.csv $D/c3_default_kernel_stats.csv
cp $P/default/default_domain_stats.csv $D/c3_default_domain_stats.csv
cp $P/fetch/fetch_counter_collection.csv $D/c3_pmc_fetch_size.csv
cp $P/write/write_counter_collection.csv $D/c3_pmc_write_size.csv
cp $P/bench_serial.json $D/bench_c3_serial_under_rocprof.json
cp $P/bench_default.json $D/bench_c3_default_under_rocprof.json
python tools/sq_summary.py $P/sq/sq_counter_collection.csv $D/c3_sq_summary.json
python - <<PY
import json
s = json.load(open("$P/summary.json"))
s["config"] = "C3"
s["source"] = ("profiles/$ROUND: tools/profile_round.sh -- rocprofv3 --kernel-trace --stats of 'python3 bench.py --streams 1 --no-cpu-baseline --no-e2e' (three queries per run()) "
               "(kernel_stats) and separate --pmc FETCH_SIZE / WRITE_SIZE passes of the same command with --steps 5 --warmup 1 --profile-steps 1")
json.dump(s, open("profiles/latest_pmc.json", "w"), indent=1, sort_keys=True)
json.dump(s, open("$D/c3_summary.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(s["kernel_stats"].items(), key=lambda kv: -kv[1]["total_ms"])[:11]:
    tr = s["traffic"].get(k, {})
    print("%-46s calls %4d avg %7.1f us  hbm/launch %6.1f MB" % (k[:46], v["calls"], v["avg_us"], tr.get("hbm_bytes_per_launch", 0) / 1e6))
for m in ("serial", "default"):
    d = json.load(open("$P/bench_%s.json" % m))
    print(m, "ct/s %.0f ms %.4f ntt GB/s %.0f avg launch us %.1f" % (d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["avg_launch_us"]))
PY
