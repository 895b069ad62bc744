#!/bin/bash
# Runs ON the GPU box (through gpurun): the rocprofv3 passes the bench numbers are judged against.
#   pass 1  --kernel-trace --stats         per-kernel durations of `python3 bench.py` (default workload)
#   pass 2  --pmc FETCH_SIZE               HBM read requests per dispatch     (counters in passes of their own,
#   pass 3  --pmc WRITE_SIZE               HBM bytes written per dispatch      with --kernel-trace only)
# Output under gpurun_out/prof_$TAG/; tools/pmc_summary.py turns it into profiles/.
set -o pipefail
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end"
python3 $BENCH > "$OUT/bench_line.json" 2> "$OUT/bench_line.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err" || exit 2
MAIN="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-residual --no-co-scheduled"   # the timed kernels only
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_main" -o run -- python3 $MAIN > "$OUT/bench_main_under_rocprof.json" 2> "$OUT/stats_main.err" || exit 5
find "$OUT/stats_main" -name "*kernel_stats.csv" -exec cp {} "$OUT/main_kernel_stats.csv" \;
rm -rf "$OUT/stats_main"
SHORT="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-co-scheduled"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o run -- python3 $SHORT > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o run -- python3 $SHORT > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err" || exit 4
# keep what is small enough to travel back: the stats summary and the per-dispatch counter rows of our kernels
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
for k in fetch write; do
  f=$(find "$OUT/pmc_$k" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$OUT/pmc_$k.csv" <<'PY'
import csv, sys
with open(sys.argv[1]) as f, open(sys.argv[2], "w", newline="") as g:
    r = csv.DictReader(f)
    w = csv.writer(g)
    w.writerow(["Grid_Size", "Kernel_Name", "Counter_Name", "Counter_Value"])
    for row in r:
        if "cabac" in row["Kernel_Name"]:
            w.writerow([row["Grid_Size"], row["Kernel_Name"], row["Counter_Name"], row["Counter_Value"]])
PY
done
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write"
ls -la "$OUT"
