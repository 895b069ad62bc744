#!/bin/bash
# GPU box: the numbers DESIGN.md / README.md quote, one file each under gpurun_out/final_$TAG/
TAG=${1:-r02}
OUT=$PWD/gpurun_out/final_$TAG
mkdir -p "$OUT"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_C4.json" 2> "$OUT/bench_C4.err" || echo "C4 failed"
for w in C2 C3 C5; do
  python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err" || echo "$w failed"
done
python3 bench.py --gpus 1 --workload C5 --strong --steps 5 --warmup 2 > "$OUT/bench_C5_strong_1gpu.json" 2> "$OUT/bench_C5_strong.err" || echo "C5 strong failed"
CABAC_BENCH_BACKEND=gloo CABAC_BENCH_SAME_DEVICE=1 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-residual > "$OUT/bench_C4_2rank_rehearsal.json" 2> "$OUT/bench_2rank.err" || echo "2-rank failed"
CABAC_BENCH_BACKEND=gloo CABAC_BENCH_SAME_DEVICE=1 python3 bench.py --gpus 2 --workload C3 --strong --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_C3_strong_2rank_rehearsal.json" 2> "$OUT/bench_2rank_strong.err" || echo "2-rank strong failed"
python3 tools/batch_scaling.py > "$OUT/batch_scaling.txt" 2>&1 || echo "batch scaling failed"
python3 tools/batch_scaling.py small > "$OUT/batch_scaling_small.txt" 2>&1 || echo "batch scaling (small) failed"
python3 tools/e2e_probe.py > "$OUT/e2e_chunks.txt" 2>&1 || echo "e2e probe failed"
python3 tools/pcie_probe.py > "$OUT/pcie_probe.txt" 2>&1 || echo "pcie probe failed"
ls -la "$OUT"
