#!/bin/bash
# GPU box: dynamic instruction mix per kernel (SQ counters) of the default bench
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/sq_mix
mkdir -p "$OUT"
BENCH="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-co-scheduled"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d "$OUT/a" -o run -- python3 $BENCH > "$OUT/a.json" 2> "$OUT/a.err" || exit 1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --kernel-trace --output-format csv -d "$OUT/b" -o run -- python3 $BENCH > "$OUT/b.json" 2> "$OUT/b.err" || exit 2
python3 - "$OUT" <<'PY' > "$OUT/mix.txt"
import csv, glob, sys, re
acc = {}
for sub in ("a", "b"):
    for f in glob.glob(sys.argv[1] + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "cabac" not in row["Kernel_Name"]: continue
            m = re.search(r"cabac::(\w+)(<[^>]*>)?", row["Kernel_Name"])
            k = m.group(1) + (m.group(2) or "")
            d = acc.setdefault(k, {})
            d.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    last = {c: v[0] for c, v in d.items()}   # the first launch of a kernel: the C4 batch of the timed region (later ones: the residual legs)
    print("%-40s " % k + "  ".join("%s %.3g" % (c.replace("SQ_INSTS_", "").replace("SQ_", ""), last[c]) for c in sorted(last)))
PY
cat "$OUT/mix.txt"
rm -rf "$OUT/a" "$OUT/b"
