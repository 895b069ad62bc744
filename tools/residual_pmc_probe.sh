#!/bin/bash
# GPU box: FETCH_SIZE of the residual binariser on homogeneous batches (tools/residual_pmc_probe.py)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_probe
mkdir -p "$OUT"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/raw" -o run -- python3 tools/residual_pmc_probe.py > "$OUT/probe.log" 2> "$OUT/probe.err" || exit 1
f=$(find "$OUT/raw" -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if "residual_kernel" in row["Kernel_Name"] or "class_" in row["Kernel_Name"]:
        name = row["Kernel_Name"].split("(")[0].replace("void cabac::", "")
        print("%-40s grid %9s  FETCH_SIZE %12.1f KiB" % (name, row["Grid_Size"], float(row["Counter_Value"])))
PY
cat "$OUT/probe.log"
rm -rf "$OUT/raw"
