#!/bin/bash
# GPU box: the request counters FETCH_SIZE is derived from, for the residual binariser on homogeneous batches
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_tcc2
mkdir -p "$OUT"
rocprofv3 --pmc TCC_MISS_sum TCC_HIT_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d "$OUT/raw" -o run -- python3 tools/residual_pmc_probe.py > "$OUT/probe.log" 2> "$OUT/probe.err" || exit 1
f=$(find "$OUT/raw" -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' > "$OUT/tcc.txt"
import csv, sys
acc = {}
order = []
for row in csv.DictReader(open(sys.argv[1])):
    if "residual_kernel" in row["Kernel_Name"] or "class_" in row["Kernel_Name"]:
        key = (row["Dispatch_Id"], row["Kernel_Name"].split("(")[0].replace("void cabac::", ""), row["Grid_Size"])
        if key not in acc:
            acc[key] = {}
            order.append(key)
        acc[key][row["Counter_Name"]] = float(row["Counter_Value"])
for key in order:
    c = acc[key]
    print("%-36s grid %9s  RDREQ %12.0f  MISS %12.0f  HIT %12.0f" % (key[1], key[2], c.get("TCC_EA0_RDREQ_sum", -1), c.get("TCC_MISS_sum", -1), c.get("TCC_HIT_sum", -1)))
PY
cat "$OUT/tcc.txt" "$OUT/probe.log"
rm -rf "$OUT/raw"
