#!/bin/bash
# GPU box: memory-side request counters of the residual binariser's RECORDS pass on homogeneous batches (the sizes pass of the
# same batch runs first; the second launch of each kernel per shape is the records pass): reads and writes of the L2's
# memory interface, per launch.  64 M coefficients = 262 144 KiB read once; the records written are printed per shape.
export TMPDIR=/tmp
export PROBE_RECORDS=1
OUT=$PWD/gpurun_out/pmc_records
mkdir -p "$OUT"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d "$OUT/raw" -o run -- python3 tools/residual_pmc_probe.py > "$OUT/probe.log" 2> "$OUT/probe.err" || exit 1
f=$(find "$OUT/raw" -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' > "$OUT/records.txt"
import csv, sys
acc = {}
order = []
for row in csv.DictReader(open(sys.argv[1])):
    if "residual_kernel" in row["Kernel_Name"]:
        key = (row["Dispatch_Id"], row["Kernel_Name"].split("(")[0].replace("void cabac::", ""), row["Grid_Size"])
        if key not in acc:
            acc[key] = {}
            order.append(key)
        acc[key][row["Counter_Name"]] = float(row["Counter_Value"])
for key in order:
    c = acc[key]
    rd, rd32 = c.get("TCC_EA0_RDREQ_sum", -1), c.get("TCC_EA0_RDREQ_32B_sum", -1)
    wr, wr64 = c.get("TCC_EA0_WRREQ_sum", -1), c.get("TCC_EA0_WRREQ_64B_sum", -1)
    # a read request is 32 or 64 bytes, a write request 32 or 64 bytes (MI355X_MICROARCH.md, HBM traffic from TCC counters)
    print("%-36s grid %9s  RDREQ %10.0f (32B %9.0f) = %9.0f KiB   WRREQ %10.0f (64B %9.0f) = %9.0f KiB" % (
        key[1], key[2], rd, rd32, (rd32 * 32 + (rd - rd32) * 64) / 1024, wr, wr64, (wr64 * 64 + (wr - wr64) * 32) / 1024))
PY
cat "$OUT/records.txt" "$OUT/probe.log"
rm -rf "$OUT/raw"
