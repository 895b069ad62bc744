#!/bin/bash
# encode variants side by side on the BASELINE workloads (GPU box): kernel times and the hash check
for w in C2 C3 C4 C5; do
  for v in 6 7; do
    python3 bench.py --workload $w --steps 5 --warmup 2 --enc-variant $v --no-residual --no-cpu-baseline --no-end-to-end 2>/dev/null > /tmp/vc.json
    python3 - "$w" "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/vc.json"))
print(sys.argv[1], "enc variant", sys.argv[2], d["kernel_ms"], "hash", d["hash_match"], "value", d["value"])
PY
  done
done
