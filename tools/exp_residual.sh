#!/bin/bash
# the residual binariser's records pass with parts of the record emission compiled out (the output is then wrong: timing only)
for f in "" "-DCABAC_EXP_NO_EP" "-DCABAC_EXP_NO_CTX" "-DCABAC_EXP_NO_EP -DCABAC_EXP_NO_CTX"; do
  CABAC_EXTRA_FLAGS="$f" python3 -c "
import sys
sys.path.insert(0,'.')
from entropy_coding_amd.build import build_library
build_library(force=True)
" > /dev/null 2>&1
  echo "== flags: $f"
  CABAC_EXTRA_FLAGS="$f" python3 bench.py --no-cpu-baseline --no-end-to-end --no-co-scheduled --steps 2 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['residual']['kernel_ms'], d['residual']['records_match_reference'])
"
done
