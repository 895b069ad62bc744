"""Where the residual parser's wave 0 spends its cycles on the bench workload.  Needs a library built with
CABAC_EXTRA_FLAGS=-DCABAC_PARSE_PROFILE (python entropy_coding_amd/build.py --force)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from entropy_coding_amd import capi  # noqa: E402

hip = capi.CabacHip(0)
L = capi.load_library()
out = (ctypes.c_ulonglong * 16)()
L.cabac_hip_debug_parse_prof(out)  # clear
r = bench.residual_leg(hip, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, reps=1)
L.cabac_hip_debug_parse_prof(out)
v = np.array(list(out), np.float64)
names = ["block setup", "last position", "group flag", "group setup", "pass 1", "pass 2+3", "signs+store", "write-out", "whole walk"]
launches = 2.0
print("parse kernel %.3f ms; per launch, wave 0 (memtime ticks = 100 MHz?):" % r["parse"]["kernel_ms"])
for k, n in enumerate(names):
    print("  %-14s %12.0f ticks  %5.1f %%" % (n, v[k] / launches, 100.0 * v[k] / max(v[8], 1)))
print("  positions in pass 1: %d, coded groups: %d, blocks: %d (per launch)" % (v[10] / launches, v[11] / launches, v[12] / launches))
