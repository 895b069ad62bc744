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
print("  wave walk time: min %d  max %d  mean %.0f ticks (2 launches pooled)" % (v[13], v[14], v[15] / (2 * 4096)))
print("  positions in pass 1: %d, coded groups: %d, blocks: %d (per launch)" % (v[10] / launches, v[11] / launches, v[12] / launches))

w = (ctypes.c_ulonglong * (3 * 8192))()
L.cabac_hip_debug_parse_waves(w)
w = np.array(list(w), np.uint64).reshape(-1, 3)[:4096]
t0 = w[:, 0].min()
start, end, hw = (w[:, 0] - t0).astype(np.int64), (w[:, 1] - t0).astype(np.int64), w[:, 2]
dur = end - start
hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
# HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
print("start spread %d, end max %d" % (start.max(), end.max()))
key_cu = xcc * 1000 + se * 100 + sh * 16 + cu
ucu, cnt = np.unique(key_cu, return_counts=True)
print("distinct CUs %d, waves per CU: min %d max %d" % (len(ucu), cnt.min(), cnt.max()), np.bincount(cnt))
key_simd = key_cu * 4 + simd
us, cs = np.unique(key_simd, return_counts=True)
print("distinct SIMDs %d, waves per SIMD histogram" % len(us), np.bincount(cs))
per = {}
for k, c in zip(us, cs):
    per.setdefault(int(c), []).append(dur[key_simd == k].mean())
for c in sorted(per):
    print("  SIMDs with %d waves: mean wave time %.0f" % (c, np.mean(per[c])))
for c in sorted(set(cnt)):
    sel = np.isin(key_cu, ucu[cnt == c])
    print("  CUs with %d waves: mean wave time %.0f" % (c, dur[sel].mean()))
for x in range(8):
    print("  xcc %d: waves %d mean %.0f" % (x, (xcc == x).sum(), dur[xcc == x].mean() if (xcc == x).any() else 0))
# when does each SIMD / CU run dry?  (the kernel ends with the last one; s_memtime has its own base on every XCC, so
# times are taken relative to the first start on the same SIMD / CU)
valid = w[:, 1] > 0
sraw, eraw = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64)
e_simd = np.array([eraw[valid & (key_simd == k)].max() - sraw[valid & (key_simd == k)].min() for k in us])
e_cu = np.array([eraw[valid & (key_cu == k)].max() - sraw[valid & (key_cu == k)].min() for k in ucu])
print("a SIMD is busy for: min %d  mean %.0f  max %d ticks" % (e_simd.min(), e_simd.mean(), e_simd.max()))
print("a CU is busy for:   min %d  mean %.0f  max %d ticks" % (e_cu.min(), e_cu.mean(), e_cu.max()))
print("wave durations: percentiles 0/25/50/75/100:", np.percentile((eraw - sraw)[valid], [0, 25, 50, 75, 100]).astype(np.int64))
