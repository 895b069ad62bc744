import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, bench
from entropy_coding_amd import capi
hip = capi.CabacHip(0)
r = bench.residual_leg(hip, 4096, reps=30)
print(r["parse"], r["kernel_ms"], r["to_bytes"])
