"""one round of fuzz_parity.py again, with the differing substreams printed:  python3 tests/fuzz_debug.py SEED"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helpers as H
from entropy_coding_amd import capi
import fuzz_parity as fz

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
desc, records, total = fz.draw_batch(rng)
enc = int(rng.choice([0, 5, 6, 7]))
orc = H.load_oracle()
hip = capi.CabacHip(0)
hip.set_variant(enc, 0)
out_o, res_o = orc.encode_batch(desc, records, total)
out_g, res_g = hip.encode_batch(desc, records, total, check=False)
print("encode equal:", np.array_equal(res_g, res_o), "init_id", hex(int(desc["init_id"][0])))
ddesc = desc.copy()
used = (res_o["n_bits"] + 7) // 8
if not (int(desc["init_id"][0]) & H.SUB_ALIGN_RBSP):
    used = np.minimum(used + 2, desc["byte_capacity"])
ddesc["byte_capacity"] = np.where(res_o["flags"] == 0, used, 0)
ddesc["n_records"] = np.where(res_o["flags"] == 0, desc["n_records"], 0)
bins_g, dg = hip.decode_batch(ddesc, records, out_o, check=False)
bins_o, do = orc.decode_batch(ddesc, records, out_o)
bad = np.flatnonzero(((dg["n_bits"] != do["n_bits"]) & ((do["flags"] & H.RES_UNDERRUN) == 0)) | (dg["flags"] != do["flags"]))
print(len(bad), "of", len(desc), "substreams differ")
for s in bad[:12]:
    lo, n = int(desc["rec_offset"][s]), int(desc["n_records"][s])
    r = records[lo:lo + n]
    nb = np.flatnonzero(bins_g[lo:lo + n] != bins_o[lo:lo + n])
    print("sub", s, "n", n, "cap", int(ddesc["byte_capacity"][s]), "enc n_bits", int(res_o["n_bits"][s]), "enc flags", int(res_o["flags"][s]),
          "| hip", int(dg["n_bits"][s]), int(dg["flags"][s]), "orc", int(do["n_bits"][s]), int(do["flags"][s]),
          "| first bin diff", nb[:3], "| align at", np.flatnonzero((r & 0x1FF) == H.REC_ALIGN)[:5], "trm0 at", np.flatnonzero(r == H.REC_TRM)[:5],
          "last recs", [hex(int(x)) for x in r[-3:]])
