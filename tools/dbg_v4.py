import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, helpers as H
from entropy_coding_amd import capi
hip = capi.CabacHip(0); hip.set_variant(4,4)
orc = H.load_oracle()
rng = np.random.default_rng(400)
lens = [1, 2, 63, 64, 65, 128, 129, 257, 1000, 5000] + [int(x) for x in rng.integers(1, 4000, size=30)]
recs = [H.random_records(rng, n - 1, ctx_frac=float(rng.choice([0.0, 0.5, 0.75, 1.0]))) for n in lens]
records = np.concatenate(recs)
desc, total = H.make_desc(lens, rng.integers(0, 64, size=len(lens)), rng.integers(0, 3, size=len(lens)), H.SUB_FINISH | H.SUB_ALIGN_RBSP)
data, res = orc.encode_batch(desc, records, total)
ddesc = desc.copy(); ddesc["byte_capacity"] = (res["n_bits"] + 7) // 8
bins_g, res_g = hip.decode_batch(ddesc, records, data, check=False)
bins_o, res_o = orc.decode_batch(ddesc, records, data)
for s in range(len(lens)):
    o=int(desc["rec_offset"][s]); n=lens[s]
    ok = np.array_equal(bins_g[o:o+n], bins_o[o:o+n])
    if res_g["flags"][s] or not ok or res_g["n_bits"][s]!=res_o["n_bits"][s]:
        bad = np.nonzero(bins_g[o:o+n]!=bins_o[o:o+n])[0]
        r = records[o:o+n]
        print(s, n, "flags", res_g["flags"][s], "nbits", res_g["n_bits"][s], res_o["n_bits"][s], "cap", ddesc["byte_capacity"][s], "first bad", bad[:3], [hex(x) for x in r[bad[:1]]] if len(bad) else "")
for s in (10, 11, 34):
    o=int(desc["rec_offset"][s]); n=lens[s]
    bad = np.nonzero(bins_g[o:o+n]!=bins_o[o:o+n])[0]
    print("sub", s, "bad positions", bad.tolist()[:20])
    for b in bad[:3]:
        g0 = (b//16)*16
        print("  group", g0, [hex(int(x)) for x in records[o+g0:o+g0+16]])
        print("  got ", bins_g[o+g0:o+g0+16].tolist())
        print("  want", bins_o[o+g0:o+g0+16].tolist())
