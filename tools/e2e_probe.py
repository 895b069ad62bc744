"""Host-pointer encode / decode (pinned caller memory) against the number of chunks (CABAC_HIP_CHUNKS), C4 batch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from entropy_coding_amd import capi
from entropy_coding_amd.workload import CONFIGS, build_batch
cfg = CONFIGS["C4"]
desc, records, total = build_batch(cfg)
n_bins = int(desc["n_records"].sum())
keep = [capi.PinnedArray(records.shape, np.uint16), capi.PinnedArray((total,), np.uint8), capi.PinnedArray((len(records),), np.uint8)]
h_rec, h_out, h_bins = (k.array for k in keep)
h_rec[:] = records
for chunks in (1, 2, 4, 8, 0):
    os.environ["CABAC_HIP_CHUNKS"] = str(chunks)
    hip = capi.CabacHip(0)
    te, td = [], []
    for _ in range(5):
        t0 = time.perf_counter(); _, res = hip.encode_batch(desc, h_rec, total, out=h_out); te.append(time.perf_counter() - t0)
    nb = (res["n_bits"].astype(np.int64) + 7) // 8
    dd = desc.copy(); dd["byte_capacity"] = nb
    for _ in range(5):
        t0 = time.perf_counter(); _, rd = hip.decode_batch(dd, h_rec, h_out, bins=h_bins); td.append(time.perf_counter() - t0)
    print("chunks %d: encode %.3f ms  decode %.3f ms (slots as the encoder left them: %d MB of byte slots H2D)" % (chunks, min(te[1:]) * 1e3, min(td[1:]) * 1e3, total >> 20))
    hip.close()
