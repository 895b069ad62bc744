"""What FETCH_SIZE reports for the residual binariser's access patterns: homogeneous batches of one block shape (sizes
pass only), 64 M coefficients = 256 MiB read exactly once if nothing is re-fetched.  Run under
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -o run -- python3 tools/residual_pmc_probe.py
and compare the raw FETCH_SIZE (KiB) of each launch with 262 144 KiB."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from entropy_coding_amd import capi, workload as W
hip = capi.CabacHip(0)
total = 64 << 20
for (w, h) in [(4, 4), (8, 8), (16, 8), (16, 16), (32, 32)]:
    n = total // (w * h)
    uniq = min(n, max(1024, (4 << 20) // (w * h)))
    rng = np.random.default_rng(1)
    blocks = W._blocks(rng, uniq, w, h, 0.12)
    tus = np.zeros(uniq, capi.TU_DTYPE)
    tus["log2_width"] = int(np.log2(w)); tus["log2_height"] = int(np.log2(h)); tus["flags"] = 3
    tus["coeff_offset"] = np.arange(uniq, dtype=np.uint64) * np.uint64(w * h)
    copies = n // uniq
    all_tus = np.tile(tus, copies)
    all_tus["coeff_offset"] += np.repeat(np.arange(copies, dtype=np.uint64) * np.uint64(uniq * w * h), uniq)
    t_tu = torch.from_numpy(all_tus.view(np.uint8).reshape(-1).copy()).cuda()
    t_co = torch.from_numpy(blocks.reshape(-1)).cuda().repeat(copies)
    t_cnt = torch.zeros(len(all_tus), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    hip.residual_device(len(all_tus), t_tu.data_ptr(), t_co.data_ptr(), 0, t_cnt.data_ptr(), 0, 0)
    hip.synchronize()
    if os.environ.get("PROBE_RECORDS") == "1":   # ... and the records pass (tools/residual_pmc_records.sh)
        cnt = t_cnt.to(torch.int64)
        t_off = torch.cumsum(cnt, 0) - cnt
        n_rec = int(cnt.sum().item())
        t_rec = torch.zeros(n_rec + 64, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        hip.residual_device(len(all_tus), t_tu.data_ptr(), t_co.data_ptr(), t_off.data_ptr(), t_cnt.data_ptr(), 0, t_rec.data_ptr())
        hip.synchronize()
        print("shape %dx%d: records pass wrote %d records = %d KiB" % (w, h, n_rec, n_rec * 2 >> 10), flush=True)
        del t_rec, t_off, cnt
    print("shape %dx%d: %d blocks, %d KiB of coefficients, %d KiB of descriptors" % (w, h, len(all_tus), total * 4 >> 10, len(all_tus) * 16 >> 10), flush=True)
hip.close()
