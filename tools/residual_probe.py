"""Per-shape cost of the residual binariser: homogeneous batches of one block size, both passes.
Usage (GPU box): python tools/residual_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from entropy_coding_amd import capi, workload as W  # noqa: E402


def run(hip, w, h, chroma, n_unique, copies, density):
    rng = np.random.default_rng(1)
    blocks = W._blocks(rng, n_unique, w, h, density)
    tus = np.zeros(n_unique, capi.TU_DTYPE)
    tus["log2_width"] = int(np.log2(w)); tus["log2_height"] = int(np.log2(h)); tus["channel"] = chroma
    tus["flags"] = 3
    tus["coeff_offset"] = np.arange(n_unique, dtype=np.uint64) * np.uint64(w * h)
    all_tus = np.tile(tus, copies)
    all_tus["coeff_offset"] += np.repeat(np.arange(copies, dtype=np.uint64) * np.uint64(n_unique * w * h), n_unique)
    n = len(all_tus)
    t_tu = torch.from_numpy(all_tus.view(np.uint8).reshape(-1).copy()).cuda()
    t_co = torch.from_numpy(blocks.reshape(-1)).cuda().repeat(copies)
    t_cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    hip.profile_enable(16)
    for _ in range(4):
        hip.residual_device(n, t_tu.data_ptr(), t_co.data_ptr(), 0, t_cnt.data_ptr(), 0, 0)
    hip.synchronize()
    cnt = t_cnt.to(torch.int64)
    t_off = torch.cumsum(cnt, 0) - cnt
    bins = int(cnt.sum().item())
    t_rec = torch.zeros(max(bins, 1), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    for _ in range(4):
        hip.residual_device(n, t_tu.data_ptr(), t_co.data_ptr(), t_off.data_ptr(), t_cnt.data_ptr(), 0, t_rec.data_ptr())
    ms = [m for k, m in hip.profile_read() if k == 5]
    p1, p2 = float(np.mean(ms[1:4])), float(np.mean(ms[5:8]))
    coefs = n * w * h
    print("%2dx%-2d ch%d dens %.2f: %8d blocks %6.1f Mcoef %6.1f Mbins (%.2f/coef) | pass1 %.3f ms pass2 %.3f ms | %.1f + %.1f ns/block | "
          "%.0f GB/s write pass" % (w, h, chroma, density, n, coefs / 1e6, bins / 1e6, bins / coefs, p1, p2, p1 * 1e6 / n, p2 * 1e6 / n,
                                   (4 * coefs + 2 * bins) / p2 / 1e6), flush=True)


def main():
    hip = capi.CabacHip(0)
    total = 64 << 20  # coefficients per batch
    for dens in (0.12, 1.5):
        for (w, h) in [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (16, 4), (2, 8)]:
            n = total // (w * h)
            uniq = min(n, max(1024, (4 << 20) // (w * h)))
            run(hip, w, h, 0 if min(w, h) >= 4 else 1, uniq, n // uniq, dens)
    hip.close()


if __name__ == "__main__":
    main()
