"""Throughput against batch size: C4-type substreams (16 384 bins, 75 % context coded), 1 024 ... 16 384 of them in
one launch.  python tools/batch_scaling.py [decode variants ...]  (on an MI355X; default: 0 = the dispatch, 4 = four
substreams per wave, 8 = sixteen, 1 = one)
python tools/batch_scaling.py small [variants ...]: 16 ... 2 048 substreams instead (the one-substream-per-wave geometry's range)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from entropy_coding_amd import capi
from entropy_coding_amd.workload import CONFIGS, build_batch

hip = capi.CabacHip(0, stream=torch.cuda.current_stream().cuda_stream)
small = len(sys.argv) > 1 and sys.argv[1] == "small"
share = len(sys.argv) > 1 and sys.argv[1] == "share"   # the shares of C5 sharded longest-first over 2, 4, 8 GPUs (long + short halves)
dec_variants = [int(v) for v in sys.argv[(2 if small or share else 1):]] or ([0, 4, 1] if small or share else [0, 4, 8])
for n_sub in ((4096, 2048, 1024) if share else (16, 64, 256, 512, 768, 1024, 1536, 2048) if small else (1024, 2048, 4096, 8192, 12288, 16384, 32768)):
    desc, records, bytes_total = build_batch(CONFIGS["C5" if share else "C4"], first=0, count=n_sub)
    n_bins = int(desc["n_records"].astype(np.int64).sum())
    t_desc = torch.from_numpy(desc.view(np.uint8)).cuda()
    t_rec = torch.from_numpy(records.view(np.int16)).cuda()
    t_bytes = torch.zeros(bytes_total, dtype=torch.uint8, device="cuda")
    t_re = torch.zeros(n_sub * 2, dtype=torch.int32, device="cuda")
    t_rd = torch.zeros(n_sub * 2, dtype=torch.int32, device="cuda")
    t_bins = torch.zeros(len(records), dtype=torch.uint8, device="cuda")
    t_est = torch.zeros(n_sub, dtype=torch.int64, device="cuda")
    for dec in dec_variants:
      hip.set_variant(0, dec)
      def step():
          hip.encode_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), t_bytes.data_ptr(), t_re.data_ptr())
          hip.decode_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), t_bytes.data_ptr(), t_bins.data_ptr(), t_rd.data_ptr())
          hip.estimate_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), t_est.data_ptr(), 0)
      step(); torch.cuda.synchronize()
      hip.profile_enable(3 * 5)
      for _ in range(5):
          step()
      prof = hip.profile_read()
      ms = {k: float(np.mean([m for kk, m in prof if kk == k])) for k in (0, 1, 4)}
      ok = bool(torch.equal(t_bins, (t_rec < 0).to(torch.uint8))) and not bool(t_re[1::2].any()) and not bool(t_rd[1::2].any())
      print("%6d substreams, decode variant %d: encode %.3f ms, decode %.3f ms, estimate %.3f ms -> %.1f Gbins/s enc+dec, estimate %.1f Gbins/s, round trip %s"
            % (n_sub, dec, ms[0], ms[1], ms[4], 2 * n_bins / ((ms[0] + ms[1]) * 1e-3) / 1e9, n_bins / (ms[4] * 1e-3) / 1e9, "ok" if ok else "MISMATCH"))
