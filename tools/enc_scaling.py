"""Encode kernel time against batch size and variant: C4-type substreams (16 384 bins), 16 ... 16 384 of them.
python tools/enc_scaling.py [variants...]   (on an MI355X)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from entropy_coding_amd import capi
from entropy_coding_amd.workload import CONFIGS, build_batch

variants = [int(v) for v in sys.argv[1:]] or [6, 7]
hip = capi.CabacHip(0, stream=torch.cuda.current_stream().cuda_stream)
for n_sub in (16, 64, 256, 1024, 2048, 4096, 8192, 16384):
    desc, records, bytes_total = build_batch(CONFIGS["C4"], first=0, count=n_sub)
    t_desc = torch.from_numpy(desc.view(np.uint8)).cuda()
    t_rec = torch.from_numpy(records.view(np.int16)).cuda()
    t_bytes = torch.zeros(bytes_total, dtype=torch.uint8, device="cuda")
    t_re = torch.zeros(n_sub * 2, dtype=torch.int32, device="cuda")
    line = "%6d substreams:" % n_sub
    for v in variants:
        hip.set_variant(v, 0)
        hip.encode_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), t_bytes.data_ptr(), t_re.data_ptr())
        torch.cuda.synchronize()
        hip.profile_enable(5)
        for _ in range(5):
            hip.encode_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), t_bytes.data_ptr(), t_re.data_ptr())
        prof = hip.profile_read()
        ms = float(np.mean([m for kk, m in prof if kk == 0]))
        line += "  v%d %.3f ms (%.0f ns / 16-bin step)" % (v, ms, ms * 1e6 / 1024)
    print(line, flush=True)
