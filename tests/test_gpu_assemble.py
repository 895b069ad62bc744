"""GPU: substream assembly / split / start-code-emulation count on the device against
numpy concatenation and the oracle's restatement of OutputBitstream::countStartCodeEmulations
(bit_stream.cpp:157-181, pinned to the reference in tests/test_oracle_vs_reference.py)."""
import ctypes

import numpy as np
import pytest

import helpers as H
from entropy_coding_amd import capi

pytestmark = pytest.mark.gpu


def test_assemble_split_count_roundtrip():
    import torch
    hip = H.gpu_ctx()
    orc = H.load_oracle()
    orc.lib.orc_count_emulations.argtypes = [H.u8p, ctypes.c_long]
    rng = np.random.default_rng(12)
    lens = [1, 2, 17, 500] + [int(x) for x in rng.integers(1, 6000, size=300)]
    # low-entropy streams (P(1) tiny on one context) give long zero runs -> start-code emulations
    recs = [H.random_records(rng, n - 1, ctx_frac=1.0, p_one=np.full(379, 0.002), ctx_pool=np.array([7])) if k % 3 == 0
            else H.random_records(rng, n - 1) for k, n in enumerate(lens)]
    records = np.concatenate(recs)
    desc, total = H.make_desc([len(r) for r in recs], rng.integers(0, 64, size=len(recs)), [2] * len(recs),
                              H.SUB_FINISH | H.SUB_ALIGN_RBSP)
    n = len(desc)
    t_desc = torch.from_numpy(desc.view(np.uint8)).cuda()
    t_rec = torch.from_numpy(records.view(np.int16)).cuda()
    t_bytes = torch.zeros(total, dtype=torch.uint8, device="cuda")
    t_res = torch.zeros(2 * n, dtype=torch.int32, device="cuda")
    hip.encode_device(n, t_desc.data_ptr(), t_rec.data_ptr(), t_bytes.data_ptr(), t_res.data_ptr())
    t_pay = torch.zeros(total, dtype=torch.uint8, device="cuda")
    t_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    hip.assemble_device(n, t_desc.data_ptr(), t_res.data_ptr(), t_bytes.data_ptr(), t_pay.data_ptr(), total, t_off.data_ptr())
    t_cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    hip.count_emulations_device(n, t_desc.data_ptr(), t_res.data_ptr(), t_bytes.data_ptr(), t_cnt.data_ptr())
    t_back = torch.zeros(total, dtype=torch.uint8, device="cuda")
    hip.split_device(n, t_desc.data_ptr(), t_off.data_ptr(), t_pay.data_ptr(), t_back.data_ptr())
    hip.synchronize()
    res = t_res.cpu().numpy().view(capi.RESULT_DTYPE)
    out = t_bytes.cpu().numpy()
    sizes = (res["n_bits"].astype(np.int64) + 7) // 8
    streams = [out[int(desc["byte_offset"][s]):int(desc["byte_offset"][s]) + int(sizes[s])] for s in range(n)]
    want = np.concatenate(streams)
    off = t_off.cpu().numpy()
    assert np.array_equal(off, np.concatenate([[0], np.cumsum(sizes)]))
    assert np.array_equal(t_pay.cpu().numpy()[: len(want)], want)            # addSubstream order and bytes
    back = t_back.cpu().numpy()
    for s in range(n):
        o = int(desc["byte_offset"][s])
        assert np.array_equal(back[o:o + int(sizes[s])], streams[s])          # extractSubstream inverse
    cnt = t_cnt.cpu().numpy()
    want_cnt = [orc.lib.orc_count_emulations(H._ptr(np.ascontiguousarray(b), H.u8p), len(b)) for b in streams]
    assert np.array_equal(cnt, np.array(want_cnt, np.int32)) and max(want_cnt) > 0
    hip.close()


def test_count_emulations_on_crafted_zero_runs():
    """Zero runs of every length and alignment (the kernel takes 64 bytes per step and carries the run length across
    steps), bytes 0..4 mixed in: against the oracle's countStartCodeEmulations (pinned to the reference)."""
    import torch
    hip = H.gpu_ctx()
    orc = H.load_oracle()
    orc.lib.orc_count_emulations.argtypes = [H.u8p, ctypes.c_long]
    rng = np.random.default_rng(5)
    streams = [np.zeros(n, np.uint8) for n in (0, 1, 2, 3, 4, 63, 64, 65, 127, 128, 129, 200, 1000)]
    for k in range(600):
        n = int(rng.integers(1, 400))
        p0 = float(rng.choice([0.3, 0.6, 0.9, 0.98]))
        b = rng.choice(np.array([0, 1, 2, 3, 4, 255], np.uint8), size=n, p=[p0] + [(1 - p0) / 5] * 5)
        streams.append(b.astype(np.uint8))
    for lead in range(0, 70, 3):           # a long run placed at every offset around the 64-byte step
        for run in (2, 3, 4, 63, 64, 65, 130):
            streams.append(np.concatenate([np.full(lead, 9, np.uint8), np.zeros(run, np.uint8), np.array([1, 0, 0, 2, 0, 0, 0, 3, 7], np.uint8)]))
    n = len(streams)
    desc = np.zeros(n, H.DESC_DTYPE)
    cap = np.array([(len(b) + 15) // 16 * 16 + 16 for b in streams], np.uint64)
    desc["byte_offset"] = np.concatenate([[0], np.cumsum(cap)[:-1]])
    desc["byte_capacity"] = cap
    res = np.zeros(n, H.RESULT_DTYPE)
    res["n_bits"] = [8 * len(b) for b in streams]
    buf = np.full(int(cap.sum()), 0, np.uint8)
    for k, b in enumerate(streams):
        buf[int(desc["byte_offset"][k]): int(desc["byte_offset"][k]) + len(b)] = b
    t_desc = torch.from_numpy(desc.view(np.uint8).copy()).cuda()
    t_res = torch.from_numpy(res.view(np.uint8).copy()).cuda()
    t_buf = torch.from_numpy(buf).cuda()
    t_cnt = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    hip.count_emulations_device(n, t_desc.data_ptr(), t_res.data_ptr(), t_buf.data_ptr(), t_cnt.data_ptr())
    hip.synchronize()
    got = t_cnt.cpu().numpy()
    want = np.array([orc.lib.orc_count_emulations(H._ptr(np.ascontiguousarray(b), H.u8p), len(b)) for b in streams], np.int32)
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]
    assert want.max() > 30
    hip.close()


def test_pack_shard_gathers_on_the_device():
    """sharding.pack_shard on a device-resident batch: one gather launch (cabac_hip_gather_records_device) gives the shard the
    host-side slicing gives — odd and even offsets and lengths, empty substreams, a permuted selection."""
    import torch
    from entropy_coding_amd import sharding
    rng = np.random.default_rng(12)
    lens = [0, 1, 2, 3, 7, 8, 9, 1000, 1001] + [int(x) for x in rng.integers(0, 5000, size=300)]
    recs = [rng.integers(0, 1 << 16, size=n, dtype=np.uint16) for n in lens]
    desc, total = H.make_desc(lens, [30] * len(lens), [2] * len(lens))
    records = np.concatenate(recs)
    for trial in range(3):
        idxs = np.sort(rng.choice(len(lens), size=len(lens) // 2, replace=False)) if trial else np.arange(len(lens))
        want_d, want_r, want_t = sharding.pack_shard(desc, records, idxs)
        got_d, got_r, got_t = sharding.pack_shard(desc, torch.from_numpy(records.view(np.int16)).cuda(), idxs)
        torch.cuda.synchronize()
        assert np.array_equal(got_d, want_d) and got_t == want_t
        assert np.array_equal(got_r.cpu().numpy().view(np.uint16), want_r)
