"""GPU: two cabac_hip contexts on two streams — the encode of batch k + 1 beside the decode of batch k (INTEGRATION.md §9,
bench.py's co_scheduled leg) — give what one context gives back to back: the contexts share nothing but the device."""
import numpy as np
import pytest

import helpers as H
from entropy_coding_amd import capi

pytestmark = pytest.mark.gpu


def _batch(rng, n_sub, top):
    lens = [int(x) for x in rng.integers(1, top, size=n_sub)]
    recs = [H.random_records(rng, n - 1, ctx_frac=0.75) for n in lens]
    desc, total = H.make_desc(lens, rng.integers(0, 64, size=n_sub), rng.integers(0, 3, size=n_sub), H.SUB_FINISH | H.SUB_ALIGN_RBSP)
    return desc, np.concatenate(recs), total


@pytest.mark.parametrize("n_sub,top", [(37, 3000), (1100, 600), (3100, 200)])
def test_encode_beside_decode_on_two_streams(n_sub, top):
    import torch
    orc = H.load_oracle()
    rng = np.random.default_rng(n_sub)
    desc, records, total = _batch(rng, n_sub, top)
    want_bytes, want_res = orc.encode_batch(desc, records, total)
    s_enc, s_dec = torch.cuda.Stream(), torch.cuda.Stream()
    h_enc = capi.CabacHip(0, stream=s_enc.cuda_stream)
    h_dec = capi.CabacHip(0, stream=s_dec.cuda_stream)
    t_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).cuda()
    t_rec = torch.from_numpy(records.view(np.int16).copy()).cuda()
    bufs = [torch.zeros(total, dtype=torch.uint8, device="cuda") for _ in range(2)]
    res_e = [torch.zeros(2 * n_sub, dtype=torch.int32, device="cuda") for _ in range(2)]
    res_d = torch.zeros(2 * n_sub, dtype=torch.int32, device="cuda")
    bins = torch.zeros(len(records), dtype=torch.uint8, device="cuda")
    coded = [torch.cuda.Event() for _ in range(2)]
    read = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()
    for k in range(12):
        b = k & 1
        if k >= 2:
            s_enc.wait_event(read[b])
        h_enc.encode_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), bufs[b].data_ptr(), res_e[b].data_ptr())
        coded[b].record(s_enc)
        s_dec.wait_event(coded[b])
        h_dec.decode_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), bufs[b].data_ptr(), bins.data_ptr(), res_d.data_ptr())
        read[b].record(s_dec)
    torch.cuda.synchronize()
    for b in range(2):
        got = res_e[b].cpu().numpy().view(H.RESULT_DTYPE)
        assert np.array_equal(got["n_bits"], want_res["n_bits"]) and not got["flags"].any()
        host = bufs[b].cpu().numpy()
        for s in range(n_sub):
            o, nb = int(desc["byte_offset"][s]), (int(want_res["n_bits"][s]) + 7) // 8
            assert np.array_equal(host[o:o + nb], want_bytes[o:o + nb]), (b, s)
    got_d = res_d.cpu().numpy().view(H.RESULT_DTYPE)
    assert not got_d["flags"].any()
    assert np.array_equal(bins.cpu().numpy(), (records >> 15).astype(np.uint8))
    h_enc.close()
    h_dec.close()
