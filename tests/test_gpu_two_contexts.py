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


@pytest.mark.gpu
def test_default_stream_is_adopted_not_replaced():
    """cabac_hip.h, stream ordering contract, form (a) on the device's DEFAULT stream: torch's current stream has the handle 0,
    which the C ABI would read as "the ctx's own stream" — capi passes CABAC_HIP_STREAM_DEFAULT for it.  With the ctx really
    on that stream a long fill in front of the launch and the read behind it need neither events nor a host synchronisation;
    on a stream of its own (what a handle of 0 silently gave before) the encoder would run beside the fills."""
    import torch
    assert torch.cuda.current_stream().cuda_stream == 0
    orc = H.load_oracle()
    rng = np.random.default_rng(98)
    desc, records, total = _batch(rng, 1500, 2500)
    want_bytes, want_res = orc.encode_batch(desc, records, total)
    hip = H.gpu_ctx()
    src_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).cuda()
    src_rec = torch.from_numpy(records.view(np.int16).copy()).cuda()
    for _ in range(3):
        big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
        big.fill_(0xA5)                          # a long fill in front, then the operands are produced ON the stream
        t_desc, t_rec = torch.zeros_like(src_desc), torch.zeros_like(src_rec)
        t_desc.copy_(src_desc), t_rec.copy_(src_rec)
        out = torch.full((total,), 0x5A, dtype=torch.uint8, device="cuda")
        res = torch.zeros(2 * len(desc), dtype=torch.int32, device="cuda")
        hip.encode_device(len(desc), t_desc.data_ptr(), t_rec.data_ptr(), out.data_ptr(), res.data_ptr())
        got = res.cpu().numpy().view(H.RESULT_DTYPE)
        host = out.cpu().numpy()
        assert np.array_equal(got["n_bits"], want_res["n_bits"]) and not got["flags"].any()
        for s in range(0, len(desc), 3):
            o, nb = int(desc["byte_offset"][s]), (int(want_res["n_bits"][s]) + 7) // 8
            assert np.array_equal(host[o:o + nb], want_bytes[o:o + nb]), s
        del big, t_desc, t_rec
    hip.close()


@pytest.mark.gpu
def test_own_stream_ordered_with_events():
    """cabac_hip.h, stream ordering contract, form (b): the ctx keeps its own (non-blocking) stream; the producer's fills are
    ordered in front of the launch with cabac_hip_wait_event and the consumer's reads behind it with cabac_hip_record_event —
    no host synchronisation anywhere between the fill and the read."""
    import torch
    orc = H.load_oracle()
    rng = np.random.default_rng(99)
    desc, records, total = _batch(rng, 1500, 2500)
    want_bytes, want_res = orc.encode_batch(desc, records, total)
    hip = capi.CabacHip(0)                       # its own stream
    t_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).cuda()
    t_rec = torch.from_numpy(records.view(np.int16).copy()).cuda()
    filled, coded = torch.cuda.Event(), torch.cuda.Event()
    coded.record()                               # creates the HIP event; re-recorded on the ctx's stream below
    for _ in range(3):
        big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
        big.fill_(0xA5)                          # a long fill in front of the buffers' own, on torch's stream
        out = torch.full((total,), 0x5A, dtype=torch.uint8, device="cuda")
        res = torch.zeros(2 * len(desc), dtype=torch.int32, device="cuda")
        filled.record()
        hip.wait_event(filled.cuda_event)
        hip.encode_device(len(desc), t_desc.data_ptr(), t_rec.data_ptr(), out.data_ptr(), res.data_ptr())
        hip.record_event(coded.cuda_event)
        torch.cuda.current_stream().wait_event(coded)
        got = res.cpu().numpy().view(H.RESULT_DTYPE)
        host = out.cpu().numpy()
        assert np.array_equal(got["n_bits"], want_res["n_bits"]) and not got["flags"].any()
        for s in range(0, len(desc), 3):
            o, nb = int(desc["byte_offset"][s]), (int(want_res["n_bits"][s]) + 7) // 8
            assert np.array_equal(host[o:o + nb], want_bytes[o:o + nb]), s
        del big
    hip.close()


def test_context_lifetime_in_a_fresh_process_exits_normally():
    """A child interpreter creates contexts, uses the device path and the host path (so that the copy / kernel streams and the
    pinned bounce rings exist), closes one context, leaves the other and a pinned array to capi.close_all(), and ends through
    the interpreter's and the libraries' ordinary teardown: exit status 0, nothing on stderr from the runtime."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import helpers as H
from entropy_coding_amd import capi
rng = np.random.default_rng(5)
recs = [H.random_records(rng, int(n)) for n in rng.integers(1, 4000, size=2500)]
desc, total = H.make_desc([len(r) for r in recs], rng.integers(0, 64, size=len(recs)), rng.integers(0, 3, size=len(recs)),
                          H.SUB_FINISH | H.SUB_ALIGN_RBSP)
records = np.concatenate(recs)
want, wres = H.load_oracle().encode_batch(desc, records, total)
a, b = capi.CabacHip(0), H.gpu_ctx()
pin = capi.PinnedArray(records.shape, np.uint16); pin.array[:] = records
for hip, src in ((a, records), (b, pin.array)):
    out, res = hip.encode_batch(desc, src, total)
    assert np.array_equal(res["n_bits"], wres["n_bits"])
    dd = desc.copy(); dd["byte_capacity"] = (res["n_bits"] + 7) // 8
    bins, rd = hip.decode_batch(dd, src, out)
    assert np.array_equal(bins, (records >> 15).astype(np.uint8))
a.close()
capi.close_all()
assert not capi._live
print("OK")
''' % {"tests": H.ROOT + "/tests", "root": H.ROOT}
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "Abort" not in r.stderr and "HSA_STATUS" not in r.stderr, r.stderr[-4000:]


def test_context_left_open_at_exit_does_not_abort():
    """... and a caller that forgets close(): the context, its streams and a pinned array are still alive when the interpreter
    finalizes (capi's __del__ does not release into a runtime that may be unloading).  The process must still end with status 0."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import helpers as H
from entropy_coding_amd import capi
rng = np.random.default_rng(6)
recs = [H.random_records(rng, 500) for _ in range(64)]
desc, total = H.make_desc([len(r) for r in recs], [30] * 64, [2] * 64, H.SUB_FINISH)
hip = capi.CabacHip(0)
keep = capi.PinnedArray((1 << 20,), np.uint8)
out, res = hip.encode_batch(desc, np.concatenate(recs), total)
assert not res["flags"].any()
print("OK")
''' % {"tests": H.ROOT + "/tests", "root": H.ROOT}
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
