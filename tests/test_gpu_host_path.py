"""The host-pointer path of the C ABI (cabac_hip_encode_batch / cabac_hip_decode_batch): pinned caller memory DMA'd
where it lies, pageable memory through the pinned bounce ring, batches cut into chunks that overlap H2D / kernel / D2H
on separate streams, coded substreams compacted on the device and copied back once per chunk.  Whatever the route, the
bytes, bit counts and flags are the oracle's."""
import os

import numpy as np
import pytest

import helpers as H
from entropy_coding_amd import capi

pytestmark = pytest.mark.gpu


def _ragged_batch(rng, n_sub, max_len):
    lens = [0, 1, 2, 17][: min(4, n_sub)] + [int(x) for x in rng.integers(0, max_len, size=max(n_sub - 4, 0))]
    recs = [H.random_records(rng, max(n - 1, 0), end_trm=(n > 0)) for n in lens]
    lens = [len(r) for r in recs]
    records = np.concatenate(recs)
    desc, total = H.make_desc(lens, rng.integers(0, 64, size=n_sub), rng.integers(0, 3, size=n_sub),
                              H.SUB_FINISH | H.SUB_ALIGN_RBSP)
    return desc, records, total


def _check_against_oracle(hip, desc, records, total, pinned):
    orc = H.load_oracle()
    keep = []
    if pinned:
        keep = [capi.PinnedArray(records.shape, np.uint16), capi.PinnedArray((max(total, 1),), np.uint8),
                capi.PinnedArray((max(len(records), 1),), np.uint8)]
        h_rec, h_out, h_bins = (k.array for k in keep)
        h_rec[:] = records
        h_out[:] = 0
        assert capi.host_is_pinned(h_rec) and capi.host_is_pinned(h_out[5:100])
    else:
        h_rec, h_out, h_bins = records, np.zeros(max(total, 1), np.uint8), np.zeros(max(len(records), 1), np.uint8)
        assert not capi.host_is_pinned(h_rec)
    out_g, res_g = hip.encode_batch(desc, h_rec, total, check=False, out=h_out)
    out_o, res_o = orc.encode_batch(desc, records, total)
    assert np.array_equal(res_g["n_bits"], res_o["n_bits"]) and np.array_equal(res_g["flags"], res_o["flags"])
    for s in range(len(desc)):
        o, nb = int(desc["byte_offset"][s]), (int(res_o["n_bits"][s]) + 7) // 8
        assert np.array_equal(out_g[o:o + nb], out_o[o:o + nb]), s
    # the payload form: the same substreams back to back (addSubstream order), no slots on the host
    nbytes = np.minimum((res_o["n_bits"].astype(np.int64) + 7) // 8, desc["byte_capacity"].astype(np.int64))
    pay = capi.PinnedArray((max(int(nbytes.sum()), 1),), np.uint8) if pinned else None
    payload = pay.array if pinned else np.zeros(max(int(nbytes.sum()), 1), np.uint8)
    offs, res_p = hip.encode_batch_payload(desc, h_rec, payload, check=False)
    assert np.array_equal(res_p["n_bits"], res_o["n_bits"]) and np.array_equal(res_p["flags"], res_o["flags"])
    assert np.array_equal(offs, np.concatenate([[0], np.cumsum(nbytes)]).astype(np.uint64))
    for s in range(0, len(desc), 7):
        o = int(desc["byte_offset"][s])
        assert np.array_equal(payload[int(offs[s]): int(offs[s + 1])], out_o[o:o + int(nbytes[s])]), s
    if pay is not None:
        pay.close()
    dd = desc.copy()
    dd["byte_capacity"] = (res_o["n_bits"] + 7) // 8
    bins_g, rd = hip.decode_batch(dd, h_rec, h_out, check=False, bins=h_bins)
    bins_o, ro = orc.decode_batch(dd, records, out_o)
    assert np.array_equal(rd["flags"], ro["flags"]) and np.array_equal(rd["n_bits"], ro["n_bits"])
    assert np.array_equal(bins_g[: len(records)], bins_o) and np.array_equal(bins_o, (records >> 15).astype(np.uint8))
    # ... and with the bins packed eight to a byte (cabac_hip_decode_batch_packed): the same bins, bit r & 7 of byte r >> 3
    pk = capi.PinnedArray(((len(records) + 7) // 8 + 1,), np.uint8) if pinned else None
    packed, rp = hip.decode_batch_packed(dd, h_rec, h_out, check=False, packed=pk.array if pinned else None)
    assert np.array_equal(rp["flags"], ro["flags"]) and np.array_equal(rp["n_bits"], ro["n_bits"])
    unpacked = np.unpackbits(packed, bitorder="little")[: len(records)]
    inside = np.zeros(len(records), bool)                    # (records between substreams, if any, are nobody's)
    for s in range(len(dd)):
        inside[int(dd["rec_offset"][s]): int(dd["rec_offset"][s]) + int(dd["n_records"][s])] = True
    assert np.array_equal(unpacked[inside], bins_o[inside])
    if pk is not None:
        pk.close()
    for k in keep:
        k.close()


@pytest.mark.parametrize("chunks", [0, 1, 2, 3, 8])
@pytest.mark.parametrize("pinned", [False, True])
def test_host_path_matches_oracle(chunks, pinned):
    """5 000 ragged substreams (empty ones among them) through every chunk count, from pinned and from pageable memory."""
    os.environ["CABAC_HIP_CHUNKS"] = str(chunks)   # read when the ctx sets up its streams
    try:
        hip = capi.CabacHip(0)
        rng = np.random.default_rng(1000 + chunks)
        desc, records, total = _ragged_batch(rng, 5000, 1500)
        _check_against_oracle(hip, desc, records, total, pinned)
        # the same ctx again with another batch: staging buffers and rings are reused
        desc, records, total = _ragged_batch(rng, 700, 9000)
        _check_against_oracle(hip, desc, records, total, pinned)
        hip.close()
    finally:
        del os.environ["CABAC_HIP_CHUNKS"]


def test_host_path_large_pageable_batch_uses_the_whole_ring():
    """More than ring depth x block size in every direction (records 40 MB, bins 20 MB) from pageable memory."""
    hip = capi.CabacHip(0)
    rng = np.random.default_rng(5)
    n_sub = 4096
    recs = H.random_records(rng, 4999)
    records = np.tile(recs, n_sub)
    desc, total = H.make_desc([len(recs)] * n_sub, [30] * n_sub, [2] * n_sub, H.SUB_FINISH | H.SUB_ALIGN_RBSP)
    out, res = hip.encode_batch(desc, records, total)
    one, nbits = H.load_oracle().encode_records(recs, 30, 2, 3)
    assert (res["n_bits"] == nbits).all() and not res["flags"].any()
    for s in range(0, n_sub, 113):
        o = int(desc["byte_offset"][s])
        assert np.array_equal(out[o:o + len(one)], one)
    dd = desc.copy()
    dd["byte_capacity"] = (res["n_bits"] + 7) // 8
    bins, rd = hip.decode_batch(dd, records, out)
    assert not rd["flags"].any() and np.array_equal(bins, (records >> 15).astype(np.uint8))
    hip.close()


def test_host_path_unordered_descriptors_fall_back_to_one_chunk():
    os.environ["CABAC_HIP_CHUNKS"] = "4"
    try:
        hip = capi.CabacHip(0)
        rng = np.random.default_rng(8)
        desc, records, total = _ragged_batch(rng, 3000, 800)
        perm = rng.permutation(len(desc))
        _check_against_oracle(hip, desc[perm].copy(), records, total, False)
        hip.close()
    finally:
        del os.environ["CABAC_HIP_CHUNKS"]


def test_host_memory_api():
    L = capi.load_library()
    a = capi.PinnedArray((1000,), np.uint16)
    assert capi.host_is_pinned(a.array) and capi.host_is_pinned(a.array[10:20])
    plain = np.zeros(1 << 16, np.uint8)
    assert not capi.host_is_pinned(plain)
    assert L.cabac_hip_host_free(capi.vp(plain.ctypes.data)) == -2          # not from cabac_hip_host_alloc
    assert L.cabac_hip_host_register(capi.vp(plain.ctypes.data), plain.nbytes) == 0
    assert capi.host_is_pinned(plain)
    hip = capi.CabacHip(0)                                                    # a registered buffer is DMA'd in place
    rec = H.random_records(np.random.default_rng(1), 3000)
    buf = plain[: 2 * len(rec)].view(np.uint16)
    buf[:] = rec
    desc, total = H.make_desc([len(rec)], [32], [2], H.SUB_FINISH)
    out, res = hip.encode_batch(desc, buf, total)
    want, nbits = H.load_oracle().encode_records(rec, 32, 2, 1)
    assert int(res["n_bits"][0]) == nbits and np.array_equal(out[: len(want)], want)
    hip.close()
    assert L.cabac_hip_host_unregister(capi.vp(plain.ctypes.data)) == 0
    assert L.cabac_hip_host_unregister(capi.vp(plain.ctypes.data)) == -2
    assert not capi.host_is_pinned(plain)
    a.close()


@pytest.mark.parametrize("pinned", [False, True])
def test_early_error_return_leaves_the_pipeline_idle(pinned):
    """cabac_hip_encode_batch_payload with two chunks and a payload that only holds the first: the call returns
    CABAC_HIP_ERR_INVALID from the middle of the pipeline.  Nothing of it may still be in flight or waiting in the bounce
    ring afterwards: the payload buffer is released at once, and the next call on the same ctx is an ordinary one."""
    os.environ["CABAC_HIP_CHUNKS"] = "2"
    try:
        hip = capi.CabacHip(0)
        rng = np.random.default_rng(77)
        desc, records, total = _ragged_batch(rng, 3000, 2500)
        out_o, res_o = H.load_oracle().encode_batch(desc, records, total)
        nbytes = np.minimum((res_o["n_bits"].astype(np.int64) + 7) // 8, desc["byte_capacity"].astype(np.int64))
        small = int(nbytes.sum()) * 2 // 3              # the first chunk (half of the records) fits, the second does not
        pay = capi.PinnedArray((small,), np.uint8) if pinned else None
        payload = pay.array if pinned else np.zeros(small, np.uint8)
        with pytest.raises(capi.CabacHipError) as e:
            hip.encode_batch_payload(desc, records, payload)
        assert e.value.status == -2 and "payload_capacity" in str(e.value)
        if pay is not None:
            pay.close()                                 # freed right after the failed call
        del payload
        junk = [np.full(small, 0xEE, np.uint8) for _ in range(4)]   # the heap block of the pageable payload is reused
        _check_against_oracle(hip, desc, records, total, pinned)
        assert all((j == 0xEE).all() for j in junk)    # no stale bounce block was copied into the old payload's memory
        hip.close()
    finally:
        del os.environ["CABAC_HIP_CHUNKS"]
