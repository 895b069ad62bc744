"""Coefficients -> bytes with the block records spliced into the substreams on the device (cabac_hip_encode_batch_residual /
cabac_hip_encode_residual_device, csrc/cabac_splice.hip; SURVEY.md section 8 row f2, the writer's side: in the reference the
bins of CABACWriter::residual_coding go straight into the encoder, cabac_writer.cpp:2424-2525).  The expected bytes are the
oracle's: each block's records (orc_residual_records, pinned to the reference's writer by tests/test_residual_oracle.py)
inserted into the host records at the splice points on the host, the whole coded by the oracle's bin encoder."""
import numpy as np
import pytest

import helpers as H
from entropy_coding_amd import capi

pytestmark = pytest.mark.gpu

SHAPES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 4), (4, 16), (32, 8), (2, 8), (64, 16), (1, 16), (2, 2)]


def build_case(rng, n_sub, max_blocks, ts=True, empty_subs=True):
    """Substreams of host records with residual blocks spliced in: returns what the C ABI takes plus the expanded record
    stream of every substream (host side, from the oracle) and the block records' ctx / EP counts."""
    orc = H.load_oracle()
    host_recs, splices, first, tus, coeffs, expanded = [], [], [0], [], [], []
    coeff_at = 0
    for s in range(n_sub):
        n_blocks = 0 if (empty_subs and s % 11 == 5) else int(rng.integers(0, max_blocks + 1))
        n_host = 0 if (empty_subs and s % 13 == 7) else int(rng.integers(0, 300))
        rec = H.random_records(rng, n_host, ctx_frac=0.6, end_trm=False, trm0_frac=0.01) if n_host else np.zeros(0, np.uint16)
        if s % 3:
            rec = np.concatenate([rec, np.array([0x81FF], np.uint16)])      # end_of_slice: TRM(1)
        ats = np.sort(rng.integers(0, len(rec) + 1 - (1 if s % 3 else 0), size=n_blocks)) if n_blocks else np.zeros(0, np.int64)
        if n_blocks > 3:
            ats[1] = ats[0]                                                   # two blocks back to back at one splice point
        parts, prev = [], 0
        for at in ats:
            w, h = SHAPES[int(rng.integers(0, len(SHAPES)))]
            fl = int(rng.integers(0, 4))
            if ts and w <= 32 and h <= 32 and rng.random() < 0.25:
                fl = (fl & 1) | H.TU_TRANSFORM_SKIP | (H.TU_BDPCM if rng.random() < 0.3 else 0)
                c = ((rng.random((h, w)) < 0.6) * rng.integers(-30, 31, (h, w))).astype(np.int32)
                if not c.any():
                    c[0, 0] = 2
            else:
                c = H.random_block(rng, w, h, density=float(rng.choice([0.05, 0.4, 1.0])), big=float(rng.choice([0.0, 0.2])),
                                   huge=0.02 if rng.random() < 0.1 else 0.0)
            ch = int(rng.integers(0, 2))
            t = np.zeros(1, H.TU_DTYPE)
            t["coeff_offset"], t["log2_width"], t["log2_height"], t["channel"], t["flags"] = coeff_at, int(np.log2(w)), int(np.log2(h)), ch, fl
            tus.append(t)
            coeffs.append(c.ravel())
            coeff_at += w * h
            splices.append((int(at), len(tus) - 1))
            parts.append(rec[prev:int(at)])
            parts.append(orc.residual_records(c, ch, fl)[0])
            prev = int(at)
        parts.append(rec[prev:])
        expanded.append(np.concatenate(parts) if parts else rec)
        host_recs.append(rec)
        first.append(len(splices))
    lens = [len(r) for r in host_recs]
    desc, _ = H.make_desc(lens, rng.integers(0, 64, size=n_sub), rng.integers(0, 3, size=n_sub), H.SUB_FINISH | H.SUB_ALIGN_RBSP)
    desc["byte_offset"] = 0
    desc["byte_capacity"] = 0                                                 # ignored by the call
    records = np.concatenate(host_recs) if sum(lens) else np.zeros(1, np.uint16)[:0]
    tus = np.concatenate(tus) if tus else np.zeros(0, H.TU_DTYPE)
    coeff = np.concatenate(coeffs).astype(np.int32) if coeffs else np.zeros(0, np.int32)
    return desc, records, np.array(first, np.uint32), np.array(splices, capi.SPLICE_DTYPE) if splices else np.zeros(0, capi.SPLICE_DTYPE), \
        tus, coeff, expanded


def expected(desc, expanded):
    orc = H.load_oracle()
    out, bits = [], []
    for s, rec in enumerate(expanded):
        b, nb = orc.encode_records(rec, int(desc["qp"][s]), int(desc["init_id"][s]) & 3, 3)
        out.append(b)
        bits.append(nb)
    return out, np.array(bits, np.uint32)


def counts_of(rec):
    ids = rec & 0x1FF
    c = np.bincount(ids[ids < H.NUM_CTX], minlength=H.NUM_CTX).astype(np.uint32)
    return np.concatenate([c, [np.count_nonzero(ids == H.REC_EP), np.count_nonzero(ids == H.REC_TRM)]]).astype(np.uint32)


@pytest.mark.parametrize("seed,n_sub,max_blocks", [(1, 40, 12), (2, 300, 40), (3, 5, 700), (4, 3100, 6)])
@pytest.mark.parametrize("pinned,narrow", [(False, False), (True, False), (False, True), (True, True)])
def test_spliced_residual_matches_oracle(seed, n_sub, max_blocks, pinned, narrow):
    """narrow: the coefficients as int16 (cabac_hip_encode_batch_residual16) — the same blocks, the same bytes."""
    hip = capi.CabacHip(0)
    cdt = np.int16 if narrow else np.int32
    rng = np.random.default_rng(0x5111CE + seed)
    desc, records, first, splices, tus, coeff, expanded = build_case(rng, n_sub, max_blocks)
    want, want_bits = expected(desc, expanded)
    total = sum(len(b) for b in want)
    keep = []
    if pinned:
        keep = [capi.PinnedArray((max(len(coeff), 1),), cdt), capi.PinnedArray((max(len(records), 1),), np.uint16),
                capi.PinnedArray((total + 64,), np.uint8)]
        keep[0].array[:len(coeff)] = coeff
        keep[1].array[:len(records)] = records
        coeff_in, rec_in, payload = keep[0].array[:len(coeff)], keep[1].array[:len(records)], keep[2].array
    else:
        coeff_in, rec_in, payload = coeff.astype(cdt), records, np.zeros(total + 64, np.uint8)
    assert np.array_equal(coeff_in.astype(np.int32), coeff)
    for rep in range(2):                                   # the same ctx again: staging is reused
        payload[:] = 0xEE
        offs, res, info, counts = hip.encode_batch_residual(desc, rec_in, first, splices, tus, coeff_in, payload, with_info=True,
                                                            with_counts=True)
        assert np.array_equal(res["n_bits"], want_bits) and not res["flags"].any()
        assert int(offs[-1]) == total
        for s in range(n_sub):
            assert np.array_equal(payload[int(offs[s]): int(offs[s + 1])], want[s]), s
        assert (payload[total:] == 0xEE).all()
        for s in range(0, n_sub, max(1, n_sub // 50)):
            assert np.array_equal(counts[s], counts_of(expanded[s])), s
        for t in range(0, len(tus), max(1, len(tus) // 200)):
            w, h = 1 << int(tus[t]["log2_width"]), 1 << int(tus[t]["log2_height"])
            c = coeff[int(tus[t]["coeff_offset"]): int(tus[t]["coeff_offset"]) + w * h].reshape(h, w)
            _, last, viol = H.load_oracle().residual_records(c, int(tus[t]["channel"]), int(tus[t]["flags"]))
            if not int(tus[t]["flags"]) & H.TU_TRANSFORM_SKIP:
                assert (int(info[t]) & 0xFFFF) == last and bool(int(info[t]) & H.TU_INFO_MTS_VIOLATION) == viol, t
    for k in keep:
        k.close()
    hip.close()


def test_spliced_residual_device_pointers_through_every_encoder():
    """The device-pointer form on torch tensors, through the dispatched encoders and the forced generations."""
    import torch
    rng = np.random.default_rng(0xD0D0)
    desc, records, first, splices, tus, coeff, expanded = build_case(rng, 200, 30)
    want, want_bits = expected(desc, expanded)
    total = sum(len(b) for b in want)
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt).reshape(-1).copy()).cuda()
    t_desc, t_rec, t_first = dev(desc, np.uint8), dev(records, np.int16), dev(first, np.int32)
    t_sp, t_tu, t_co = dev(splices, np.uint8), dev(tus, np.uint8), dev(coeff, np.int32)
    t_co16 = dev(coeff.astype(np.int16), np.int16)
    for variant in (0, 4, 6, 7, 16):                       # (16: the dispatched encoder on int16 coefficients)
        hip = H.gpu_ctx()
        narrow, variant = variant == 16, variant % 16
        hip.set_variant(variant, 0)
        t_pay = torch.full((total + 64,), 0xEE, dtype=torch.uint8, device="cuda")
        t_off = torch.zeros(len(desc) + 1, dtype=torch.int64, device="cuda")
        t_res = torch.zeros(2 * len(desc), dtype=torch.int32, device="cuda")
        t_cnt = torch.zeros(len(desc) * capi.BIN_COUNT_WORDS, dtype=torch.int32, device="cuda")
        hip.encode_residual_device(len(desc), t_desc.data_ptr(), t_rec.data_ptr(), t_first.data_ptr(), t_sp.data_ptr(), len(splices),
                                   len(tus), t_tu.data_ptr(), (t_co16 if narrow else t_co).data_ptr(), t_pay.data_ptr(), total + 64,
                                   t_off.data_ptr(), t_res.data_ptr(), 0, t_cnt.data_ptr(), int16=narrow)
        hip.synchronize()
        res = t_res.cpu().numpy().view(H.RESULT_DTYPE)
        offs = t_off.cpu().numpy()
        pay = t_pay.cpu().numpy()
        assert np.array_equal(res["n_bits"], want_bits) and not res["flags"].any(), variant
        for s in range(len(desc)):
            assert np.array_equal(pay[int(offs[s]): int(offs[s + 1])], want[s]), (variant, s)
        cnt = t_cnt.cpu().numpy().view(np.uint32).reshape(len(desc), -1)
        assert np.array_equal(cnt[7], counts_of(expanded[7]))
        hip.close()
    # a splice_first[] on the device that is not a partition of the splice list (the device form cannot look at it on the host):
    # refused by the plan kernel without reading anything through it
    hip = H.gpu_ctx()
    for wrong in (first[::-1].copy(), np.where(np.arange(len(first)) == len(first) // 2, 1 << 30, first).astype(np.uint32),
                  (first + 1).astype(np.uint32)):
        t_wrong = dev(wrong, np.int32)
        with pytest.raises(capi.CabacHipError) as e:
            hip.encode_residual_device(len(desc), t_desc.data_ptr(), t_rec.data_ptr(), t_wrong.data_ptr(), t_sp.data_ptr(), len(splices),
                                       len(tus), t_tu.data_ptr(), t_co.data_ptr(), t_pay.data_ptr(), total + 64, t_off.data_ptr(),
                                       t_res.data_ptr(), 0, 0)
        assert "splice list" in str(e.value)
    hip.close()


def test_spliced_residual_rejects_bad_splice_lists():
    hip = capi.CabacHip(0)
    rng = np.random.default_rng(9)
    desc, records, first, splices, tus, coeff, expanded = build_case(rng, 30, 8, empty_subs=False)
    payload = np.zeros(1 << 20, np.uint8)

    def call(sp=splices, fi=first, tu=tus, co=coeff):
        return hip.encode_batch_residual(desc, records, fi, sp, tu, co, payload)

    bad = splices.copy(); bad["tu"][3] = bad["tu"][4]                      # one block twice, one never
    with pytest.raises(capi.CabacHipError) as e:
        call(sp=bad)
    assert e.value.status == -2
    bad = splices.copy(); bad["at"][0] = 1 << 20                            # outside its substream
    with pytest.raises(capi.CabacHipError):
        call(sp=bad)
    s = next(k for k in range(len(desc)) if first[k + 1] - first[k] >= 2 and splices["at"][first[k]] != splices["at"][first[k + 1] - 1])
    bad = splices.copy(); j = int(first[s]); bad["at"][j], bad["at"][first[s + 1] - 1] = bad["at"][first[s + 1] - 1], bad["at"][j]
    with pytest.raises(capi.CabacHipError):                                 # not sorted
        call(sp=bad)
    with pytest.raises(capi.CabacHipError):                                 # fewer splices than blocks
        call(fi=np.minimum(first, first[-1] - 1).astype(np.uint32))
    bad_tu = tus.copy(); bad_tu["coeff_offset"][2] = len(coeff)             # coefficients out of range
    with pytest.raises(capi.CabacHipError) as e:
        call(tu=bad_tu)
    assert "coefficients out of range" in str(e.value)
    # an all-zero block: the reference throws for it; here its splice adds nothing and the call says so
    zero = coeff.copy()
    t0 = int(splices["tu"][0]); w, h = 1 << int(tus[t0]["log2_width"]), 1 << int(tus[t0]["log2_height"])
    zero[int(tus[t0]["coeff_offset"]): int(tus[t0]["coeff_offset"]) + w * h] = 0
    offs, res, info = hip.encode_batch_residual(desc, records, first, splices, tus, zero, payload, check=False, with_info=True)
    assert int(info[t0]) & H.TU_INFO_EMPTY
    with pytest.raises(capi.CabacHipError) as e:
        call(co=zero)
    assert e.value.status == -5
    # and the ctx is as good as new afterwards
    want, want_bits = expected(desc, expanded)
    offs, res = call()
    assert np.array_equal(res["n_bits"], want_bits)
    for k in range(len(desc)):
        assert np.array_equal(payload[int(offs[k]): int(offs[k + 1])], want[k])
    hip.close()
