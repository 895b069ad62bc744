"""GPU: the residual parser (cabac_hip_residual_parse_device, csrc/cabac_residual_parse.hip) — bytes -> coefficient blocks with
the contexts derived on the device — through the C ABI against the oracle's parser (orc_residual_decode, pinned to the
reference's CABACReader::residual_coding by tests/test_residual_oracle.py) and against the coefficients that were coded."""
import numpy as np
import pytest

import helpers as H
from entropy_coding_amd import capi

pytestmark = pytest.mark.gpu

SHAPES = [(w, h) for w in (1, 2, 4, 8, 16, 32, 64) for h in (1, 2, 4, 8, 16, 32, 64)]


@pytest.fixture(scope="module", params=["int32", "int16"])
def hip(request):
    """Every test of this file with the blocks stored as int32 (cabac_hip_residual_parse_device) and as int16
    (cabac_hip_residual_parse16_device: the same levels, half the bytes)."""
    c = H.gpu_ctx()
    c.parse_int16 = request.param == "int16"
    yield c
    c.close()


def build(rng, n_sub, flags_of, qps, max_blocks=12):
    """n_sub substreams of random blocks -> (metas per substream, blocks per substream, bytes per substream)."""
    orc = H.load_oracle()
    subs = []
    for s in range(n_sub):
        blocks, metas = [], []
        for k in range(int(rng.integers(1, max_blocks + 1))):
            w, h = SHAPES[int(rng.integers(0, len(SHAPES)))]
            blocks.append(H.random_block(rng, w, h, density=float(rng.choice([0.05, 0.3, 0.7, 1.0])), big=float(rng.choice([0.0, 0.05, 0.3])),
                                         huge=0.02 if rng.random() < 0.1 else 0.0, last_frac=float(rng.choice([1.0, 0.5, 0.2]))))
            metas.append((w, h, int(rng.integers(0, 2)), flags_of(s)))
        rec = np.concatenate([orc.residual_records(c, metas[i][2], metas[i][3])[0] for i, c in enumerate(blocks)] + [np.array([0x81FF], np.uint16)])
        data, _ = orc.encode_records(rec, int(qps[s]), 2, 3)
        subs.append((metas, blocks, data))
    return subs


def parse(hip, subs, qps, capacities=None, finish=True, mutate=None):
    import torch
    n_sub = len(subs)
    metas = [m for s in subs for m in s[0]]
    tus = np.zeros(len(metas), H.TU_DTYPE)
    off = 0
    for i, (w, h, ch, fl) in enumerate(metas):
        tus[i]["coeff_offset"], tus[i]["log2_width"], tus[i]["log2_height"], tus[i]["channel"], tus[i]["flags"] = off, int(np.log2(w)), int(np.log2(h)), ch, fl
        off += w * h
    tile_first = np.concatenate([[0], np.cumsum([len(s[0]) for s in subs])]).astype(np.uint32)
    desc = np.zeros(n_sub, H.DESC_DTYPE)
    caps = np.array([len(s[2]) for s in subs], np.uint64) if capacities is None else np.asarray(capacities, np.uint64)
    slots = (np.array([len(s[2]) for s in subs], np.uint64) + 15) // 16 * 16 + 16
    desc["byte_offset"] = np.concatenate([[0], np.cumsum(slots)[:-1]])
    desc["byte_capacity"] = caps
    desc["qp"] = qps
    desc["init_id"] = 2 | (H.SUB_FINISH if finish else 0)
    buf = np.zeros(int(slots.sum()), np.uint8)
    for s in range(n_sub):
        buf[int(desc["byte_offset"][s]): int(desc["byte_offset"][s]) + len(subs[s][2])] = subs[s][2]
    if mutate:
        mutate(buf, desc)
    t_desc = torch.from_numpy(desc.view(np.uint8).copy()).cuda()
    t_buf = torch.from_numpy(buf).cuda()
    t_first = torch.from_numpy(tile_first.view(np.int32).copy()).cuda()
    t_tu = torch.from_numpy(tus.view(np.uint8).reshape(-1).copy()).cuda()
    narrow = getattr(hip, "parse_int16", False)
    t_co = torch.full((max(off, 1),), 0x5A5A, dtype=torch.int16, device="cuda") if narrow else torch.full((max(off, 1),), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    t_res = torch.full((2 * n_sub,), -1, dtype=torch.int32, device="cuda")
    t_info = torch.full((max(len(metas), 1),), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    hip.residual_parse_device(n_sub, t_desc.data_ptr(), t_buf.data_ptr(), t_first.data_ptr(), t_tu.data_ptr(), t_co.data_ptr(), t_res.data_ptr(),
                              d_tu_info=t_info.data_ptr(), int16=narrow)
    hip.synchronize()
    parse.last_info = t_info.cpu().numpy().view(np.uint32)
    co = t_co.cpu().numpy()
    co = co.astype(np.int32)
    res = t_res.cpu().numpy().view(H.RESULT_DTYPE)
    out, o = [], 0
    for s in subs:
        blocks = []
        for (w, h, _, _) in s[0]:
            blocks.append(co[o:o + w * h].reshape(h, w))
            o += w * h
        out.append(blocks)
    return out, res


@pytest.mark.parametrize("flags", [0, H.TU_DEP_QUANT])
def test_parse_gives_the_coded_blocks_back(hip, flags):
    """Without sign hiding decode(encode(block)) == block: every shape, several blocks per substream, ragged substreams."""
    orc = H.load_oracle()
    rng = np.random.default_rng(0x9A + flags)
    n_sub = 150
    qps = rng.integers(0, 64, n_sub)
    subs = build(rng, n_sub, lambda s: flags, qps)
    got, res = parse(hip, subs, qps)
    assert not res["flags"].any()
    for s, (metas, blocks, data) in enumerate(subs):
        rc, want, nbits = orc.residual_decode(data, int(qps[s]), metas)
        assert rc == 0 and int(res["n_bits"][s]) == nbits, s
        for k, c in enumerate(blocks):
            w, h = metas[k][0], metas[k][1]
            we, he = min(w, 32), min(h, 32)
            assert np.array_equal(got[s][k][:he, :we], c[:he, :we]), (s, k, c.shape)       # the coded region comes back
            assert np.array_equal(got[s][k][:he, :we], want[k][:he, :we]), (s, k)


def test_parse_with_sign_hiding_matches_the_oracle(hip):
    orc = H.load_oracle()
    rng = np.random.default_rng(0x51)
    n_sub = 80
    qps = rng.integers(20, 40, n_sub)
    subs = build(rng, n_sub, lambda s: H.TU_SIGN_HIDING | (H.TU_DEP_QUANT if s & 1 else 0), qps)
    got, res = parse(hip, subs, qps)
    assert not res["flags"].any()
    for s, (metas, blocks, data) in enumerate(subs):
        rc, want, nbits = orc.residual_decode(data, int(qps[s]), metas)
        assert rc == 0 and int(res["n_bits"][s]) == nbits
        for k in range(len(blocks)):
            we, he = min(metas[k][0], 32), min(metas[k][1], 32)
            assert np.array_equal(got[s][k][:he, :we], want[k][:he, :we]), (s, k)


def _ts_block(rng, w, h, kind):
    c = ((rng.random((h, w)) < [0.3, 1.0, 0.8, 0.05][kind]) *
         rng.integers(-[4, 40, 3, 3000][kind], [4, 40, 3, 3000][kind] + 1, (h, w))).astype(np.int32)
    if not c.any():
        c[rng.integers(0, h), rng.integers(0, w)] = 1
    return c


def build_mixed(rng, n_sub, qps, max_blocks=14):
    """Substreams that mix regular blocks (transform_skip_flag absent or coded 0) with transform-skip blocks (flag coded 1,
    not coded, BDPCM); dependent quantisation per block; no sign hiding, so every block must come back exactly."""
    orc = H.load_oracle()
    sizes = [1, 2, 4, 8, 16, 32]
    subs = []
    for s in range(n_sub):
        blocks, metas = [], []
        for k in range(int(rng.integers(1, max_blocks + 1))):
            kind = int(rng.integers(0, 5))
            dq = int(rng.integers(0, 2))
            if kind < 2:
                w, h = SHAPES[int(rng.integers(0, len(SHAPES)))]
                if kind == 1:
                    w, h = min(w, 32), min(h, 32)
                c = H.random_block(rng, w, h, density=float(rng.choice([0.1, 0.5, 1.0])), big=float(rng.choice([0.0, 0.2])))
                fl = dq | (H.TU_TS_FLAG if kind == 1 else 0)
            else:
                w, h = sizes[int(rng.integers(0, 6))], sizes[int(rng.integers(0, 6))]
                if w * h == 1:
                    h = 4
                c = _ts_block(rng, w, h, int(rng.integers(0, 4)))
                fl = dq | H.TU_TRANSFORM_SKIP | [H.TU_TS_FLAG, 0, H.TU_BDPCM][kind - 2]
            blocks.append(c)
            metas.append((w, h, int(rng.integers(0, 2)), fl))
        rec = np.concatenate([orc.residual_records(c, metas[i][2], metas[i][3])[0] for i, c in enumerate(blocks)] + [np.array([0x81FF], np.uint16)])
        data, _ = orc.encode_records(rec, int(qps[s]), 2, 3)
        subs.append((metas, blocks, data))
    return subs


def test_transform_skip_and_regular_blocks_mixed(hip):
    """residual_codingTS / BDPCM / ts_flag on the device: coded blocks come back exactly, bit counts and the per-block info
    (scanPosLast | MTS violation, or TS) equal the oracle's (pinned to the reference reader in tests/test_residual_oracle.py)."""
    orc = H.load_oracle()
    rng = np.random.default_rng(0x7511)
    n_sub = 160
    qps = rng.integers(0, 64, n_sub)
    subs = build_mixed(rng, n_sub, qps)
    got, res = parse(hip, subs, qps)
    info = parse.last_info
    assert not res["flags"].any()
    t = 0
    for s, (metas, blocks, data) in enumerate(subs):
        rc, want, nbits, winfo = orc.residual_decode(data, int(qps[s]), metas, with_info=True)
        assert rc == 0 and int(res["n_bits"][s]) == nbits, s
        for k, c in enumerate(blocks):
            we, he = min(metas[k][0], 32), min(metas[k][1], 32)
            assert np.array_equal(got[s][k][:he, :we], c[:he, :we]), (s, k, metas[k])
            assert int(info[t]) == int(winfo[k]), (s, k, metas[k], hex(int(info[t])), hex(int(winfo[k])))
            t += 1


def test_transform_skip_budget_exhaustion_and_escapes(hip):
    """Saturated transform-skip blocks: the 7/4 bins-per-sample budget runs out (bypass levels with bypass signs), 32-bin
    escape codes; with the flag in the stream, without, and with BDPCM."""
    subs, qps = [], []
    orc = H.load_oracle()
    for w, h in [(4, 4), (32, 32), (8, 16), (2, 2), (1, 16), (16, 1)]:
        for v in (1, -7, 2000, -32768):
            c = np.full((h, w), v, np.int32)
            c[::2, 1::2] = -v if v != -32768 else 32767
            for fl in (H.TU_TRANSFORM_SKIP | H.TU_TS_FLAG, H.TU_TRANSFORM_SKIP | H.TU_BDPCM, H.TU_TRANSFORM_SKIP):
                metas = [(w, h, 0, fl), (w, h, 1, fl)]
                rec = np.concatenate([orc.residual_records(c, m[2], m[3])[0] for m in metas] + [np.array([0x81FF], np.uint16)])
                data, _ = orc.encode_records(rec, 30, 2, 3)
                subs.append((metas, [c, c], data))
                qps.append(30)
    got, res = parse(hip, subs, np.array(qps))
    assert not res["flags"].any()
    for s, (metas, blocks, data) in enumerate(subs):
        rc, want, nbits = orc.residual_decode(data, 30, metas)
        assert rc == 0 and int(res["n_bits"][s]) == nbits
        for k in range(2):
            assert np.array_equal(got[s][k], blocks[k]), (s, k, metas[k])


def test_ts_flag_in_the_stream_decides(hip):
    """A block whose descriptor only says "transform_skip_flag is coded" is parsed as the decoded bin says — the real
    decoder's situation (cabac_reader.cpp:2737-2752) — whatever its TRANSFORM_SKIP bit claims."""
    orc = H.load_oracle()
    rng = np.random.default_rng(99)
    c_ts, c_reg = _ts_block(rng, 8, 8, 1), H.random_block(rng, 8, 8, density=0.6)
    coded = [(8, 8, 0, H.TU_TS_FLAG | H.TU_TRANSFORM_SKIP), (8, 8, 0, H.TU_TS_FLAG)]
    rec = np.concatenate([orc.residual_records(c_ts, 0, coded[0][3])[0], orc.residual_records(c_reg, 0, coded[1][3])[0], np.array([0x81FF], np.uint16)])
    data, _ = orc.encode_records(rec, 28, 2, 3)
    lying = [(8, 8, 0, H.TU_TS_FLAG), (8, 8, 0, H.TU_TS_FLAG | H.TU_TRANSFORM_SKIP)]   # descriptor bits swapped
    got, res = parse(hip, [(lying, [c_ts, c_reg], data)], np.array([28]))
    assert not res["flags"].any()
    assert np.array_equal(got[0][0], c_ts) and np.array_equal(got[0][1], c_reg)
    assert int(parse.last_info[0]) == H.TU_INFO_TS and not (int(parse.last_info[1]) & H.TU_INFO_TS)


def test_parse_error_flags(hip):
    rng = np.random.default_rng(3)
    qps = np.full(6, 30)
    subs = build(rng, 6, lambda s: 0, qps, max_blocks=4)
    # 0: intact; 1: truncated input; 2: stop pattern destroyed; 3: a transform-skip block beyond 32 x 32 (not covered); 4, 5: intact
    caps = [len(s[2]) for s in subs]
    caps[1] = max(1, caps[1] // 3)
    subs[3] = ([(64, 16, m[2], H.TU_TRANSFORM_SKIP) if i == 0 else m for i, m in enumerate(subs[3][0])],
               [np.zeros((16, 64), np.int32) if i == 0 else c for i, c in enumerate(subs[3][1])], subs[3][2])

    def mutate(buf, desc):
        o = int(desc["byte_offset"][2]) + len(subs[2][2]) - 1
        buf[o] = 0x00 if buf[o] != 0 else 0x55

    got, res = parse(hip, subs, qps, capacities=caps, mutate=mutate)
    assert int(res["flags"][0]) == 0 and int(res["flags"][4]) == 0 and int(res["flags"][5]) == 0
    assert int(res["flags"][1]) & H.RES_UNDERRUN
    assert int(res["flags"][2]) & H.RES_BAD_STOP
    assert int(res["flags"][3]) & H.RES_BAD_RECORD
    for s in (0, 4, 5):
        for k, c in enumerate(subs[s][1]):
            we, he = min(c.shape[1], 32), min(c.shape[0], 32)
            assert np.array_equal(got[s][k][:he, :we], c[:he, :we])


def test_golden_substreams_from_the_reference_reader(hip):
    """bytes coded by the reference's writer -> the coefficients its reader decodes (tests/golden/residual_parse.npz)."""
    import os
    g = np.load(os.path.join(H.GOLDEN, "residual_parse.npz"))
    n_sub = int(g["n_sub"][0])
    subs, qps = [], []
    for s in range(n_sub):
        metas = [tuple(int(x) for x in m) for m in g["s%d_meta" % s]]
        co = g["s%d_coeff" % s]
        blocks, o = [], 0
        for (w, h, _, _) in metas:
            blocks.append(co[o:o + w * h].reshape(h, w))
            o += w * h
        subs.append((metas, blocks, g["s%d_bytes" % s]))
        qps.append(int(g["s%d_qp" % s][0]))
    got, res = parse(hip, subs, np.array(qps))
    info = parse.last_info
    assert not res["flags"].any()
    t = 0
    for s in range(n_sub):    # mtsIdx / CUCtx as the reference reader left them (ref_residual_decode's info)
        for k, m in enumerate(subs[s][0]):
            ts, lfnst_viol, lfnst_last, mts_viol, mts_last = (int(x) for x in g["s%d_refinfo" % s][k])
            w, h, chroma = m[0], m[1], m[2]
            word = int(info[t])
            t += 1
            assert bool(word & H.TU_INFO_TS) == bool(ts), (s, k)
            if not ts:
                last, big = word & 0xFFFF, w >= 4 and h >= 4
                assert lfnst_viol == (int(big and last > (7 if (w, h) in ((4, 4), (8, 8)) else 15)) << (1 if chroma else 0))
                assert lfnst_last == int(big and last >= 1) and mts_last == int((not chroma) and last >= 1)
                assert mts_viol == int(bool(word & H.TU_INFO_MTS_VIOLATION))
    for s in range(n_sub):
        assert int(res["n_bits"][s]) == int(g["s%d_qp" % s][1])
        for k, c in enumerate(subs[s][1]):
            we, he = min(c.shape[1], 32), min(c.shape[0], 32)
            assert np.array_equal(got[s][k][:he, :we], c[:he, :we]), (s, k)


def test_host_pointer_parse_batch(hip):
    """cabac_hip_residual_parse_batch (host arrays, staging inside) gives the coded blocks back; a truncated substream is
    reported through the status and the flags."""
    rng = np.random.default_rng(21)
    qps = rng.integers(10, 50, 12)
    subs = build(rng, 12, lambda s: H.TU_DEP_QUANT if s & 1 else 0, qps, max_blocks=6)
    metas = [m for s in subs for m in s[0]]
    tus = np.zeros(len(metas), H.TU_DTYPE)
    off = 0
    for i, (w, h, ch, fl) in enumerate(metas):
        tus[i]["coeff_offset"], tus[i]["log2_width"], tus[i]["log2_height"], tus[i]["channel"], tus[i]["flags"] = off, int(np.log2(w)), int(np.log2(h)), ch, fl
        off += w * h
    first = np.concatenate([[0], np.cumsum([len(s[0]) for s in subs])]).astype(np.uint32)
    desc = np.zeros(len(subs), H.DESC_DTYPE)
    slots = (np.array([len(s[2]) for s in subs], np.uint64) + 15) // 16 * 16
    desc["byte_offset"] = np.concatenate([[0], np.cumsum(slots)[:-1]])
    desc["byte_capacity"] = [len(s[2]) for s in subs]
    desc["qp"] = qps
    desc["init_id"] = 2 | H.SUB_FINISH
    buf = np.zeros(int(slots.sum()), np.uint8)
    for s in range(len(subs)):
        buf[int(desc["byte_offset"][s]): int(desc["byte_offset"][s]) + len(subs[s][2])] = subs[s][2]
    narrow = getattr(hip, "parse_int16", False)
    co, res = hip.residual_parse_batch(desc, buf, first, tus, off, int16=narrow)
    assert not res["flags"].any() and co.dtype == (np.int16 if narrow else np.int32)
    o = 0
    for s in subs:
        for c in s[1]:
            h, w = c.shape
            got = co[o:o + w * h].reshape(h, w)
            assert np.array_equal(got[:min(h, 32), :min(w, 32)], c[:min(h, 32), :min(w, 32)])
            o += w * h
    bad = desc.copy()
    bad["byte_capacity"][3] = 2
    with pytest.raises(capi.CabacHipError):
        hip.residual_parse_batch(bad, buf, first, tus, off, int16=narrow)
    co2, res2 = hip.residual_parse_batch(bad, buf, first, tus, off, check=False, int16=narrow)
    assert int(res2["flags"][3]) & H.RES_UNDERRUN and not res2["flags"][[0, 1, 2, 4]].any()


def test_sbt_zero_out_blocks_parse_back(hip):
    """CABAC_TU_SBT_ZERO_OUT on the parser's side (cabac_reader.cpp:2880-2891, :2718-2727): substreams of zero-out blocks mixed
    with blocks the flag does not touch come back as the oracle's parser — pinned to the reference's reader — gives them."""
    orc = H.load_oracle()
    rng = np.random.default_rng(0x5B8)
    shapes = [(32, 32), (32, 8), (8, 32), (32, 16), (16, 32), (32, 4), (4, 32), (16, 16), (8, 8), (4, 4), (64, 32)]
    subs, qps = [], rng.integers(10, 50, 60)
    for s in range(60):
        blocks, metas = [], []
        for k in range(int(rng.integers(1, 10))):
            w, h = shapes[int(rng.integers(0, len(shapes)))]
            ch = int(rng.random() < 0.15)
            c = H.random_block(rng, w, h, density=float(rng.choice([0.05, 0.4, 1.0])), big=float(rng.choice([0.0, 0.3])))
            if not ch and max(w, h) <= 32:
                if w == 32:
                    c[:, 16:] = 0
                if h == 32:
                    c[16:, :] = 0
                if not c.any():
                    c[0, 0] = 1
            blocks.append(c)
            metas.append((w, h, ch, (H.TU_DEP_QUANT if s & 1 else 0) | H.TU_SBT_ZERO_OUT))
        rec = np.concatenate([orc.residual_records(c, metas[i][2], metas[i][3])[0] for i, c in enumerate(blocks)] + [np.array([0x81FF], np.uint16)])
        data, _ = orc.encode_records(rec, int(qps[s]), 2, 3)
        subs.append((metas, blocks, data))
    got, res = parse(hip, subs, qps)
    assert not res["flags"].any()
    for s, (metas, blocks, data) in enumerate(subs):
        rc, want, nbits = orc.residual_decode(data, int(qps[s]), metas)
        assert rc == 0 and int(res["n_bits"][s]) == nbits, s
        for k, c in enumerate(blocks):
            we, he = min(metas[k][0], 32), min(metas[k][1], 32)
            assert np.array_equal(got[s][k][:he, :we], c[:he, :we]) and np.array_equal(got[s][k][:he, :we], want[k][:he, :we]), (s, k)


def test_int16_output_reports_a_level_that_does_not_fit():
    """cabac_hip_residual_parse_batch16 on a stream of extended dynamic range (max_log2_tr_range 20) whose block holds a level
    beyond int16: CABAC_RES_RANGE in that substream's result (the level is stored truncated), the other substream is not
    touched by it; the 32-bit form gives the block back exactly.  (HipBatch::residualParse falls back on that flag.)"""
    orc = H.load_oracle()
    c_big = np.zeros((8, 8), np.int32)
    c_big[0, 0], c_big[1, 2], c_big[3, 3] = 100000, -7, 2
    c_ok = np.zeros((8, 8), np.int32)
    c_ok[0, 0], c_ok[2, 1] = -31000, 5
    datas = []
    for c in (c_big, c_ok):
        rec = np.concatenate([orc.residual_records(c, 0, 0, max_log2_range=20)[0], np.array([0x81FF], np.uint16)])
        datas.append(orc.encode_records(rec, 30, 2, 3)[0])
    tus = np.zeros(2, H.TU_DTYPE)
    tus["log2_width"], tus["log2_height"], tus["max_log2_tr_range"] = 3, 3, 20
    tus["coeff_offset"] = [0, 64]
    desc = np.zeros(2, H.DESC_DTYPE)
    desc["byte_offset"] = [0, (len(datas[0]) + 15) // 16 * 16 + 16]
    desc["byte_capacity"] = [len(d) for d in datas]
    desc["qp"], desc["init_id"] = 30, 2 | H.SUB_FINISH
    buf = np.zeros(int(desc["byte_offset"][1]) + len(datas[1]) + 32, np.uint8)
    for s in range(2):
        buf[int(desc["byte_offset"][s]): int(desc["byte_offset"][s]) + len(datas[s])] = datas[s]
    first = np.array([0, 1, 2], np.uint32)
    hip = capi.CabacHip(0)
    co, res = hip.residual_parse_batch(desc, buf, first, tus, 128)
    assert not res["flags"].any() and np.array_equal(co[:64].reshape(8, 8), c_big) and np.array_equal(co[64:].reshape(8, 8), c_ok)
    with pytest.raises(capi.CabacHipError):
        hip.residual_parse_batch(desc, buf, first, tus, 128, int16=True)
    co16, res16 = hip.residual_parse_batch(desc, buf, first, tus, 128, int16=True, check=False)
    assert int(res16["flags"][0]) == capi.RES_RANGE and int(res16["flags"][1]) == 0
    assert np.array_equal(co16[64:].reshape(8, 8), c_ok) and np.array_equal(res16["n_bits"], res["n_bits"])
    assert np.array_equal(co16[:64].reshape(8, 8), c_big.astype(np.int16))
    hip.close()
