"""GPU: the residual binariser (cabac_hip_residual_device, csrc/cabac_residual.hip) through the C ABI against the
oracle (orc_residual_records, pinned to the reference's CABACWriter::residual_coding by tests/test_residual_oracle.py)
and against the golden blocks the compiled reference produced (tests/golden/residual.npz); then coefficient blocks ->
records -> device encoder == the oracle's bytes for the same blocks."""
import os

import numpy as np
import pytest

import helpers as H
from entropy_coding_amd import capi

pytestmark = pytest.mark.gpu

SIZES = [(w, h) for w in (1, 2, 4, 8, 16, 32, 64) for h in (1, 2, 4, 8, 16, 32, 64)]


@pytest.fixture(scope="module")
def hip():
    c = H.gpu_ctx()
    yield c
    c.close()


def make_tus(blocks, chromas, flags, max_log2=None):
    tus = np.zeros(len(blocks), H.TU_DTYPE)
    off = 0
    for i, c in enumerate(blocks):
        h, w = c.shape
        tus[i]["coeff_offset"] = off
        tus[i]["log2_width"] = int(np.log2(w))
        tus[i]["log2_height"] = int(np.log2(h))
        tus[i]["channel"] = chromas[i]
        tus[i]["flags"] = flags[i]
        tus[i]["max_log2_tr_range"] = 0 if max_log2 is None else max_log2[i]
        off += w * h
    coeff = np.concatenate([c.ravel() for c in blocks]).astype(np.int32) if blocks else np.zeros(1, np.int32)
    return tus, coeff


def residual(hip, tus, coeff, slack=0):
    """Two passes as the header describes: sizes, then records at the prefix-summed offsets."""
    import torch
    n = len(tus)
    t_tu = torch.from_numpy(tus.view(np.uint8).reshape(-1).copy()).cuda()
    t_co = torch.from_numpy(coeff).cuda()
    t_cnt = torch.full((max(n, 1),), -1, dtype=torch.int32, device="cuda")
    t_info = torch.full((max(n, 1),), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()        # torch's stream filled the buffers; the library runs on its own stream
    hip.residual_device(n, t_tu.data_ptr(), t_co.data_ptr(), 0, t_cnt.data_ptr(), t_info.data_ptr(), 0)
    hip.synchronize()
    cnt = t_cnt.cpu().numpy().view(np.uint32)[:n].astype(np.uint64)
    info1 = t_info.cpu().numpy().view(np.uint32)[:n].copy()
    roff = (np.concatenate([[0], np.cumsum(cnt + slack)[:-1]]) if n else np.zeros(0)).astype(np.uint64)
    total = int((cnt + slack).sum())
    t_roff = torch.from_numpy(roff.view(np.int64).copy()).cuda() if n else torch.zeros(1, dtype=torch.int64, device="cuda")
    t_rec = torch.full((max(total, 1),), 0x5555, dtype=torch.int16, device="cuda")
    t_cnt2 = torch.full((max(n, 1),), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    hip.residual_device(n, t_tu.data_ptr(), t_co.data_ptr(), t_roff.data_ptr(), t_cnt2.data_ptr(), t_info.data_ptr(),
                        t_rec.data_ptr())
    hip.synchronize()
    assert np.array_equal(t_cnt2.cpu().numpy().view(np.uint32)[:n], cnt.astype(np.uint32))
    assert np.array_equal(t_info.cpu().numpy().view(np.uint32)[:n], info1)
    rec = t_rec.cpu().numpy().view(np.uint16)
    if slack:   # nothing is written outside a block's own range
        for i in range(n):
            assert (rec[int(roff[i] + cnt[i]): int(roff[i] + cnt[i]) + slack] == 0x5555).all(), i
    return [rec[int(roff[i]): int(roff[i] + cnt[i])] for i in range(n)], info1


def check_against_oracle(hip, blocks, chromas, flags, max_log2=None, slack=0):
    orc = H.load_oracle()
    tus, coeff = make_tus(blocks, chromas, flags, max_log2)
    recs, info = residual(hip, tus, coeff, slack)
    for i, c in enumerate(blocks):
        want, last, mts = orc.residual_records(c, chromas[i], flags[i], 15 if max_log2 is None else (max_log2[i] or 15))
        assert len(recs[i]) == len(want) and np.array_equal(recs[i], want), (i, c.shape, chromas[i], flags[i])
        assert int(info[i]) == (last | (H.TU_INFO_MTS_VIOLATION if mts else 0)), i
    return recs


def test_golden_blocks_from_the_reference(hip):
    g = np.load(os.path.join(H.GOLDEN, "residual.npz"))
    n = int(g["n_blocks"][0])
    blocks = [g["coeff"][g["coeff_off"][k]: g["coeff_off"][k + 1]].reshape(1 << int(g["meta"][k][1]), 1 << int(g["meta"][k][0]))
              for k in range(n)]
    tus, coeff = make_tus(blocks, [int(m[2]) for m in g["meta"]], [int(m[3]) for m in g["meta"]])
    recs, _ = residual(hip, tus, coeff, slack=3)
    for k in range(n):
        assert np.array_equal(recs[k], g["records"][g["rec_off"][k]: g["rec_off"][k + 1]]), k


@pytest.mark.parametrize("seed", range(3))
def test_random_blocks_all_sizes(hip, seed):
    rng = np.random.default_rng(0x2F00 + seed)
    blocks, chromas, flags = [], [], []
    for w, h in SIZES:
        for k in range(10):
            blocks.append(H.random_block(rng, w, h, density=[0.05, 0.3, 0.7, 1.0][k % 4], big=[0.0, 0.05, 0.3][k % 3],
                                         huge=0.02 if k % 5 == 4 else 0.0, last_frac=[1.0, 0.5, 0.2][k % 3]))
            chromas.append(int(rng.integers(0, 2)))
            flags.append(int(rng.integers(0, 8)))
    order = rng.permutation(len(blocks))            # mix sizes inside a wave: rows of one wave run different group counts
    check_against_oracle(hip, [blocks[i] for i in order], [chromas[i] for i in order], [flags[i] for i in order], slack=seed)


def test_edge_blocks(hip):
    blocks, chromas, flags = [], [], []
    for w, h in [(4, 4), (8, 8), (32, 32), (64, 64), (2, 8), (8, 2), (16, 1), (1, 16), (4, 32), (64, 4), (1, 1), (2, 2)]:
        we, he = min(w, 32), min(h, 32)
        z = np.zeros((h, w), np.int32)
        cases = []
        for (y, x) in [(0, 0), (he - 1, we - 1), (0, we - 1), (he - 1, 0)]:
            for v in (1, -1, 2, -3, 4, 5, 32767, -32768):
                c = z.copy(); c[y, x] = v; cases.append(c)
        c = z.copy(); c[:he, :we] = 1; cases.append(c)
        c = z.copy(); c[:he, :we] = -32768; cases.append(c)          # escapes everywhere; the context-bin budget runs out
        c = z.copy(); c[:he, :we] = 3; c[0, 0] = -7; cases.append(c)
        c = z.copy(); c[:he, :we] = np.where((np.add.outer(np.arange(he), np.arange(we)) & 1) == 0, 2, -1); cases.append(c)
        for c in cases:
            for fl in (0, 1, 2, 3, 7):
                for ch in (0, 1):
                    blocks.append(c); chromas.append(ch); flags.append(fl)
    check_against_oracle(hip, blocks, chromas, flags)


def test_extended_dynamic_range_and_bad_descriptors(hip):
    rng = np.random.default_rng(77)
    blocks = [H.random_block(rng, 16, 16, density=0.8, big=0.3, huge=0.2) * 17 for _ in range(8)]
    check_against_oracle(hip, blocks, [0, 1] * 4, [3] * 8, max_log2=[20, 18, 17, 15, 0, 17, 19, 20])
    # an all-zero block (the reference throws, cabac_writer.cpp:2458) and bad descriptors produce no records and a flag
    tus, coeff = make_tus([np.zeros((8, 8), np.int32), np.ones((4, 4), np.int32), np.ones((4, 4), np.int32), np.ones((4, 4), np.int32)],
                          [0, 0, 2, 0], [0, 0, 0, 0])
    tus[1]["log2_width"] = 7
    recs, info = residual(hip, tus, coeff, slack=2)
    assert [len(r) for r in recs[:3]] == [0, 0, 0] and len(recs[3]) > 0
    assert int(info[0]) == H.TU_INFO_EMPTY and int(info[1]) == H.TU_INFO_BAD_DESC and int(info[2]) == H.TU_INFO_BAD_DESC
    # n_tu == 0 is a no-op
    hip.residual_device(0, 0, 0, 0, 0, 0, 0)


def test_blocks_to_bytes_through_the_device_encoder(hip):
    """coefficients -> residual binariser -> bin encoder, all on the device, equals the oracle's bytes for the blocks'
    records (one substream per 'frame' of blocks, terminated by TRM(1))."""
    import torch
    orc = H.load_oracle()
    rng = np.random.default_rng(0xB10C)
    n_sub, per = 24, 40
    blocks, chromas, flags = [], [], []
    for s in range(n_sub * per):
        w, h = [(4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (16, 8), (4, 16), (2, 8)][int(rng.integers(0, 8))]
        blocks.append(H.random_block(rng, w, h, density=0.4, big=0.05))
        chromas.append(int(rng.integers(0, 2)))
        flags.append(3)
    tus, coeff = make_tus(blocks, chromas, flags)
    n = len(tus)
    t_tu = torch.from_numpy(tus.view(np.uint8).reshape(-1).copy()).cuda()
    t_co = torch.from_numpy(coeff).cuda()
    t_cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    hip.residual_device(n, t_tu.data_ptr(), t_co.data_ptr(), 0, t_cnt.data_ptr(), 0, 0)
    hip.synchronize()
    cnt = t_cnt.cpu().numpy().astype(np.int64)
    # substream s = blocks [s*per, (s+1)*per) followed by one TRM(1) record
    sub_len = cnt.reshape(n_sub, per).sum(1) + 1
    sub_off = np.concatenate([[0], np.cumsum(sub_len)[:-1]])
    roff = (np.repeat(sub_off, per) + (np.cumsum(cnt.reshape(n_sub, per), 1) - cnt.reshape(n_sub, per)).ravel()).astype(np.uint64)
    t_rec = torch.zeros(int(sub_len.sum()), dtype=torch.int16, device="cuda")
    trm = torch.from_numpy((sub_off + sub_len - 1).astype(np.int64)).cuda()
    t_rec[trm] = torch.tensor(np.array([0x81FF], np.uint16).view(np.int16)[0], dtype=torch.int16, device="cuda")
    t_roff = torch.from_numpy(roff.view(np.int64).copy()).cuda()
    torch.cuda.synchronize()        # torch's stream filled t_rec; the library runs on its own stream
    hip.residual_device(n, t_tu.data_ptr(), t_co.data_ptr(), t_roff.data_ptr(), t_cnt.data_ptr(), 0, t_rec.data_ptr())
    desc, total = H.make_desc([int(x) for x in sub_len], [32] * n_sub, [2] * n_sub, H.SUB_FINISH | H.SUB_ALIGN_RBSP)
    t_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).cuda()
    t_bytes = torch.zeros(total, dtype=torch.uint8, device="cuda")
    t_res = torch.zeros(2 * n_sub, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    hip.encode_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), t_bytes.data_ptr(), t_res.data_ptr())
    hip.synchronize()
    out = t_bytes.cpu().numpy()
    res = t_res.cpu().numpy().view(H.RESULT_DTYPE)
    for s in range(n_sub):
        want_rec = np.concatenate([orc.residual_records(blocks[i], chromas[i], flags[i])[0] for i in range(s * per, (s + 1) * per)]
                                  + [np.array([0x81FF], np.uint16)])
        want, nbits = orc.encode_records(want_rec, 32, 2, 3)
        o = int(desc["byte_offset"][s])
        assert int(res["n_bits"][s]) == nbits and np.array_equal(out[o:o + len(want)], want), s


def test_host_pointer_batch_api(hip):
    """cabac_hip_residual_batch (host arrays, both passes inside) == the device API == the oracle."""
    orc = H.load_oracle()
    rng = np.random.default_rng(99)
    blocks = [H.random_block(rng, w, h, density=0.5, big=0.1) for (w, h) in [(4, 4), (16, 16), (32, 8), (64, 64), (8, 8)] * 6]
    chromas = [int(x) for x in rng.integers(0, 2, len(blocks))]
    flags = [int(x) for x in rng.integers(0, 8, len(blocks))]
    tus, coeff = make_tus(blocks, chromas, flags)
    rec, off, info = hip.residual_batch(tus, coeff)
    assert int(off[0]) == 0 and int(off[-1]) == len(rec)
    for i, c in enumerate(blocks):
        want, last, mts = orc.residual_records(c, chromas[i], flags[i])
        assert np.array_equal(rec[int(off[i]): int(off[i + 1])], want), i
        assert int(info[i]) == (last | (H.TU_INFO_MTS_VIOLATION if mts else 0))
    # an all-zero block: flagged, status CABAC_HIP_ERR_SUBSTREAM, the other blocks still coded
    tus2, coeff2 = make_tus([blocks[0], np.zeros((4, 4), np.int32), blocks[1]], [0, 0, 1], [0, 0, 0])
    with pytest.raises(capi.CabacHipError):
        hip.residual_batch(tus2, coeff2)
    rec2, off2, info2 = hip.residual_batch(tus2, coeff2, check=False)
    assert int(info2[1]) == H.TU_INFO_EMPTY and int(off2[2]) == int(off2[1])
    assert np.array_equal(rec2[int(off2[2]): int(off2[3])], orc.residual_records(blocks[1], 1, 0)[0])
    # coefficients outside the buffer are refused before anything is launched
    tus3 = tus2.copy(); tus3[2]["coeff_offset"] = len(coeff2)
    with pytest.raises(capi.CabacHipError):
        hip.residual_batch(tus3, coeff2)


def _ts_block(rng, w, h, kind):
    if kind == 0:
        c = (rng.random((h, w)) < 0.3) * rng.integers(-4, 5, (h, w))
    elif kind == 1:
        c = rng.integers(-40, 41, (h, w))
    elif kind == 2:
        c = np.repeat(rng.integers(-3, 4, (h, 1)), w, 1) * (rng.random((h, w)) < 0.8)
    else:
        c = (rng.random((h, w)) < 0.05) * rng.integers(-3000, 3000, (h, w))
    c = c.astype(np.int32)
    if not c.any():
        c[rng.integers(0, h), rng.integers(0, w)] = 1
    return c


def test_transform_skip_blocks(hip):
    """residual_codingTS on the device against the oracle (pinned to the reference by test_residual_oracle.py), transform-skip
    and regular blocks mixed in one batch, BDPCM, budget exhaustion, escapes; a 64-wide TS block is refused."""
    rng = np.random.default_rng(0x7575)
    blocks, chromas, flags = [], [], []
    for w in (1, 2, 4, 8, 16, 32):
        for h in (1, 2, 4, 8, 16, 32):
            for k in range(6):
                blocks.append(_ts_block(rng, w, h, k % 4))
                chromas.append(int(rng.integers(0, 2)))
                flags.append(H.TU_TRANSFORM_SKIP | [H.TU_TS_FLAG, H.TU_BDPCM, 0][k % 3] | int(rng.integers(0, 4)))
            blocks.append(H.random_block(rng, w, h, density=0.5, big=0.1))      # a regular block in between
            chromas.append(0)
            flags.append(int(rng.integers(0, 8)))
    for w, h in [(4, 4), (32, 32), (8, 16)]:
        for v in (1, -7, 2000, -32768):
            c = np.full((h, w), v, np.int32)
            c[::2, 1::2] = -v if v != -32768 else 32767
            for fl in (H.TU_TRANSFORM_SKIP | H.TU_TS_FLAG, H.TU_TRANSFORM_SKIP | H.TU_BDPCM):
                blocks.append(c); chromas.append(0); flags.append(fl)
    order = rng.permutation(len(blocks))
    check_against_oracle(hip, [blocks[i] for i in order], [chromas[i] for i in order], [flags[i] for i in order], slack=2)
    tus, coeff = make_tus([np.ones((8, 64), np.int32)], [0], [H.TU_TRANSFORM_SKIP])
    recs, info = residual(hip, tus, coeff)
    assert len(recs[0]) == 0 and int(info[0]) == H.TU_INFO_BAD_DESC


def test_many_random_blocks_of_every_kind(hip):
    """40 000 blocks — every shape, regular and transform-skip, all flag combinations, sparse to saturated — in one
    batch (so that every launch variant, the ordering pre-pass with all classes, staged and direct rows, and rows of
    different kinds inside one wave are exercised) against the oracle."""
    rng = np.random.default_rng(0xA11)
    shapes = [(w, h) for w in (1, 2, 4, 8, 16, 32, 64) for h in (1, 2, 4, 8, 16, 32, 64)]
    weights = np.array([1.0 / (1 + (w * h) / 64.0) for w, h in shapes])
    weights /= weights.sum()
    blocks, chromas, flags = [], [], []
    for i in range(40000):
        w, h = shapes[int(rng.choice(len(shapes), p=weights))]
        ts = max(w, h) <= 32 and rng.random() < 0.25
        if ts:
            blocks.append(_ts_block(rng, w, h, int(rng.integers(0, 4))))
            flags.append(H.TU_TRANSFORM_SKIP | int(rng.choice([H.TU_TS_FLAG, H.TU_BDPCM, 0])) | int(rng.integers(0, 4)))
        else:
            blocks.append(H.random_block(rng, w, h, density=float(rng.choice([0.05, 0.3, 0.7, 1.0])), big=float(rng.choice([0.0, 0.05, 0.3])),
                                         huge=0.02 if rng.random() < 0.1 else 0.0, last_frac=float(rng.choice([1.0, 0.5, 0.2]))))
            fl = int(rng.integers(0, 8))
            flags.append(fl & ~H.TU_TS_FLAG if max(w, h) > 32 else fl)
        chromas.append(int(rng.integers(0, 2)))
    check_against_oracle(hip, blocks, chromas, flags)


def test_sbt_zero_out_blocks(hip):
    """CABAC_TU_SBT_ZERO_OUT: 32-wide / 32-tall luma blocks coded as their left / upper 16 (last-position clamp, groups passed
    over without a flag, reduced budget: cabac_writer.cpp:2660-2667, :2507-2516, unit.cpp:465-479) against the oracle — which
    test_residual_oracle.py pins to the reference's writer —, mixed with blocks the flag leaves alone, both passes."""
    rng = np.random.default_rng(0x5B7)
    shapes = [(32, 32), (32, 8), (8, 32), (32, 16), (16, 32), (32, 4), (4, 32), (32, 2), (2, 32), (32, 1), (16, 16), (8, 8), (4, 4), (64, 64)]
    blocks, chromas, flags = [], [], []
    for k in range(600):
        w, h = shapes[k % len(shapes)]
        ch = 1 if k % 11 == 10 else 0
        c = H.random_block(rng, w, h, density=[0.05, 0.4, 1.0][k % 3], big=[0.0, 0.2, 0.6][(k // 3) % 3], huge=0.05 if k % 17 == 0 else 0.0)
        if not ch and max(w, h) <= 32:
            if w == 32:
                c[:, 16:] = 0
            if h == 32:
                c[16:, :] = 0
            if not c.any():
                c[0, min(w, 16) - 1] = -2
        blocks.append(c); chromas.append(ch); flags.append(int(rng.integers(0, 4)) | H.TU_SBT_ZERO_OUT)
    check_against_oracle(hip, blocks, chromas, flags, slack=3)
