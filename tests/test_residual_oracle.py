"""CPU: the residual-coding restatement (oracle/cabac_oracle.c, orc_residual_records) pinned to the reference's own
CABACWriter::residual_coding (cabac_writer.cpp:2424-2872) compiled from its sources (oracle/_ref, build container
only), and everywhere else to the golden vectors that compiled reference produced (tests/golden/residual.npz,
oracle/gen_golden.py)."""
import os

import numpy as np
import pytest

import helpers as H

SIZES = [(w, h) for w in (1, 2, 4, 8, 16, 32, 64) for h in (1, 2, 4, 8, 16, 32, 64)]
needs_ref = pytest.mark.skipif(not H.ref_available(), reason="compiled reference only exists in the build container")


@needs_ref
def test_scan_order_matches_reference_rom():
    orc, ref = H.load_oracle(), H.load_ref()
    for w, h in SIZES:
        assert np.array_equal(orc.scan_order(w, h), ref.scan_order(w, h)), (w, h)


def _check(orc, ref, c, chroma, flags):
    want, info = ref.residual_records(c, chroma, flags)   # flags bit2 there: transform skip enabled in the SPS, max size 32
    if max(c.shape) > 32:
        flags &= ~H.TU_TS_FLAG                             # TU::isTSAllowed (unit_tools.cpp:651-664) is the caller's to evaluate
    got, last, mts = orc.residual_records(c, chroma, flags)
    assert np.array_equal(got, want), (c.shape, chroma, flags)
    # what residual_coding leaves in its CUCtx (cabac_writer.cpp:2461-2477, :2519-2522), from scanPosLast
    h, w = c.shape
    if w >= 4 and h >= 4:
        thr = 7 if (w, h) in ((4, 4), (8, 8)) else 15
        assert bool(info[1] >> chroma & 1) == (last > thr)
        assert bool(info[2]) == (last >= 1)
    if not chroma:
        assert bool(info[4]) == (last >= 1) and bool(info[3]) == mts


@needs_ref
@pytest.mark.parametrize("chroma", [0, 1])
def test_random_blocks_match_reference(chroma):
    orc, ref = H.load_oracle(), H.load_ref()
    rng = np.random.default_rng(0xF2 + chroma)
    for w, h in SIZES:
        for k in range(12):
            c = H.random_block(rng, w, h, density=[0.05, 0.3, 0.7, 1.0][k % 4], big=[0.0, 0.05, 0.3][k % 3],
                               huge=0.02 if k % 5 == 4 else 0.0, last_frac=[1.0, 0.5, 0.2][k % 3])
            _check(orc, ref, c, chroma, k % 8)


@needs_ref
def test_edge_blocks_match_reference():
    orc, ref = H.load_oracle(), H.load_ref()
    for w, h in [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (2, 8), (8, 2), (16, 1), (1, 16), (4, 32), (64, 4)]:
        we, he = min(w, 32), min(h, 32)
        cases = []
        z = np.zeros((h, w), np.int32)
        for (y, x) in [(0, 0), (he - 1, we - 1), (0, we - 1), (he - 1, 0)]:  # a single coefficient at each corner
            for v in (1, -1, 2, -3, 4, 5, 32767, -32768):
                c = z.copy(); c[y, x] = v; cases.append(c)
        c = z.copy(); c[:he, :we] = 1; cases.append(c)                 # dense ones
        c = z.copy(); c[:he, :we] = -32768; cases.append(c)            # every level an escape: the context-bin budget runs out
        c = z.copy(); c[:he, :we] = 3; c[0, 0] = -7; cases.append(c)
        c = z.copy(); c[:he, :we] = np.where((np.add.outer(np.arange(he), np.arange(we)) & 1) == 0, 2, -1); cases.append(c)
        for c in cases:
            for flags in (0, 1, 2, 3, 7):
                for chroma in (0, 1):
                    _check(orc, ref, c, chroma, flags)
    with pytest.raises(ValueError):
        orc.residual_records(np.zeros((8, 8), np.int32))
    with pytest.raises(ValueError):
        ref.residual_records(np.zeros((8, 8), np.int32))


@needs_ref
def test_extended_precision_dynamic_range_matches_reference():
    """maxLog2TrDynamicRange > 15 (SPS extended_precision_processing_flag, slice.hpp:180-192) changes the escape code
    length limits of encodeRemAbsEP (arith_codec.cpp:437-447)."""
    orc, ref = H.load_oracle(), H.load_ref()
    rng = np.random.default_rng(0xE0)
    for rng_bits in (17, 18, 19, 20):
        for w, h in [(4, 4), (16, 16), (8, 32)]:
            for k in range(4):
                c = H.random_block(rng, w, h, density=0.8, big=0.3, huge=0.3) * (1 << (rng_bits - 15))
                want, _ = ref.residual_records(c, k & 1, 3, max_log2_range=rng_bits)
                got, _, _ = orc.residual_records(c, k & 1, 3, max_log2_range=rng_bits)
                assert np.array_equal(got, want), (rng_bits, w, h, k)


def _ts_block(rng, w, h, kind):
    """Transform-skip blocks are spatial residuals: no fall-off towards high frequencies, runs of equal values."""
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


@needs_ref
def test_transform_skip_blocks_match_reference():
    """residual_codingTS (cabac_writer.cpp:2874-3046): mtsIdx == MTS_SKIP, with and without BDPCM, ts_flag coded or not."""
    orc, ref = H.load_oracle(), H.load_ref()
    rng = np.random.default_rng(0x75)
    for w in (1, 2, 4, 8, 16, 32):
        for h in (1, 2, 4, 8, 16, 32):
            for k in range(8):
                c = _ts_block(rng, w, h, k % 4)
                for chroma in (0, 1):
                    for extra in (H.TU_TS_FLAG, H.TU_BDPCM, 0):          # the reference codes ts_flag iff allowed: never with BDPCM
                        flags = H.TU_TRANSFORM_SKIP | extra | (k & 3)
                        want, _ = ref.residual_records(c, chroma, flags)
                        got, _, _ = orc.residual_records(c, chroma, flags)
                        assert np.array_equal(got, want), (w, h, k, chroma, flags)
    # budget exhaustion (7/4 context bins per sample) and 32-bin escapes
    for w, h in [(4, 4), (32, 32), (8, 16)]:
        for v in (1, -7, 2000, -32768):
            c = np.full((h, w), v, np.int32)
            c[::2, 1::2] = -v if v != -32768 else 32767
            for flags in (H.TU_TRANSFORM_SKIP | H.TU_TS_FLAG, H.TU_TRANSFORM_SKIP | H.TU_BDPCM):
                want, _ = ref.residual_records(c, 0, flags)
                got, _, _ = orc.residual_records(c, 0, flags)
                assert np.array_equal(got, want), (w, h, v, flags)


def test_golden_blocks():
    orc = H.load_oracle()
    g = np.load(os.path.join(H.GOLDEN, "residual.npz"))
    n = int(g["n_blocks"][0])
    assert n >= 100
    for k in range(n):
        lw, lh, chroma, flags = [int(x) for x in g["meta"][k]]
        c = g["coeff"][g["coeff_off"][k]: g["coeff_off"][k + 1]].reshape(1 << lh, 1 << lw)
        want = g["records"][g["rec_off"][k]: g["rec_off"][k + 1]]
        got, last, mts = orc.residual_records(c, chroma, flags)
        assert np.array_equal(got, want), k


def _encode_blocks(orc, blocks, chromas, flags, qp=32):
    """blocks -> records (oracle, pinned above) + TRM(1) -> the substream's bytes (finish + byte alignment)."""
    rec = np.concatenate([orc.residual_records(c, chromas[i], flags[i])[0] for i, c in enumerate(blocks)] + [np.array([0x81FF], np.uint16)])
    data, nbits = orc.encode_records(rec, qp, 2, 3)
    return data, len(rec)


@needs_ref
@pytest.mark.parametrize("flags", [0, H.TU_DEP_QUANT])
def test_residual_parser_matches_reference_reader(flags):
    """bytes -> coefficients: the oracle's parser against the reference's CABACReader::residual_coding, several blocks per
    substream so that contexts and the arithmetic decoder carry over; without sign hiding the blocks come back exactly."""
    orc, ref = H.load_oracle(), H.load_ref()
    rng = np.random.default_rng(0xDEC + flags)
    shapes = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 4), (4, 16), (32, 8), (2, 8), (64, 16), (1, 16), (16, 1)]
    for trial in range(6):
        blocks, metas = [], []
        for k in range(14):
            w, h = shapes[int(rng.integers(0, len(shapes)))]
            c = H.random_block(rng, w, h, density=[0.1, 0.4, 0.9][k % 3], big=[0.0, 0.1, 0.3][k % 3],
                               huge=0.02 if k % 7 == 6 else 0.0, last_frac=[1.0, 0.4][k % 2])
            blocks.append(c)
            metas.append((w, h, int(rng.integers(0, 2)), flags))
        data, _ = _encode_blocks(orc, blocks, [m[2] for m in metas], [m[3] for m in metas])
        rc_r, got_r, nb_r = ref.residual_decode(data, 32, metas)
        rc_o, got_o, nb_o = orc.residual_decode(data, 32, metas)
        assert rc_r == 0 and rc_o == 0 and nb_r == nb_o
        for k, c in enumerate(blocks):
            assert np.array_equal(got_o[k], got_r[k]), (trial, k, c.shape)
            assert np.array_equal(got_o[k], c), (trial, k, c.shape)


@needs_ref
def test_residual_parser_with_sign_hiding_matches_reference_reader():
    orc, ref = H.load_oracle(), H.load_ref()
    rng = np.random.default_rng(0x51D)
    for fl in (H.TU_SIGN_HIDING, H.TU_SIGN_HIDING | H.TU_DEP_QUANT):
        blocks, metas = [], []
        for k in range(20):
            w, h = [(4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (4, 16)][k % 6]
            blocks.append(H.random_block(rng, w, h, density=0.6, big=0.1))
            metas.append((w, h, k & 1, fl))
        data, _ = _encode_blocks(orc, blocks, [m[2] for m in metas], [m[3] for m in metas])
        rc_r, got_r, nb_r = ref.residual_decode(data, 32, metas)
        rc_o, got_o, nb_o = orc.residual_decode(data, 32, metas)
        assert rc_r == 0 and rc_o == 0 and nb_r == nb_o
        for k, c in enumerate(blocks):
            assert np.array_equal(got_o[k], got_r[k]), k            # the hidden sign follows the parity rule in both
            assert np.array_equal(np.abs(got_o[k]), np.abs(c)), k   # magnitudes always come back


def _expected_info(rinfo, w, h, chroma, info):
    """What the reference reader left in TransformUnit / CUCtx (ref_residual_decode's info) against the parser's info word."""
    ts, lfnst_viol, lfnst_last, mts_viol, mts_last = (int(x) for x in rinfo)
    assert bool(info & H.TU_INFO_TS) == bool(ts)
    if ts:
        return
    last = int(info) & 0xFFFF
    big = w >= 4 and h >= 4
    max_lfnst = 7 if (w, h) in ((4, 4), (8, 8)) else 15
    assert lfnst_viol == (int(big and last > max_lfnst) << (1 if chroma else 0))
    assert lfnst_last == int(big and last >= 1)
    assert mts_last == int((not chroma) and last >= 1)
    assert mts_viol == int(bool(info & H.TU_INFO_MTS_VIOLATION))


@needs_ref
def test_transform_skip_parser_matches_reference_reader():
    """residual_codingTS / residual_coding_subblockTS (cabac_reader.cpp:3130-3339) and ts_flag (:2737-2752): substreams
    that mix regular blocks, transform-skip blocks whose flag is in the stream, transform-skip blocks without a coded flag
    and BDPCM blocks, contexts and the arithmetic decoder carrying over; every block comes back exactly (no sign hiding),
    and the parser reports what the reader leaves in mtsIdx / CUCtx."""
    orc, ref = H.load_oracle(), H.load_ref()
    rng = np.random.default_rng(0x7500)
    sizes = [1, 2, 4, 8, 16, 32]
    for trial in range(12):
        blocks, metas = [], []
        for k in range(16):
            kind = int(rng.integers(0, 5))
            slice_fl = int(rng.integers(0, 2))                    # dependent quantisation on / off (per block in the rig)
            if kind == 0:                                         # regular, flag not coded
                w, h = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 32), (8, 4), (2, 8)][int(rng.integers(0, 7))]
                c, fl = H.random_block(rng, w, h, density=0.5, big=0.1), slice_fl
            elif kind == 1:                                       # regular, transform_skip_flag = 0 in the stream
                w, h = [(4, 4), (8, 8), (16, 16), (32, 32), (16, 4)][int(rng.integers(0, 5))]
                c, fl = H.random_block(rng, w, h, density=0.5, big=0.1), slice_fl | H.TU_TS_FLAG
            else:
                w, h = sizes[int(rng.integers(0, 6))], sizes[int(rng.integers(0, 6))]
                if w * h == 1:
                    w = 2
                c = _ts_block(rng, w, h, int(rng.integers(0, 4)))
                fl = slice_fl | H.TU_TRANSFORM_SKIP | [H.TU_TS_FLAG, 0, H.TU_BDPCM][kind - 2]
            blocks.append(c)
            metas.append((w, h, int(rng.integers(0, 2)), fl))
        qp = int(rng.integers(20, 45))
        data, _ = _encode_blocks(orc, blocks, [m[2] for m in metas], [m[3] for m in metas], qp=qp)
        rc_o, got_o, nb_o, info_o = orc.residual_decode(data, qp, metas, with_info=True)
        assert rc_o == 0, trial
        rc_r, got_r, nb_r, info_r = ref.residual_decode(data, qp, metas, with_info=True)
        assert rc_r == 0 and nb_r == nb_o, trial
        for k, c in enumerate(blocks):
            assert np.array_equal(got_r[k], c) and np.array_equal(got_o[k], c), (trial, k, metas[k])
            _expected_info(info_r[k], metas[k][0], metas[k][1], metas[k][2], int(info_o[k]))


@needs_ref
def test_transform_skip_parser_budget_and_escapes():
    """Blocks that exhaust the 7/4 bins-per-sample budget (bypass-coded levels with bypass signs) and 32-bin escapes."""
    orc, ref = H.load_oracle(), H.load_ref()
    for w, h in [(4, 4), (32, 32), (8, 16), (2, 2), (1, 16)]:
        for v in (1, -7, 2000, -32768):
            c = np.full((h, w), v, np.int32)
            c[::2, 1::2] = -v if v != -32768 else 32767
            for fl in (H.TU_TRANSFORM_SKIP | H.TU_TS_FLAG, H.TU_TRANSFORM_SKIP | H.TU_BDPCM, H.TU_TRANSFORM_SKIP):
                metas = [(w, h, 0, fl), (w, h, 1, fl)]
                data, _ = _encode_blocks(orc, [c, c], [0, 1], [fl, fl])
                rc_r, got_r, nb_r = ref.residual_decode(data, 32, metas)
                rc_o, got_o, nb_o = orc.residual_decode(data, 32, metas)
                assert rc_r == 0 and rc_o == 0 and nb_r == nb_o, (w, h, v, fl)
                for k in range(2):
                    assert np.array_equal(got_r[k], c) and np.array_equal(got_o[k], c), (w, h, v, fl, k)


def test_residual_parser_errors():
    orc = H.load_oracle()
    c = np.array([[3, 0, 0, 0], [0, -1, 0, 0], [0, 0, 0, 0], [0, 0, 0, 2]], np.int32)
    data, _ = _encode_blocks(orc, [c], [0], [0])
    rc, got, _ = orc.residual_decode(data, 32, [(4, 4, 0, 0)])
    assert rc == 0 and np.array_equal(got[0], c)
    assert orc.residual_decode(data[:1], 32, [(4, 4, 0, 0)])[0] == -4                 # read past the end
    assert orc.residual_decode(data, 32, [(64, 64, 0, H.TU_TRANSFORM_SKIP)])[0] == -2  # transform skip stops at 32 x 32


def test_golden_parse_substreams():
    """The oracle's parser against what the compiled reference reader decoded (tests/golden/residual_parse.npz)."""
    orc = H.load_oracle()
    g = np.load(os.path.join(H.GOLDEN, "residual_parse.npz"))
    for s in range(int(g["n_sub"][0])):
        metas = [tuple(int(x) for x in m) for m in g["s%d_meta" % s]]
        qp, nbits = [int(x) for x in g["s%d_qp" % s]]
        rc, dec, nb, info = orc.residual_decode(g["s%d_bytes" % s], qp, metas, with_info=True)
        assert rc == 0 and nb == nbits
        assert np.array_equal(np.concatenate([d.ravel() for d in dec]), g["s%d_coeff" % s]), s
        for k, m in enumerate(metas):      # mtsIdx / CUCtx as the reference reader left them
            _expected_info(g["s%d_refinfo" % s][k], m[0], m[1], m[2], int(info[k]))


def _sbt_block(rng, w, h, density, big=0.1):
    """A luma block as an encoder with SBT + MTS leaves it: nothing outside the left 16 columns of a 32-wide block / the upper
    16 rows of a 32-tall one (unit.cpp:465-479)."""
    c = H.random_block(rng, w, h, density=density, big=big)
    if w == 32:
        c[:, 16:] = 0
    if h == 32:
        c[16:, :] = 0
    if not c.any():
        c[0, 0] = 3
    return c


@needs_ref
def test_sbt_zero_out_matches_reference():
    """CABAC_TU_SBT_ZERO_OUT (sps.getUseMTS() && cu.sbtInfo != 0, luma, at most 32 x 32): the clamp of the last position's
    prefix (cabac_writer.cpp:2660-2667), the coefficient groups passed over without a flag (:2507-2516) and the budget of the
    reduced area (unit.cpp:465-479) — records and CUCtx against the reference's writer, bytes -> coefficients against its
    reader (cabac_reader.cpp:2880-2891, :2718-2727); chroma blocks and small blocks are not affected by the flag."""
    orc, ref = H.load_oracle(), H.load_ref()
    rng = np.random.default_rng(0x5B7)
    shapes = [(32, 32), (32, 8), (8, 32), (32, 16), (16, 32), (32, 4), (4, 32), (32, 2), (16, 16), (8, 8), (4, 4)]
    for trial in range(40):
        blocks, metas = [], []
        for k in range(8):
            w, h = shapes[int(rng.integers(0, len(shapes)))]
            chroma = 1 if k == 7 else 0
            fl = int(rng.integers(0, 4)) | H.TU_SBT_ZERO_OUT
            c = H.random_block(rng, w, h, density=0.4) if chroma else _sbt_block(rng, w, h, [0.05, 0.4, 1.0][k % 3], big=[0.0, 0.2, 0.6][k % 3])
            if trial == 0 and not chroma and w == 32:
                c[:, :] = 0
                c[0 if h < 32 else min(h, 16) - 1, 15] = -2          # the last position ON the clamp: no terminating prefix bin
            if chroma:
                fl &= ~H.TU_SBT_ZERO_OUT                              # (the rig applies SBT to the CU; chroma is untouched by it)
                want, info = ref.residual_records(c, 1, fl | H.TU_SBT_ZERO_OUT)
            else:
                want, info = ref.residual_records(c, 0, fl)
            got, last, mts = orc.residual_records(c, chroma, fl)
            assert np.array_equal(got, want), (trial, k, (w, h), chroma, fl)
            if not chroma:
                assert bool(info[3]) == mts
            blocks.append(c)
            metas.append((w, h, chroma, fl & ~H.TU_SIGN_HIDING))
        rec = np.concatenate([orc.residual_records(c, m[2], m[3])[0] for c, m in zip(blocks, metas)] + [np.array([0x81FF], np.uint16)])
        data, _ = orc.encode_records(rec, 30, 2, 3)
        ref_metas = [(w, h, ch, fl | H.TU_SBT_ZERO_OUT) for (w, h, ch, fl) in metas]   # SBT is a property of the CU in the rig
        rc_r, got_r, nb_r = ref.residual_decode(data, 30, ref_metas)
        rc_o, got_o, nb_o = orc.residual_decode(data, 30, metas)
        assert rc_r == 0 and rc_o == 0 and nb_r == nb_o
        for k, c in enumerate(blocks):
            assert np.array_equal(got_o[k], got_r[k]) and np.array_equal(got_o[k], c), (trial, k)
    # without the flag the same 32-wide block is coded differently (one more prefix bin, the full budget)
    c = np.zeros((8, 32), np.int32); c[0, 15] = 1
    assert len(orc.residual_records(c, 0, H.TU_SBT_ZERO_OUT)[0]) < len(orc.residual_records(c, 0, 0)[0])
