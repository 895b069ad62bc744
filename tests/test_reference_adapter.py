"""The reference's OWN CABACWriter class running on top of the GPU recording encoder
(integration/reference_adapter.hpp: BinEncoderHipRef IS-A EntropyCoding::BinEncIf), compared in the same
process with the reference's BinEncoder_Std under the same CABACWriter.  Needs
oracle/_ref/libadapter_test.so (built by `make -C oracle` where /root/reference exists; it travels to the
GPU box as a prebuilt library)."""
import ctypes
import os
import tempfile

import numpy as np
import pytest

import helpers as H

SAN = os.environ.get("CABAC_TEST_SANITIZED_ADAPTER")   # tests/test_sanitizers.py: everything in one library under ASan + UBSan
SO = SAN or os.path.join(H.ORACLE_DIR, "_ref", "libadapter_test.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SO), reason="oracle/_ref/libadapter_test.so not built")


@pytest.fixture(scope="module")
def adp():
    if not SAN:
        from entropy_coding_amd import capi
        capi.load_library()
    so = SAN or H.ref_test_library("libadapter_test.so")   # rebuilt, or refused, when it is older than the host headers
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp(prefix="cabac_ref_"))   # reference log.cpp:3-4 creates bin_log.txt/bit_log.txt in CWD
    try:
        L = ctypes.CDLL(so)
    finally:
        os.chdir(cwd)
    L.adapter_last_error.restype = ctypes.c_char_p
    L.adapter_encode.restype = ctypes.c_long
    L.adapter_encode.argtypes = [ctypes.c_int, H.u32p, ctypes.c_long, ctypes.c_int, ctypes.c_int, H.u8p, ctypes.c_long,
                                 H.u32p, H.u32p]
    L.adapter_estimate.restype = ctypes.c_long
    L.adapter_estimate.argtypes = [ctypes.c_int, H.u32p, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_int),
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    L.adapter_residual.restype = ctypes.c_long
    L.adapter_residual.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                   ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int, H.u8p, ctypes.c_long,
                                   ctypes.POINTER(ctypes.c_int32)]
    L.adapter_record.restype = ctypes.c_long
    L.adapter_record.argtypes = [H.u32p, ctypes.c_long, H.u16p, ctypes.c_long, H.u32p]
    return L


def _enc(adp, which, ops, qp, iid):
    ops = np.ascontiguousarray(ops, np.uint32)
    out = np.zeros(64 + 40 * len(ops), np.uint8)
    nbits, nbins = ctypes.c_uint32(), ctypes.c_uint32()
    n = adp.adapter_encode(which, H._ptr(ops, H.u32p), len(ops), qp, iid, H._ptr(out, H.u8p), len(out),
                           ctypes.byref(nbits), ctypes.byref(nbins))
    assert n >= 0, adp.adapter_last_error()
    return out[:n].copy(), nbits.value, nbins.value


def test_adapter_records_under_reference_cabacwriter(adp):
    """CPU: the reference CABACWriter's binarisers + our recorder give the oracle's record stream."""
    orc = H.load_oracle()
    rng = np.random.default_rng(123)
    ops = H.random_ops(rng, 4000, ctx_frac=0.5, end_trm=False)
    rec = np.zeros(40 * len(ops), np.uint16)
    nb = ctypes.c_uint32()
    n = adp.adapter_record(H._ptr(ops, H.u32p), len(ops), H._ptr(rec, H.u16p), len(rec), ctypes.byref(nb))
    assert n >= 0, adp.adapter_last_error()
    assert np.array_equal(rec[:n], orc.ops_to_records(ops)) and nb.value == n


def test_bin_store_is_kept_as_the_reference_keeps_it(adp):
    """CPU: setBinStorage(true) / getBinStore() / getTestBinEncoder() (arith_codec.cpp:585-601, the window-size training
    path): the recording encoder keeps the reference's own BinStore filled bin for bin as TBinEncoder does."""
    rng = np.random.default_rng(55)
    ops = H.random_ops(rng, 5000, ctx_frac=0.7, end_trm=False)
    f = adp.adapter_bin_store
    f.restype = ctypes.c_long
    f.argtypes = [ctypes.c_int, H.u32p, ctypes.c_long, H.u32p, ctypes.POINTER(ctypes.c_int)]
    got = []
    for which in (0, 1):
        out = np.zeros(H.NUM_CTX, np.uint32)
        has = ctypes.c_int(0)
        n = f(which, H._ptr(ops, H.u32p), len(ops), H._ptr(out, H.u32p), ctypes.byref(has))
        assert n > 1000 and has.value == 1, (n, adp.adapter_last_error())
        got.append((n, out))
    assert got[0][0] == got[1][0] and np.array_equal(got[0][1], got[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_reference_cabacwriter_on_gpu_encoder_is_bit_exact(adp, seed):
    rng = np.random.default_rng(500 + seed)
    ops = H.random_ops(rng, int(rng.choice([0, 10, 3000, 12000])), ctx_frac=float(rng.choice([0.3, 0.7])), end_trm=False)
    qp, iid = int(rng.integers(0, 64)), int(rng.integers(0, 3))
    ref_bytes, ref_bits, ref_bins = _enc(adp, 0, ops, qp, iid)     # reference encoder
    gpu_bytes, gpu_bits, gpu_bins = _enc(adp, 1, ops, qp, iid)     # GPU encoder behind the same interface
    assert ref_bits == gpu_bits and ref_bins == gpu_bins and np.array_equal(ref_bytes, gpu_bytes)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_reference_cabacwriter_on_gpu_estimator_matches_bitestimator_std(adp, seed):
    """The reference's own CABACWriter driving BitEstimatorHipRef (recorded, costed on the GPU) reports the same
    getEstFracBits() as on the reference's BitEstimator_Std — after every segment, with resetBits() / start() /
    restart() in between and the contexts carrying on."""
    rng = np.random.default_rng(880 + seed)
    ops = H.random_ops(rng, 1200, ctx_frac=[0.5, 0.8, 0.3][seed], with_align=True)
    ends = np.array(np.sort(rng.integers(0, len(ops) + 1, size=5)).tolist() + [len(ops)], np.int64)
    kinds = np.array([0] + [int(k) for k in rng.integers(0, 3, size=5)], np.int32)
    res = []
    for which in (0, 1):
        costs = np.zeros(len(ends), np.uint64)
        rc = adp.adapter_estimate(which, H._ptr(ops, H.u32p), ends.ctypes.data_as(ctypes.POINTER(ctypes.c_long)),
                                  kinds.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(ends), 29, seed % 3,
                                  costs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
        assert rc == 0, adp.adapter_last_error()
        res.append(costs)
    assert np.array_equal(res[0], res[1])


@pytest.mark.gpu
def test_get_num_written_bits_matches_bin_encoder_std(adp):
    """getNumWrittenBits() (arith_codec.cpp:482-485, what estBits asks for): the reference's BinEncoder_Std and BinEncoderHipRef
    in Immediate mode (one probing launch per question) give the same answers all along a substream."""
    rng = np.random.default_rng(41)
    f = adp.adapter_num_written_bits
    f.restype = ctypes.c_long
    f.argtypes = [ctypes.c_int, H.u32p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, H.u32p, ctypes.c_long]
    for lead, every in ((0, 5), (3, 33)):
        ops = H.random_ops(rng, 300, ctx_frac=0.65, end_trm=False, with_align=True)
        got = []
        for which in (0, 1):
            ans = np.zeros(len(ops), np.uint32)
            n = f(which, H._ptr(ops, H.u32p), len(ops), 33, 1, every, lead, H._ptr(ans, H.u32p), len(ans))
            assert n > 0, adp.adapter_last_error()
            got.append(ans[:n].copy())
        assert np.array_equal(got[0], got[1]) and got[0][-1] > lead


def _residual(adp, which, blocks, comps, rig_flags, qp=32):
    wh = np.array([[c.shape[1], c.shape[0]] for c in blocks], np.int32).ravel()
    comp = np.array(comps, np.int32)
    coeff = np.concatenate([c.ravel() for c in blocks]).astype(np.int32)
    out = np.zeros(64 + 8 * len(coeff), np.uint8)
    cu = np.zeros(8, np.int32)
    ip = ctypes.POINTER(ctypes.c_int)
    n = adp.adapter_residual(which, len(blocks), wh.ctypes.data_as(ip), comp.ctypes.data_as(ip), rig_flags,
                             coeff.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), qp, H._ptr(out, H.u8p), len(out),
                             cu.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    assert n >= 0, adp.adapter_last_error()
    return out[:n].copy(), cu[:4].copy()


@pytest.mark.gpu
@pytest.mark.parametrize("rig_flags", [0, 1, 2, 3, 7])
def test_residual_coding_through_the_gpu_binariser_matches_reference_writer(adp, rig_flags):
    """The reference's CABACWriter::residual_coding on its BinEncoder_Std against ResidualCoderHipRef (GPU binariser
    feeding the same BinEncoder_Std) over the reference's own TransformUnit objects: same bytes, same CUCtx."""
    rng = np.random.default_rng(0x5E5 + rig_flags)
    sizes = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 4), (4, 16), (32, 8), (2, 8), (64, 16)]
    blocks, comps = [], []
    for k in range(40):
        w, h = sizes[k % len(sizes)]
        comp = int(rng.integers(0, 3))
        blocks.append(H.random_block(rng, w, h, density=[0.1, 0.4, 0.9][k % 3], big=[0.0, 0.1, 0.3][k % 3],
                                     huge=0.02 if k % 7 == 6 else 0.0, last_frac=[1.0, 0.4][k % 2]))
        comps.append(comp)
    want, cu_want = _residual(adp, 0, blocks, comps, rig_flags)
    one, cu_one = _residual(adp, 1, blocks, comps, rig_flags)          # block by block
    many, cu_many = _residual(adp, 2, blocks, comps, rig_flags)        # queued, one launch
    assert np.array_equal(one, want) and np.array_equal(cu_one, cu_want)
    assert np.array_equal(many, want) and np.array_equal(cu_many, cu_want)


@pytest.mark.gpu
@pytest.mark.parametrize("rig_flags", [0x14, 0x10, 0x30, 0x17])
def test_transform_skip_residual_through_the_gpu_binariser_matches_reference_writer(adp, rig_flags):
    """mtsIdx == MTS_SKIP (rig bit 4; bit 5 BDPCM, bit 2 transform skip enabled in the SPS so that ts_flag is coded):
    CABACWriter::residual_coding takes its residual_codingTS branch; ResidualCoderHipRef must produce the same bytes."""
    rng = np.random.default_rng(0x7A + rig_flags)
    blocks, comps = [], []
    for k in range(30):
        w, h = [(4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (4, 16), (2, 8)][k % 7]
        kind = k % 4
        c = ((rng.random((h, w)) < [0.3, 1.0, 0.8, 0.05][kind]) * rng.integers(-[4, 40, 3, 3000][kind], [4, 40, 3, 3000][kind] + 1, (h, w))).astype(np.int32)
        if not c.any():
            c[0, 0] = 1
        blocks.append(c)
        comps.append(int(rng.integers(0, 3)))
    want, cu_want = _residual(adp, 0, blocks, comps, rig_flags)
    many, cu_many = _residual(adp, 2, blocks, comps, rig_flags)
    assert np.array_equal(many, want) and np.array_equal(cu_many, cu_want)


@pytest.mark.gpu
@pytest.mark.parametrize("rig_flags", [0, 3, 7, 0x14, 0x30])
def test_residual_blocks_spliced_on_the_device_match_reference_writer(adp, rig_flags):
    """Row f2 closed end to end: the reference's CABACWriter on BinEncoderHipRef, residual_coding replaced by
    ResidualCoderHipRef's splice form — coefficients to the device, the block bins spliced into the recorded ones there, one
    flush — against the reference's CABACWriter::residual_coding on BinEncoder_Std with other syntax elements in between:
    same bytes, same CUCtx (read before the flush), same getNumBins() / getNumBins(ctxId) (after it)."""
    rng = np.random.default_rng(0xF2 + rig_flags)
    ts = bool(rig_flags & 0x10)
    sizes = [(4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (4, 16), (2, 8)] + ([] if ts else [(64, 64), (64, 16)])
    blocks, comps, op_list = [], [], []
    for k in range(24):
        w, h = sizes[k % len(sizes)]
        if ts:
            c = ((rng.random((h, w)) < 0.5) * rng.integers(-20, 21, (h, w))).astype(np.int32)
            if not c.any():
                c[0, 0] = 1
        else:
            c = H.random_block(rng, w, h, density=[0.1, 0.4, 0.9][k % 3], big=[0.0, 0.1, 0.3][k % 3], last_frac=[1.0, 0.4][k % 2])
        blocks.append(c)
        comps.append(int(rng.integers(0, 3)))
        op_list.append(H.random_ops(rng, int(rng.integers(0, 6)), ctx_frac=0.7, end_trm=False))
    op_list.append(H.random_ops(rng, 5, ctx_frac=0.7, end_trm=False))
    wh = np.array([[c.shape[1], c.shape[0]] for c in blocks], np.int32).ravel()
    comp = np.array(comps, np.int32)
    coeff = np.concatenate([c.ravel() for c in blocks]).astype(np.int32)
    ops = np.concatenate(op_list).astype(np.uint32)
    op_off = np.concatenate([[0], np.cumsum([len(o) for o in op_list])]).astype(np.int64)
    probe = np.array([90, 102, 150, 182, 246, 269, 310, 373], np.int32)
    ip, lp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_long)
    f = adp.adapter_residual_spliced
    f.restype = ctypes.c_long
    f.argtypes = [ctypes.c_int, ctypes.c_int, ip, ip, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), H.u32p, lp, ctypes.c_int, H.u8p,
                  ctypes.c_long, ctypes.POINTER(ctypes.c_int32), ip, H.u32p]
    got = []
    for which in (0, 1):
        out = np.zeros(64 + 8 * len(coeff), np.uint8)
        cu = np.zeros(8, np.int32)
        counts = np.zeros(9, np.uint32)
        n = f(which, len(blocks), wh.ctypes.data_as(ip), comp.ctypes.data_as(ip), rig_flags, coeff.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
              H._ptr(ops, H.u32p), op_off.ctypes.data_as(lp), 30, H._ptr(out, H.u8p), len(out), cu.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
              probe.ctypes.data_as(ip), H._ptr(counts, H.u32p))
        assert n >= 0, adp.adapter_last_error()
        got.append((out[:n].copy(), cu[:4].copy(), counts.copy()))
    assert np.array_equal(got[1][0], got[0][0])
    assert np.array_equal(got[1][1], got[0][1])
    assert np.array_equal(got[1][2], got[0][2]) and got[0][2][0] > 100


@pytest.mark.gpu
def test_residual_adapter_empty_block_throws_like_the_reference(adp):
    z = [np.zeros((8, 8), np.int32)]
    for which in (0, 1):
        wh = np.array([8, 8], np.int32); comp = np.array([0], np.int32); cu = np.zeros(8, np.int32); out = np.zeros(64, np.uint8)
        ip = ctypes.POINTER(ctypes.c_int)
        n = adp.adapter_residual(which, 1, wh.ctypes.data_as(ip), comp.ctypes.data_as(ip), 0,
                                 z[0].ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), 32, H._ptr(out, H.u8p), len(out),
                                 cu.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        assert n == -1 and b"empty TU" in adp.adapter_last_error()


# ---- decoder side: ResidualParserHipRef / BinDecoderHipRef behind the reference's reader-side types ----------------------
def _parse(adp, which, metas, rig, data, qp):
    n = len(metas)
    wh = np.array([[w, h] for w, h, _ in metas], np.int32).ravel()
    comp = np.array([c for _, _, c in metas], np.int32)
    rig = np.asarray(rig, np.int32)
    total = int(sum(w * h for w, h, _ in metas))
    co = np.full(total, 0x5A5A5A5A, np.int32)
    tu = np.zeros(n, np.int32)
    cu = np.zeros(8, np.int32)
    ip, i32p = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int32)
    f = adp.adapter_residual_parse
    f.restype = ctypes.c_long
    f.argtypes = [ctypes.c_int, ctypes.c_int, ip, ip, ip, H.u8p, ctypes.c_long, ctypes.c_int, i32p, i32p, i32p]
    data = np.ascontiguousarray(data, np.uint8)
    rc = f(which, n, wh.ctypes.data_as(ip), comp.ctypes.data_as(ip), rig.ctypes.data_as(ip), H._ptr(data, H.u8p), len(data), qp,
           co.ctypes.data_as(i32p), tu.ctypes.data_as(i32p), cu.ctypes.data_as(i32p))
    assert rc >= 0, adp.adapter_last_error()
    return co, tu, cu[:4].copy(), rc


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_residual_parser_adapter_matches_reference_reader(adp, seed):
    """CABACReader::residual_coding (reference, on BinDecoder_Std) and ResidualParserHipRef (device) over the same bytes —
    written by the reference's own CABACWriter::residual_coding —: identical TransformUnit coefficients, mtsIdx and CUCtx.
    Regular blocks, transform_skip_flag coded 0 / 1, transform skip without a flag, BDPCM; dependent quantisation and sign
    hiding per block."""
    ref = H.load_ref()
    rng = np.random.default_rng(0xAD0 + seed)
    metas, rig, recs = [], [], []
    for k in range(36):
        kind = int(rng.integers(0, 5))
        comp = int(rng.integers(0, 3))
        slice_fl = int(rng.integers(0, 4))
        if kind < 2:
            w, h = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 4), (4, 16), (2, 8)][int(rng.integers(0, 8))]
            if kind == 1:
                w, h = min(w, 32), min(h, 32)
            c = H.random_block(rng, w, h, density=float(rng.choice([0.1, 0.5, 1.0])), big=float(rng.choice([0.0, 0.2])))
            fl = slice_fl | (H.TU_TS_FLAG if kind == 1 else 0)
        else:
            w, h = int(rng.choice([2, 4, 8, 16, 32])), int(rng.choice([2, 4, 8, 16, 32]))
            c = ((rng.random((h, w)) < 0.6) * rng.integers(-30, 31, (h, w))).astype(np.int32)
            if not c.any():
                c[0, 0] = -2
            fl = (slice_fl & 1) | H.TU_TRANSFORM_SKIP | [H.TU_TS_FLAG, 0, H.TU_BDPCM][kind - 2]
        metas.append((w, h, comp))
        rig.append((fl & 3) | (4 if fl & H.TU_TS_FLAG else 0) | (0x10 if fl & H.TU_TRANSFORM_SKIP else 0) | (0x20 if fl & H.TU_BDPCM else 0))
        recs.append(ref.residual_records(c, 1 if comp else 0, fl)[0])     # the reference's writer
    data, _ = ref.encode_records(np.concatenate(recs + [np.array([0x81FF], np.uint16)]), 31, 2, 3)
    co_r, tu_r, cu_r, pos_r = _parse(adp, 0, metas, rig, data, 31)
    co_g, tu_g, cu_g, pos_g = _parse(adp, 1, metas, rig, data, 31)
    assert np.array_equal(co_g, co_r) and np.array_equal(tu_g, tu_r) and np.array_equal(cu_g, cu_r) and pos_g == pos_r


@pytest.mark.gpu
def test_bin_decoder_adapter_serves_the_planned_sequence(adp):
    """BinDecoderHipRef IS-A BinDecoderBase: the planned ctxId / bypass / terminate sequence decoded on the device comes back
    through decodeBin (the interface's virtual) and the class's own bypass / terminate calls exactly as BinDecoder_Std
    decodes it, and the bitstream is left at the same byte."""
    ref = H.load_ref()
    rng = np.random.default_rng(77)
    rec = H.random_records(rng, 6000)
    data, _ = ref.encode_records(rec, 29, 2, 3)
    buf = np.concatenate([data, np.array([9, 9, 9], np.uint8)])
    f = adp.adapter_decode_replay
    f.restype = ctypes.c_long
    f.argtypes = [ctypes.c_int, H.u16p, ctypes.c_long, ctypes.c_int, ctypes.c_int, H.u8p, ctypes.c_long, H.u8p, H.u32p]
    got = {}
    for which in (0, 1):
        bins = np.zeros(len(rec), np.uint8)
        idx = ctypes.c_uint32()
        rc = f(which, H._ptr(rec, H.u16p), len(rec), 29, 2, H._ptr(buf, H.u8p), len(buf), H._ptr(bins, H.u8p), ctypes.byref(idx))
        assert rc == 0, adp.adapter_last_error()
        got[which] = (bins, idx.value)
    assert np.array_equal(got[1][0], got[0][0]) and np.array_equal(got[0][0], (rec >> 15).astype(np.uint8))
    assert got[1][1] == got[0][1] == len(data)
