"""Pins oracle/cabac_oracle.c against the reference's OWN compiled sources
(oracle/_ref/libcabac_ref.so = /root/reference/src/**/*.cpp + oracle/ref_harness.cpp).
Runs wherever that library exists (the build container; it also travels to the GPU box)."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.skipif(not H.ref_available(), reason="oracle/_ref/libcabac_ref.so not built "
                                "(needs /root/reference; run `make -C oracle`)")


def test_reference_facts():
    ref = H.load_ref()
    assert ref.lib.ref_num_contexts() == 379
    assert ref.lib.ref_sizeof_prob_model() == 6


@pytest.mark.parametrize("init_id", [0, 1, 2])
def test_ctx_init_all_qp(init_id):
    ref, orc = H.load_ref(), H.load_oracle()
    for qp in range(-3, 70):
        a, b = ref.ctx_init(qp, init_id), orc.ctx_init(qp, init_id)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), (qp, init_id)


def test_ctx_update_traces():
    ref, orc = H.load_ref(), H.load_oracle()
    rng = np.random.default_rng(11)
    for trial in range(200):
        qp, init_id, ctx = int(rng.integers(0, 64)), int(rng.integers(0, 3)), int(rng.integers(0, 379))
        p = rng.choice([0.0, 0.02, 0.3, 0.5, 0.9, 1.0])
        bins = (rng.random(400) < p).astype(np.uint8)
        rg = int(rng.integers(256, 511))
        for x, y in zip(ref.ctx_trace(qp, init_id, ctx, bins, rg), orc.ctx_trace(qp, init_id, ctx, bins, rg)):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("seed", range(12))
def test_encode_ops_bit_exact(seed):
    ref, orc = H.load_ref(), H.load_oracle()
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([0, 1, 7, 100, 3000, 20000]))
    ops = H.random_ops(rng, n, ctx_frac=float(rng.choice([0.0, 0.3, 0.6, 0.9])), with_align=(seed % 4 == 3))
    qp, init_id = int(rng.integers(0, 64)), int(rng.integers(0, 3))
    for flags in (0, 1, 3):
        rb, rbits, rbins = ref.encode_ops(ops, qp, init_id, flags)
        ob, obits, obins = orc.encode_ops(ops, qp, init_id, flags)
        assert rbits == obits and np.array_equal(rb, ob)
        assert np.array_equal(rbins, obins)
    # flattening ops -> bin records is bit-exact in the reference itself
    rec = orc.ops_to_records(ops)
    assert len(rec) == int(obins.sum()) + int((ops[:, 0] == H.OP_ALIGN).sum())
    fb, fbits = ref.encode_records(rec, qp, init_id, 3)
    ob3, obits3, _ = orc.encode_ops(ops, qp, init_id, 3)
    assert fbits == obits3 and np.array_equal(fb, ob3)
    xb, xbits = orc.encode_records(rec, qp, init_id, 3)
    assert xbits == fbits and np.array_equal(xb, fb)


def test_num_written_bits_probe():
    """flags bit 2 (CABAC_SUB_PROBE on the device): the oracle's count against the reference's own
    BinEncoderBase::getNumWrittenBits() (arith_codec.cpp:482-485) at many points along bin strings, long 0xFF runs included
    (buffered bytes count)."""
    ref, orc = H.load_ref(), H.load_oracle()
    rng = np.random.default_rng(4321)
    for trial in range(40):
        if trial % 5 == 4:      # an all-MPS single-context stream: long runs of outstanding bytes
            rec = np.full(4000, 17 | (0x8000 if trial & 1 else 0), np.uint16)
        else:
            rec = H.random_records(rng, int(rng.integers(1, 3000)), ctx_frac=float(rng.choice([0.0, 0.5, 0.9, 1.0])), end_trm=False)
        qp, init_id = int(rng.integers(0, 64)), int(rng.integers(0, 3))
        for m in sorted(set([0, 1, 2, len(rec)] + [int(x) for x in rng.integers(0, len(rec) + 1, size=12)])):
            _, rbits = ref.encode_records(rec[:m], qp, init_id, 4)
            _, obits = orc.encode_records(rec[:m], qp, init_id, 4)
            assert rbits == obits, (trial, m)
    assert orc.encode_records(np.zeros(0, np.uint16), 30, 2, 4)[1] == 0


@pytest.mark.parametrize("seed", range(8))
def test_decode_records_and_ops(seed):
    ref, orc = H.load_ref(), H.load_oracle()
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.choice([1, 50, 5000, 30000]))
    ops = H.random_ops(rng, n, ctx_frac=float(rng.choice([0.2, 0.6, 0.85])), with_align=(seed % 4 == 1))
    qp, init_id = int(rng.integers(0, 64)), int(rng.integers(0, 3))
    data, nbits, _ = ref.encode_ops(ops, qp, init_id, 3)
    rec = orc.ops_to_records(ops)
    rc_r, bins_r, nread_r = ref.decode_records(rec, qp, init_id, data, 1)
    rc_o, bins_o, nread_o = orc.decode_records(rec, qp, init_id, data, 1)
    assert rc_r == 0 and rc_o == 0
    assert np.array_equal(bins_r, bins_o) and nread_r == nread_o
    keep = (rec & 0x1FF) != H.REC_ALIGN
    assert np.array_equal(bins_o[keep], (rec[keep] >> 15).astype(np.uint8))   # round trip
    rc_r, v_r = ref.decode_ops(ops, qp, init_id, data, 1)
    rc_o, v_o = orc.decode_ops(ops, qp, init_id, data, 1)
    assert rc_r == 0 and rc_o == 0 and np.array_equal(v_r, v_o)
    # decoded symbols equal the encoded ones for every op kind
    want = ops[:, 1].copy()
    want[ops[:, 0] == H.OP_ALIGN] = 0
    assert np.array_equal(v_o, want)


def test_decode_error_paths():
    ref, orc = H.load_ref(), H.load_oracle()
    rng = np.random.default_rng(5)
    rec = H.random_records(rng, 4000)
    data, _ = ref.encode_records(rec, 32, 2, 3)
    # truncated input -> reference throws "FIFO exceeded"; oracle reports -4
    rc_r, _, _ = ref.decode_records(rec, 32, 2, data[: len(data) // 2], 1)
    rc_o, _, _ = orc.decode_records(rec, 32, 2, data[: len(data) // 2], 1)
    assert rc_r == -1 and rc_o == -4
    # missing stop pattern -> reference finish() throws; oracle -5
    raw, _ = ref.encode_records(rec, 32, 2, 1)            # finish() but no writeByteAlignment
    padded = np.concatenate([raw, np.zeros(4, np.uint8)])
    rc_r, _, _ = ref.decode_records(rec, 32, 2, padded, 1)
    rc_o, _, _ = orc.decode_records(rec, 32, 2, padded, 1)
    assert (rc_r == -1) == (rc_o == -5)


def test_long_ff_runs_and_carry():
    """writeOut's delayed carry: long outstanding-0xFF runs resolved both ways
    (arith_codec.cpp:524-546, :339-357)."""
    ref, orc = H.load_ref(), H.load_oracle()
    rng = np.random.default_rng(77)
    hits_ff = 0
    for trial in range(60):
        # all-ones bypass bins at range close to 512 keep producing 0xFF lead bytes
        n = int(rng.integers(50, 400))
        ops = np.zeros((n + 3, 4), np.uint32)
        ops[:n] = (H.OP_EP, 1, 0, 0)
        ops[n] = (H.OP_BIN, int(rng.integers(0, 2)), int(rng.integers(0, 379)), 0)
        ops[n + 1] = (H.OP_BINS_EP, int(rng.integers(0, 1 << 16)), 16, 0)
        ops[n + 2] = (H.OP_TRM, 1, 0, 0)
        rb, rbits, _ = ref.encode_ops(ops, 30, 2, 3)
        ob, obits, _ = orc.encode_ops(ops, 30, 2, 3)
        assert rbits == obits and np.array_equal(rb, ob)
        hits_ff += int((rb == 0xFF).sum() > 8)
    assert hits_ff > 0


def test_count_start_code_emulations():
    import ctypes
    ref, orc = H.load_ref(), H.load_oracle()
    ref.lib.ref_count_emulations.argtypes = [H.u8p, ctypes.c_long]
    orc.lib.orc_count_emulations.argtypes = [H.u8p, ctypes.c_long]
    rng = np.random.default_rng(31)
    for trial in range(400):
        n = int(rng.integers(1, 200))
        data = rng.choice(np.array([0, 0, 0, 1, 2, 3, 4, 255], np.uint8), size=n)
        a = ref.lib.ref_count_emulations(H._ptr(data, H.u8p), n)
        b = orc.lib.orc_count_emulations(H._ptr(data, H.u8p), n)
        assert a == b, (trial, data.tolist())


def test_whole_batch_digests_agree():
    """bench.py's whole-batch hash: the reference's per-substream digests (ref_digest_mt), the restatement's (orc_digest_mt)
    and the digest of bytes in slots (orc_digest_slots, what is applied to the device's output) are the same numbers."""
    import ctypes
    ref, orc = H.load_ref(), H.load_oracle()
    rng = np.random.default_rng(31)
    recs = [H.random_records(rng, int(n)) for n in rng.integers(1, 3000, size=64)]
    desc, total = H.make_desc([len(r) for r in recs], rng.integers(0, 64, size=64), rng.integers(0, 3, size=64), H.SUB_FINISH | H.SUB_ALIGN_RBSP)
    records = np.concatenate(recs)
    out, res = orc.encode_batch(desc, records, total)
    a, b, c = np.zeros(64, np.uint64), np.zeros(64, np.uint64), np.zeros(64, np.uint64)
    for lib, fn, dst in ((ref.lib, "ref_digest_mt", a), (orc.lib, "orc_digest_mt", b)):
        f = getattr(lib, fn)
        f.restype = ctypes.c_uint64
        f.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        assert f(desc.ctypes.data, 0, 64, records.ctypes.data, 3, dst.ctypes.data) == 0
    g = orc.lib.orc_digest_slots
    g.restype = None
    g.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    g(desc.ctypes.data, res.ctypes.data, 64, out.ctypes.data, c.ctypes.data)
    assert np.array_equal(a, b) and np.array_equal(a, c) and len(set(a.tolist())) == 64
    out[int(desc["byte_offset"][5]) + 1] ^= 1
    g(desc.ctypes.data, res.ctypes.data, 64, out.ctypes.data, c.ctypes.data)
    assert (a != c).sum() == 1 and a[5] != c[5]
