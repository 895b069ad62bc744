"""Bit estimator: the oracle's restatement of BitEstimator_Std against the reference's own
compiled sources (build container) and against golden vectors generated from them (everywhere)."""
import os

import numpy as np
import pytest

import helpers as H

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
needs_ref = pytest.mark.skipif(not H.ref_available(), reason="compiled reference (oracle/_ref) not present")


def test_frac_bits_table_is_minus_log2_p():
    """The cost table pinned as data: 256 x {bin 0, bin 1} = -log2(p) * 2^15 at the state's midpoint."""
    t = np.fromfile(os.path.join(GOLDEN, "frac_bits_table.bin"), "<u4").reshape(256, 2)
    q = (np.arange(256) + 0.5) / 256.0
    assert np.all(np.abs(t[:, 1] - (-np.log2(q)) * 32768.0) <= 1.0)
    assert np.all(np.abs(t[:, 0] - (-np.log2(1.0 - q)) * 32768.0) <= 1.0)
    assert np.array_equal(t[:, 0], t[::-1, 1])  # symmetric


@needs_ref
@pytest.mark.parametrize("seed", range(6))
def test_estimate_ops_matches_reference(seed):
    rng = np.random.default_rng(9000 + seed)
    orc, ref = H.load_oracle(), H.load_ref()
    for _ in range(12):
        n = int(rng.integers(0, 600))
        ops = H.random_ops(rng, n, ctx_frac=float(rng.choice([0.0, 0.4, 0.7, 0.95])), with_align=True)
        qp, iid = int(rng.integers(-3, 70)), int(rng.integers(0, 3))
        assert orc.estimate_ops(ops, qp, iid) == ref.estimate_ops(ops, qp, iid)


@needs_ref
def test_estimate_records_matches_reference():
    rng = np.random.default_rng(9100)
    orc, ref = H.load_oracle(), H.load_ref()
    for n in [0, 1, 2, 17, 64, 1000, 20000]:
        for frac in (0.0, 0.5, 1.0):
            rec = H.random_records(rng, n, ctx_frac=frac)
            qp, iid = int(rng.integers(0, 64)), int(rng.integers(0, 3))
            assert orc.estimate_records(rec, qp, iid) == ref.estimate_records(rec, qp, iid)
    # one context hammered (state saturates both ways), and align records in between
    rec = np.array([7 | 0x8000] * 300 + [0x1FD] + [7] * 300 + [0x1FE, 0x1FD, 0x81FF], np.uint16)
    assert orc.estimate_records(rec, 30, 2) == ref.estimate_records(rec, 30, 2)
    # resetBits() / start() (cost := 0) and restart() (cost rounded down to a whole bit) in between; contexts carry on
    for seed in range(4):
        r2 = np.random.default_rng(9200 + seed)
        rec = H.random_records(r2, 3000, ctx_frac=0.7)
        pos = r2.integers(0, 3000, size=40)
        rec[pos[:15]] = 0x1FC
        rec[pos[15:30]] = 0x1FB
        rec[pos[30:]] = 0x1FD
        assert orc.estimate_records(rec, 27, seed % 3) == ref.estimate_records(rec, 27, seed % 3)
    bad = np.array([3, 0x1F0, 4], np.uint16)
    assert orc.estimate_records(bad, 30, 2)[0] == -2 and ref.estimate_records(bad, 30, 2)[0] == -2


def test_estimate_golden():
    g = np.load(os.path.join(GOLDEN, "vectors.npz"))
    orc = H.load_oracle()
    n_cases = int(g["est_n_cases"])
    assert n_cases >= 8
    for k in range(n_cases):
        qp, iid = (int(x) for x in g["est%d_meta" % k])
        rc, bits = orc.estimate_ops(g["est%d_ops" % k], qp, iid)
        assert rc == 0 and bits == int(g["est%d_bits" % k]), k
        rec = orc.ops_to_records(g["est%d_ops" % k])
        assert orc.estimate_records(rec, qp, iid) == (0, bits)


@needs_ref
def test_estimate_from_given_contexts_matches_reference():
    """RDO use: the estimator's contexts assigned from a coder that has already adapted (Ctx::operator=), then
    resetBits() and the candidate string.  The oracle starts from the dumped states."""
    rng = np.random.default_rng(9300)
    orc, ref = H.load_oracle(), H.load_ref()
    for _ in range(10):
        hist = H.random_records(rng, int(rng.integers(0, 5000)), ctx_frac=0.8, end_trm=False)
        rec = H.random_records(rng, int(rng.integers(0, 600)), ctx_frac=0.7)
        qp, iid = int(rng.integers(0, 64)), int(rng.integers(0, 3))
        rc, bits, s0, s1, rate = ref.estimate_from_history(hist, rec, qp, iid)
        assert rc == 0
        assert orc.estimate_records_from(rec, s0, s1, rate) == (0, bits)
        # the states are those the oracle's own update reaches
        assert orc.estimate_records(np.concatenate([hist, np.array([0x1FC], np.uint16), rec]), qp, iid) == (0, bits)
