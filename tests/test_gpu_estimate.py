"""Bit estimator on the GPU (cabac_hip_estimate_device) against the oracle (pinned to the reference's
BitEstimator_Std by tests/test_estimator_oracle.py): bit-exact fractional-bit totals."""
import os

import numpy as np
import pytest

import helpers as H
from entropy_coding_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    return H.gpu_ctx()


def _batch(rng, lens, ctx_fracs, with_align=True):
    recs = []
    for n, f in zip(lens, ctx_fracs):
        r = H.random_records(rng, n - 1, ctx_frac=f) if n > 0 else np.zeros(0, np.uint16)
        if with_align and n > 40:  # sprinkle align() / resetBits() / restart(): they act on the running total
            pos = rng.integers(0, n - 1, size=max(3, n // 100))
            r[pos] = rng.choice([0x1FD, 0x1FC, 0x1FB], size=len(pos))
        recs.append(r)
    records = np.concatenate(recs) if recs else np.zeros(0, np.uint16)
    lens = [len(r) for r in recs]
    desc, _ = H.make_desc(lens, rng.integers(0, 64, size=len(lens)), rng.integers(0, 3, size=len(lens)), H.SUB_FINISH)
    return desc, records


@pytest.mark.parametrize("seed", range(3))
def test_estimate_random_batches(hip, seed):
    rng = np.random.default_rng(7700 + seed)
    lens = [0, 1, 2, 15, 16, 17, 63, 64, 65, 257, 1000, 5000] + [int(x) for x in rng.integers(0, 3000, size=37)]
    fracs = [float(rng.choice([0.0, 0.5, 0.75, 1.0])) for _ in lens]
    desc, records = _batch(rng, lens, fracs)
    want, wflags = H.load_oracle().estimate_batch(desc, records)
    got, gflags = hip.estimate_batch(desc, records)
    assert np.array_equal(gflags, wflags) and not gflags.any()
    assert np.array_equal(got, want)


def test_estimate_single_context_and_saturation(hip):
    # one context hammered in both directions (the state saturates), 16 identical contexts per step
    rec = np.array([7 | 0x8000] * 1000 + [7] * 1000 + [0x1FD] + [300] * 77 + [0x81FE] * 5 + [0x81FF], np.uint16)
    desc, _ = H.make_desc([len(rec)], [30], [2], H.SUB_FINISH)
    want, _ = H.load_oracle().estimate_batch(desc, rec)
    got, flags = hip.estimate_batch(desc, rec)
    assert not flags.any() and np.array_equal(got, want)


def test_estimate_bad_record_flag(hip):
    rng = np.random.default_rng(7800)
    lens = [100, 100, 100, 100, 100]
    desc, records = _batch(rng, lens, [0.5] * 5, with_align=False)
    records = records.copy()
    records[int(desc["rec_offset"][2]) + 40] = 0x1F0  # neither a ctxId nor EP / TRM / align
    want, wflags = H.load_oracle().estimate_batch(desc, records)
    got, gflags = hip.estimate_batch(desc, records)
    assert np.array_equal(gflags, wflags) and gflags[2] == capi.RES_BAD_RECORD
    ok = gflags == 0
    assert np.array_equal(got[ok], want[ok])


def test_estimate_golden(hip):
    """The reference's own numbers (tests/golden/vectors.npz, generated from the compiled reference)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vectors.npz"))
    orc = H.load_oracle()
    recs, metas = [], []
    for k in range(int(g["est_n_cases"])):
        recs.append(orc.ops_to_records(g["est%d_ops" % k]))
        metas.append((int(g["est%d_meta" % k][0]), int(g["est%d_meta" % k][1]), int(g["est%d_bits" % k])))
    desc, _ = H.make_desc([len(r) for r in recs], [m[0] for m in metas], [m[1] for m in metas], H.SUB_FINISH)
    got, flags = hip.estimate_batch(desc, np.concatenate(recs))
    assert not flags.any() and got.tolist() == [m[2] for m in metas]


def test_estimate_full_size_additivity(hip):
    """C4-sized batch: the sum over substreams equals the oracle's on a sample, and a substream split in two
    halves (the second started from fresh contexts) is NOT additive unless contexts carry over — checked the
    other way: bypass-only strings cost exactly one bit per bin."""
    from entropy_coding_amd.workload import CONFIGS, build_batch
    desc, records, _ = build_batch(CONFIGS["C4"], first=0, count=512)
    got, flags = hip.estimate_batch(desc, records)
    assert not flags.any()
    sample = [0, 1, 255, 511]
    want, _ = H.load_oracle().estimate_batch(desc[sample].copy(), records)
    assert np.array_equal(got[sample], want)
    ep = np.full(4096, 0x81FE, np.uint16)
    d1, _ = H.make_desc([4096], [32], [2], H.SUB_FINISH)
    b, _ = hip.estimate_batch(d1, ep)
    assert int(b[0]) == 4096 << 15


def test_estimate_from_given_contexts(hip):
    """cabac_hip_estimate_from_device: many candidate strings sharing a few start states (contexts reached by coding
    a history), against the oracle started from the same arrays."""
    import torch
    rng = np.random.default_rng(7900)
    orc = H.load_oracle()
    n_sets, n_cand = 5, 300
    sets = []
    for k in range(n_sets):  # a start state = the contexts after a history (oracle: reset + history, then dump via trace)
        hist = H.random_records(rng, int(rng.integers(100, 4000)), ctx_frac=0.9, end_trm=False)
        qp, iid = int(rng.integers(0, 64)), int(rng.integers(0, 3))
        s0, s1, rate = orc.ctx_init(qp, iid)
        s0, s1 = s0.astype(np.int64), s1.astype(np.int64)
        for r in hist:  # update(), contexts.cpp:903-913
            i, b = int(r) & 0x1FF, int(r) >> 15
            if i < 379:
                r0, r1 = int(rate[i]) >> 4, int(rate[i]) & 15
                s0[i] -= (s0[i] >> r0) & 0x7FE0
                s1[i] -= (s1[i] >> r1) & 0x7FFE
                if b:
                    s0[i] += (0x7FFF >> r0) & 0x7FE0
                    s1[i] += (0x7FFF >> r1) & 0x7FFE
        sets.append((s0.astype(np.uint16), s1.astype(np.uint16), rate))
    recs = [H.random_records(rng, int(rng.integers(0, 300)), ctx_frac=0.75) for _ in range(n_cand)]
    which = rng.integers(0, n_sets, size=n_cand).astype(np.uint32)
    desc, _ = H.make_desc([len(r) for r in recs], [0] * n_cand, [0] * n_cand, H.SUB_FINISH)
    records = np.concatenate(recs + [np.zeros(1, np.uint16)])
    state = np.concatenate([(s[0].astype(np.uint32) | (s[1].astype(np.uint32) << 16)) for s in sets])
    rate = np.concatenate([s[2] for s in sets]).astype(np.uint8)
    dev = "cuda:0"
    t = lambda a, dt: torch.from_numpy(a.view(dt).copy()).to(dev)
    t_desc, t_rec = t(desc.view(np.uint8).reshape(-1), np.uint8), t(records, np.int16)
    t_state, t_rate, t_set = t(state, np.int32), t(rate, np.uint8), t(which, np.int32)
    t_bits = torch.zeros(n_cand, dtype=torch.int64, device=dev)
    t_flags = torch.zeros(n_cand, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    hip.estimate_from_device(n_cand, t_desc.data_ptr(), t_rec.data_ptr(), t_state.data_ptr(), t_rate.data_ptr(),
                             t_set.data_ptr(), t_bits.data_ptr(), t_flags.data_ptr())
    hip.synchronize()
    got = t_bits.cpu().numpy().view(np.uint64)
    assert not t_flags.cpu().numpy().any()
    for i in range(n_cand):
        s0, s1, rt = sets[int(which[i])]
        assert orc.estimate_records_from(recs[i], s0, s1, rt) == (0, int(got[i])), i
