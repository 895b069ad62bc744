"""GPU: the device binariser (syntax-element records -> bin records) against the oracle's expander
(oracle/cabac_oracle.c: orc_ops_to_records, itself pinned to the reference's CABACWriter helpers by
tests/test_oracle_vs_reference.py), then through the encoder to the reference-generated golden bytes."""
import os

import numpy as np
import pytest

import helpers as H
from entropy_coding_amd import capi

pytestmark = pytest.mark.gpu


def ops_to_se(ops):
    """Operation stream (oracle format) -> syntax-element records (include/cabac_hip.h)."""
    ops = np.asarray(ops, np.uint64).reshape(-1, 4)
    se = np.zeros((len(ops), 2), np.uint32)
    for i, (code, a, b, c) in enumerate(ops):
        code, a, b, c = int(code), int(a), int(b), int(c)
        if code == H.OP_BIN:
            se[i] = (0 | (b << 4), a)
        elif code == H.OP_EP:
            se[i] = (1 | (1 << 4), a)
        elif code == H.OP_BINS_EP:
            se[i] = (1 | (b << 4), a)
        elif code == H.OP_REM_ABS:
            se[i] = (2 | (b << 4) | ((c & 0xff) << 9) | ((c >> 8) << 14), a)
        elif code == H.OP_TRM:
            se[i] = (3, a)
        elif code == H.OP_ALIGN:
            se[i] = (8, 0)
        elif code == H.OP_UNARY_MAX:
            se[i] = (4 | ((b & 0xffff) << 4) | ((b >> 16) << 13) | (c << 22), a)
        elif code == H.OP_UNARY_EP:
            se[i] = (5 | (b << 4), a)
        elif code == H.OP_EXP_GOLOMB:
            se[i] = (6 | (b << 4), a)
        elif code == H.OP_TRUNC_BIN:
            se[i] = (7 | (b << 4), a)
    return se


def binarize(hip, se_list):
    import torch
    n = len(se_list)
    se = np.concatenate(se_list) if n else np.zeros((0, 2), np.uint32)
    off = np.concatenate([[0], np.cumsum([len(s) for s in se_list])]).astype(np.uint64)
    t_se = torch.from_numpy(se.view(np.int32).reshape(-1)).cuda() if len(se) else torch.zeros(2, dtype=torch.int32, device="cuda")
    t_off = torch.from_numpy(off.view(np.int64)).cuda()
    t_cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    hip.binarize_device(n, t_off.data_ptr(), t_se.data_ptr(), 0, t_cnt.data_ptr(), 0)      # pass 1: sizes
    hip.synchronize()
    cnt = t_cnt.cpu().numpy().view(np.uint32).astype(np.uint64)
    roff = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.uint64)
    t_roff = torch.from_numpy(roff.view(np.int64)).cuda()
    t_rec = torch.zeros(max(int(cnt.sum()), 1), dtype=torch.int16, device="cuda")
    hip.binarize_device(n, t_off.data_ptr(), t_se.data_ptr(), t_roff.data_ptr(), t_cnt.data_ptr(), t_rec.data_ptr())
    hip.synchronize()
    rec = t_rec.cpu().numpy().view(np.uint16)
    return [rec[int(roff[s]):int(roff[s] + cnt[s])] for s in range(n)], cnt


@pytest.fixture(scope="module")
def hip():
    c = H.gpu_ctx()
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(4))
def test_binarize_matches_oracle(hip, seed):
    orc = H.load_oracle()
    rng = np.random.default_rng(800 + seed)
    op_list = [H.random_ops(rng, int(n), ctx_frac=float(rng.choice([0.1, 0.5, 0.8])), with_align=(seed == 3))
               for n in [0, 1, 2, 255, 256, 257, 1000] + [int(x) for x in rng.integers(1, 3000, size=20)]]
    recs, cnt = binarize(hip, [ops_to_se(o) for o in op_list])
    for s, ops in enumerate(op_list):
        want = orc.ops_to_records(ops)
        assert int(cnt[s]) == len(want) and np.array_equal(recs[s], want), s


def test_binarize_then_encode_reproduces_reference_bytes(hip):
    """SE records -> device binariser -> device encoder == the reference's own bytes (golden vectors)."""
    gold = np.load(os.path.join(H.GOLDEN, "vectors.npz"))
    ks = list(range(int(gold["n_cases"][0])))
    recs, cnt = binarize(hip, [ops_to_se(gold["case%d_ops" % k]) for k in ks])
    metas = [[int(x) for x in gold["case%d_meta" % k]] for k in ks]
    records = np.concatenate(recs)
    desc, total = H.make_desc([len(r) for r in recs], [m[0] for m in metas], [m[1] for m in metas],
                              H.SUB_FINISH | H.SUB_ALIGN_RBSP)
    out, res = hip.encode_batch(desc, records, total)
    for i, k in enumerate(ks):
        nb = (int(res["n_bits"][i]) + 7) // 8
        o = int(desc["byte_offset"][i])
        assert int(res["n_bits"][i]) == metas[i][2] and np.array_equal(out[o:o + nb], gold["case%d_bytes_aligned" % k]), k


def test_binarize_edge_values(hip):
    orc = H.load_oracle()
    ops = np.array([
        [H.OP_BINS_EP, 0xFFFFFFFF, 32, 0], [H.OP_BINS_EP, 0, 0, 0], [H.OP_UNARY_EP, 31, 31, 0], [H.OP_UNARY_EP, 0, 0, 0],
        [H.OP_UNARY_EP, 5, 31, 0], [H.OP_REM_ABS, 32767, 0, 5 | (15 << 8)], [H.OP_REM_ABS, 4099, 0, 5 | (15 << 8)],
        [H.OP_REM_ABS, 0, 3, 5 | (15 << 8)], [H.OP_REM_ABS, 100, 1, 0 | (17 << 8)], [H.OP_EXP_GOLOMB, 0, 0, 0],
        [H.OP_EXP_GOLOMB, 100000, 2, 0], [H.OP_TRUNC_BIN, 0, 1, 0], [H.OP_TRUNC_BIN, 699, 700, 0], [H.OP_TRUNC_BIN, 255, 256, 0],
        [H.OP_UNARY_MAX, 0, 5 | (9 << 16), 0], [H.OP_UNARY_MAX, 7, 5 | (9 << 16), 7], [H.OP_UNARY_MAX, 3, 378 | (377 << 16), 200],
        [H.OP_TRM, 0, 0, 0], [H.OP_ALIGN, 0, 0, 0], [H.OP_BIN, 1, 378, 0], [H.OP_TRM, 1, 0, 0]], np.uint32)
    recs, cnt = binarize(hip, [ops_to_se(ops)])
    assert np.array_equal(recs[0], orc.ops_to_records(ops))
