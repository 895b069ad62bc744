"""CPU: the oracle (oracle/cabac_oracle.c) against the committed golden vectors, which were produced by
the reference's own compiled sources (oracle/gen_golden.py).  This is what pins the oracle on the GPU
box, where /root/reference does not exist."""
import hashlib
import json
import os

import numpy as np
import pytest

import helpers as H
from entropy_coding_amd import capi
from entropy_coding_amd.workload import CONFIGS


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(H.GOLDEN, "vectors.npz"))


def test_init_tables_header_matches_golden_blob():
    blob = open(os.path.join(H.GOLDEN, "ctx_init_tables.bin"), "rb").read()
    assert len(blob) == 4 * 379 and hashlib.md5(blob).hexdigest() == "96d432c564d403e474dd39432f771182"
    hdr = open(os.path.join(H.ROOT, "include", "cabac_ctx_tables.h")).read()
    body = hdr.split("#define CABAC_CTX_INIT_TABLE_VALUES")[1].split("#define CABAC_FRAC_BITS_TABLE_VALUES")[0]
    import re
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    vals = [int(x) for x in re.findall(r"\d+", body)]
    assert bytes(vals) == blob
    # the bit-estimator cost table in the same header, against its own golden blob
    fb = np.fromfile(os.path.join(H.GOLDEN, "frac_bits_table.bin"), "<u4")
    body = re.sub(r"/\*.*?\*/", "", hdr.split("#define CABAC_FRAC_BITS_TABLE_VALUES")[1], flags=re.S)
    assert [int(x, 16) for x in re.findall(r"0x[0-9a-f]+", body)] == fb.tolist()


def test_ctx_init(gold):
    orc = H.load_oracle()
    for i, qp in enumerate(gold["ctx_init_qps"]):
        for iid in range(3):
            s0, s1, rate = orc.ctx_init(int(qp), iid)
            g = gold["ctx_init"][i, iid]
            assert np.array_equal(s0, g[0]) and np.array_equal(s1, g[1]) and np.array_equal(rate, g[2].astype(np.uint8))
    # pinned facts from SURVEY.md §8c (QP32, I slice)
    s0, s1, rate = orc.ctx_init(32, 2)
    assert (s0[0], s1[0], rate[0]) == (9984, 9984, 0x58)
    assert (s0[1], s1[1], rate[1]) == (16640, 16640, 0x59)
    assert (s0[378], s1[378], rate[378]) == (29952, 29952, 0x47)


def test_ctx_traces(gold):
    orc = H.load_oracle()
    for meta, bins, want in zip(gold["trace_meta"], gold["trace_bins"], gold["trace_out"]):
        qp, iid, ctx, rg = [int(x) for x in meta]
        st, lps, a, b = orc.ctx_trace(qp, iid, ctx, bins, rg)
        assert np.array_equal(st, want[0]) and np.array_equal(lps, want[1])
        assert np.array_equal(a, want[2]) and np.array_equal(b, want[3])


def test_op_stream_cases(gold):
    orc = H.load_oracle()
    for k in range(int(gold["n_cases"][0])):
        ops = gold["case%d_ops" % k]
        qp, iid, nbits3, nbits1 = [int(x) for x in gold["case%d_meta" % k]]
        b3, n3, nbins = orc.encode_ops(ops, qp, iid, 3)
        b1, n1, _ = orc.encode_ops(ops, qp, iid, 1)
        assert n3 == nbits3 and np.array_equal(b3, gold["case%d_bytes_aligned" % k])
        assert n1 == nbits1 and np.array_equal(b1, gold["case%d_bytes_finish" % k])
        assert np.array_equal(nbins, gold["case%d_nbins" % k])
        rc, vals = orc.decode_ops(ops, qp, iid, b3, 1)
        assert rc == 0 and np.array_equal(vals, gold["case%d_values" % k])
        # the flattened record stream gives the same bytes, and decodes back to the recorded bins
        rec = orc.ops_to_records(ops)
        fb, fn = orc.encode_records(rec, qp, iid, 3)
        assert fn == nbits3 and np.array_equal(fb, b3)
        rc, bins, _ = orc.decode_records(rec, qp, iid, b3, 1)
        keep = (rec & 0x1FF) != H.REC_ALIGN
        assert rc == 0 and np.array_equal(bins[keep], (rec[keep] >> 15).astype(np.uint8))


@pytest.mark.parametrize("name", ["C1", "C2", "C3", "C4", "C5"])
def test_synthetic_config_md5(name):
    orc = H.load_oracle()
    gold = json.load(open(os.path.join(H.GOLDEN, "synth_md5.json")))[name]
    cfg = CONFIGS[name]
    assert gold["seed"] == cfg.seed
    cat = hashlib.md5()
    for g in gold["substreams"]:
        n, permille, qp = cfg.substream(g["index"])
        assert n == g["n_records"]
        rec = capi.synth_records(cfg.seed, g["index"], n, permille)
        assert hashlib.md5(rec.tobytes()).hexdigest() == g["records_md5"]  # generator is pinned too
        b, nbits = orc.encode_records(rec, qp, 2, 3)
        assert nbits == g["n_bits"] and hashlib.md5(b.tobytes()).hexdigest() == g["md5"]
        cat.update(b.tobytes())
    assert cat.hexdigest() == gold["concat_md5"]
