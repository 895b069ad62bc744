"""Differential fuzz of the HIP bin codec against the oracle (GPU box; tests/test_gpu_fuzz.py runs 20 s of it under -m gpu, longer runs by hand).

    python3 tests/fuzz_parity.py --seconds 300 [--seed 1] > gpurun_out/fuzz.txt

Each round draws a batch shape the unit tests do not pin down — substream counts on both sides of the kernels' geometry
switches (1 024 / 3 072 substreams), lengths from 0 to tens of thousands of bins in one batch, context pools of one to
all 379 contexts (a pool of one or two contexts makes every bin depend on the bin before it), probabilities down to
1/1000 (long MPS runs: 0xFF runs and carries), align and terminate records in the middle, capacities cut short (the
overflow flag) — and sends it through encode (every encoder variant), decode and the estimator.  Bytes, bit counts,
flags, bins and bit costs must equal the oracle's.  The first mismatch stops the run with the seed of its round."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helpers as H  # noqa: E402
from entropy_coding_amd import capi  # noqa: E402


def draw_batch(rng):
    shape = rng.integers(0, 6)
    if shape == 0:
        n_sub = int(rng.integers(1, 40))
    elif shape == 1:
        n_sub = int(rng.choice([63, 64, 65, 255, 256, 257]))
    elif shape == 2:
        n_sub = int(rng.choice([1023, 1024, 1025, 1100]))
    elif shape == 3:
        n_sub = int(rng.choice([3071, 3072, 3073, 3200]))
    else:
        n_sub = int(rng.integers(40, 700))
    budget = 2_000_000
    top = int(rng.choice([40, 300, 3000, 30000]))
    lens = rng.integers(0, top + 1, size=n_sub)
    if rng.random() < 0.3:
        lens[rng.integers(0, n_sub, size=max(1, n_sub // 8))] = 0
    if lens.sum() > budget:
        lens = (lens * (budget / lens.sum())).astype(np.int64)
    pool_kind = rng.integers(0, 4)
    pool = (np.arange(H.NUM_CTX) if pool_kind == 0 else
            rng.choice(H.NUM_CTX, size=int(rng.choice([1, 2, 3, 6, 17])), replace=False) if pool_kind == 1 else
            np.arange(90, 246) if pool_kind == 2 else rng.choice(H.NUM_CTX, size=40, replace=False))
    p_kind = rng.integers(0, 3)
    p_one = (rng.choice([0.03, 0.1, 0.25, 0.5, 0.75, 0.9], size=H.NUM_CTX) if p_kind == 0 else
             rng.choice([0.001, 0.999, 0.01, 0.5], size=H.NUM_CTX) if p_kind == 1 else rng.random(H.NUM_CTX))
    ctx_frac = float(rng.choice([0.0, 0.3, 0.75, 0.9, 1.0]))
    trm0 = float(rng.choice([0.0, 0.002, 0.05]))
    recs = []
    for n in lens:
        n = int(n)
        r = H.random_records(rng, max(n - 1, 0), ctx_frac=ctx_frac, p_one=p_one, ctx_pool=pool, end_trm=n > 0, trm0_frac=trm0)
        recs.append(r)
    if rng.random() < 0.3:   # align records in the middle (their bin bit is ignored)
        for r in recs:
            if len(r) > 2:
                at = rng.integers(0, len(r) - 1, size=max(1, len(r) // 200))
                r[at] = H.REC_ALIGN
    lens = np.array([len(r) for r in recs], np.uint64)
    records = np.concatenate(recs) if len(recs) else np.zeros(0, np.uint16)
    caps = None
    if rng.random() < 0.15:  # some substreams do not fit: flags must agree, the others' bytes too
        caps = (lens * 3) // 4 + 64
        cut = rng.integers(0, n_sub, size=max(1, n_sub // 10))
        caps[cut] = lens[cut] // 16
    flags = H.SUB_FINISH | (H.SUB_ALIGN_RBSP if rng.random() < 0.5 else 0)
    desc, total = H.make_desc(lens, rng.integers(-3, 67, size=n_sub), rng.integers(0, 3, size=n_sub), flags, capacities=caps)
    return desc, records, total


def stream(out, desc, res, s):
    o = int(desc["byte_offset"][s])
    return out[o:o + (int(res["n_bits"][s]) + 7) // 8]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    print(run(ap.parse_args()))


def run(a):
    """a.seconds / a.seed; returns the summary line, raises SystemExit("MISMATCH ...") on the first difference."""
    orc = H.load_oracle()
    hip = H.gpu_ctx()
    t0, rounds, bins = time.time(), 0, 0
    shapes = {}
    while time.time() - t0 < a.seconds:
        seed = a.seed * 1_000_003 + rounds
        rng = np.random.default_rng(seed)
        desc, records, total = draw_batch(rng)
        enc = int(rng.choice([0, 4, 6, 7]))
        hip.set_variant(enc, int(rng.choice([0, 4, 8, 1])))
        out_o, res_o = orc.encode_batch(desc, records, total)
        out_g, res_g = hip.encode_batch(desc, records, total, check=False)
        what = "round %d (seed %d, %d substreams, encoder %d)" % (rounds, seed, len(desc), enc)
        if not (np.array_equal(res_g["n_bits"], res_o["n_bits"]) and np.array_equal(res_g["flags"], res_o["flags"])):
            raise SystemExit("MISMATCH encode results, " + what)
        good = np.flatnonzero(res_o["flags"] == 0)
        for s in good:
            if not np.array_equal(stream(out_g, desc, res_g, s), stream(out_o, desc, res_o, s)):
                raise SystemExit("MISMATCH encode bytes of substream %d, %s" % (s, what))
        # decode what fitted: from exactly its valid bytes if it ends with the RBSP stop bit, else with the two (zero)
        # bytes behind it that the decoder's read-ahead wants (arith_codec.cpp:60-66; without them: the underrun flag)
        ddesc = desc.copy()
        used = (res_o["n_bits"] + 7) // 8
        if not (int(desc["init_id"][0]) & H.SUB_ALIGN_RBSP):
            used = np.minimum(used + 2, desc["byte_capacity"])
        ddesc["byte_capacity"] = np.where(res_o["flags"] == 0, used, 0)
        ddesc["n_records"] = np.where(res_o["flags"] == 0, desc["n_records"], 0)
        bins_g, dres_g = hip.decode_batch(ddesc, records, out_o, check=False)
        bins_o, dres_o = orc.decode_batch(ddesc, records, out_o)
        ran = (dres_o["flags"] & H.RES_UNDERRUN) == 0   # (after an underrun the reference has thrown: n_bits and bins unspecified)
        if not (np.array_equal(dres_g["n_bits"][ran], dres_o["n_bits"][ran]) and np.array_equal(dres_g["flags"], dres_o["flags"])):
            raise SystemExit("MISMATCH decode results, " + what)
        for s in np.flatnonzero((res_o["flags"] == 0) & ((dres_o["flags"] & np.uint32(0xFFFFFFFF ^ H.RES_BAD_STOP)) == 0)):
            lo, n = int(desc["rec_offset"][s]), int(desc["n_records"][s])
            keep = (records[lo:lo + n] & 0x1FF) != H.REC_ALIGN   # an align record has no bin
            if not np.array_equal(bins_g[lo:lo + n][keep], bins_o[lo:lo + n][keep]):
                raise SystemExit("MISMATCH decoded bins of substream %d, %s" % (s, what))
        bits_g, fl_g = hip.estimate_batch(desc, records)
        bits_o, fl_o = orc.estimate_batch(desc, records)
        if not (np.array_equal(bits_g[:len(desc)], bits_o) and np.array_equal(fl_g[:len(desc)], fl_o)):
            raise SystemExit("MISMATCH estimate, " + what)
        rounds += 1
        bins += len(records)
        key = "%d substreams" % (1 << int(np.ceil(np.log2(max(len(desc), 1)))))
        shapes[key] = shapes.get(key, 0) + 1
        if rounds % 50 == 0:
            print("%d rounds, %.1f M bins, %.0f s" % (rounds, bins / 1e6, time.time() - t0), flush=True)
    hip.close()
    return ("fuzz ok: %d rounds, %.1f M bins in %.0f s, seed %d; rounds by batch size (up to): %s"
            % (rounds, bins / 1e6, time.time() - t0, a.seed, dict(sorted(shapes.items(), key=lambda kv: int(kv[0].split()[0])))))


if __name__ == "__main__":
    main()
