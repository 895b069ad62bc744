"""Differential fuzz of the residual binariser and parser against the oracle (GPU box; run by hand, not collected):

    python3 tests/fuzz_residual.py --seconds 300 [--seed 1]

A round is what tests/test_gpu_residual.py::test_many_random_blocks_of_every_kind and
tests/test_gpu_residual_parse.py::test_transform_skip_and_regular_blocks_mixed / ..._with_sign_hiding do once with a fixed
seed, with a fresh seed: a batch of random blocks of every shape and kind through the binariser (records, last position,
MTS violation == the oracle's), then random substreams of regular and transform-skip blocks through the parser
(coefficients, bit counts and per-block info == the oracle's parser)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helpers as H  # noqa: E402
import test_gpu_residual as B  # noqa: E402
import test_gpu_residual_parse as P  # noqa: E402
from entropy_coding_amd import capi  # noqa: E402


def binariser_round(hip, rng, n_blocks):
    shapes = B.SIZES
    weights = np.array([1.0 / (1 + (w * h) / 64.0) for w, h in shapes])
    weights /= weights.sum()
    blocks, chromas, flags = [], [], []
    for _ in range(n_blocks):
        w, h = shapes[int(rng.choice(len(shapes), p=weights))]
        if max(w, h) <= 32 and rng.random() < 0.25:
            blocks.append(B._ts_block(rng, w, h, int(rng.integers(0, 4))))
            flags.append(H.TU_TRANSFORM_SKIP | int(rng.choice([H.TU_TS_FLAG, H.TU_BDPCM, 0])) | int(rng.integers(0, 4)))
        else:
            blocks.append(H.random_block(rng, w, h, density=float(rng.choice([0.02, 0.05, 0.3, 0.7, 1.0])), big=float(rng.choice([0.0, 0.05, 0.3])),
                                         huge=0.02 if rng.random() < 0.1 else 0.0, last_frac=float(rng.choice([1.0, 0.5, 0.2, 0.05]))))
            fl = int(rng.integers(0, 8))
            fl = fl & ~H.TU_TS_FLAG if max(w, h) > 32 else fl
            if max(w, h) <= 32 and rng.random() < 0.2:       # SBT / MTS zero-out: what lies outside the left / upper 16 is zero
                fl |= H.TU_SBT_ZERO_OUT
                c = blocks[-1]
                if w == 32:
                    c[:, 16:] = 0
                if h == 32:
                    c[16:, :] = 0
                if not c.any():
                    c[0, 0] = 3
            flags.append(fl)
        chromas.append(int(rng.integers(0, 2)))
    B.check_against_oracle(hip, blocks, chromas, flags, slack=int(rng.integers(0, 2)))
    return sum(b.size for b in blocks)


def parser_round(hip, rng, n_sub):
    orc = H.load_oracle()
    qps = rng.integers(0, 64, n_sub)
    if rng.random() < 0.5:
        subs = P.build_mixed(rng, n_sub, qps)
        exact = True
    else:
        fl = int(rng.choice([0, H.TU_DEP_QUANT, H.TU_SIGN_HIDING, H.TU_SIGN_HIDING | H.TU_DEP_QUANT]))
        subs = P.build(rng, n_sub, lambda s: fl, qps)
        exact = not (fl & H.TU_SIGN_HIDING)   # random blocks are not arranged for sign hiding: the oracle's parse is the truth
    hip.parse_int16 = bool(rng.integers(0, 2))   # the blocks stored as int16 (cabac_hip_residual_parse16_device) or as int32
    got, res = P.parse(hip, subs, qps)
    info = P.parse.last_info
    assert not res["flags"].any()
    t = n = 0
    for s, (metas, blocks, data) in enumerate(subs):
        rc, want, nbits, winfo = orc.residual_decode(data, int(qps[s]), metas, with_info=True)
        assert rc == 0 and int(res["n_bits"][s]) == nbits, s
        for k, c in enumerate(blocks):
            we, he = min(metas[k][0], 32), min(metas[k][1], 32)
            assert np.array_equal(got[s][k][:he, :we], want[k][:he, :we]), (s, k, metas[k])
            if exact:
                assert np.array_equal(got[s][k][:he, :we], c[:he, :we]), (s, k, metas[k])
            assert int(info[t]) == int(winfo[k]), (s, k, metas[k])
            t += 1
            n += c.size
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    print(run(ap.parse_args()))


def run(a):
    """a.seconds / a.seed; returns the summary line, raises SystemExit("MISMATCH ...") on the first difference."""
    hip = H.gpu_ctx()
    t0, rounds, coeffs = time.time(), 0, 0
    while time.time() - t0 < a.seconds:
        seed = a.seed * 1_000_003 + rounds
        rng = np.random.default_rng(seed)
        try:
            coeffs += binariser_round(hip, rng, int(rng.choice([50, 700, 4000])))
            coeffs += parser_round(hip, rng, int(rng.choice([3, 40, 130])))
        except AssertionError as e:
            raise SystemExit("MISMATCH in round %d (seed %d): %r" % (rounds, seed, e.args))
        rounds += 1
        if rounds % 5 == 0:
            print("%d rounds, %.1f M coefficients, %.0f s" % (rounds, coeffs / 1e6, time.time() - t0), flush=True)
    hip.close()
    return "fuzz ok: %d rounds (binariser batch + parser batch each), %.1f M coefficients in %.0f s, seed %d" % (rounds, coeffs / 1e6, time.time() - t0, a.seed)


if __name__ == "__main__":
    main()
