"""Child process of tests/test_bin_log_parity.py (a process of its own, so that the ENABLE_LOGGING build of the
reference is the only copy of the reference loaded and bin_log.txt is created in a scratch directory).

usage: bin_log_walk.py <mode> <out.json>     mode: cpu | gpu
Drives integration/reference_adapter_test.cpp::adapter_walk — the reference's own CABACWriter over several substreams of
mvd_coding / cu_qp_delta / cu_chroma_qp_offset / residual_coding items — on BinEncoder_Std (which 0), on BinEncoderHipRef
recording only (which 2, its records then coded by the oracle) and, in gpu mode, on BinEncoderHipRef with the substreams
coded on the device (which 1); reports the bin_log.txt segment and the bytes of every run."""
import ctypes
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import helpers as H  # noqa: E402


def build_walk(seed=0xB1A106, n_sub=6):
    rng = np.random.default_rng(seed)
    items, coeffs, first, qps = [], [], [0], []
    for s in range(n_sub):
        for k in range(int(rng.integers(3, 10))):
            kind = int(rng.integers(0, 4))
            comp = int(rng.integers(0, 3))
            slice_fl = int(rng.integers(0, 4))                      # dep quant | sign hiding
            if kind < 3:
                w, h = [(4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (4, 16), (2, 8), (64, 32)][int(rng.integers(0, 8))]
                c = H.random_block(rng, w, h, density=float(rng.choice([0.1, 0.5, 1.0])), big=float(rng.choice([0.0, 0.2])))
                rig = slice_fl | (4 if kind == 2 and max(w, h) <= 32 else 0)       # kind 2: transform_skip_flag coded as 0
            else:
                w, h = int(rng.choice([2, 4, 8, 16])), int(rng.choice([2, 4, 8, 16]))
                c = (rng.integers(-6, 7, (h, w)) * (rng.random((h, w)) < 0.5)).astype(np.int32)
                if not c.any():
                    c[0, 0] = 3
                rig = (slice_fl & 1) | 4 | 0x10                                    # transform skip, flag coded as 1
            mvh, mvv = int(rng.integers(-40, 41)), int(rng.integers(-40, 41))
            pred = int(rng.integers(20, 40))
            items.append((w, h, comp, rig, mvh, mvv, pred, pred + int(rng.integers(-9, 10))))
            coeffs.append(c.ravel())
        first.append(len(items))
        qps.append(int(rng.integers(18, 45)))
    return (np.array(items, np.int32), np.concatenate(coeffs).astype(np.int32), np.array(first, np.int32),
            np.array(qps, np.int32))


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    so = os.path.join(H.ORACLE_DIR, "_ref", "libadapter_test_log.so")
    newest = max(os.path.getmtime(p) for p in H.HOST_ABI_DEPS if os.path.exists(p))
    if os.path.getmtime(so) < newest:      # (tests/helpers.py::ref_test_library says why this must not run)
        raise SystemExit("oracle/_ref/libadapter_test_log.so is older than the host headers it was compiled from: run `make -C oracle`")
    if mode == "gpu":
        from entropy_coding_amd import capi
        capi.load_library()                       # torch's HIP runtime first (see capi.load_library)
    scratch = tempfile.mkdtemp(prefix="bin_log_")
    os.chdir(scratch)                             # bin_log.txt / bit_log.txt are created where the library is loaded
    L = ctypes.CDLL(so)
    L.adapter_log_mark.restype = ctypes.c_long
    L.adapter_walk.restype = ctypes.c_long
    L.adapter_last_error.restype = ctypes.c_char_p
    items, coeff, first, qps = build_walk()
    n_sub = len(qps)
    lp = ctypes.POINTER(ctypes.c_long)

    def run(which):
        out = np.zeros(1 << 20, np.uint8)
        out_off = np.zeros(n_sub + 1, np.int64)
        nbits = np.zeros(n_sub, np.uint32)
        rec = np.zeros(1 << 21, np.uint16)
        rec_off = np.zeros(n_sub + 1, np.int64)
        m0 = L.adapter_log_mark()
        rc = L.adapter_walk(which, n_sub, first.ctypes.data_as(ctypes.c_void_p), qps.ctypes.data_as(ctypes.c_void_p),
                            items.ctypes.data_as(ctypes.c_void_p), coeff.ctypes.data_as(ctypes.c_void_p),
                            out.ctypes.data_as(ctypes.c_void_p), len(out), out_off.ctypes.data_as(lp),
                            nbits.ctypes.data_as(ctypes.c_void_p), rec.ctypes.data_as(ctypes.c_void_p), len(rec),
                            rec_off.ctypes.data_as(lp))
        assert rc == 0, (which, rc, L.adapter_last_error())
        m1 = L.adapter_log_mark()
        with open(os.path.join(scratch, "bin_log.txt"), "rb") as f:
            f.seek(m0)
            log = f.read(m1 - m0)
        if which == 2:                            # the recorded bins, coded on the CPU by the oracle
            orc = H.load_oracle()
            streams = []
            for s in range(n_sub):
                b, nb = orc.encode_records(rec[int(rec_off[s]): int(rec_off[s + 1])], int(qps[s]), 2, 3)
                streams.append(b.tobytes())
        else:
            streams = [out[int(out_off[s]): int(out_off[s + 1])].tobytes() for s in range(n_sub)]
        return log, streams

    res = {}
    for name, which in [("std", 0), ("recorded", 2)] + ([("device", 1)] if mode == "gpu" else []):
        log, streams = run(which)
        res[name] = {"log_md5": hashlib.md5(log).hexdigest(), "log_bytes": len(log), "log_lines": log.count(b"\n"),
                     "log_head": log[:200].decode(errors="replace"),
                     "stream_md5": [hashlib.md5(b).hexdigest() for b in streams], "stream_bytes": [len(b) for b in streams]}
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
