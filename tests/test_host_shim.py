"""The C++ host shim (entropy_coding_amd/host: BinEncoderHip / BinDecoderHip / HipBatch / bitstream
mirrors), driven like the reference's CABACWriter drives BinEncIf.  CPU tests cover the recording and
container logic; GPU tests cover the full path shim -> C ABI -> HIP kernels against the oracle and the
reference-generated golden vectors."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import helpers as H

CSRC = os.path.join(H.ROOT, "tests", "csrc")
u8p, u16p, u32p = H.u8p, H.u16p, H.u32p
lp = ctypes.POINTER(ctypes.c_long)
ip = ctypes.POINTER(ctypes.c_int)


@pytest.fixture(scope="module")
def drv():
    san = os.environ.get("CABAC_TEST_SANITIZED_SHIM")   # tests/test_sanitizers.py: shim + driver + oracle-backed stub under ASan
    if san:
        L = ctypes.CDLL(san)
    else:
        from entropy_coding_amd import capi
        capi.load_library()          # builds libcabac_hip.so if stale (and loads torch's HIP runtime first)
        so = os.path.join(CSRC, "libhost_shim_driver.so")
        src = os.path.join(CSRC, "host_shim_driver.cpp")
        lib = os.path.join(H.ROOT, "entropy_coding_amd", "libcabac_hip.so")
        hdrs = [os.path.join(H.ROOT, "entropy_coding_amd", "host", h) for h in ("cabac_hip_host.hpp", "cabac_rem_abs.hpp")] + \
               [os.path.join(H.ROOT, "include", "cabac_hip.h")]
        # (older than a header = another layout of the shim's classes than the library's: never run, always rebuilt)
        if not os.path.exists(so) or os.path.getmtime(so) < max([os.path.getmtime(src), os.path.getmtime(lib)] + [os.path.getmtime(h) for h in hdrs]):
            subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I" + os.path.join(H.ROOT, "include"),
                                   "-I" + os.path.join(H.ROOT, "entropy_coding_amd", "host"), src,
                                   "-L" + os.path.dirname(lib), "-lcabac_hip", "-Wl,-rpath,$ORIGIN/../../entropy_coding_amd",
                                   "-o", so])
        L = ctypes.CDLL(so)
    L.shim_last_error.restype = ctypes.c_char_p
    L.shim_record_ops.restype = ctypes.c_long
    L.shim_record_ops.argtypes = [u32p, ctypes.c_long, u16p, ctypes.c_long, u32p]
    L.shim_encode_streams.argtypes = [ctypes.c_int, u32p, lp, ip, ip, ctypes.c_int, ctypes.c_int, u8p, lp, u32p]
    L.shim_decode_replay.argtypes = [u16p, ctypes.c_long, ctypes.c_int, ctypes.c_int, u8p, ctypes.c_long,
                                     ctypes.c_int, u8p, u32p]
    u64p = ctypes.POINTER(ctypes.c_uint64)
    L.shim_estimate_segments.restype = ctypes.c_long
    L.shim_estimate_segments.argtypes = [u32p, lp, ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, u64p, ctypes.c_int, u16p,
                                         ctypes.c_long]
    L.shim_estimate_many.argtypes = [u16p, lp, ctypes.c_int, ip, ip, u64p]
    L.shim_rem_abs_round_trip.argtypes = [u32p, ctypes.c_long, ctypes.c_int, ctypes.c_int, u32p]
    L.shim_input_bitstream_script.restype = ctypes.c_long
    L.shim_input_bitstream_script.argtypes = [u8p, ctypes.c_long, u32p, ctypes.c_long, u32p, u8p, ctypes.c_long]
    L.shim_bitstream_writes.restype = ctypes.c_long
    L.shim_bitstream_writes.argtypes = [u32p, u32p, ctypes.c_long, ctypes.c_int, u8p, ctypes.c_long, u32p]
    return L


def _record(drv, ops):
    ops = np.ascontiguousarray(ops, np.uint32)
    rec = np.zeros(40 * len(ops) + 8, np.uint16)
    counts = np.zeros(4, np.uint32)
    n = drv.shim_record_ops(H._ptr(ops, u32p), len(ops), H._ptr(rec, u16p), len(rec), H._ptr(counts, u32p))
    assert n >= 0, drv.shim_last_error()
    return rec[:n], counts


@pytest.mark.parametrize("seed", range(6))
def test_recorder_matches_oracle_binarisation(drv, seed):
    """BinEncoderHip + helper binarisers record exactly the bin sequence the oracle's expander gives,
    and BinCounter totals equal the reference's (arith_codec.cpp:281-316)."""
    orc = H.load_oracle()
    rng = np.random.default_rng(50 + seed)
    ops = H.random_ops(rng, 3000, ctx_frac=[0.0, 0.4, 0.8][seed % 3], with_align=(seed == 5))
    rec, counts = _record(drv, ops)
    want = orc.ops_to_records(ops)
    assert np.array_equal(rec, want)
    _, _, nbins = orc.encode_ops(ops, 32, 2, 1)
    assert list(counts[:3]) == list(nbins) and counts[3] == nbins.sum()


def test_recorder_golden_counts(drv):
    gold = np.load(os.path.join(H.GOLDEN, "vectors.npz"))
    for k in range(int(gold["n_cases"][0])):
        rec, counts = _record(drv, gold["case%d_ops" % k])
        assert list(counts[:3]) == list(gold["case%d_nbins" % k])     # the reference's own BinCounter values


def test_recorder_rejects_bad_arguments(drv):
    for bad in ([H.OP_BIN, 1, 379, 0], [H.OP_BINS_EP, 4, 2, 0], [H.OP_UNARY_MAX, 5, 0, 3], [H.OP_TRUNC_BIN, 7, 7, 0]):
        ops = np.array([bad], np.uint32)
        rec = np.zeros(64, np.uint16)
        counts = np.zeros(4, np.uint32)
        assert drv.shim_record_ops(H._ptr(ops, u32p), 1, H._ptr(rec, u16p), 64, H._ptr(counts, u32p)) == -1
        assert b"ERROR" in drv.shim_last_error()


def test_caller_built_from_another_header_version_is_refused(tmp_path):
    """A caller whose compiler laid the shim's classes out from another version of cabac_hip_host.hpp (here: the same header
    with one more member in HipBatch) would corrupt memory silently; the constructor hands the library its view of the
    layout and the library refuses it with an exception before anything else happens."""
    if os.environ.get("CABAC_TEST_SANITIZED_SHIM"):
        return      # (a test of the real library's check; the sanitizer job has the stand-in)
    from entropy_coding_amd import capi
    lib = capi.build_library()
    src = tmp_path / "stale_caller.cpp"
    src.write_text("""
#include <cstring>
#include "cabac_hip_host.hpp"
extern "C" int make_batch(char *msg, int cap) {
  try { EntropyCodingAMD::HipBatch b(0); return 0; }
  catch (std::exception &e) { strncpy(msg, e.what(), cap - 1); return -1; }
}
""")
    got = {}
    for name, flag in (("same", []), ("other", ["-DCABAC_HOST_TEST_OTHER_LAYOUT"])):
        so = str(tmp_path / ("libcaller_%s.so" % name))
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared"] + flag + ["-I" + os.path.join(H.ROOT, "include"),
                               "-I" + os.path.join(H.ROOT, "entropy_coding_amd", "host"), str(src), "-L" + os.path.dirname(lib),
                               "-lcabac_hip", "-Wl,-rpath," + os.path.dirname(lib), "-o", so])
        capi.load_library()
        L = ctypes.CDLL(so)
        msg = ctypes.create_string_buffer(512)
        got[name] = (L.make_batch(msg, 512), msg.value.decode())
    assert got["same"][0] == 0
    assert got["other"][0] == -1 and "another version of cabac_hip_host.hpp" in got["other"][1]


def test_output_bitstream_mirror(drv):
    """write / addSubstream into a non-aligned parent / writeByteAlignment (bit_stream.cpp:70-155)."""
    rng = np.random.default_rng(3)
    for trial in range(50):
        n = int(rng.integers(0, 40))
        nb = rng.integers(0, 33, size=n).astype(np.uint32)
        vals = np.array([int(rng.integers(0, 1 << 32)) & ((1 << int(b)) - 1) if b else 0 for b in nb], np.uint32)
        align = trial & 1
        out = np.zeros(4 * n + 16, np.uint8)
        tb = ctypes.c_uint32()
        got = drv.shim_bitstream_writes(H._ptr(vals, u32p), H._ptr(nb, u32p), n, align, H._ptr(out, u8p), len(out),
                                        ctypes.byref(tb))
        bits = "101" + "".join(format(int(v), "0%db" % int(b)) if b else "" for v, b in zip(vals, nb))
        if align:
            bits += "1"
            bits += "0" * (-len(bits) % 8)
        assert tb.value == len(bits)
        padded = bits + "0" * (-len(bits) % 8)
        want = bytes(int(padded[i:i + 8], 2) for i in range(0, len(padded), 8))
        assert got == len(want) and out[:got].tobytes() == want


def _input_script_model(data, script):
    """Independent bit-string model of InputBitstream (bit_stream.cpp:183-430)."""
    bits = "".join(format(b, "08b") for b in data)
    pos, out, sub = 0, [], bytearray()

    def take(n):
        nonlocal pos
        need_bytes = -(-(pos + n) // 8)
        if need_bytes > len(data):
            raise IndexError
        v = int(bits[pos:pos + n], 2) if n else 0
        pos += n
        return v

    for op, arg in script:
        try:
            if op == 0:
                out.append(take(arg))
            elif op == 1:            # readByte only ever runs byte aligned in the reference's callers
                assert pos % 8 == 0
                out.append(take(8))
            elif op == 2:
                nbytes, tail = arg // 8, arg % 8
                chunk = bytearray()
                if pos % 8 == 0:     # aligned: copies what is there, pads with zeros
                    avail = min(nbytes, len(data) - pos // 8)
                    chunk += bytes(data[pos // 8: pos // 8 + avail]) + bytes(nbytes - avail)
                    pos += 8 * avail
                else:
                    for _ in range(nbytes):
                        chunk.append(take(8))
                if tail:
                    chunk.append(take(tail) << (8 - tail))
                out.append(len(chunk))
                sub += chunk
            elif op == 3:
                n = 0
                while 8 * len(data) - pos > 0 and pos % 8:
                    take(1)
                    n += 1
                out.append(n)
            elif op == 4:
                out.append(8 * len(data) - pos)
            elif op == 5:
                out.append(-pos % 8)
            elif op == 6:
                if take(1) != 1:
                    raise IndexError
                n = -pos % 8
                if n and take(n) != 0:
                    raise IndexError
                out.append(n + 1)
        except IndexError:
            out.append(0xFFFFFFFF)
            break
    return out, bytes(sub)


def _run_input_script(fn, data, script):
    sc = np.ascontiguousarray(script, np.uint32).reshape(-1, 2)
    out = np.zeros(len(sc), np.uint32)
    sub = np.zeros(len(data) + 8 * len(sc) + 8, np.uint8)
    n = fn(H._ptr(data, u8p), len(data), H._ptr(sc, u32p), len(sc), H._ptr(out, u32p), H._ptr(sub, u8p), len(sub))
    assert n >= 0
    return out, sub[:n].tobytes()


def test_input_bitstream_mirror(drv):
    """read / readOutTrailingBits / extractSubstream at any bit position / readByteAlignment of the InputBitstream
    mirror against a bit-string model and, where it is built, against the reference's own class."""
    rng = np.random.default_rng(22)
    ref = None
    if H.ref_available():
        ref = H.load_ref().lib.ref_input_bitstream_script
        ref.restype = ctypes.c_long
        ref.argtypes = drv.shim_input_bitstream_script.argtypes
    for trial in range(200):
        data = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)
        if trial % 3 == 0:
            data[rng.integers(0, len(data))] = 0x80      # a byte that passes readByteAlignment
        script, aligned = [], True
        for _ in range(int(rng.integers(1, 12))):
            op = int(rng.choice([0, 0, 0, 2, 2, 3, 4, 5, 6] + ([1] if aligned else [])))
            arg = int(rng.integers(0, 33)) if op == 0 else int(rng.integers(0, 8 * len(data) + 9)) if op == 2 else 0
            script.append((op, arg))
            aligned = op in (3, 6) or (aligned and op in (1, 4, 5)) or (aligned and op in (0, 2) and arg % 8 == 0)
        want, want_sub = _input_script_model(data, script)
        got, got_sub = _run_input_script(drv.shim_input_bitstream_script, data, script)
        n = len(want)
        assert list(got[:n]) == want, (trial, script, want, list(got[:n]))
        if want[-1] != 0xFFFFFFFF:
            assert got_sub == want_sub
        if ref is not None:
            rgot, rsub = _run_input_script(ref, data, script)
            assert list(rgot[:n]) == want and (want[-1] == 0xFFFFFFFF or rsub == want_sub), (trial, script)


def test_input_bitstream_state_after_a_failed_read(drv):
    """A read that ends exactly on the last byte succeeds; a read past the end throws "Exceeded FIFO size" (readByte: "FIFO
    exceeded") and leaves the byte position and the held bits where they were — only the bit counter has moved
    (bit_stream.cpp:205-208, :240-242).  The scripts carry on after the throw (op 10) and probe the state (ops 7 / 8 / 9 / 5 /
    4); the mirror and, where it is built, the reference's own class must answer alike."""
    ref = None
    if H.ref_available():
        ref = H.load_ref().lib.ref_input_bitstream_script
        ref.restype = ctypes.c_long
        ref.argtypes = drv.shim_input_bitstream_script.argtypes
    probe = [(7, 0), (8, 0), (9, 0), (5, 0), (4, 0)]
    rng = np.random.default_rng(23)
    cases = []
    data4 = np.array([0xA5, 0x3C, 0x81, 0x7E], np.uint8)
    cases.append((data4, [(10, 0), (0, 24), (0, 8)] + probe))                         # ends exactly on the last byte
    cases.append((data4, [(10, 0), (0, 24), (0, 9)] + probe + [(0, 8)] + probe))       # one bit too many, then what is there
    cases.append((data4, [(10, 0), (0, 3), (0, 32)] + probe + [(0, 29)] + probe))      # held bits + 4 bytes > what is left
    cases.append((data4, [(10, 0), (0, 32), (1, 0)] + probe + [(0, 1)] + probe))       # readByte / read at the very end
    cases.append((data4, [(10, 0), (0, 5), (2, 40)] + probe))                          # extractSubstream running dry
    for _ in range(60):
        data = rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8)
        script = [(10, 0)]
        for _ in range(int(rng.integers(2, 10))):
            script.append((0, int(rng.integers(0, 33))))
            if rng.random() < 0.5:
                script += probe
        cases.append((data, script + probe))
    for data, script in cases:
        got, _ = _run_input_script(drv.shim_input_bitstream_script, data, script)
        # model: bit position, with a failed read leaving it alone but counting its bits
        pos, nread, want = 0, 0, []
        bits = "".join(format(int(b), "08b") for b in data)
        for op, arg in script:
            if op == 0:
                nread += arg
                if -(-(pos + arg) // 8) > len(data):
                    want.append(0xFFFFFFFF)
                else:
                    want.append(int(bits[pos:pos + arg], 2) if arg else 0)
                    pos += arg
            elif op == 7:
                want.append(-(-pos // 8))
            elif op == 8:
                want.append(nread)
            elif op == 5:
                want.append(-pos % 8)
            elif op == 4:
                want.append(8 * len(data) - pos)
            else:
                want.append(None)                       # compared with the reference only
        if all(op != 2 for op, _ in script):            # (the model above does not follow extractSubstream)
            for g, w in zip(got, want):
                assert w is None or int(g) == w, (data.tolist(), script, list(got), want)
        if ref is not None:
            rgot, _ = _run_input_script(ref, data, script)
            # the held byte itself (op 9) is only defined while some of its bits are unread
            for i, (op, _) in enumerate(script):
                if op == 9 and got[i + 1] == 0 == rgot[i + 1]:
                    continue
                assert got[i] == rgot[i], (data.tolist(), script, list(got), list(rgot))


# ------------------------------------------------------------------ GPU
def _encode_streams(drv, op_list, qps, ids, mode, flags):
    n = len(op_list)
    ops = np.concatenate(op_list).astype(np.uint32)
    op_off = np.concatenate([[0], np.cumsum([len(o) for o in op_list])]).astype(np.int64)
    caps = [64 + 6 * len(o) * 5 for o in op_list]
    out_off = np.concatenate([[0], np.cumsum(caps)]).astype(np.int64)
    out = np.zeros(int(out_off[-1]), np.uint8)
    nbits = np.zeros(n, np.uint32)
    rc = drv.shim_encode_streams(n, H._ptr(ops, u32p), op_off.ctypes.data_as(lp),
                                 np.asarray(qps, np.int32).ctypes.data_as(ip), np.asarray(ids, np.int32).ctypes.data_as(ip),
                                 mode, flags, H._ptr(out, u8p), out_off.ctypes.data_as(lp), H._ptr(nbits, u32p))
    assert rc == 0, drv.shim_last_error()
    return [out[int(out_off[s]):int(out_off[s]) + (int(nbits[s]) + 7) // 8] for s in range(n)], nbits


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_shim_encode_matches_oracle(drv, mode):
    orc = H.load_oracle()
    rng = np.random.default_rng(70 + mode)
    op_list = [H.random_ops(rng, int(n), ctx_frac=0.6, end_trm=False) for n in rng.integers(0, 1500, size=12)]
    qps, ids = rng.integers(0, 64, size=12), rng.integers(0, 3, size=12)
    for flags in (0, 2, 4, 6):     # bit1: writeByteAlignment afterwards; bit2: pinned mirrors (usePinnedMirrors)
        got, nbits = _encode_streams(drv, op_list, qps, ids, mode, flags)
        for s, ops in enumerate(op_list):
            full = np.concatenate([ops.reshape(-1, 4), np.array([[H.OP_TRM, 1, 0, 0]], np.uint32)])
            want, wbits, _ = orc.encode_ops(full, int(qps[s]), int(ids[s]), 1 | (flags & 2))
            assert wbits == nbits[s] and np.array_equal(got[s], want), s


@pytest.mark.gpu
def test_shim_encode_matches_reference_golden(drv):
    """Golden op streams end with TRM(1) themselves; strip it (the driver adds end_of_slice)."""
    gold = np.load(os.path.join(H.GOLDEN, "vectors.npz"))
    ks = [k for k in range(int(gold["n_cases"][0])) if len(gold["case%d_ops" % k])]
    op_list = [gold["case%d_ops" % k][:-1] for k in ks]
    metas = [[int(x) for x in gold["case%d_meta" % k]] for k in ks]
    got, nbits = _encode_streams(drv, op_list, [m[0] for m in metas], [m[1] for m in metas], 0, 2)
    for i, k in enumerate(ks):
        assert nbits[i] == metas[i][2] and np.array_equal(got[i], gold["case%d_bytes_aligned" % k]), k


@pytest.mark.gpu
def test_shim_get_num_written_bits_immediate_mode(drv):
    """BinEncoderHip::getNumWrittenBits() in Immediate mode (arith_codec.cpp:482-485): after every few calls the answer is the
    oracle's — pinned to the reference's own getNumWrittenBits() by tests/test_oracle_vs_reference.py — plus the bits the
    bitstream held before; Deferred mode throws."""
    orc = H.load_oracle()
    rng = np.random.default_rng(93)
    drv.shim_num_written_bits.restype = ctypes.c_long
    drv.shim_num_written_bits.argtypes = [u32p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, u32p, ctypes.c_long]
    for lead, every in ((0, 7), (5, 40)):
        ops = H.random_ops(rng, 280, ctx_frac=0.6, end_trm=False, with_align=True)
        ans = np.zeros(len(ops), np.uint32)
        n = drv.shim_num_written_bits(H._ptr(ops, u32p), len(ops), 28, 2, every, lead, H._ptr(ans, u32p), len(ans))
        assert n > 0, drv.shim_last_error()
        marks = [i + 1 for i in range(len(ops)) if (i + 1) % every == 0 or i + 1 == len(ops)]
        assert n == len(marks)
        for k, m in enumerate(marks):
            rec = orc.ops_to_records(ops[:m])
            _, want = orc.encode_records(rec, 28, 2, 4)
            assert int(ans[k]) == lead + want, (lead, every, m)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,wide", [(0, False), (1, False), (2, False), (0, True), (2, True)])
def test_shim_spliced_residual_blocks(drv, mode, wide):
    """BinEncoderHip::encodeResidual: coefficients handed over where the reference's writer would call residual_coding
    (cabac_writer.cpp:2424-2525), their bins spliced in between the recorded ones on the device at flush().  Bytes and
    BinCounter totals against the oracle coding the same substreams from host-side records.
    The batch's staging area narrows the coefficients to int16 on the way in; wide: a block in the middle of the batch holds a
    coefficient that does not fit, which turns the staging area — the blocks already in it included — back into 32 bits."""
    orc = H.load_oracle()
    rng = np.random.default_rng(170 + mode)
    n = 9
    op_list, blk_first, blk_at, geom, coeffs, want, want_counts = [], [0], [], [], [], [], []
    qps, ids = rng.integers(0, 64, size=n), rng.integers(0, 3, size=n)
    for s in range(n):
        ops = H.random_ops(rng, int(rng.integers(0, 60)), ctx_frac=0.6, end_trm=False)
        ats = np.sort(rng.integers(0, len(ops) + 1, size=0 if s == 4 else int(rng.integers(1, 9))))
        parts, prev = [], 0
        for at in ats:
            w, h = [(4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (2, 8), (64, 64)][int(rng.integers(0, 7))]
            ts = w <= 32 and rng.random() < 0.3
            fl = (int(rng.integers(0, 2)) | H.TU_TRANSFORM_SKIP) if ts else int(rng.integers(0, 4))
            c = H.random_block(rng, w, h, density=0.5, big=0.1)
            if wide and s == 3 and not ts:
                c[0, 0] = 40000
            ch = int(rng.integers(0, 2))
            geom.append((w, h, ch, fl)); coeffs.append(c.ravel()); blk_at.append(int(at))
            parts.append(orc.ops_to_records(ops[prev:int(at)]) if at > prev else np.zeros(0, np.uint16))
            parts.append(orc.residual_records(c, ch, fl)[0])
            prev = int(at)
        parts.append(orc.ops_to_records(ops[prev:]) if len(ops) > prev else np.zeros(0, np.uint16))
        parts.append(np.array([0x81FF], np.uint16))
        rec = np.concatenate(parts)
        want.append(orc.encode_records(rec, int(qps[s]), int(ids[s]), 3))
        idv = rec & 0x1FF
        want_counts.append((int((idv < H.NUM_CTX).sum()), int((idv == H.REC_EP).sum()), int((idv == H.REC_TRM).sum())))
        op_list.append(ops); blk_first.append(len(geom))
    ops = np.concatenate(op_list).astype(np.uint32) if sum(len(o) for o in op_list) else np.zeros((1, 4), np.uint32)
    op_off = np.concatenate([[0], np.cumsum([len(o) for o in op_list])]).astype(np.int64)
    out_off = np.concatenate([[0], np.cumsum([len(w[0]) + 64 for w in want])]).astype(np.int64)
    out = np.zeros(int(out_off[-1]), np.uint8)
    nbits = np.zeros(n, np.uint32)
    counts = np.zeros(4 * n, np.uint32)
    i32p = ctypes.POINTER(ctypes.c_int32)
    drv.shim_spliced_streams.argtypes = [ctypes.c_int, u32p, lp, ip, ip, ip, i32p, ip, ip, ctypes.c_int, u8p, lp, u32p, u32p]
    rc = drv.shim_spliced_streams(n, H._ptr(ops, u32p), op_off.ctypes.data_as(lp), np.array(blk_first, np.int32).ctypes.data_as(ip),
                                  np.array(blk_at + [0], np.int32).ctypes.data_as(ip), np.array(geom, np.int32).ravel().ctypes.data_as(ip),
                                  np.concatenate(coeffs).astype(np.int32).ctypes.data_as(i32p), np.asarray(qps, np.int32).ctypes.data_as(ip),
                                  np.asarray(ids, np.int32).ctypes.data_as(ip), mode, H._ptr(out, u8p), out_off.ctypes.data_as(lp),
                                  H._ptr(nbits, u32p), H._ptr(counts, u32p))
    assert rc == 0, drv.shim_last_error()
    for s in range(n):
        wb, wbits = want[s]
        assert int(nbits[s]) == wbits and np.array_equal(out[int(out_off[s]): int(out_off[s]) + len(wb)], wb), s
        assert tuple(int(x) for x in counts[4 * s: 4 * s + 3]) == want_counts[s] and int(counts[4 * s + 3]) == sum(want_counts[s]), s


@pytest.mark.gpu
@pytest.mark.parametrize("n_dev", [2, 3])
def test_shim_multi_device_batch(drv, n_dev):
    """HipBatch over several devices (here the same GPU listed n_dev times: n_dev contexts, n_dev host threads at flush): the
    substreams are dealt out longest first, every share is coded on its own context at the same time, and each substream's
    bitstream receives exactly the oracle's bytes whatever context coded it; HipBatch::decode deals the jobs out likewise."""
    orc = H.load_oracle()
    rng = np.random.default_rng(210 + n_dev)
    n = 37
    op_list = [H.random_ops(rng, int(x), ctx_frac=0.6, end_trm=False) for x in rng.integers(0, 2500, size=n)]
    qps, ids = rng.integers(0, 64, size=n), rng.integers(0, 3, size=n)
    ops = np.concatenate(op_list).astype(np.uint32)
    op_off = np.concatenate([[0], np.cumsum([len(o) for o in op_list])]).astype(np.int64)
    out_off = np.concatenate([[0], np.cumsum([64 + 30 * len(o) for o in op_list])]).astype(np.int64)
    out = np.zeros(int(out_off[-1]), np.uint8)
    nbits = np.zeros(n, np.uint32)
    ok = np.zeros(n, np.int32)
    devs = np.zeros(n_dev, np.int32)
    drv.shim_multi_device_round_trip.argtypes = [ctypes.c_int, ip, ctypes.c_int, u32p, lp, ip, ip, u8p, lp, u32p, ip]
    rc = drv.shim_multi_device_round_trip(n_dev, devs.ctypes.data_as(ip), n, H._ptr(ops, u32p), op_off.ctypes.data_as(lp),
                                          np.asarray(qps, np.int32).ctypes.data_as(ip), np.asarray(ids, np.int32).ctypes.data_as(ip),
                                          H._ptr(out, u8p), out_off.ctypes.data_as(lp), H._ptr(nbits, u32p), ok.ctypes.data_as(ip))
    assert rc == 0, drv.shim_last_error()
    for s, o in enumerate(op_list):
        full = np.concatenate([o.reshape(-1, 4), np.array([[H.OP_TRM, 1, 0, 0]], np.uint32)])
        want, wbits, _ = orc.encode_ops(full, int(qps[s]), int(ids[s]), 3)
        assert wbits == nbits[s] and np.array_equal(out[int(out_off[s]): int(out_off[s]) + len(want)], want), s
    assert ok.all()


@pytest.mark.gpu
def test_shim_decode_replay(drv):
    orc = H.load_oracle()
    rng = np.random.default_rng(91)
    rec = H.random_records(rng, 5000)
    data, nbits = orc.encode_records(rec, 27, 2, 3)
    tail = np.array([1, 2, 3], np.uint8)                       # following bytes must stay unread
    buf = np.concatenate([data, tail])
    bins = np.zeros(len(rec), np.uint8)
    idx = ctypes.c_uint32()
    rc = drv.shim_decode_replay(H._ptr(rec, u16p), len(rec), 27, 2, H._ptr(buf, u8p), len(buf), 1, H._ptr(bins, u8p),
                                ctypes.byref(idx))
    assert rc == 0, drv.shim_last_error()
    assert np.array_equal(bins, (rec >> 15).astype(np.uint8)) and idx.value == len(data)
    # reference error behaviour: truncated input throws "FIFO exceeded"
    rc = drv.shim_decode_replay(H._ptr(rec, u16p), len(rec), 27, 2, H._ptr(buf, u8p), len(data) // 2, 1,
                                H._ptr(bins, u8p), ctypes.byref(idx))
    assert rc == -1 and b"FIFO exceeded" in drv.shim_last_error()


@pytest.mark.gpu
def test_shim_decode_rem_abs(drv):
    """BinDecoderHip::decodeRemAbsEP (arith_codec.cpp:153-179) returns what encodeRemAbsEP coded: short and escape
    codes, every Rice parameter, the longest prefix, extended dynamic range."""
    rng = np.random.default_rng(92)
    vals = []
    for i in range(400):
        rice = int(rng.integers(0, 5))
        max_log2 = int(rng.choice([15, 15, 15, 17, 20]))
        kind = i % 4
        v = int(rng.integers(0, 5 << rice)) if kind == 0 else int(rng.integers(0, 300)) if kind == 1 else \
            int(rng.integers(0, 1 << 15)) if kind == 2 else (1 << max_log2) - 1 - int(rng.integers(0, 3))
        vals.append((v, rice, max_log2))
    a = np.array(vals, np.uint32)
    got = np.zeros(len(a), np.uint32)
    rc = drv.shim_rem_abs_round_trip(H._ptr(a, u32p), len(a), 30, 2, H._ptr(got, u32p))
    assert rc == 0, drv.shim_last_error()
    assert np.array_equal(got, a[:, 0])


# ---- bit estimator shim (BitEstimatorHip / HipBatch::estimate) ----------------------------------------
def _segments(rng, n_ops, n_seg):
    ends = np.sort(rng.integers(0, n_ops + 1, size=n_seg - 1)).tolist() + [n_ops]
    kinds = [0] + [int(k) for k in rng.integers(0, 3, size=n_seg - 1)]
    return np.array(ends, np.int64), np.array(kinds, np.int32)


def test_estimator_recorder_matches_reference_semantics(drv):
    """CPU: what BitEstimatorHip records (pseudo-records for resetBits / start / restart included), costed by the
    oracle, equals the reference's BitEstimator_Std driven with the same calls."""
    orc = H.load_oracle()
    rng = np.random.default_rng(610)
    for _ in range(6):
        ops = H.random_ops(rng, 800, ctx_frac=0.7, with_align=True)
        ends, kinds = _segments(rng, len(ops), 6)
        rec = np.zeros(40 * len(ops) + 64, np.uint16)
        n = drv.shim_estimate_segments(H._ptr(ops, u32p), ends.ctypes.data_as(lp), kinds.ctypes.data_as(ip), len(ends),
                                       31, 2, None, 1, H._ptr(rec, u16p), len(rec))
        assert n >= 0, drv.shim_last_error()
        rec = rec[:n]
        # the same calls on the oracle: the op segments expanded to records, pseudo-records in between
        want, begin = [], 0
        for e, k in zip(ends, kinds):
            if begin or len(want):
                pass
            seg = orc.ops_to_records(ops[begin:e]) if e > begin else np.zeros(0, np.uint16)
            want.append((k, seg))
            begin = int(e)
        parts = []
        for i, (k, seg) in enumerate(want):
            if i > 0:
                parts.append(np.array([0x1FB if k == 2 else 0x1FC], np.uint16))
            parts.append((seg & 0x81FF).astype(np.uint16))
        flat = np.concatenate(parts)
        # the estimator records bypass bins without their values (their cost does not depend on them)
        ep = (flat & 0x1FF) == 0x1FE
        got_ep = (rec & 0x1FF) == 0x1FE
        assert len(rec) == len(flat) and np.array_equal(ep, got_ep)
        assert np.array_equal(rec[~got_ep], flat[~ep])
        assert orc.estimate_records(rec, 31, 2) == orc.estimate_records(flat, 31, 2)
        if H.ref_available():
            assert H.load_ref().estimate_records(flat, 31, 2) == orc.estimate_records(rec, 31, 2)


@pytest.mark.gpu
def test_estimator_shim_on_gpu(drv):
    """GPU: getEstFracBits() after every segment equals the oracle's running value; HipBatch::estimate in bulk."""
    orc = H.load_oracle()
    rng = np.random.default_rng(611)
    ops = H.random_ops(rng, 1500, ctx_frac=0.6, with_align=True)
    ends, kinds = _segments(rng, len(ops), 5)
    costs = np.zeros(len(ends), np.uint64)
    rec = np.zeros(40 * len(ops) + 64, np.uint16)
    n = drv.shim_estimate_segments(H._ptr(ops, u32p), ends.ctypes.data_as(lp), kinds.ctypes.data_as(ip), len(ends), 22, 1,
                                   costs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), 0, H._ptr(rec, u16p), len(rec))
    assert n >= 0, drv.shim_last_error()
    rec = rec[:n]
    # prefix of the recording up to the end of each segment = everything before the next pseudo-record
    cuts = [i for i, r in enumerate(rec) if (r & 0x1FF) in (0x1FB, 0x1FC)] + [len(rec)]
    assert len(cuts) == len(ends)
    for i, c in enumerate(cuts):
        assert orc.estimate_records(rec[:c], 22, 1) == (0, int(costs[i])), i
    # bulk: 50 candidate strings, one launch
    recs = [H.random_records(rng, int(rng.integers(0, 400)), ctx_frac=0.7) for _ in range(50)]
    off = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.int64)
    qps = rng.integers(0, 64, size=50).astype(np.int32)
    ids = rng.integers(0, 3, size=50).astype(np.int32)
    out = np.zeros(50, np.uint64)
    allrec = np.concatenate(recs + [np.zeros(1, np.uint16)])
    rc = drv.shim_estimate_many(H._ptr(allrec, u16p), off.ctypes.data_as(lp), 50, qps.ctypes.data_as(ip),
                                ids.ctypes.data_as(ip), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
    assert rc == 0, drv.shim_last_error()
    for i in range(50):
        assert orc.estimate_records(recs[i], int(qps[i]), int(ids[i])) == (0, int(out[i]))


@pytest.mark.gpu
def test_residual_round_trip_through_the_shim(drv):
    """HipBatch::residual -> BinEncoderHip -> HipBatch::residualParse: coefficient blocks to bytes and back, C++ only."""
    rng = np.random.default_rng(77)
    blocks, geom, first = [], [], [0]
    for j in range(5):
        for k in range(int(rng.integers(1, 7))):
            w, h = [(4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (2, 8), (64, 64)][int(rng.integers(0, 7))]
            blocks.append(H.random_block(rng, w, h, density=0.5, big=0.1))
            geom.append((w, h, int(rng.integers(0, 2)), H.TU_DEP_QUANT if j & 1 else 0))
        first.append(len(blocks))
    cin = np.concatenate([b.ravel() for b in blocks]).astype(np.int32)
    cout = np.zeros_like(cin)
    g = np.array(geom, np.int32).ravel()
    f = np.array(first, np.int32)
    nb = ctypes.c_long(0)
    i32p, ip = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int)
    drv.shim_residual_round_trip.argtypes = [ctypes.c_int, ip, ip, i32p, ctypes.c_int, i32p, ctypes.POINTER(ctypes.c_long)]
    rc = drv.shim_residual_round_trip(5, f.ctypes.data_as(ip), g.ctypes.data_as(ip), cin.ctypes.data_as(i32p), 30,
                                      cout.ctypes.data_as(i32p), ctypes.byref(nb))
    assert rc == 0, drv.shim_last_error()
    o = 0
    for b in blocks:
        h, w = b.shape
        got = cout[o:o + w * h].reshape(h, w)
        assert np.array_equal(got[:min(h, 32), :min(w, 32)], b[:min(h, 32), :min(w, 32)])
        o += w * h
    assert nb.value > 0
