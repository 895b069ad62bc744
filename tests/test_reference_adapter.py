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

SO = os.path.join(H.ORACLE_DIR, "_ref", "libadapter_test.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SO), reason="oracle/_ref/libadapter_test.so not built")


@pytest.fixture(scope="module")
def adp():
    from entropy_coding_amd import capi
    capi.load_library()
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp(prefix="cabac_ref_"))   # reference log.cpp:3-4 creates bin_log.txt/bit_log.txt in CWD
    try:
        L = ctypes.CDLL(SO)
    finally:
        os.chdir(cwd)
    L.adapter_last_error.restype = ctypes.c_char_p
    L.adapter_encode.restype = ctypes.c_long
    L.adapter_encode.argtypes = [ctypes.c_int, H.u32p, ctypes.c_long, ctypes.c_int, ctypes.c_int, H.u8p, ctypes.c_long,
                                 H.u32p, H.u32p]
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


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_reference_cabacwriter_on_gpu_encoder_is_bit_exact(adp, seed):
    rng = np.random.default_rng(500 + seed)
    ops = H.random_ops(rng, int(rng.choice([0, 10, 3000, 12000])), ctx_frac=float(rng.choice([0.3, 0.7])), end_trm=False)
    qp, iid = int(rng.integers(0, 64)), int(rng.integers(0, 3))
    ref_bytes, ref_bits, ref_bins = _enc(adp, 0, ops, qp, iid)     # reference encoder
    gpu_bytes, gpu_bits, gpu_bins = _enc(adp, 1, ops, qp, iid)     # GPU encoder behind the same interface
    assert ref_bits == gpu_bits and ref_bins == gpu_bins and np.array_equal(ref_bytes, gpu_bytes)
