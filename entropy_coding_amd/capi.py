"""ctypes binding of include/cabac_hip.h — test/bench plumbing; the product is the shared library."""
import ctypes
import sys
import os
import weakref

import numpy as np

from .build import build_library

NUM_CTX = 379
REC_BIN = 0x8000
REC_ALIGN, REC_EP, REC_TRM = 0x1FD, 0x1FE, 0x1FF
SUB_FINISH, SUB_ALIGN_RBSP = 0x100, 0x200
CABAC_INIT_B, CABAC_INIT_P, CABAC_INIT_I = 0, 1, 2
RES_OVERFLOW, RES_BAD_RECORD, RES_UNDERRUN, RES_BAD_STOP, RES_RANGE = 1, 2, 4, 8, 16

DESC_DTYPE = np.dtype([("rec_offset", "<u8"), ("byte_offset", "<u8"), ("n_records", "<u4"),
                       ("byte_capacity", "<u4"), ("qp", "<i4"), ("init_id", "<u4")])
RESULT_DTYPE = np.dtype([("n_bits", "<u4"), ("flags", "<u4")])
assert DESC_DTYPE.itemsize == 32 and RESULT_DTYPE.itemsize == 8
# cabac_tu_desc (include/cabac_hip.h): one transform block for the residual binariser
TU_DTYPE = np.dtype([("coeff_offset", "<u8"), ("log2_width", "u1"), ("log2_height", "u1"), ("channel", "u1"),
                     ("flags", "u1"), ("max_log2_tr_range", "u1"), ("reserved", "u1", (3,))])
assert TU_DTYPE.itemsize == 16
TU_DEP_QUANT, TU_SIGN_HIDING, TU_TS_FLAG, TU_TRANSFORM_SKIP, TU_BDPCM, TU_SBT_ZERO_OUT = 1, 2, 4, 8, 16, 32
TU_INFO_MTS_VIOLATION, TU_INFO_EMPTY, TU_INFO_BAD_DESC = 0x10000, 0x80000000, 0x40000000
TU_INFO_TS = 0x20000
SPLICE_DTYPE = np.dtype([("at", "<u4"), ("tu", "<u4")])          # cabac_splice
BIN_COUNT_WORDS = NUM_CTX + 2                                    # CABAC_BIN_COUNT_WORDS
SUB_PROBE = 0x400

EXPORTS = [
    "cabac_hip_encode_bound", "cabac_hip_init", "cabac_hip_destroy", "cabac_hip_strerror",
    "cabac_hip_last_error", "cabac_hip_set_stream", "cabac_hip_synchronize", "cabac_hip_set_variant",
    "cabac_hip_encode_device", "cabac_hip_decode_device", "cabac_hip_ctx_init_device",
    "cabac_hip_binarize_device", "cabac_hip_residual_device", "cabac_hip_residual_batch", "cabac_hip_residual_parse_device", "cabac_hip_residual_parse_batch", "cabac_hip_encode_batch", "cabac_hip_decode_batch",
    "cabac_hip_last_kernel_ms", "cabac_synth_records", "cabac_hip_profile_enable", "cabac_hip_profile_read",
    "cabac_hip_assemble_device", "cabac_hip_split_device", "cabac_hip_count_emulations_device",
    "cabac_hip_estimate_device", "cabac_hip_estimate_batch", "cabac_hip_estimate_from_device",
    "cabac_hip_host_alloc", "cabac_hip_host_free", "cabac_hip_host_register", "cabac_hip_host_unregister",
    "cabac_hip_host_is_pinned", "cabac_hip_encode_batch_payload", "cabac_hip_wait_event", "cabac_hip_record_event",
    "cabac_hip_encode_residual_device", "cabac_hip_encode_batch_residual", "cabac_hip_encode_residual16_device",
    "cabac_hip_encode_batch_residual16", "cabac_hip_decode_batch_packed", "cabac_hip_residual_parse16_device",
    "cabac_hip_residual_parse_batch16", "cabac_hip_gather_records_device",
]

_lib = None
vp = ctypes.c_void_p
# every CabacHip / PinnedArray that has not been closed yet: close_all() ends them in a defined order (contexts first, then
# the pinned buffers they may still have been copying from) while the HIP runtime is certainly alive
_live = weakref.WeakSet()


def close_all():
    """Close every live context, then every live pinned array.  Call before the process ends (a test session's last
    fixture, a server's shutdown hook): teardown then happens here, in this order, and not in whatever order the
    interpreter and the loaded libraries are finalized in."""
    objs = list(_live)
    for o in objs:
        if isinstance(o, CabacHip):
            o.close()
    for o in objs:
        if isinstance(o, PinnedArray):
            o.close()


def load_library():
    """Load libcabac_hip.so (building it first if the sources are newer). Loading does not need a GPU;
    cabac_hip_init does."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch bundles its own libamdhip64.so.7; whichever copy is loaded first serves the whole
        # process (same SONAME), and torch only works with its own.  Load it first so that this
        # library and torch share ONE HIP runtime (device memory, streams) in python processes.
        import torch  # noqa: F401
    except ImportError:
        pass
    # CABAC_HIP_LIBRARY: load this build of the library instead (experiments: the same sources under other compiler flags)
    L = ctypes.CDLL(os.environ.get("CABAC_HIP_LIBRARY") or build_library())
    L.cabac_hip_encode_bound.restype = ctypes.c_size_t
    L.cabac_hip_encode_bound.argtypes = [ctypes.c_uint64] * 3
    L.cabac_hip_init.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.cabac_hip_destroy.argtypes = [vp]
    L.cabac_hip_destroy.restype = None
    L.cabac_hip_strerror.restype = ctypes.c_char_p
    L.cabac_hip_strerror.argtypes = [ctypes.c_int]
    L.cabac_hip_last_error.restype = ctypes.c_char_p
    L.cabac_hip_last_error.argtypes = [vp]
    L.cabac_hip_set_stream.argtypes = [vp, vp]
    L.cabac_hip_synchronize.argtypes = [vp]
    L.cabac_hip_wait_event.argtypes = [vp, vp]
    L.cabac_hip_record_event.argtypes = [vp, vp]
    L.cabac_hip_set_variant.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.cabac_hip_encode_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp]
    L.cabac_hip_decode_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp, vp]
    L.cabac_hip_ctx_init_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp]
    L.cabac_hip_binarize_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp, vp]
    L.cabac_hip_residual_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp, vp, vp]
    L.cabac_hip_residual_parse_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp, vp, vp, vp]
    L.cabac_hip_residual_parse_batch.argtypes = [vp, ctypes.c_uint32, vp, vp, ctypes.c_uint64, vp, vp, vp, ctypes.c_uint64, vp, vp]
    L.cabac_hip_residual_parse_batch16.argtypes = [vp, ctypes.c_uint32, vp, vp, ctypes.c_uint64, vp, vp, vp, ctypes.c_uint64, vp, vp]
    L.cabac_hip_residual_parse16_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp, vp, vp, vp]
    L.cabac_hip_residual_batch.argtypes = [vp, ctypes.c_uint32, vp, vp, ctypes.c_uint64, vp, vp, vp, ctypes.c_uint64]
    L.cabac_hip_encode_batch.argtypes = [vp, ctypes.c_uint32, vp, vp, ctypes.c_uint64, vp, ctypes.c_uint64, vp]
    L.cabac_hip_decode_batch.argtypes = [vp, ctypes.c_uint32, vp, vp, ctypes.c_uint64, vp, ctypes.c_uint64, vp, vp]
    L.cabac_hip_decode_batch_packed.argtypes = [vp, ctypes.c_uint32, vp, vp, ctypes.c_uint64, vp, ctypes.c_uint64, vp, vp]
    L.cabac_hip_encode_batch_payload.argtypes = [vp, ctypes.c_uint32, vp, vp, ctypes.c_uint64, vp, ctypes.c_uint64, vp, vp]
    L.cabac_hip_encode_residual_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp, ctypes.c_uint32, ctypes.c_uint32, vp, vp,
                                                   vp, ctypes.c_uint64, vp, vp, vp, vp]
    L.cabac_hip_encode_residual16_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp, ctypes.c_uint32, ctypes.c_uint32, vp, vp,
                                                   vp, ctypes.c_uint64, vp, vp, vp, vp]
    L.cabac_hip_encode_batch_residual.argtypes = [vp, ctypes.c_uint32, vp, vp, ctypes.c_uint64, vp, vp, ctypes.c_uint32, vp, vp,
                                                  ctypes.c_uint64, vp, ctypes.c_uint64, vp, vp, vp, vp]
    L.cabac_hip_encode_batch_residual16.argtypes = [vp, ctypes.c_uint32, vp, vp, ctypes.c_uint64, vp, vp, ctypes.c_uint32, vp, vp,
                                                  ctypes.c_uint64, vp, ctypes.c_uint64, vp, vp, vp, vp]
    L.cabac_hip_gather_records_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp, vp]
    L.cabac_hip_last_kernel_ms.restype = ctypes.c_float
    L.cabac_hip_last_kernel_ms.argtypes = [vp]
    L.cabac_hip_assemble_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp, ctypes.c_uint64, vp]
    L.cabac_hip_split_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp]
    L.cabac_hip_count_emulations_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp]
    L.cabac_hip_estimate_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp]
    L.cabac_hip_estimate_batch.argtypes = [vp, ctypes.c_uint32, vp, vp, ctypes.c_uint64, vp, vp]
    L.cabac_hip_estimate_from_device.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp, vp, vp, vp]
    L.cabac_hip_profile_enable.argtypes = [vp, ctypes.c_uint32]
    L.cabac_hip_profile_read.argtypes = [vp, vp, vp, ctypes.c_uint32]
    L.cabac_hip_host_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
    L.cabac_hip_host_free.argtypes = [vp]
    L.cabac_hip_host_register.argtypes = [vp, ctypes.c_size_t]
    L.cabac_hip_host_unregister.argtypes = [vp]
    L.cabac_hip_host_is_pinned.argtypes = [vp, ctypes.c_size_t]
    L.cabac_synth_records.restype = None
    L.cabac_synth_records.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, vp]
    _lib = L
    return L


class CabacHipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("cabac_hip status %d (%s): %s" % (status, load_library().cabac_hip_strerror(status).decode(), msg))
        self.status = status


def synth_records(seed, substream_index, n_bins, ctx_permille):
    L = load_library()
    out = np.empty(n_bins, np.uint16)
    L.cabac_synth_records(seed, substream_index, n_bins, ctx_permille, out.ctypes.data)
    return out


class PinnedArray:
    """A numpy array in page-locked host memory from cabac_hip_host_alloc (freed with close() / garbage collection):
    buffers the host-pointer entry points DMA without a staging copy."""

    def __init__(self, shape, dtype):
        L = load_library()
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = vp()
        rc = L.cabac_hip_host_alloc(max(n, 1), ctypes.byref(p))
        if rc != 0:
            raise CabacHipError(rc, "cabac_hip_host_alloc(%d)" % n)
        self._p = p
        _live.add(self)
        buf = (ctypes.c_uint8 * max(n, 1)).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            load_library().cabac_hip_host_free(self._p)
            self._p = None
            _live.discard(self)

    def __del__(self):
        # not while the interpreter is shutting down: the HIP runtime (torch's, ours) may already be unloading then, and a
        # stream / event / pinned-memory release into a half-torn-down runtime can abort the process; the OS reclaims it all
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass


def host_is_pinned(a):
    a = np.asarray(a)
    return bool(load_library().cabac_hip_host_is_pinned(vp(a.ctypes.data), a.nbytes))


def encode_bound(n_ctx, n_ep, n_trm):
    return load_library().cabac_hip_encode_bound(n_ctx, n_ep, n_trm)


class CabacHip:
    """One codec context = one device + one HIP stream (stream=None: a non-blocking stream of the ctx's own; otherwise the
    caller's stream handle, 0 being the device's default stream).  Raises if there is no GPU (no CPU fallback)."""

    def __init__(self, device=0, stream=None):
        self.L = load_library()
        h = vp()
        rc = self.L.cabac_hip_init(device, ctypes.byref(h))
        if rc != 0:
            raise CabacHipError(rc, "cabac_hip_init(device=%d)" % device)
        self.h = h
        self.device = device
        _live.add(self)
        if stream is not None:
            # a HIP stream handle; 0 is the device's default stream (torch's current stream unless it was changed), which the
            # C ABI takes as CABAC_HIP_STREAM_DEFAULT — a null pointer there means "the ctx's own stream"
            self._check(self.L.cabac_hip_set_stream(self.h, vp(stream if stream else 1)))

    def close(self):
        if getattr(self, "h", None):
            self.L.cabac_hip_destroy(self.h)   # waits for the ctx's streams, then releases them
            self.h = None
            _live.discard(self)

    def __del__(self):
        # not while the interpreter is shutting down: the HIP runtime (torch's, ours) may already be unloading then, and a
        # stream / event / pinned-memory release into a half-torn-down runtime can abort the process; the OS reclaims it all
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    def _check(self, rc, allow_substream=False):
        if rc != 0 and not (allow_substream and rc == -5):
            raise CabacHipError(rc, self.L.cabac_hip_last_error(self.h).decode())
        return rc

    def set_variant(self, enc=0, dec=0):
        self._check(self.L.cabac_hip_set_variant(self.h, enc, dec))

    def synchronize(self):
        self._check(self.L.cabac_hip_synchronize(self.h))

    def wait_event(self, hip_event):
        """The ctx's stream waits for `hip_event` (a hipEvent_t, e.g. torch.cuda.Event(...).cuda_event after record())."""
        self._check(self.L.cabac_hip_wait_event(self.h, vp(hip_event)))

    def record_event(self, hip_event):
        self._check(self.L.cabac_hip_record_event(self.h, vp(hip_event)))

    def last_kernel_ms(self):
        return float(self.L.cabac_hip_last_kernel_ms(self.h))

    def profile_enable(self, capacity):
        self._check(self.L.cabac_hip_profile_enable(self.h, capacity))
        self._prof_cap = capacity

    def profile_read(self):
        """[(kind, ms)] of the device calls since the last read; kind 0 encode, 1 decode, 2 binarize."""
        cap = getattr(self, "_prof_cap", 0)
        kind = np.zeros(max(cap, 1), np.int32)
        ms = np.zeros(max(cap, 1), np.float32)
        n = self.L.cabac_hip_profile_read(self.h, kind.ctypes.data, ms.ctypes.data, cap)
        if n < 0:
            self._check(n)
        return list(zip(kind[:n].tolist(), ms[:n].tolist()))

    # ---- host-pointer entry points (numpy) --------------------------------------------------
    def encode_batch(self, desc, records, bytes_total, check=True, out=None):
        """`out` (optional): the caller's byte buffer (e.g. PinnedArray(...).array), else a fresh zeroed array."""
        desc = np.ascontiguousarray(desc, DESC_DTYPE)
        records = np.ascontiguousarray(records, np.uint16)
        if out is None:
            out = np.zeros(max(int(bytes_total), 1), np.uint8)
        res = np.zeros(len(desc), RESULT_DTYPE)
        rc = self.L.cabac_hip_encode_batch(self.h, len(desc), desc.ctypes.data, records.ctypes.data, len(records),
                                           out.ctypes.data, int(bytes_total), res.ctypes.data)
        self._check(rc, allow_substream=not check)
        return out, res

    def encode_batch_payload(self, desc, records, payload, check=True):
        """cabac_hip_encode_batch_payload: the coded substreams back to back in `payload` (uint8 array, e.g. pinned);
        returns (offsets uint64[n + 1], results)."""
        desc = np.ascontiguousarray(desc, DESC_DTYPE)
        records = np.ascontiguousarray(records, np.uint16)
        offsets = np.zeros(len(desc) + 1, np.uint64)
        res = np.zeros(len(desc), RESULT_DTYPE)
        rc = self.L.cabac_hip_encode_batch_payload(self.h, len(desc), desc.ctypes.data, records.ctypes.data, len(records),
                                                   payload.ctypes.data, payload.nbytes, offsets.ctypes.data, res.ctypes.data)
        self._check(rc, allow_substream=not check)
        return offsets, res

    def decode_batch(self, desc, records, data, check=True, bins=None):
        desc = np.ascontiguousarray(desc, DESC_DTYPE)
        records = np.ascontiguousarray(records, np.uint16)
        data = np.ascontiguousarray(data, np.uint8)
        if bins is None:
            bins = np.zeros(max(len(records), 1), np.uint8)
        res = np.zeros(len(desc), RESULT_DTYPE)
        rc = self.L.cabac_hip_decode_batch(self.h, len(desc), desc.ctypes.data, records.ctypes.data, len(records),
                                           data.ctypes.data, len(data), bins.ctypes.data, res.ctypes.data)
        self._check(rc, allow_substream=not check)
        return bins[: len(records)], res

    def decode_batch_packed(self, desc, records, data, check=True, packed=None):
        """cabac_hip_decode_batch_packed: (packed uint8[(n + 7) // 8] — bit r & 7 of byte r >> 3 = the bin of record r —, results)."""
        desc = np.ascontiguousarray(desc, DESC_DTYPE)
        records = np.ascontiguousarray(records, np.uint16)
        data = np.ascontiguousarray(data, np.uint8)
        if packed is None:
            packed = np.zeros((len(records) + 7) // 8 + 1, np.uint8)
        res = np.zeros(len(desc), RESULT_DTYPE)
        rc = self.L.cabac_hip_decode_batch_packed(self.h, len(desc), desc.ctypes.data, records.ctypes.data, len(records),
                                                  data.ctypes.data, len(data), packed.ctypes.data, res.ctypes.data)
        self._check(rc, allow_substream=not check)
        return packed[: (len(records) + 7) // 8], res

    # ---- device-pointer entry points (raw addresses, e.g. torch tensor .data_ptr()) ----------
    def encode_device(self, n_sub, d_desc, d_records, d_bytes, d_results):
        self._check(self.L.cabac_hip_encode_device(self.h, n_sub, vp(d_desc), vp(d_records), vp(d_bytes), vp(d_results)))

    def decode_device(self, n_sub, d_desc, d_records, d_bytes, d_bins, d_results):
        self._check(self.L.cabac_hip_decode_device(self.h, n_sub, vp(d_desc), vp(d_records), vp(d_bytes), vp(d_bins),
                                                   vp(d_results)))

    def estimate_device(self, n_sub, d_desc, d_records, d_frac_bits, d_flags=0):
        """BitEstimator_Std over a batch of bin strings: d_frac_bits[s] (uint64) = cost in 1/32768 bit."""
        self._check(self.L.cabac_hip_estimate_device(self.h, n_sub, vp(d_desc), vp(d_records), vp(d_frac_bits),
                                                     vp(d_flags) if d_flags else None))

    def estimate_from_device(self, n_sub, d_desc, d_records, d_state, d_rate, d_set, d_frac_bits, d_flags=0):
        """estimate_device started from given context sets (format of ctx_init_device) instead of reset(qp, initId)."""
        self._check(self.L.cabac_hip_estimate_from_device(self.h, n_sub, vp(d_desc), vp(d_records), vp(d_state), vp(d_rate),
                                                          vp(d_set), vp(d_frac_bits), vp(d_flags) if d_flags else None))

    def estimate_batch(self, desc, records, check=False):
        """Host arrays through cabac_hip_estimate_batch: (frac_bits uint64[n], flags uint32[n])."""
        desc = np.ascontiguousarray(desc, DESC_DTYPE)
        records = np.ascontiguousarray(records, np.uint16)
        bits = np.zeros(max(len(desc), 1), np.uint64)
        flags = np.zeros(max(len(desc), 1), np.uint32)
        rc = self.L.cabac_hip_estimate_batch(self.h, len(desc), desc.ctypes.data, records.ctypes.data, len(records),
                                             bits.ctypes.data, flags.ctypes.data)
        self._check(rc, allow_substream=not check)
        return bits[: len(desc)], flags[: len(desc)]

    def binarize_device(self, n_sub, d_se_offset, d_se, d_rec_offset, d_n_records, d_records):
        self._check(self.L.cabac_hip_binarize_device(self.h, n_sub, vp(d_se_offset), vp(d_se), vp(d_rec_offset),
                                                     vp(d_n_records), vp(d_records)))

    def residual_device(self, n_tu, d_tu, d_coeff, d_rec_offset, d_n_records, d_info, d_records):
        """cabac_hip_residual_device: coefficient blocks -> bin records (pass 1 when d_records == 0)."""
        self._check(self.L.cabac_hip_residual_device(self.h, n_tu, vp(d_tu), vp(d_coeff), vp(d_rec_offset),
                                                     vp(d_n_records), vp(d_info), vp(d_records)))

    def residual_parse_device(self, n_sub, d_desc, d_bytes, d_tile_first, d_tu, d_coeff, d_results, d_tu_info=0, int16=False):
        """cabac_hip_residual_parse_device (int16: cabac_hip_residual_parse16_device): bytes -> coefficient blocks, contexts
        derived on the device."""
        self._check((self.L.cabac_hip_residual_parse16_device if int16 else self.L.cabac_hip_residual_parse_device)(self.h, n_sub, vp(d_desc), vp(d_bytes), vp(d_tile_first), vp(d_tu),
                                                           vp(d_coeff), vp(d_tu_info) if d_tu_info else None, vp(d_results)))

    def residual_parse_batch(self, desc, data, tile_first, tus, n_coeff_total, check=True, with_info=False, int16=False, coeff=None):
        """Host arrays in, (coeff, results[, info]) out (cabac_hip_residual_parse_batch, synchronous; int16:
        cabac_hip_residual_parse_batch16, the blocks as int16)."""
        desc = np.ascontiguousarray(desc, DESC_DTYPE)
        data = np.ascontiguousarray(data, np.uint8)
        tile_first = np.ascontiguousarray(tile_first, np.uint32)
        tus = np.ascontiguousarray(tus, TU_DTYPE)
        if coeff is None:
            coeff = np.zeros(max(int(n_coeff_total), 1), np.int16 if int16 else np.int32)
        res = np.zeros(max(len(desc), 1), RESULT_DTYPE)
        info = np.zeros(max(len(tus), 1), np.uint32)
        rc = (self.L.cabac_hip_residual_parse_batch16 if int16 else self.L.cabac_hip_residual_parse_batch)(self.h, len(desc), desc.ctypes.data, data.ctypes.data, len(data),
                                                   tile_first.ctypes.data, tus.ctypes.data, coeff.ctypes.data, int(n_coeff_total),
                                                   info.ctypes.data, res.ctypes.data)
        self._check(rc, allow_substream=not check)
        if with_info:
            return coeff[: int(n_coeff_total)], res[: len(desc)], info[: len(tus)]
        return coeff[: int(n_coeff_total)], res[: len(desc)]

    def residual_batch(self, tus, coeff, check=True):
        """Host arrays in, (records, offsets, info) out (cabac_hip_residual_batch: both passes, synchronous)."""
        tus = np.ascontiguousarray(tus, TU_DTYPE)
        coeff = np.ascontiguousarray(coeff, np.int32)
        n = len(tus)
        offsets = np.zeros(n + 1, np.uint64)
        info = np.zeros(max(n, 1), np.uint32)
        rc = self.L.cabac_hip_residual_batch(self.h, n, tus.ctypes.data, coeff.ctypes.data, len(coeff), offsets.ctypes.data,
                                             info.ctypes.data, None, 0)
        self._check(rc, allow_substream=not check)
        records = np.zeros(max(int(offsets[n]), 1), np.uint16)
        rc = self.L.cabac_hip_residual_batch(self.h, n, tus.ctypes.data, coeff.ctypes.data, len(coeff), offsets.ctypes.data,
                                             info.ctypes.data, records.ctypes.data, len(records))
        self._check(rc, allow_substream=not check)
        return records[: int(offsets[n])], offsets, info[:n]

    def encode_residual_device(self, n_sub, d_desc, d_records, d_splice_first, d_splices, n_splice, n_tu, d_tu, d_coeff,
                               d_payload, payload_capacity, d_payload_offsets, d_results, d_tu_info=0, d_bin_counts=0, int16=False):
        """cabac_hip_encode_residual_device (int16: cabac_hip_encode_residual16_device, d_coeff holds int16 coefficients): host
        records + spliced coefficient blocks -> compacted coded substreams."""
        self._check((self.L.cabac_hip_encode_residual16_device if int16 else self.L.cabac_hip_encode_residual_device)(
            self.h, n_sub, vp(d_desc), vp(d_records), vp(d_splice_first), vp(d_splices) if d_splices else None, n_splice, n_tu,
            vp(d_tu) if d_tu else None, vp(d_coeff) if d_coeff else None, vp(d_payload), payload_capacity, vp(d_payload_offsets),
            vp(d_results), vp(d_tu_info) if d_tu_info else None, vp(d_bin_counts) if d_bin_counts else None))

    def encode_batch_residual(self, desc, records, splice_first, splices, tus, coeff, payload, check=True, with_info=False,
                              with_counts=False):
        """cabac_hip_encode_batch_residual (host arrays, synchronous): (offsets uint64[n + 1], results[, tu_info][, counts]).
        int16 coefficients go through cabac_hip_encode_batch_residual16."""
        narrow = isinstance(coeff, np.ndarray) and coeff.dtype == np.int16
        desc = np.ascontiguousarray(desc, DESC_DTYPE)
        records = np.ascontiguousarray(records, np.uint16)
        splice_first = np.ascontiguousarray(splice_first, np.uint32)
        splices = np.ascontiguousarray(splices, SPLICE_DTYPE)
        tus = np.ascontiguousarray(tus, TU_DTYPE)
        coeff = np.ascontiguousarray(coeff, np.int16 if narrow else np.int32)
        n = len(desc)
        offsets = np.zeros(n + 1, np.uint64)
        res = np.zeros(max(n, 1), RESULT_DTYPE)
        info = np.zeros(max(len(tus), 1), np.uint32) if with_info else None
        counts = np.zeros((max(n, 1), BIN_COUNT_WORDS), np.uint32) if with_counts else None
        rc = (self.L.cabac_hip_encode_batch_residual16 if narrow else self.L.cabac_hip_encode_batch_residual)(
            self.h, n, desc.ctypes.data, records.ctypes.data, len(records), splice_first.ctypes.data, splices.ctypes.data,
            len(tus), tus.ctypes.data, coeff.ctypes.data, len(coeff), payload.ctypes.data, payload.nbytes, offsets.ctypes.data,
            res.ctypes.data, info.ctypes.data if with_info else None, counts.ctypes.data if with_counts else None)
        self._check(rc, allow_substream=not check)
        out = [offsets, res[:n]]
        if with_info:
            out.append(info[: len(tus)])
        if with_counts:
            out.append(counts[:n])
        return tuple(out)

    def assemble_device(self, n_sub, d_desc, d_results, d_bytes, d_payload, payload_capacity, d_offsets):
        self._check(self.L.cabac_hip_assemble_device(self.h, n_sub, vp(d_desc), vp(d_results), vp(d_bytes), vp(d_payload),
                                                     payload_capacity, vp(d_offsets)))

    def gather_records_device(self, n_seg, d_src_off, d_dst_off, d_len, d_src, d_dst):
        self._check(self.L.cabac_hip_gather_records_device(self.h, n_seg, vp(d_src_off), vp(d_dst_off), vp(d_len), vp(d_src), vp(d_dst)))

    def split_device(self, n_sub, d_desc, d_offsets, d_payload, d_bytes):
        self._check(self.L.cabac_hip_split_device(self.h, n_sub, vp(d_desc), vp(d_offsets), vp(d_payload), vp(d_bytes)))

    def count_emulations_device(self, n_sub, d_desc, d_results, d_bytes, d_counts):
        self._check(self.L.cabac_hip_count_emulations_device(self.h, n_sub, vp(d_desc), vp(d_results), vp(d_bytes),
                                                             vp(d_counts)))

    def ctx_init_device(self, n_sub, d_qp, d_init_id, d_state, d_rate):
        self._check(self.L.cabac_hip_ctx_init_device(self.h, n_sub, vp(d_qp), vp(d_init_id), vp(d_state), vp(d_rate)))
