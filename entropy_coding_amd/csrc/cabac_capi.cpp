// C ABI of libcabac_hip.so (declared in include/cabac_hip.h).  Host-side plumbing only: argument
// checks, stream/event handling, pinned staging for the host-pointer entry points.  There is no
// CPU codec in this library — without a GPU cabac_hip_init fails.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "cabac_hip.h"
#include "cabac_kernels.h"

struct cabac_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  bool timed = false;
  int enc_variant = 0, dec_variant = 0;
  std::string last_error;
  // per-launch profiling ring (cabac_hip_profile_enable)
  std::vector<hipEvent_t> prof_ev;  // 2 per slot
  std::vector<int32_t> prof_kind;
  uint32_t prof_n = 0;
  // staging for the host-pointer entry points (grown on demand)
  void *d_buf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [5]: scratch of the residual binariser
  size_t d_cap[6] = {0, 0, 0, 0, 0, 0};
};

namespace {

int fail_hip(cabac_hip_ctx *c, hipError_t e, const char *what) {
  if (c) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    c->last_error = buf;
  }
  return CABAC_HIP_ERR_HIP;
}

int fail(cabac_hip_ctx *c, int status, const char *what) {
  if (c) c->last_error = what;
  return status;
}

#define HIP_TRY(c, expr)                                   \
  do {                                                     \
    hipError_t _e = (expr);                                \
    if (_e != hipSuccess) return fail_hip((c), _e, #expr); \
  } while (0)

int ensure(cabac_hip_ctx *c, int slot, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (c->d_cap[slot] >= bytes) return CABAC_HIP_OK;
  if (c->d_buf[slot]) {
    HIP_TRY(c, hipFree(c->d_buf[slot]));
    c->d_buf[slot] = nullptr;
    c->d_cap[slot] = 0;
  }
  size_t want = bytes + bytes / 4 + 256;
  hipError_t e = hipMalloc(&c->d_buf[slot], want);
  if (e != hipSuccess) {
    c->last_error = "hipMalloc failed";
    return CABAC_HIP_ERR_NOMEM;
  }
  c->d_cap[slot] = want;
  return CABAC_HIP_OK;
}

// Event bracket of one launch: the ring slot when profiling is on, else the ctx's single pair.
struct Bracket {
  hipEvent_t a, b;
};
Bracket bracket_for(cabac_hip_ctx *c, int kind) {
  if (!c->prof_ev.empty() && c->prof_n < c->prof_kind.size()) {
    uint32_t i = c->prof_n++;
    c->prof_kind[i] = kind;
    return {c->prof_ev[2 * i], c->prof_ev[2 * i + 1]};
  }
  return {c->ev_start, c->ev_stop};
}

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int check_desc_host(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, uint64_t n_records_total,
                    uint64_t bytes_total) {
  for (uint32_t s = 0; s < n_sub; s++) {
    const cabac_substream_desc &d = desc[s];
    if (d.rec_offset > n_records_total || d.n_records > n_records_total - d.rec_offset)
      return fail(c, CABAC_HIP_ERR_INVALID, "records out of range");
    if (d.byte_offset > bytes_total || d.byte_capacity > bytes_total - d.byte_offset)
      return fail(c, CABAC_HIP_ERR_INVALID, "bytes out of range");
    if (d.byte_offset & 15u) return fail(c, CABAC_HIP_ERR_INVALID, "byte_offset must be 16-byte aligned");
    if ((d.init_id & 3u) > 2u) return fail(c, CABAC_HIP_ERR_INVALID, "init_id must be 0..2");
  }
  return CABAC_HIP_OK;
}

}  // namespace

extern "C" {

size_t cabac_hip_encode_bound(uint64_t n_ctx_bins, uint64_t n_ep_bins, uint64_t n_trm_bins) {
  // <= 6 bits per context bin (contexts.cpp:787-789), 1 per bypass bin, <= 7 per terminate bin,
  // finish() <= 2 buffered + 2 flushed bytes, alignment <= 1; rounded up to 16.
  uint64_t bits = 6 * n_ctx_bins + n_ep_bins + 7 * n_trm_bins;
  uint64_t bytes = (bits + 7) / 8 + 8;
  return (size_t)((bytes + 15) / 16 * 16);
}

int cabac_hip_init(int device, cabac_hip_ctx **out) {
  if (!out) return CABAC_HIP_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CABAC_HIP_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return CABAC_HIP_ERR_INVALID;
  cabac_hip_ctx *c = new (std::nothrow) cabac_hip_ctx;
  if (!c) return CABAC_HIP_ERR_NOMEM;
  c->device = device;
  DeviceGuard g(device);
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&c->ev_start) != hipSuccess || hipEventCreate(&c->ev_stop) != hipSuccess) {
    delete c;
    return CABAC_HIP_ERR_HIP;
  }
  c->own_stream = true;
  *out = c;
  return CABAC_HIP_OK;
}

void cabac_hip_destroy(cabac_hip_ctx *c) {
  if (!c) return;
  DeviceGuard g(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (int i = 0; i < 6; i++)
    if (c->d_buf[i]) (void)hipFree(c->d_buf[i]);
  for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
  if (c->ev_start) (void)hipEventDestroy(c->ev_start);
  if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *cabac_hip_strerror(int status) {
  switch (status) {
  case CABAC_HIP_OK: return "ok";
  case CABAC_HIP_ERR_NO_DEVICE: return "no HIP device (this library has no CPU path)";
  case CABAC_HIP_ERR_INVALID: return "invalid argument";
  case CABAC_HIP_ERR_HIP: return "HIP runtime error";
  case CABAC_HIP_ERR_NOMEM: return "out of memory";
  case CABAC_HIP_ERR_SUBSTREAM: return "a substream reported an error flag";
  default: return "unknown status";
  }
}

const char *cabac_hip_last_error(const cabac_hip_ctx *c) { return c ? c->last_error.c_str() : ""; }

int cabac_hip_set_stream(cabac_hip_ctx *c, void *hip_stream) {
  if (!c) return CABAC_HIP_ERR_INVALID;
  DeviceGuard g(c->device);
  if (c->own_stream && c->stream) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
    c->own_stream = false;
    c->stream = nullptr;
  }
  if (hip_stream) {
    c->stream = (hipStream_t)hip_stream;
  } else {
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  return CABAC_HIP_OK;
}

int cabac_hip_synchronize(cabac_hip_ctx *c) {
  if (!c) return CABAC_HIP_ERR_INVALID;
  DeviceGuard g(c->device);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CABAC_HIP_OK;
}

int cabac_hip_set_variant(cabac_hip_ctx *c, int enc, int dec) {
  if (!c) return CABAC_HIP_ERR_INVALID;
  c->enc_variant = enc;
  c->dec_variant = dec;
  return CABAC_HIP_OK;
}

int cabac_hip_encode_device(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc,
                            const uint16_t *d_records, uint8_t *d_bytes, cabac_substream_result *d_results) {
  if (!c || (n_sub && (!d_desc || !d_records || !d_bytes || !d_results))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  DeviceGuard g(c->device);
  Bracket br = bracket_for(c, 0);
  HIP_TRY(c, hipEventRecord(br.a, c->stream));
  HIP_TRY(c, cabac::launch_encode(c->stream, c->enc_variant, n_sub, d_desc, d_records, d_bytes, d_results));
  HIP_TRY(c, hipEventRecord(br.b, c->stream));
  c->timed = (br.a == c->ev_start);
  return CABAC_HIP_OK;
}

int cabac_hip_decode_device(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc,
                            const uint16_t *d_records, const uint8_t *d_bytes, uint8_t *d_bins,
                            cabac_substream_result *d_results) {
  if (!c || (n_sub && (!d_desc || !d_records || !d_bytes || !d_bins || !d_results)))
    return fail(c, CABAC_HIP_ERR_INVALID, "null");
  DeviceGuard g(c->device);
  Bracket br = bracket_for(c, 1);
  HIP_TRY(c, hipEventRecord(br.a, c->stream));
  HIP_TRY(c, cabac::launch_decode(c->stream, c->dec_variant, n_sub, d_desc, d_records, d_bytes, d_bins, d_results));
  HIP_TRY(c, hipEventRecord(br.b, c->stream));
  c->timed = (br.a == c->ev_start);
  return CABAC_HIP_OK;
}

int cabac_hip_estimate_device(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc,
                              const uint16_t *d_records, uint64_t *d_frac_bits, uint32_t *d_flags) {
  if (!c || (n_sub && (!d_desc || !d_records || !d_frac_bits))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  DeviceGuard g(c->device);
  Bracket br = bracket_for(c, 4);
  HIP_TRY(c, hipEventRecord(br.a, c->stream));
  HIP_TRY(c, cabac::launch_estimate(c->stream, n_sub, d_desc, d_records, d_frac_bits, d_flags));
  HIP_TRY(c, hipEventRecord(br.b, c->stream));
  c->timed = (br.a == c->ev_start);
  return CABAC_HIP_OK;
}

int cabac_hip_estimate_from_device(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc,
                                   const uint16_t *d_records, const uint32_t *d_state, const uint8_t *d_rate,
                                   const uint32_t *d_set, uint64_t *d_frac_bits, uint32_t *d_flags) {
  if (!c || (n_sub && (!d_desc || !d_records || !d_frac_bits || !d_state || !d_rate || !d_set)))
    return fail(c, CABAC_HIP_ERR_INVALID, "null");
  DeviceGuard g(c->device);
  Bracket br = bracket_for(c, 4);
  HIP_TRY(c, hipEventRecord(br.a, c->stream));
  HIP_TRY(c, cabac::launch_estimate(c->stream, n_sub, d_desc, d_records, d_frac_bits, d_flags, d_state, d_rate, d_set));
  HIP_TRY(c, hipEventRecord(br.b, c->stream));
  c->timed = (br.a == c->ev_start);
  return CABAC_HIP_OK;
}

int cabac_hip_ctx_init_device(cabac_hip_ctx *c, uint32_t n_sub, const int32_t *d_qp, const uint32_t *d_init_id,
                              uint32_t *d_state, uint8_t *d_rate) {
  if (!c || (n_sub && (!d_qp || !d_init_id || !d_state || !d_rate))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  DeviceGuard g(c->device);
  HIP_TRY(c, cabac::launch_ctx_init(c->stream, n_sub, d_qp, d_init_id, d_state, d_rate));
  return CABAC_HIP_OK;
}

int cabac_hip_profile_enable(cabac_hip_ctx *c, uint32_t capacity) {
  if (!c) return CABAC_HIP_ERR_INVALID;
  DeviceGuard g(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
  c->prof_ev.clear();
  c->prof_kind.clear();
  c->prof_n = 0;
  for (uint32_t i = 0; i < 2 * capacity; i++) {
    hipEvent_t e;
    HIP_TRY(c, hipEventCreate(&e));
    c->prof_ev.push_back(e);
  }
  c->prof_kind.assign(capacity, 0);
  return CABAC_HIP_OK;
}

int cabac_hip_profile_read(cabac_hip_ctx *c, int32_t *kind, float *ms, uint32_t max_entries) {
  if (!c || !kind || !ms) return CABAC_HIP_ERR_INVALID;
  DeviceGuard g(c->device);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  uint32_t n = c->prof_n < max_entries ? c->prof_n : max_entries;
  for (uint32_t i = 0; i < n; i++) {
    kind[i] = c->prof_kind[i];
    HIP_TRY(c, hipEventElapsedTime(&ms[i], c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
  }
  c->prof_n = 0;
  return (int)n;
}

int cabac_hip_assemble_device(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc,
                              const cabac_substream_result *d_results, const uint8_t *d_bytes, uint8_t *d_payload,
                              uint64_t payload_capacity, uint64_t *d_offsets) {
  if (!c || !d_offsets || (n_sub && (!d_desc || !d_results || !d_bytes || !d_payload)))
    return fail(c, CABAC_HIP_ERR_INVALID, "null");
  DeviceGuard g(c->device);
  Bracket br = bracket_for(c, 6);
  HIP_TRY(c, hipEventRecord(br.a, c->stream));
  HIP_TRY(c, cabac::launch_assemble(c->stream, n_sub, d_desc, d_results, d_bytes, d_payload, payload_capacity, d_offsets));
  HIP_TRY(c, hipEventRecord(br.b, c->stream));
  c->timed = (br.a == c->ev_start);
  return CABAC_HIP_OK;
}

int cabac_hip_split_device(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc, const uint64_t *d_offsets,
                           const uint8_t *d_payload, uint8_t *d_bytes) {
  if (!c || (n_sub && (!d_desc || !d_offsets || !d_payload || !d_bytes))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  DeviceGuard g(c->device);
  Bracket br = bracket_for(c, 7);
  HIP_TRY(c, hipEventRecord(br.a, c->stream));
  HIP_TRY(c, cabac::launch_split(c->stream, n_sub, d_desc, d_offsets, d_payload, d_bytes));
  HIP_TRY(c, hipEventRecord(br.b, c->stream));
  c->timed = (br.a == c->ev_start);
  return CABAC_HIP_OK;
}

int cabac_hip_count_emulations_device(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc,
                                      const cabac_substream_result *d_results, const uint8_t *d_bytes,
                                      uint32_t *d_counts) {
  if (!c || (n_sub && (!d_desc || !d_results || !d_bytes || !d_counts))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  DeviceGuard g(c->device);
  Bracket br = bracket_for(c, 8);
  HIP_TRY(c, hipEventRecord(br.a, c->stream));
  HIP_TRY(c, cabac::launch_count_emulations(c->stream, n_sub, d_desc, d_results, d_bytes, d_counts));
  HIP_TRY(c, hipEventRecord(br.b, c->stream));
  c->timed = (br.a == c->ev_start);
  return CABAC_HIP_OK;
}

float cabac_hip_last_kernel_ms(cabac_hip_ctx *c) {
  if (!c || !c->timed) return -1.0f;
  DeviceGuard g(c->device);
  if (hipEventSynchronize(c->ev_stop) != hipSuccess) return -1.0f;
  float ms = -1.0f;
  if (hipEventElapsedTime(&ms, c->ev_start, c->ev_stop) != hipSuccess) return -1.0f;
  return ms;
}

int cabac_hip_encode_batch(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                           uint64_t n_records_total, uint8_t *bytes, uint64_t bytes_total,
                           cabac_substream_result *results) {
  if (!c || (n_sub && (!desc || !results))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  if (n_sub == 0) return CABAC_HIP_OK;
  int rc = check_desc_host(c, n_sub, desc, n_records_total, bytes_total);
  if (rc) return rc;
  DeviceGuard g(c->device);
  if ((rc = ensure(c, 0, n_sub * sizeof(cabac_substream_desc)))) return rc;
  if ((rc = ensure(c, 1, n_records_total * 2))) return rc;
  if ((rc = ensure(c, 2, bytes_total))) return rc;
  if ((rc = ensure(c, 3, n_sub * sizeof(cabac_substream_result)))) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->d_buf[0], desc, n_sub * sizeof(cabac_substream_desc), hipMemcpyHostToDevice, c->stream));
  if (n_records_total)
    HIP_TRY(c, hipMemcpyAsync(c->d_buf[1], records, n_records_total * 2, hipMemcpyHostToDevice, c->stream));
  rc = cabac_hip_encode_device(c, n_sub, (const cabac_substream_desc *)c->d_buf[0], (const uint16_t *)c->d_buf[1],
                               (uint8_t *)c->d_buf[2], (cabac_substream_result *)c->d_buf[3]);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(results, c->d_buf[3], n_sub * sizeof(cabac_substream_result), hipMemcpyDeviceToHost,
                            c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  // copy back only what was produced, substream by substream (payload is ~0.1 B/bin)
  int status = CABAC_HIP_OK;
  for (uint32_t s = 0; s < n_sub; s++) {
    size_t nbytes = (results[s].n_bits + 7) / 8;
    if (nbytes > desc[s].byte_capacity) nbytes = desc[s].byte_capacity;
    if (nbytes)
      HIP_TRY(c, hipMemcpyAsync(bytes + desc[s].byte_offset, (uint8_t *)c->d_buf[2] + desc[s].byte_offset, nbytes,
                                hipMemcpyDeviceToHost, c->stream));
    if (results[s].flags) status = CABAC_HIP_ERR_SUBSTREAM;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (status) c->last_error = "substream flag set (see results[].flags)";
  return status;
}

int cabac_hip_estimate_batch(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                             uint64_t n_records_total, uint64_t *frac_bits, uint32_t *flags) {
  if (!c || (n_sub && (!desc || !frac_bits))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  if (n_sub == 0) return CABAC_HIP_OK;
  for (uint32_t s = 0; s < n_sub; s++) {
    if (desc[s].rec_offset > n_records_total || desc[s].n_records > n_records_total - desc[s].rec_offset)
      return fail(c, CABAC_HIP_ERR_INVALID, "records out of range");
    if ((desc[s].init_id & 3u) > 2u) return fail(c, CABAC_HIP_ERR_INVALID, "init_id must be 0..2");
  }
  DeviceGuard g(c->device);
  int rc;
  if ((rc = ensure(c, 0, n_sub * sizeof(cabac_substream_desc)))) return rc;
  if ((rc = ensure(c, 1, n_records_total * 2))) return rc;
  if ((rc = ensure(c, 3, n_sub * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, 4, n_sub * sizeof(uint32_t)))) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->d_buf[0], desc, n_sub * sizeof(cabac_substream_desc), hipMemcpyHostToDevice, c->stream));
  if (n_records_total)
    HIP_TRY(c, hipMemcpyAsync(c->d_buf[1], records, n_records_total * 2, hipMemcpyHostToDevice, c->stream));
  rc = cabac_hip_estimate_device(c, n_sub, (const cabac_substream_desc *)c->d_buf[0], (const uint16_t *)c->d_buf[1],
                                 (uint64_t *)c->d_buf[3], (uint32_t *)c->d_buf[4]);
  if (rc) return rc;
  std::vector<uint32_t> fl(n_sub);
  HIP_TRY(c, hipMemcpyAsync(frac_bits, c->d_buf[3], n_sub * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(fl.data(), c->d_buf[4], n_sub * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int status = CABAC_HIP_OK;
  for (uint32_t s = 0; s < n_sub; s++) {
    if (flags) flags[s] = fl[s];
    if (fl[s]) status = CABAC_HIP_ERR_SUBSTREAM;
  }
  if (status) c->last_error = "substream flag set (see flags[])";
  return status;
}

int cabac_hip_decode_batch(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                           uint64_t n_records_total, const uint8_t *bytes, uint64_t bytes_total, uint8_t *bins,
                           cabac_substream_result *results) {
  if (!c || (n_sub && (!desc || !results))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  if (n_sub == 0) return CABAC_HIP_OK;
  int rc = check_desc_host(c, n_sub, desc, n_records_total, bytes_total);
  if (rc) return rc;
  DeviceGuard g(c->device);
  if ((rc = ensure(c, 0, n_sub * sizeof(cabac_substream_desc)))) return rc;
  if ((rc = ensure(c, 1, n_records_total * 2))) return rc;
  if ((rc = ensure(c, 2, bytes_total))) return rc;
  if ((rc = ensure(c, 3, n_sub * sizeof(cabac_substream_result)))) return rc;
  if ((rc = ensure(c, 4, n_records_total))) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->d_buf[0], desc, n_sub * sizeof(cabac_substream_desc), hipMemcpyHostToDevice, c->stream));
  if (n_records_total)
    HIP_TRY(c, hipMemcpyAsync(c->d_buf[1], records, n_records_total * 2, hipMemcpyHostToDevice, c->stream));
  if (bytes_total) HIP_TRY(c, hipMemcpyAsync(c->d_buf[2], bytes, bytes_total, hipMemcpyHostToDevice, c->stream));
  rc = cabac_hip_decode_device(c, n_sub, (const cabac_substream_desc *)c->d_buf[0], (const uint16_t *)c->d_buf[1],
                               (const uint8_t *)c->d_buf[2], (uint8_t *)c->d_buf[4],
                               (cabac_substream_result *)c->d_buf[3]);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(results, c->d_buf[3], n_sub * sizeof(cabac_substream_result), hipMemcpyDeviceToHost,
                            c->stream));
  if (n_records_total)
    HIP_TRY(c, hipMemcpyAsync(bins, c->d_buf[4], n_records_total, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (uint32_t s = 0; s < n_sub; s++)
    if (results[s].flags) {
      c->last_error = "substream flag set (see results[].flags)";
      return CABAC_HIP_ERR_SUBSTREAM;
    }
  return CABAC_HIP_OK;
}

int cabac_hip_binarize_device(cabac_hip_ctx *c, uint32_t n_sub, const uint64_t *d_se_offset, const uint32_t *d_se,
                              const uint64_t *d_rec_offset, uint32_t *d_n_records, uint16_t *d_records) {
  if (!c || (n_sub && (!d_se_offset || !d_se || !d_n_records || (d_records && !d_rec_offset))))
    return fail(c, CABAC_HIP_ERR_INVALID, "null");
  DeviceGuard g(c->device);
  Bracket br = bracket_for(c, 2);
  HIP_TRY(c, hipEventRecord(br.a, c->stream));
  HIP_TRY(c, cabac::launch_binarize(c->stream, n_sub, d_se_offset, d_se, d_rec_offset, d_n_records, d_records));
  HIP_TRY(c, hipEventRecord(br.b, c->stream));
  c->timed = (br.a == c->ev_start);
  return CABAC_HIP_OK;
}

int cabac_hip_residual_device(cabac_hip_ctx *c, uint32_t n_tu, const cabac_tu_desc *d_tu, const int32_t *d_coeff,
                              const uint64_t *d_rec_offset, uint32_t *d_n_records, uint32_t *d_info,
                              uint16_t *d_records) {
  if (!c || (n_tu && (!d_tu || !d_coeff || !d_n_records || (d_records && !d_rec_offset))))
    return fail(c, CABAC_HIP_ERR_INVALID, "null");
  DeviceGuard g(c->device);
  if (int rc = ensure(c, 5, cabac::residual_scratch_bytes(n_tu))) return rc;
  Bracket br = bracket_for(c, 5);
  HIP_TRY(c, hipEventRecord(br.a, c->stream));
  HIP_TRY(c, cabac::launch_residual(c->stream, n_tu, d_tu, d_coeff, d_rec_offset, d_n_records, d_info, d_records,
                                    c->d_buf[5]));
  HIP_TRY(c, hipEventRecord(br.b, c->stream));
  c->timed = (br.a == c->ev_start);
  return CABAC_HIP_OK;
}

int cabac_hip_residual_batch(cabac_hip_ctx *c, uint32_t n_tu, const cabac_tu_desc *tus, const int32_t *coeff,
                             uint64_t n_coeff_total, uint64_t *offsets, uint32_t *info, uint16_t *records,
                             uint64_t records_capacity) {
  if (!c || !offsets || (n_tu && (!tus || !coeff))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  offsets[0] = 0;
  if (n_tu == 0) return CABAC_HIP_OK;
  for (uint32_t t = 0; t < n_tu; t++) {
    if (tus[t].log2_width > 6 || tus[t].log2_height > 6) continue;  // flagged by the kernel, reads nothing
    const uint64_t n = uint64_t(1) << (tus[t].log2_width + tus[t].log2_height);
    if (tus[t].coeff_offset > n_coeff_total || n > n_coeff_total - tus[t].coeff_offset)
      return fail(c, CABAC_HIP_ERR_INVALID, "coefficients out of range");
  }
  DeviceGuard g(c->device);
  int rc;
  // staging: [0] descriptors, [1] coefficients, [3] record offsets, [4] counts then info, [2] records
  if ((rc = ensure(c, 0, n_tu * sizeof(cabac_tu_desc)))) return rc;
  if ((rc = ensure(c, 1, n_coeff_total * sizeof(int32_t)))) return rc;
  if ((rc = ensure(c, 3, n_tu * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, 4, 2 * size_t(n_tu) * sizeof(uint32_t)))) return rc;
  uint32_t *d_cnt = static_cast<uint32_t *>(c->d_buf[4]), *d_info = d_cnt + n_tu;
  HIP_TRY(c, hipMemcpyAsync(c->d_buf[0], tus, n_tu * sizeof(cabac_tu_desc), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_buf[1], coeff, n_coeff_total * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  rc = cabac_hip_residual_device(c, n_tu, (const cabac_tu_desc *)c->d_buf[0], (const int32_t *)c->d_buf[1], nullptr, d_cnt,
                                 d_info, nullptr);
  if (rc) return rc;
  std::vector<uint32_t> cnt(2 * size_t(n_tu));
  HIP_TRY(c, hipMemcpyAsync(cnt.data(), d_cnt, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int status = CABAC_HIP_OK;
  for (uint32_t t = 0; t < n_tu; t++) {
    offsets[t + 1] = offsets[t] + cnt[t];
    if (info) info[t] = cnt[n_tu + t];
    if (cnt[n_tu + t] & (CABAC_TU_INFO_EMPTY | CABAC_TU_INFO_BAD_DESC)) status = CABAC_HIP_ERR_SUBSTREAM;
  }
  if (status) c->last_error = "empty block or bad descriptor (see info[])";
  if (!records) return status;
  if (records_capacity < offsets[n_tu]) return fail(c, CABAC_HIP_ERR_INVALID, "records_capacity too small");
  if (offsets[n_tu] == 0) return status;
  if ((rc = ensure(c, 2, offsets[n_tu] * sizeof(uint16_t)))) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->d_buf[3], offsets, n_tu * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  rc = cabac_hip_residual_device(c, n_tu, (const cabac_tu_desc *)c->d_buf[0], (const int32_t *)c->d_buf[1],
                                 (const uint64_t *)c->d_buf[3], d_cnt, d_info, (uint16_t *)c->d_buf[2]);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(records, c->d_buf[2], offsets[n_tu] * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return status;
}

int cabac_hip_residual_parse_device(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc,
                                    const uint8_t *d_bytes, const uint32_t *d_tile_first, const cabac_tu_desc *d_tu,
                                    int32_t *d_coeff, cabac_substream_result *d_results) {
  if (!c || (n_sub && (!d_desc || !d_bytes || !d_tile_first || !d_tu || !d_coeff || !d_results)))
    return fail(c, CABAC_HIP_ERR_INVALID, "null");
  DeviceGuard g(c->device);
  Bracket br = bracket_for(c, 9);
  HIP_TRY(c, hipEventRecord(br.a, c->stream));
  HIP_TRY(c, cabac::launch_residual_parse(c->stream, n_sub, d_desc, d_bytes, d_tile_first, d_tu, d_coeff, d_results));
  HIP_TRY(c, hipEventRecord(br.b, c->stream));
  c->timed = (br.a == c->ev_start);
  return CABAC_HIP_OK;
}

int cabac_hip_residual_parse_batch(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint8_t *bytes,
                                   uint64_t bytes_total, const uint32_t *tile_first, const cabac_tu_desc *tus, int32_t *coeff,
                                   uint64_t n_coeff_total, cabac_substream_result *results) {
  if (!c || (n_sub && (!desc || !bytes || !tile_first || !tus || !coeff || !results))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  if (n_sub == 0) return CABAC_HIP_OK;
  const uint32_t n_tu = tile_first[n_sub];
  for (uint32_t s = 0; s < n_sub; s++) {
    if (tile_first[s] > tile_first[s + 1]) return fail(c, CABAC_HIP_ERR_INVALID, "tile_first must not decrease");
    if (desc[s].byte_offset > bytes_total || desc[s].byte_capacity > bytes_total - desc[s].byte_offset)
      return fail(c, CABAC_HIP_ERR_INVALID, "bytes out of range");
    if ((desc[s].init_id & 3u) > 2u) return fail(c, CABAC_HIP_ERR_INVALID, "init_id must be 0..2");
  }
  for (uint32_t t = 0; t < n_tu; t++) {
    if (tus[t].log2_width > 6 || tus[t].log2_height > 6) continue;  // flagged by the kernel, writes nothing
    const uint64_t n = uint64_t(1) << (tus[t].log2_width + tus[t].log2_height);
    if (tus[t].coeff_offset > n_coeff_total || n > n_coeff_total - tus[t].coeff_offset)
      return fail(c, CABAC_HIP_ERR_INVALID, "coefficients out of range");
  }
  DeviceGuard g(c->device);
  int rc;
  // staging: [0] substream descriptors, [2] bytes, [3] tile_first then block descriptors, [1] coefficients, [4] results
  const size_t first_bytes = (size_t(n_sub) + 1) * sizeof(uint32_t), first_pad = (first_bytes + 15) / 16 * 16;
  if ((rc = ensure(c, 0, n_sub * sizeof(cabac_substream_desc)))) return rc;
  if ((rc = ensure(c, 2, bytes_total))) return rc;
  if ((rc = ensure(c, 3, first_pad + size_t(n_tu) * sizeof(cabac_tu_desc)))) return rc;
  if ((rc = ensure(c, 1, n_coeff_total * sizeof(int32_t)))) return rc;
  if ((rc = ensure(c, 4, n_sub * sizeof(cabac_substream_result)))) return rc;
  uint8_t *d_first = static_cast<uint8_t *>(c->d_buf[3]);
  HIP_TRY(c, hipMemcpyAsync(c->d_buf[0], desc, n_sub * sizeof(cabac_substream_desc), hipMemcpyHostToDevice, c->stream));
  if (bytes_total) HIP_TRY(c, hipMemcpyAsync(c->d_buf[2], bytes, bytes_total, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d_first, tile_first, first_bytes, hipMemcpyHostToDevice, c->stream));
  if (n_tu) HIP_TRY(c, hipMemcpyAsync(d_first + first_pad, tus, size_t(n_tu) * sizeof(cabac_tu_desc), hipMemcpyHostToDevice, c->stream));
  // what the parser does not write (outside the coded region of 64-wide blocks) keeps the caller's values
  if (n_coeff_total) HIP_TRY(c, hipMemcpyAsync(c->d_buf[1], coeff, n_coeff_total * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  rc = cabac_hip_residual_parse_device(c, n_sub, (const cabac_substream_desc *)c->d_buf[0], (const uint8_t *)c->d_buf[2],
                                       (const uint32_t *)d_first, (const cabac_tu_desc *)(d_first + first_pad),
                                       (int32_t *)c->d_buf[1], (cabac_substream_result *)c->d_buf[4]);
  if (rc) return rc;
  if (n_coeff_total) HIP_TRY(c, hipMemcpyAsync(coeff, c->d_buf[1], n_coeff_total * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(results, c->d_buf[4], n_sub * sizeof(cabac_substream_result), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int status = CABAC_HIP_OK;
  for (uint32_t s = 0; s < n_sub; s++)
    if (results[s].flags) status = CABAC_HIP_ERR_SUBSTREAM;
  if (status) c->last_error = "substream flag set (see results[].flags)";
  return status;
}

}  // extern "C"
