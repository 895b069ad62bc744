// C ABI of libcabac_hip.so (declared in include/cabac_hip.h).  Host-side plumbing only: argument
// checks, stream/event handling, and the host-pointer entry points' path over PCIe: pinned host memory
// (cabac_hip_host_alloc / _register) is DMA'd directly, pageable memory goes through a ring of pinned bounce
// blocks, a batch is cut into chunks whose H2D copy, kernel and D2H copy run on separate streams, and coded
// substreams leave the device compacted (cabac_assemble.hip) in one copy per chunk.  There is no
// CPU codec in this library — without a GPU cabac_hip_init fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "cabac_hip.h"
#include "cabac_kernels.h"

#ifdef CABAC_PARSE_PROFILE
namespace cabac {
hipError_t debug_read_parse_prof(unsigned long long *out);
hipError_t debug_read_parse_waves(unsigned long long *out);
}
#endif

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
  // device staging for the host-pointer entry points (grown on demand)
  // [5]: scratch of the residual binariser, [6]: compacted payload, [7]: payload offsets; [8..]: the spliced-residual path
  // (kSp* below)
  static constexpr int kSlots = 24;
  void *d_buf[kSlots] = {};
  size_t d_cap[kSlots] = {};
  void *h_totals = nullptr;  // pinned, 64 bytes: what the spliced-residual path reads back in the middle
  uint32_t *d_select = nullptr;  // kMaxChunks + 1 device words: the decode dispatch's on-device choice, one per launch in flight
  // ---- the PCIe path of the host-pointer entry points -------------------------------------------------
  static constexpr int kKernelStreams = 4, kMaxChunks = 8, kBounceDepth = 4;
  static constexpr size_t kBounceBlock = size_t(4) << 20;
  hipStream_t s_in = nullptr, s_out = nullptr, s_k[kKernelStreams] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_in[kMaxChunks] = {}, ev_k[kMaxChunks] = {}, ev_out[kMaxChunks] = {};
  struct Bounce {  // ring of pinned blocks between pageable caller memory and the DMA engines
    void *blk[kBounceDepth] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[kBounceDepth] = {nullptr, nullptr, nullptr, nullptr};
    bool busy[kBounceDepth] = {false, false, false, false};
    void *dst[kBounceDepth] = {nullptr, nullptr, nullptr, nullptr};  // D2H: where the block goes once it has arrived
    size_t len[kBounceDepth] = {0, 0, 0, 0};
    int next = 0;
  } bounce_in, bounce_out;
  void *h_pin[2] = {nullptr, nullptr};  // pinned: [0] results + payload offsets coming back, [1] compacted payload
  size_t h_cap[2] = {0, 0};
  bool pipe_ready = false;
  int chunks_override = 0;  // CABAC_HIP_CHUNKS (experiments); 0 = by batch size
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


// ---- pinned host memory handed out by cabac_hip_host_alloc / pinned in place by cabac_hip_host_register ----
std::mutex g_host_mu;
std::unordered_map<const void *, size_t> g_host_owned, g_host_registered;

bool host_is_pinned(const void *p, size_t bytes) {
  if (!p) return false;
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    for (const auto *m : {&g_host_owned, &g_host_registered})
      for (const auto &kv : *m) {
        const uint8_t *b = static_cast<const uint8_t *>(kv.first), *q = static_cast<const uint8_t *>(p);
        if (q >= b && q + bytes <= b + kv.second) return true;
      }
  }
  // pinned by somebody else (e.g. torch's pin_memory): the first AND the last byte must be host memory the runtime knows, inside
  // ONE allocation whose extent the runtime can tell — a range that starts in a pinned allocation and runs past its end, or
  // whose end cannot be established, is treated as pageable (it then goes through the bounce ring, which is always safe)
  const uint8_t *q = static_cast<const uint8_t *>(p);
  hipPointerAttribute_t a, z;
  if (hipPointerGetAttributes(&a, q) != hipSuccess || hipPointerGetAttributes(&z, q + (bytes ? bytes - 1 : 0)) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (a.type != hipMemoryTypeHost || z.type != hipMemoryTypeHost) return false;
  hipDeviceptr_t base = nullptr;
  size_t extent = 0;
  if (hipMemGetAddressRange(&base, &extent, const_cast<uint8_t *>(q)) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  const uint8_t *b = static_cast<const uint8_t *>(base);
  return q >= b && q + bytes <= b + extent;
}

// Nothing of the ctx's own streams is in flight afterwards and no bounce block is waiting to be handed over: the state a
// host-pointer call must leave behind when it returns early (the blocks' destinations point into the caller's buffers,
// which the caller may free as soon as the call has returned — they are dropped, not copied).
void pipe_quiesce(cabac_hip_ctx *c) {
  (void)hipStreamSynchronize(c->stream);  // null = the adopted default stream: synchronised like any other
  for (hipStream_t st : {c->s_in, c->s_out, c->s_k[0], c->s_k[1], c->s_k[2], c->s_k[3]})
    if (st) (void)hipStreamSynchronize(st);
  for (cabac_hip_ctx::Bounce *b : {&c->bounce_in, &c->bounce_out})
    for (int i = 0; i < cabac_hip_ctx::kBounceDepth; i++) {
      b->busy[i] = false;
      b->dst[i] = nullptr;
      b->len[i] = 0;
    }
}

// the exit of every host-pointer entry point: an error other than "a substream has a flag set" may have come from the middle
// of the pipeline
int host_call_exit(cabac_hip_ctx *c, int rc) {
  if (c && c->pipe_ready && rc != CABAC_HIP_OK && rc != CABAC_HIP_ERR_SUBSTREAM) {
    const std::string keep = c->last_error;
    DeviceGuard g(c->device);
    pipe_quiesce(c);
    (void)hipGetLastError();
    c->last_error = keep;
  }
  return rc;
}

void pipe_destroy(cabac_hip_ctx *c) {
  pipe_quiesce(c);
  for (hipStream_t st : {c->s_in, c->s_out, c->s_k[0], c->s_k[1], c->s_k[2], c->s_k[3]})
    if (st) (void)hipStreamDestroy(st);
  for (int i = 0; i < cabac_hip_ctx::kMaxChunks; i++)
    for (hipEvent_t e : {c->ev_in[i], c->ev_k[i], c->ev_out[i]})
      if (e) (void)hipEventDestroy(e);
  for (cabac_hip_ctx::Bounce *b : {&c->bounce_in, &c->bounce_out})
    for (int i = 0; i < cabac_hip_ctx::kBounceDepth; i++) {
      if (b->blk[i]) (void)hipHostFree(b->blk[i]);
      if (b->ev[i]) (void)hipEventDestroy(b->ev[i]);
    }
  for (void *p : c->h_pin)
    if (p) (void)hipHostFree(p);
  c->pipe_ready = false;
}

int pipe_init(cabac_hip_ctx *c) {
  if (c->pipe_ready) return CABAC_HIP_OK;
  HIP_TRY(c, hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
  HIP_TRY(c, hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
  for (hipStream_t &st : c->s_k) HIP_TRY(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (int i = 0; i < cabac_hip_ctx::kMaxChunks; i++) {
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_k[i], hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming));
  }
  for (cabac_hip_ctx::Bounce *b : {&c->bounce_in, &c->bounce_out})
    for (int i = 0; i < cabac_hip_ctx::kBounceDepth; i++) {
      HIP_TRY(c, hipHostMalloc(&b->blk[i], cabac_hip_ctx::kBounceBlock, hipHostMallocDefault));
      HIP_TRY(c, hipEventCreateWithFlags(&b->ev[i], hipEventDisableTiming));
    }
  if (const char *e = getenv("CABAC_HIP_CHUNKS")) c->chunks_override = atoi(e);
  c->pipe_ready = true;
  return CABAC_HIP_OK;
}

int ensure_pinned(cabac_hip_ctx *c, int slot, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (c->h_cap[slot] >= bytes) return CABAC_HIP_OK;
  if (c->h_pin[slot]) {
    HIP_TRY(c, hipHostFree(c->h_pin[slot]));
    c->h_pin[slot] = nullptr;
    c->h_cap[slot] = 0;
  }
  const size_t want = bytes + bytes / 4 + 4096;
  if (hipHostMalloc(&c->h_pin[slot], want, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    c->last_error = "hipHostMalloc failed";
    return CABAC_HIP_ERR_NOMEM;
  }
  c->h_cap[slot] = want;
  return CABAC_HIP_OK;
}

// grow pinned slot `slot` to `bytes`, keeping its first `keep` bytes (copies into them may still be in flight on s_out)
int ensure_pinned_keep(cabac_hip_ctx *c, int slot, size_t bytes, size_t keep) {
  if (c->h_cap[slot] >= bytes && c->h_pin[slot]) return CABAC_HIP_OK;
  void *old = c->h_pin[slot];
  if (old && keep) HIP_TRY(c, hipStreamSynchronize(c->s_out));
  c->h_pin[slot] = nullptr;
  c->h_cap[slot] = 0;
  int rc = ensure_pinned(c, slot, bytes + bytes);  // chunks are about equal: leave room for the ones to come
  if (rc == CABAC_HIP_OK && old && keep) std::memcpy(c->h_pin[slot], old, keep);
  if (old) (void)hipHostFree(old);
  return rc;
}

// host -> device on `st`: pinned memory is DMA'd where it lies; pageable memory is copied block by block into the
// pinned ring, each block's DMA overlapping the host copy of the next one
int h2d(cabac_hip_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t st) {
  if (bytes == 0) return CABAC_HIP_OK;
  if (host_is_pinned(src, bytes)) {
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    return CABAC_HIP_OK;
  }
  cabac_hip_ctx::Bounce &b = c->bounce_in;
  for (size_t off = 0; off < bytes; off += cabac_hip_ctx::kBounceBlock) {
    const size_t n = std::min(cabac_hip_ctx::kBounceBlock, bytes - off);
    const int i = b.next;
    b.next = (i + 1) % cabac_hip_ctx::kBounceDepth;
    if (b.busy[i]) HIP_TRY(c, hipEventSynchronize(b.ev[i]));
    std::memcpy(b.blk[i], static_cast<const uint8_t *>(src) + off, n);
    HIP_TRY(c, hipMemcpyAsync(static_cast<uint8_t *>(dst) + off, b.blk[i], n, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipEventRecord(b.ev[i], st));
    b.busy[i] = true;
  }
  return CABAC_HIP_OK;
}

// one block of the outgoing ring has arrived: hand it to the caller's memory
int d2h_retire(cabac_hip_ctx *c, int i) {
  cabac_hip_ctx::Bounce &b = c->bounce_out;
  if (!b.busy[i]) return CABAC_HIP_OK;
  HIP_TRY(c, hipEventSynchronize(b.ev[i]));
  std::memcpy(b.dst[i], b.blk[i], b.len[i]);
  b.busy[i] = false;
  return CABAC_HIP_OK;
}

// device -> host on `st`; for pageable memory the data is complete only after d2h_drain
int d2h(cabac_hip_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t st) {
  if (bytes == 0) return CABAC_HIP_OK;
  if (host_is_pinned(dst, bytes)) {
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
    return CABAC_HIP_OK;
  }
  cabac_hip_ctx::Bounce &b = c->bounce_out;
  for (size_t off = 0; off < bytes; off += cabac_hip_ctx::kBounceBlock) {
    const size_t n = std::min(cabac_hip_ctx::kBounceBlock, bytes - off);
    const int i = b.next;
    b.next = (i + 1) % cabac_hip_ctx::kBounceDepth;
    if (int rc = d2h_retire(c, i)) return rc;
    HIP_TRY(c, hipMemcpyAsync(b.blk[i], static_cast<const uint8_t *>(src) + off, n, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipEventRecord(b.ev[i], st));
    b.busy[i] = true;
    b.dst[i] = static_cast<uint8_t *>(dst) + off;
    b.len[i] = n;
  }
  return CABAC_HIP_OK;
}

int d2h_drain(cabac_hip_ctx *c) {
  for (int k = 0; k < cabac_hip_ctx::kBounceDepth; k++) {
    const int i = (c->bounce_out.next + k) % cabac_hip_ctx::kBounceDepth;  // oldest first
    if (int rc = d2h_retire(c, i)) return rc;
  }
  return CABAC_HIP_OK;
}

// A batch cut at substream boundaries into chunks of about equal record counts.  Chunks need the substreams' record
// and byte slots in ascending, non-overlapping order (what every packer produces); otherwise the batch is one chunk.
struct Chunk {
  uint32_t s0, s1;
  uint64_t rec_lo, rec_hi, byte_lo, byte_hi;
};

std::vector<Chunk> plan_chunks(const cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc) {
  uint64_t rec_end = 0, byte_end = 0, total = 0;
  bool ordered = true;
  for (uint32_t s = 0; s < n_sub; s++) {
    ordered = ordered && desc[s].rec_offset >= rec_end && desc[s].byte_offset >= byte_end;
    rec_end = std::max<uint64_t>(rec_end, desc[s].rec_offset + desc[s].n_records);
    byte_end = std::max<uint64_t>(byte_end, desc[s].byte_offset + desc[s].byte_capacity);
    total += desc[s].n_records;
  }
  // The kernels' time does not shrink below ~1 024 substreams (it is the length of the serial chains), and kernels of
  // different chunks were measured to run one after the other even on separate streams, so a batch of c chunks costs about
  // copy / c + c x kernel + tail: with the C4 batch (2.35 ms of H2D at 57 GB/s, 0.9 ms of kernel) two chunks are the
  // minimum (measured: 1 chunk 4.26 ms, 2 chunks 4.08, 4 chunks 4.86, 8 chunks 6.4).  Default: two chunks from 2 048
  // substreams and 16 M records up, else one; CABAC_HIP_CHUNKS overrides.
  int want = c->chunks_override > 0 ? c->chunks_override : (n_sub >= 2048u && (total >> 24) != 0u ? 2 : 1);
  want = std::max(1, std::min(want, (int)cabac_hip_ctx::kMaxChunks));
  if (!ordered || (uint32_t)want > n_sub) want = 1;
  std::vector<Chunk> out;
  uint32_t s = 0;
  uint64_t done = 0;
  for (int k = 0; k < want; k++) {
    Chunk ch;
    ch.s0 = s;
    const uint64_t goal = total * uint64_t(k + 1) / uint64_t(want);
    while (s < n_sub && (done < goal || k == want - 1)) done += desc[s++].n_records;
    if (k == want - 1) s = n_sub;
    ch.s1 = s;
    if (ch.s1 == ch.s0) continue;
    ch.rec_lo = ch.byte_lo = ~0ull;
    ch.rec_hi = ch.byte_hi = 0;
    for (uint32_t q = ch.s0; q < ch.s1; q++) {
      ch.rec_lo = std::min<uint64_t>(ch.rec_lo, desc[q].rec_offset);
      ch.rec_hi = std::max<uint64_t>(ch.rec_hi, desc[q].rec_offset + desc[q].n_records);
      ch.byte_lo = std::min<uint64_t>(ch.byte_lo, desc[q].byte_offset);
      ch.byte_hi = std::max<uint64_t>(ch.byte_hi, desc[q].byte_offset + desc[q].byte_capacity);
    }
    out.push_back(ch);
  }
  return out;
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
  if (hipMalloc(&c->d_select, 256) != hipSuccess) {
    (void)hipGetLastError();
    cabac_hip_destroy(c);
    return CABAC_HIP_ERR_NOMEM;
  }
  *out = c;
  return CABAC_HIP_OK;
}

void cabac_hip_destroy(cabac_hip_ctx *c) {
  if (!c) return;
  DeviceGuard g(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (int i = 0; i < cabac_hip_ctx::kSlots; i++)
    if (c->d_buf[i]) (void)hipFree(c->d_buf[i]);
  if (c->h_totals) (void)hipHostFree(c->h_totals);
  if (c->d_select) (void)hipFree(c->d_select);
  pipe_destroy(c);
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
  if (hip_stream == CABAC_HIP_STREAM_DEFAULT) {
    c->stream = nullptr;  // the device's default stream (every launch, event and wait below takes a null stream as that)
  } else if (hip_stream) {
    c->stream = (hipStream_t)hip_stream;
  } else {
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  return CABAC_HIP_OK;
}

int cabac_hip_wait_event(cabac_hip_ctx *c, void *hip_event) {
  if (!c || !hip_event) return CABAC_HIP_ERR_INVALID;
  DeviceGuard g(c->device);
  HIP_TRY(c, hipStreamWaitEvent(c->stream, (hipEvent_t)hip_event, 0));
  return CABAC_HIP_OK;
}

int cabac_hip_record_event(cabac_hip_ctx *c, void *hip_event) {
  if (!c || !hip_event) return CABAC_HIP_ERR_INVALID;
  DeviceGuard g(c->device);
  HIP_TRY(c, hipEventRecord((hipEvent_t)hip_event, c->stream));
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
  HIP_TRY(c, cabac::launch_decode(c->stream, c->dec_variant, n_sub, d_desc, d_records, d_bytes, d_bins, d_results, 0, c->d_select));
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

int cabac_hip_gather_records_device(cabac_hip_ctx *c, uint32_t n_seg, const uint64_t *d_src_off, const uint64_t *d_dst_off,
                                    const uint32_t *d_len, const uint16_t *d_src, uint16_t *d_dst) {
  if (!c || (n_seg && (!d_src_off || !d_dst_off || !d_len || !d_src || !d_dst))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  DeviceGuard g(c->device);
  HIP_TRY(c, cabac::launch_gather_records(c->stream, n_seg, d_src_off, d_dst_off, d_len, d_src, d_dst));
  c->timed = false;
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

// bytes != NULL: every substream's bytes at its byte_offset (the reference's FIFOs); payload != NULL: the substreams back to
// back in descriptor order with payload_offsets[0..n_sub] (what OutputBitstream::addSubstream makes of byte-aligned
// substreams, bit_stream.cpp:139-150)
static int encode_batch_impl(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                             uint64_t n_records_total, uint8_t *bytes, uint64_t bytes_total, uint8_t *payload,
                             uint64_t payload_capacity, uint64_t *payload_offsets, cabac_substream_result *results) {
  if (!c || (n_sub && (!desc || !results))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  if (payload_offsets) payload_offsets[0] = 0;
  if (n_sub == 0) return CABAC_HIP_OK;
  int rc = check_desc_host(c, n_sub, desc, n_records_total, bytes_total);
  if (rc) return rc;
  DeviceGuard g(c->device);
  if ((rc = pipe_init(c))) return rc;
  const std::vector<Chunk> chunks = plan_chunks(c, n_sub, desc);
  const uint32_t nc = uint32_t(chunks.size());
  // device: [0] descriptors, [1] records, [2] byte slots, [3] results, [6] compacted payload (chunk k from its byte_lo),
  // [7] payload offsets (chunk k: n_k + 1 entries from s0 + k).  pinned: [0] results then offsets, [1] payload.
  const size_t res_bytes = size_t(n_sub) * sizeof(cabac_substream_result);
  const size_t off_count = size_t(n_sub) + nc;
  if ((rc = ensure(c, 0, n_sub * sizeof(cabac_substream_desc)))) return rc;
  if ((rc = ensure(c, 1, n_records_total * 2))) return rc;
  if ((rc = ensure(c, 2, bytes_total))) return rc;
  if ((rc = ensure(c, 3, res_bytes))) return rc;
  if ((rc = ensure(c, 6, bytes_total))) return rc;
  if ((rc = ensure(c, 7, off_count * sizeof(uint64_t)))) return rc;
  if ((rc = ensure_pinned(c, 0, res_bytes + off_count * sizeof(uint64_t)))) return rc;
  auto *d_desc = static_cast<const cabac_substream_desc *>(c->d_buf[0]);
  auto *d_rec = static_cast<uint16_t *>(c->d_buf[1]);
  auto *d_slots = static_cast<uint8_t *>(c->d_buf[2]);
  auto *d_res = static_cast<cabac_substream_result *>(c->d_buf[3]);
  auto *d_pay = static_cast<uint8_t *>(c->d_buf[6]);
  auto *d_off = static_cast<uint64_t *>(c->d_buf[7]);
  auto *h_res = static_cast<cabac_substream_result *>(c->h_pin[0]);
  auto *h_off = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(c->h_pin[0]) + res_bytes);

  // what came before on the caller's stream is finished first; everything below runs on the library's own streams
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if ((rc = h2d(c, c->d_buf[0], desc, n_sub * sizeof(cabac_substream_desc), c->s_in))) return rc;
  for (uint32_t k = 0; k < nc; k++) {
    const Chunk &ch = chunks[k];
    const uint32_t n_k = ch.s1 - ch.s0;
    if ((rc = h2d(c, d_rec + ch.rec_lo, records + ch.rec_lo, (ch.rec_hi - ch.rec_lo) * 2, c->s_in))) return rc;
    HIP_TRY(c, hipEventRecord(c->ev_in[k], c->s_in));
    hipStream_t ks = c->s_k[k % cabac_hip_ctx::kKernelStreams];
    HIP_TRY(c, hipStreamWaitEvent(ks, c->ev_in[k], 0));
    // (timed only through the profile ring: the ctx's single event pair would tie the chunk streams together)
    const bool ring = !c->prof_ev.empty() && c->prof_n < c->prof_kind.size();
    Bracket br = ring ? bracket_for(c, 0) : Bracket{nullptr, nullptr};
    if (ring) HIP_TRY(c, hipEventRecord(br.a, ks));
    HIP_TRY(c, cabac::launch_encode(ks, c->enc_variant, n_k, d_desc + ch.s0, d_rec, d_slots, d_res + ch.s0, n_sub));
    if (ring) HIP_TRY(c, hipEventRecord(br.b, ks));
    c->timed = false;
    // compaction: the coded substreams of the chunk back to back, so that they leave in ONE copy instead of one per
    // substream (the payload is ~0.1 B/bin; the slots are sized for the worst case)
    HIP_TRY(c, cabac::launch_assemble(ks, n_k, d_desc + ch.s0, d_res + ch.s0, d_slots, d_pay + ch.byte_lo,
                                      ch.byte_hi - ch.byte_lo, d_off + ch.s0 + k));
    HIP_TRY(c, hipMemcpyAsync(h_res + ch.s0, d_res + ch.s0, n_k * sizeof(cabac_substream_result), hipMemcpyDeviceToHost, ks));
    HIP_TRY(c, hipMemcpyAsync(h_off + ch.s0 + k, d_off + ch.s0 + k, (size_t(n_k) + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ks));
    HIP_TRY(c, hipEventRecord(c->ev_k[k], ks));
  }
  // the sizes of chunk k are known once its kernel is through: its payload follows while later chunks are still coded
  std::vector<uint64_t> pay_base(nc + 1, 0);
  for (uint32_t k = 0; k < nc; k++) {
    const Chunk &ch = chunks[k];
    HIP_TRY(c, hipEventSynchronize(c->ev_k[k]));
    const uint64_t n_pay = h_off[ch.s1 + k];
    pay_base[k + 1] = pay_base[k] + n_pay;
    if (payload) {  // straight into the caller's payload (DMA if it is pinned, else through the bounce ring)
      if (pay_base[k + 1] > payload_capacity) return fail(c, CABAC_HIP_ERR_INVALID, "payload_capacity too small");
      if ((rc = d2h(c, payload + pay_base[k], d_pay + ch.byte_lo, n_pay, c->s_out))) return rc;
    } else {
      if ((rc = ensure_pinned_keep(c, 1, pay_base[k + 1], pay_base[k]))) return rc;
      if (n_pay)
        HIP_TRY(c, hipMemcpyAsync(static_cast<uint8_t *>(c->h_pin[1]) + pay_base[k], d_pay + ch.byte_lo, n_pay, hipMemcpyDeviceToHost, c->s_out));
    }
    HIP_TRY(c, hipEventRecord(c->ev_out[k], c->s_out));
  }
  if (payload && (rc = d2h_drain(c))) return rc;
  int status = CABAC_HIP_OK;
  for (uint32_t k = 0; k < nc; k++) {
    const Chunk &ch = chunks[k];
    HIP_TRY(c, hipEventSynchronize(c->ev_out[k]));
    const uint8_t *pay = payload ? payload + pay_base[k] : static_cast<const uint8_t *>(c->h_pin[1]) + pay_base[k];
    const uint64_t *off = h_off + ch.s0 + k;
    for (uint32_t s = ch.s0; s < ch.s1; s++) {
      const uint64_t o = off[s - ch.s0], n = off[s - ch.s0 + 1] - o;
      if (n && bytes) std::memcpy(bytes + desc[s].byte_offset, pay + o, n);  // into the caller's slots
      if (payload_offsets) payload_offsets[s + 1] = pay_base[k] + o + n;
      results[s] = h_res[s];
      if (h_res[s].flags) status = CABAC_HIP_ERR_SUBSTREAM;
    }
  }
  if (status) c->last_error = "substream flag set (see results[].flags)";
  return status;
}

int cabac_hip_encode_batch(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                           uint64_t n_records_total, uint8_t *bytes, uint64_t bytes_total,
                           cabac_substream_result *results) {
  return host_call_exit(c, encode_batch_impl(c, n_sub, desc, records, n_records_total, bytes, bytes_total, nullptr, 0, nullptr, results));
}

int cabac_hip_encode_batch_payload(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                                   uint64_t n_records_total, uint8_t *payload, uint64_t payload_capacity,
                                   uint64_t *payload_offsets, cabac_substream_result *results) {
  if (!payload || !payload_offsets) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  uint64_t slots_end = 0;  // the device-side slots still follow the descriptors' byte_offset / byte_capacity
  for (uint32_t s = 0; s < n_sub && desc; s++) slots_end = std::max<uint64_t>(slots_end, desc[s].byte_offset + desc[s].byte_capacity);
  return host_call_exit(c, encode_batch_impl(c, n_sub, desc, records, n_records_total, nullptr, slots_end, payload, payload_capacity,
                                             payload_offsets, results));
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

// packed: `bins` receives (n_records_total + 7) / 8 bytes, bit (r & 7) of byte r >> 3 = the bin of record r
static int decode_batch_impl(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                             uint64_t n_records_total, const uint8_t *bytes, uint64_t bytes_total, uint8_t *bins,
                             cabac_substream_result *results, bool packed) {
  if (!c || (n_sub && (!desc || !results))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  if (n_sub == 0) return CABAC_HIP_OK;
  int rc = check_desc_host(c, n_sub, desc, n_records_total, bytes_total);
  if (rc) return rc;
  DeviceGuard g(c->device);
  if ((rc = pipe_init(c))) return rc;
  const std::vector<Chunk> chunks = plan_chunks(c, n_sub, desc);
  const uint32_t nc = uint32_t(chunks.size());
  const size_t res_bytes = size_t(n_sub) * sizeof(cabac_substream_result);
  // the kernel reads whole aligned dwords: room for the dword that holds the last byte
  if ((rc = ensure(c, 0, n_sub * sizeof(cabac_substream_desc)))) return rc;
  if ((rc = ensure(c, 1, n_records_total * 2))) return rc;
  if ((rc = ensure(c, 2, bytes_total + 4))) return rc;
  if ((rc = ensure(c, 3, res_bytes))) return rc;
  if ((rc = ensure(c, 4, n_records_total + 8))) return rc;
  if (packed && bins && (rc = ensure(c, 6, (n_records_total + 7) / 8 + 8))) return rc;
  if ((rc = ensure_pinned(c, 0, res_bytes))) return rc;
  auto *d_desc = static_cast<const cabac_substream_desc *>(c->d_buf[0]);
  auto *d_rec = static_cast<uint16_t *>(c->d_buf[1]);
  auto *d_slots = static_cast<uint8_t *>(c->d_buf[2]);
  auto *d_res = static_cast<cabac_substream_result *>(c->d_buf[3]);
  auto *d_bins = static_cast<uint8_t *>(c->d_buf[4]);
  auto *h_res = static_cast<cabac_substream_result *>(c->h_pin[0]);

  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if ((rc = h2d(c, c->d_buf[0], desc, n_sub * sizeof(cabac_substream_desc), c->s_in))) return rc;
  for (uint32_t k = 0; k < nc; k++) {
    const Chunk &ch = chunks[k];
    const uint32_t n_k = ch.s1 - ch.s0;
    if ((rc = h2d(c, d_rec + ch.rec_lo, records + ch.rec_lo, (ch.rec_hi - ch.rec_lo) * 2, c->s_in))) return rc;
    if (bytes && (rc = h2d(c, d_slots + ch.byte_lo, bytes + ch.byte_lo, ch.byte_hi - ch.byte_lo, c->s_in))) return rc;
    HIP_TRY(c, hipEventRecord(c->ev_in[k], c->s_in));
    hipStream_t ks = c->s_k[k % cabac_hip_ctx::kKernelStreams];
    HIP_TRY(c, hipStreamWaitEvent(ks, c->ev_in[k], 0));
    const bool ring = !c->prof_ev.empty() && c->prof_n < c->prof_kind.size();
    Bracket br = ring ? bracket_for(c, 1) : Bracket{nullptr, nullptr};
    if (ring) HIP_TRY(c, hipEventRecord(br.a, ks));
    HIP_TRY(c, cabac::launch_decode(ks, c->dec_variant, n_k, d_desc + ch.s0, d_rec, d_slots, d_bins, d_res + ch.s0, n_sub, c->d_select + 1 + k));
    if (ring) HIP_TRY(c, hipEventRecord(br.b, ks));
    c->timed = false;
    HIP_TRY(c, hipMemcpyAsync(h_res + ch.s0, d_res + ch.s0, n_k * sizeof(cabac_substream_result), hipMemcpyDeviceToHost, ks));
    HIP_TRY(c, hipEventRecord(c->ev_k[k], ks));
    // the decoded bins of the chunk leave while the next chunk is decoded
    HIP_TRY(c, hipStreamWaitEvent(c->s_out, c->ev_k[k], 0));
    if (bins && !packed && (rc = d2h(c, bins + ch.rec_lo, d_bins + ch.rec_lo, ch.rec_hi - ch.rec_lo, c->s_out))) return rc;
  }
  if (bins && packed) {  // all chunks decoded (s_out waits for each of them above): eight bins to a byte, one small copy
    HIP_TRY(c, cabac::launch_pack_bins(c->s_out, n_records_total, d_bins, static_cast<uint8_t *>(c->d_buf[6])));
    if ((rc = d2h(c, bins, c->d_buf[6], (n_records_total + 7) / 8, c->s_out))) return rc;
  }
  if ((rc = d2h_drain(c))) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->s_out));
  for (uint32_t k = 0; k < nc; k++) HIP_TRY(c, hipEventSynchronize(c->ev_k[k]));
  int status = CABAC_HIP_OK;
  for (uint32_t s = 0; s < n_sub; s++) {
    results[s] = h_res[s];
    if (h_res[s].flags) status = CABAC_HIP_ERR_SUBSTREAM;
  }
  if (status) c->last_error = "substream flag set (see results[].flags)";
  return status;
}

int cabac_hip_decode_batch(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                           uint64_t n_records_total, const uint8_t *bytes, uint64_t bytes_total, uint8_t *bins,
                           cabac_substream_result *results) {
  return host_call_exit(c, decode_batch_impl(c, n_sub, desc, records, n_records_total, bytes, bytes_total, bins, results, false));
}

int cabac_hip_decode_batch_packed(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                                  uint64_t n_records_total, const uint8_t *bytes, uint64_t bytes_total, uint8_t *packed_bins,
                                  cabac_substream_result *results) {
  return host_call_exit(c, decode_batch_impl(c, n_sub, desc, records, n_records_total, bytes, bytes_total, packed_bins, results, true));
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
  HIP_TRY(c, cabac::launch_residual(c->stream, n_tu, d_tu, d_coeff, 4, d_rec_offset, d_n_records, d_info, d_records,
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

static int residual_parse_device_impl(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc, const uint8_t *d_bytes,
                                      const uint32_t *d_tile_first, const cabac_tu_desc *d_tu, void *d_coeff, int coeff_bytes,
                                      uint32_t *d_tu_info, cabac_substream_result *d_results) {
  if (!c || (n_sub && (!d_desc || !d_bytes || !d_tile_first || !d_tu || !d_coeff || !d_results)))
    return fail(c, CABAC_HIP_ERR_INVALID, "null");
  DeviceGuard g(c->device);
  Bracket br = bracket_for(c, 9);
  HIP_TRY(c, hipEventRecord(br.a, c->stream));
  HIP_TRY(c, cabac::launch_residual_parse(c->stream, n_sub, d_desc, d_bytes, d_tile_first, d_tu, d_coeff, coeff_bytes, d_tu_info, d_results));
  HIP_TRY(c, hipEventRecord(br.b, c->stream));
  c->timed = (br.a == c->ev_start);
  return CABAC_HIP_OK;
}

int cabac_hip_residual_parse_device(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc, const uint8_t *d_bytes,
                                    const uint32_t *d_tile_first, const cabac_tu_desc *d_tu, int32_t *d_coeff, uint32_t *d_tu_info,
                                    cabac_substream_result *d_results) {
  return residual_parse_device_impl(c, n_sub, d_desc, d_bytes, d_tile_first, d_tu, d_coeff, 4, d_tu_info, d_results);
}

int cabac_hip_residual_parse16_device(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc, const uint8_t *d_bytes,
                                      const uint32_t *d_tile_first, const cabac_tu_desc *d_tu, int16_t *d_coeff, uint32_t *d_tu_info,
                                      cabac_substream_result *d_results) {
  return residual_parse_device_impl(c, n_sub, d_desc, d_bytes, d_tile_first, d_tu, d_coeff, 2, d_tu_info, d_results);
}

static int residual_parse_batch_impl(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint8_t *bytes,
                                     uint64_t bytes_total, const uint32_t *tile_first, const cabac_tu_desc *tus, void *coeff, int coeff_bytes,
                                     uint64_t n_coeff_total, uint32_t *tu_info, cabac_substream_result *results) {
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
  if ((rc = ensure(c, 1, (n_coeff_total + 4) * size_t(coeff_bytes)))) return rc;
  if ((rc = ensure(c, 4, n_sub * sizeof(cabac_substream_result)))) return rc;
  if ((rc = ensure(c, 6, size_t(n_tu) * sizeof(uint32_t)))) return rc;
  uint8_t *d_first = static_cast<uint8_t *>(c->d_buf[3]);
  HIP_TRY(c, hipMemcpyAsync(c->d_buf[0], desc, n_sub * sizeof(cabac_substream_desc), hipMemcpyHostToDevice, c->stream));
  if (bytes_total) HIP_TRY(c, hipMemcpyAsync(c->d_buf[2], bytes, bytes_total, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d_first, tile_first, first_bytes, hipMemcpyHostToDevice, c->stream));
  if (n_tu) HIP_TRY(c, hipMemcpyAsync(d_first + first_pad, tus, size_t(n_tu) * sizeof(cabac_tu_desc), hipMemcpyHostToDevice, c->stream));
  // int32: what the parser does not write (outside the coded region of 64-wide blocks) keeps the caller's values;
  // int16: output only, zero where nothing is written
  if (n_coeff_total && coeff_bytes == 4) HIP_TRY(c, hipMemcpyAsync(c->d_buf[1], coeff, n_coeff_total * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  if (n_coeff_total && coeff_bytes == 2) HIP_TRY(c, hipMemsetAsync(c->d_buf[1], 0, n_coeff_total * sizeof(int16_t), c->stream));
  rc = residual_parse_device_impl(c, n_sub, (const cabac_substream_desc *)c->d_buf[0], (const uint8_t *)c->d_buf[2],
                                  (const uint32_t *)d_first, (const cabac_tu_desc *)(d_first + first_pad), c->d_buf[1], coeff_bytes,
                                  (uint32_t *)c->d_buf[6], (cabac_substream_result *)c->d_buf[4]);
  if (rc) return rc;
  if (tu_info && n_tu) HIP_TRY(c, hipMemcpyAsync(tu_info, c->d_buf[6], size_t(n_tu) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  if (n_coeff_total) HIP_TRY(c, hipMemcpyAsync(coeff, c->d_buf[1], n_coeff_total * size_t(coeff_bytes), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(results, c->d_buf[4], n_sub * sizeof(cabac_substream_result), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int status = CABAC_HIP_OK;
  for (uint32_t s = 0; s < n_sub; s++)
    if (results[s].flags) status = CABAC_HIP_ERR_SUBSTREAM;
  if (status) c->last_error = "substream flag set (see results[].flags)";
  return status;
}

int cabac_hip_residual_parse_batch(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint8_t *bytes,
                                   uint64_t bytes_total, const uint32_t *tile_first, const cabac_tu_desc *tus, int32_t *coeff,
                                   uint64_t n_coeff_total, uint32_t *tu_info, cabac_substream_result *results) {
  return residual_parse_batch_impl(c, n_sub, desc, bytes, bytes_total, tile_first, tus, coeff, 4, n_coeff_total, tu_info, results);
}

int cabac_hip_residual_parse_batch16(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint8_t *bytes,
                                     uint64_t bytes_total, const uint32_t *tile_first, const cabac_tu_desc *tus, int16_t *coeff,
                                     uint64_t n_coeff_total, uint32_t *tu_info, cabac_substream_result *results) {
  return residual_parse_batch_impl(c, n_sub, desc, bytes, bytes_total, tile_first, tus, coeff, 2, n_coeff_total, tu_info, results);
}

// ---- coefficients -> bytes (cabac_splice.hip) -------------------------------------------------------------------
namespace {
// device slots of the spliced-residual path
enum { kSpCnt = 8, kSpPlan = 9, kSpDesc = 10, kSpTuOff = 11, kSpRec = 12, kSpBytes = 13, kSpFlag = 14,
       // ... and the staging of its host-pointer form
       kSpInDesc = 15, kSpInRec = 16, kSpInFirst = 17, kSpInSplice = 18, kSpInTu = 19, kSpInCoeff = 20, kSpOutRes = 21,
       kSpOutInfo = 22, kSpOutCnt = 23 };

struct Timed {  // profile-ring bracket around a group of launches (nothing when the ring is off or full)
  cabac_hip_ctx *c;
  Bracket br{nullptr, nullptr};
  Timed(cabac_hip_ctx *ctx, int kind) : c(ctx) {
    if (!c->prof_ev.empty() && c->prof_n < c->prof_kind.size()) {
      br = bracket_for(c, kind);
      (void)hipEventRecord(br.a, c->stream);
    }
  }
  ~Timed() {
    if (br.b) (void)hipEventRecord(br.b, c->stream);
  }
};
}  // namespace

static int encode_residual_device_impl(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc, const uint16_t *d_records,
                                       const uint32_t *d_splice_first, const cabac_splice *d_splices, uint32_t n_splice, uint32_t n_tu,
                                       const cabac_tu_desc *d_tu, const void *d_coeff, int coeff_bytes, uint8_t *d_payload,
                                       uint64_t payload_capacity, uint64_t *d_payload_offsets, cabac_substream_result *d_results,
                                       uint32_t *d_tu_info, uint32_t *d_bin_counts) {
  if (!c || !d_payload_offsets || (n_sub && (!d_desc || !d_records || !d_splice_first || !d_payload || !d_results)) ||
      (n_splice && !d_splices) || (n_tu && (!d_tu || !d_coeff)))
    return fail(c, CABAC_HIP_ERR_INVALID, "null");
  if (n_splice != n_tu) return fail(c, CABAC_HIP_ERR_INVALID, "every block must be spliced exactly once (n_splice != n_tu)");
  DeviceGuard g(c->device);
  int rc;
  const size_t n_pre = size_t(n_splice) + n_sub + 1;
  const size_t plan32 = n_pre + 2 * size_t(n_sub) + size_t(n_tu ? n_tu : 1) + 2;  // pre, sub_n, sub_cap, seen, err (+ pad)
  const size_t plan32_pad = (plan32 + 1) & ~size_t(1);
  if ((rc = ensure(c, kSpCnt, 2 * size_t(n_tu ? n_tu : 1) * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, 5, cabac::residual_scratch_bytes(n_tu)))) return rc;
  if ((rc = ensure(c, kSpPlan, plan32_pad * sizeof(uint32_t) + (2 * size_t(n_sub) + 3) * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, kSpDesc, size_t(n_sub ? n_sub : 1) * sizeof(cabac_substream_desc)))) return rc;
  if ((rc = ensure(c, kSpTuOff, size_t(n_tu ? n_tu : 1) * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, kSpFlag, 64))) return rc;
  if (!c->h_totals) HIP_TRY(c, hipHostMalloc(&c->h_totals, 64, hipHostMallocDefault));
  uint32_t *d_cnt = static_cast<uint32_t *>(c->d_buf[kSpCnt]), *d_info = d_cnt + (n_tu ? n_tu : 1);
  uint32_t *pre = static_cast<uint32_t *>(c->d_buf[kSpPlan]), *sub_n = pre + n_pre, *sub_cap = sub_n + n_sub, *seen = sub_cap + n_sub,
           *err = seen + (n_tu ? n_tu : 1);
  uint64_t *rec_base = reinterpret_cast<uint64_t *>(pre + plan32_pad), *byte_base = rec_base + n_sub, *totals = byte_base + n_sub;
  auto *desc2 = static_cast<cabac_substream_desc *>(c->d_buf[kSpDesc]);
  auto *tu_off = static_cast<uint64_t *>(c->d_buf[kSpTuOff]);
  c->timed = false;
  {  // block sizes (pass 1 of the binariser)
    Timed t(c, 5);
    HIP_TRY(c, cabac::launch_residual(c->stream, n_tu, d_tu, d_coeff, coeff_bytes, nullptr, d_cnt, d_info, nullptr, c->d_buf[5]));
  }
  {
    Timed t(c, 10);
    HIP_TRY(c, cabac::launch_splice_plan(c->stream, n_sub, n_tu, d_desc, d_splice_first, d_splices, n_splice, d_cnt, pre, sub_n, sub_cap, seen, err,
                                         rec_base, byte_base, totals));
  }
  uint64_t *h_tot = static_cast<uint64_t *>(c->h_totals);
  HIP_TRY(c, hipMemcpyAsync(h_tot, totals, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the one wait: sizes decide the buffers of the second half
  if (h_tot[2]) return fail(c, CABAC_HIP_ERR_INVALID, "splice list: not sorted, outside its substream, or a block not spliced exactly once");
  if ((rc = ensure(c, kSpRec, (h_tot[0] + 64) * sizeof(uint16_t)))) return rc;
  if ((rc = ensure(c, kSpBytes, h_tot[1] + 64))) return rc;
  auto *exp_rec = static_cast<uint16_t *>(c->d_buf[kSpRec]);
  auto *slots = static_cast<uint8_t *>(c->d_buf[kSpBytes]);
  {
    Timed t(c, 10);
    HIP_TRY(c, cabac::launch_splice_expand(c->stream, n_sub, d_desc, d_records, d_splice_first, d_splices, pre, sub_n, sub_cap, rec_base,
                                           byte_base, desc2, tu_off, exp_rec));
  }
  {  // block records (pass 2), straight into the expanded substreams
    Timed t(c, 5);
    HIP_TRY(c, cabac::launch_residual(c->stream, n_tu, d_tu, d_coeff, coeff_bytes, tu_off, d_cnt, d_info, exp_rec, c->d_buf[5],
                                      /*order_ready=*/true));  // the block order of the sizes pass above: same tus[], same scratch
  }
  {
    Timed t(c, 0);
    HIP_TRY(c, cabac::launch_encode(c->stream, c->enc_variant, n_sub, desc2, exp_rec, slots, d_results));
  }
  if (d_bin_counts) {
    Timed t(c, 11);
    HIP_TRY(c, cabac::launch_bin_count(c->stream, n_sub, desc2, exp_rec, d_bin_counts));
  }
  {
    Timed t(c, 6);
    HIP_TRY(c, cabac::launch_assemble(c->stream, n_sub, desc2, d_results, slots, d_payload, payload_capacity, d_payload_offsets));
  }
  if (d_tu_info && n_tu) HIP_TRY(c, hipMemcpyAsync(d_tu_info, d_info, size_t(n_tu) * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
  return CABAC_HIP_OK;
}

int cabac_hip_encode_residual_device(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc, const uint16_t *d_records,
                                     const uint32_t *d_splice_first, const cabac_splice *d_splices, uint32_t n_splice, uint32_t n_tu,
                                     const cabac_tu_desc *d_tu, const int32_t *d_coeff, uint8_t *d_payload, uint64_t payload_capacity,
                                     uint64_t *d_payload_offsets, cabac_substream_result *d_results, uint32_t *d_tu_info,
                                     uint32_t *d_bin_counts) {
  return encode_residual_device_impl(c, n_sub, d_desc, d_records, d_splice_first, d_splices, n_splice, n_tu, d_tu, d_coeff, 4, d_payload,
                                     payload_capacity, d_payload_offsets, d_results, d_tu_info, d_bin_counts);
}

int cabac_hip_encode_residual16_device(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *d_desc, const uint16_t *d_records,
                                       const uint32_t *d_splice_first, const cabac_splice *d_splices, uint32_t n_splice, uint32_t n_tu,
                                       const cabac_tu_desc *d_tu, const int16_t *d_coeff, uint8_t *d_payload, uint64_t payload_capacity,
                                       uint64_t *d_payload_offsets, cabac_substream_result *d_results, uint32_t *d_tu_info,
                                       uint32_t *d_bin_counts) {
  return encode_residual_device_impl(c, n_sub, d_desc, d_records, d_splice_first, d_splices, n_splice, n_tu, d_tu, d_coeff, 2, d_payload,
                                     payload_capacity, d_payload_offsets, d_results, d_tu_info, d_bin_counts);
}

static int encode_batch_residual_impl(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                                      uint64_t n_records_total, const uint32_t *splice_first, const cabac_splice *splices, uint32_t n_tu,
                                      const cabac_tu_desc *tus, const void *coeff, int coeff_bytes, uint64_t n_coeff_total, uint8_t *payload,
                                      uint64_t payload_capacity, uint64_t *payload_offsets, cabac_substream_result *results,
                                      uint32_t *tu_info, uint32_t *bin_counts) {
  if (!c || !payload_offsets || (n_sub && (!desc || !splice_first || !results || !payload)) || (n_tu && (!tus || !coeff || !splices)))
    return fail(c, CABAC_HIP_ERR_INVALID, "null");
  payload_offsets[0] = 0;
  if (n_sub == 0) return n_tu ? fail(c, CABAC_HIP_ERR_INVALID, "blocks without a substream") : CABAC_HIP_OK;
  for (uint32_t s = 0; s < n_sub; s++) {
    if (desc[s].rec_offset > n_records_total || desc[s].n_records > n_records_total - desc[s].rec_offset)
      return fail(c, CABAC_HIP_ERR_INVALID, "records out of range");
    if ((desc[s].init_id & 3u) > 2u) return fail(c, CABAC_HIP_ERR_INVALID, "init_id must be 0..2");
    if (splice_first[s] > splice_first[s + 1]) return fail(c, CABAC_HIP_ERR_INVALID, "splice_first must not decrease");
  }
  const uint32_t n_splice = splice_first[n_sub];
  if (splice_first[0] != 0 || n_splice != n_tu) return fail(c, CABAC_HIP_ERR_INVALID, "every block must be spliced exactly once");
  DeviceGuard g(c->device);
  int rc;
  if ((rc = pipe_init(c))) return rc;
  const size_t first_bytes = (size_t(n_sub) + 1) * sizeof(uint32_t);
  if ((rc = ensure(c, kSpInDesc, n_sub * sizeof(cabac_substream_desc)))) return rc;
  if ((rc = ensure(c, kSpInRec, (n_records_total + 8) * sizeof(uint16_t)))) return rc;
  if ((rc = ensure(c, kSpInFirst, first_bytes))) return rc;
  if ((rc = ensure(c, kSpInSplice, size_t(n_splice ? n_splice : 1) * sizeof(cabac_splice)))) return rc;
  if ((rc = ensure(c, kSpInTu, size_t(n_tu ? n_tu : 1) * sizeof(cabac_tu_desc)))) return rc;
  if ((rc = ensure(c, kSpInCoeff, (n_coeff_total + 4) * size_t(coeff_bytes)))) return rc;
  if ((rc = ensure(c, kSpOutRes, n_sub * sizeof(cabac_substream_result)))) return rc;
  if ((rc = ensure(c, kSpOutInfo, size_t(n_tu ? n_tu : 1) * sizeof(uint32_t)))) return rc;
  if (bin_counts && (rc = ensure(c, kSpOutCnt, size_t(n_sub) * CABAC_BIN_COUNT_WORDS * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, 6, payload_capacity ? payload_capacity : 16))) return rc;
  if ((rc = ensure(c, 7, (size_t(n_sub) + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, kSpFlag, 64))) return rc;
  const size_t res_bytes = size_t(n_sub) * sizeof(cabac_substream_result), off_bytes = (size_t(n_sub) + 1) * sizeof(uint64_t);
  if ((rc = ensure_pinned(c, 0, res_bytes + off_bytes + 64))) return rc;
  auto *h_res = static_cast<cabac_substream_result *>(c->h_pin[0]);
  auto *h_off = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(c->h_pin[0]) + res_bytes);
  auto *h_flag = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(h_off) + off_bytes);
  uint32_t *d_flag = static_cast<uint32_t *>(c->d_buf[kSpFlag]);  // [0] descriptor range check, [1] empty / bad block

  // what came before on the caller's stream is finished first; uploads run on the copy stream, the kernels on the ctx's
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  // small things first: the block descriptors are checked on the device while the coefficients are still on the wire
  if ((rc = h2d(c, c->d_buf[kSpInTu], tus, size_t(n_tu) * sizeof(cabac_tu_desc), c->s_in))) return rc;
  HIP_TRY(c, hipEventRecord(c->ev_in[0], c->s_in));
  HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_in[0], 0));
  HIP_TRY(c, cabac::launch_tu_range_check(c->stream, n_tu, static_cast<const cabac_tu_desc *>(c->d_buf[kSpInTu]), n_coeff_total, d_flag));
  HIP_TRY(c, hipMemcpyAsync(h_flag, d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipEventRecord(c->ev_k[0], c->stream));
  if ((rc = h2d(c, c->d_buf[kSpInDesc], desc, n_sub * sizeof(cabac_substream_desc), c->s_in))) return rc;
  if ((rc = h2d(c, c->d_buf[kSpInFirst], splice_first, first_bytes, c->s_in))) return rc;
  if ((rc = h2d(c, c->d_buf[kSpInSplice], splices, size_t(n_splice) * sizeof(cabac_splice), c->s_in))) return rc;
  if ((rc = h2d(c, c->d_buf[kSpInRec], records, n_records_total * sizeof(uint16_t), c->s_in))) return rc;
  if ((rc = h2d(c, c->d_buf[kSpInCoeff], coeff, n_coeff_total * size_t(coeff_bytes), c->s_in))) return rc;
  HIP_TRY(c, hipEventRecord(c->ev_in[1], c->s_in));
  HIP_TRY(c, hipEventSynchronize(c->ev_k[0]));
  if (*h_flag) return fail(c, CABAC_HIP_ERR_INVALID, "coefficients out of range");
  HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_in[1], 0));
  rc = encode_residual_device_impl(c, n_sub, static_cast<const cabac_substream_desc *>(c->d_buf[kSpInDesc]),
                                        static_cast<const uint16_t *>(c->d_buf[kSpInRec]), static_cast<const uint32_t *>(c->d_buf[kSpInFirst]),
                                        static_cast<const cabac_splice *>(c->d_buf[kSpInSplice]), n_splice, n_tu,
                                        static_cast<const cabac_tu_desc *>(c->d_buf[kSpInTu]), c->d_buf[kSpInCoeff], coeff_bytes,
                                        static_cast<uint8_t *>(c->d_buf[6]), payload_capacity, static_cast<uint64_t *>(c->d_buf[7]),
                                        static_cast<cabac_substream_result *>(c->d_buf[kSpOutRes]), static_cast<uint32_t *>(c->d_buf[kSpOutInfo]),
                                        bin_counts ? static_cast<uint32_t *>(c->d_buf[kSpOutCnt]) : nullptr);
  if (rc) return rc;
  HIP_TRY(c, cabac::launch_tu_info_any(c->stream, n_tu, static_cast<const uint32_t *>(c->d_buf[kSpOutInfo]), d_flag + 1));
  HIP_TRY(c, hipMemcpyAsync(h_res, c->d_buf[kSpOutRes], res_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(h_off, c->d_buf[7], off_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(h_flag, d_flag + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipEventRecord(c->ev_k[1], c->stream));
  // the big things leave on the outgoing copy stream once the offsets are known
  HIP_TRY(c, hipEventSynchronize(c->ev_k[1]));
  const uint64_t n_pay = h_off[n_sub];
  if (n_pay > payload_capacity) return fail(c, CABAC_HIP_ERR_INVALID, "payload_capacity too small");
  if ((rc = d2h(c, payload, c->d_buf[6], n_pay, c->s_out))) return rc;
  if (tu_info && (rc = d2h(c, tu_info, c->d_buf[kSpOutInfo], size_t(n_tu) * sizeof(uint32_t), c->s_out))) return rc;
  if (bin_counts && (rc = d2h(c, bin_counts, c->d_buf[kSpOutCnt], size_t(n_sub) * CABAC_BIN_COUNT_WORDS * sizeof(uint32_t), c->s_out))) return rc;
  if ((rc = d2h_drain(c))) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->s_out));
  int status = *h_flag ? CABAC_HIP_ERR_SUBSTREAM : CABAC_HIP_OK;
  for (uint32_t s = 0; s < n_sub; s++) {
    results[s] = h_res[s];
    payload_offsets[s + 1] = h_off[s + 1];
    if (h_res[s].flags) status = CABAC_HIP_ERR_SUBSTREAM;
  }
  if (status) c->last_error = "substream flag set, or an empty / badly described block (see results[].flags, tu_info[])";
  return status;
}

int cabac_hip_encode_batch_residual(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                                    uint64_t n_records_total, const uint32_t *splice_first, const cabac_splice *splices, uint32_t n_tu,
                                    const cabac_tu_desc *tus, const int32_t *coeff, uint64_t n_coeff_total, uint8_t *payload,
                                    uint64_t payload_capacity, uint64_t *payload_offsets, cabac_substream_result *results,
                                    uint32_t *tu_info, uint32_t *bin_counts) {
  return host_call_exit(c, encode_batch_residual_impl(c, n_sub, desc, records, n_records_total, splice_first, splices, n_tu, tus, coeff, 4,
                                                      n_coeff_total, payload, payload_capacity, payload_offsets, results, tu_info,
                                                      bin_counts));
}

int cabac_hip_encode_batch_residual16(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                                      uint64_t n_records_total, const uint32_t *splice_first, const cabac_splice *splices, uint32_t n_tu,
                                      const cabac_tu_desc *tus, const int16_t *coeff, uint64_t n_coeff_total, uint8_t *payload,
                                      uint64_t payload_capacity, uint64_t *payload_offsets, cabac_substream_result *results,
                                      uint32_t *tu_info, uint32_t *bin_counts) {
  return host_call_exit(c, encode_batch_residual_impl(c, n_sub, desc, records, n_records_total, splice_first, splices, n_tu, tus, coeff, 2,
                                                      n_coeff_total, payload, payload_capacity, payload_offsets, results, tu_info,
                                                      bin_counts));
}

// ---- pinned host memory for the caller's buffers ------------------------------------------------------------
int cabac_hip_host_alloc(size_t bytes, void **out) {
  if (!out) return CABAC_HIP_ERR_INVALID;
  *out = nullptr;
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return CABAC_HIP_ERR_NOMEM;
  }
  std::lock_guard<std::mutex> lk(g_host_mu);
  g_host_owned[p] = bytes ? bytes : 16;
  *out = p;
  return CABAC_HIP_OK;
}

int cabac_hip_host_free(void *p) {
  if (!p) return CABAC_HIP_OK;
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!g_host_owned.erase(p)) return CABAC_HIP_ERR_INVALID;
  }
  return hipHostFree(p) == hipSuccess ? CABAC_HIP_OK : CABAC_HIP_ERR_HIP;
}

int cabac_hip_host_register(void *p, size_t bytes) {
  if (!p || !bytes) return CABAC_HIP_ERR_INVALID;
  if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();
    return CABAC_HIP_ERR_HIP;
  }
  std::lock_guard<std::mutex> lk(g_host_mu);
  g_host_registered[p] = bytes;
  return CABAC_HIP_OK;
}

int cabac_hip_host_unregister(void *p) {
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!g_host_registered.erase(p)) return CABAC_HIP_ERR_INVALID;
  }
  return hipHostUnregister(p) == hipSuccess ? CABAC_HIP_OK : CABAC_HIP_ERR_HIP;
}

int cabac_hip_host_is_pinned(const void *p, size_t bytes) { return host_is_pinned(p, bytes) ? 1 : 0; }

#ifdef CABAC_PARSE_PROFILE
int cabac_hip_debug_parse_waves(unsigned long long *out) { return cabac::debug_read_parse_waves(out) == hipSuccess ? 0 : -3; }
int cabac_hip_debug_parse_prof(unsigned long long *out) { return cabac::debug_read_parse_prof(out) == hipSuccess ? 0 : -3; }
#endif

}  // extern "C"
