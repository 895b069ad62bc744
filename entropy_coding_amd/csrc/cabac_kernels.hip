// MI355X (gfx950) CABAC bin codec kernels — one 64-lane wavefront per independent substream.
//
// What is restated here, with the reference lines each piece must match bit for bit
// (paths relative to /root/reference/src):
//   context init        BinProbModel_Std::init / setLog2WindowSize / CtxStore::init
//                       common/contexts.cpp:893-901, :915-920, :996-1015
//   probability model   state / mps / getLPS / update / getRenormBitsLPS
//                       common/contexts.cpp:903-913, :939-954, :787-789
//   bin encoder         start / encodeBin / encodeBinEP / encodeBinTrm / writeOut / finish
//                       entropy_codec/arith_codec.cpp:329-337, :553-582, :389-399, :460-478,
//                       :524-546, :339-357
//   bin decoder         start / decodeBin / decodeBinEP / decodeBinTrm / finish
//                       entropy_codec/arith_codec.cpp:60-73, :242-277, :100-114, :181-197
//   byte I/O            OutputBitstream::write / writeByteAlignment, InputBitstream::readByte
//                       common/bit_stream.cpp:70-117, :152-155, :268-274
//
// Execution model (v1, "wave-serial"): the low/range/bitsLeft recurrence of a substream is a
// strict serial chain, so one wavefront walks it with *wave-uniform* (SGPR/SALU) arithmetic while
// the 64 lanes do everything around it in parallel:
//   * 64 bin records are fetched per step with one coalesced load (prefetched one step ahead);
//   * the 64 context states those bins touch are gathered from the LDS context store at once;
//     inside the step a freshly updated state is forwarded to every lane that holds the same
//     ctxId with one v_cmp + v_cndmask, so the serial chain never waits on LDS;
//   * output bytes are assembled in an SGPR word and dropped into one lane of a VGPR (v_cmp +
//     v_cndmask); every 256 bytes the wave stores them with one coalesced 4-B-per-lane store;
//   * decode mirrors this: 256 input bytes per coalesced load, v_readlane per consumed byte,
//     decoded bins collected in a 64-bit scalar mask and stored as one byte per lane.
// No MFMA (there is no contraction here), no atomics, no inter-wave communication.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cabac_ctx_tables.h"
#include "cabac_hip.h"
#include "cabac_kernels.h"

namespace cabac {

__constant__ uint8_t c_init_tables[CABAC_CTX_TABLE_ROWS * CABAC_CTX_TABLE_COLS] = {CABAC_CTX_INIT_TABLE_VALUES};

constexpr int kNumCtx = CABAC_NUM_CONTEXTS;
constexpr uint32_t kMask0 = 0x7FE0u;  // contexts.hpp:18-19
constexpr uint32_t kMask1 = 0x7FFEu;  // contexts.hpp:20-21

// LDS context entry: x = state0 | state1 << 16 (the two 15-bit estimators exactly as the
// reference keeps them), y = rate0 | (16 + rate1) << 8 | add1 << 16 is not needed: y = rates.
struct CtxEntry {
  uint32_t state;  // s0 | s1 << 16
  uint32_t rates;  // r0 | r1 << 8
};

__device__ __forceinline__ uint32_t ctx_init_state(int qp, uint32_t init_value) {
  // BinProbModel_Std::init, contexts.cpp:893-901
  int slope = (int)(init_value >> 3) - 4;
  int offset = (int)(init_value & 7) * 18 + 1;
  int st = ((slope * (qp - 16)) >> 1) + offset;
  st = st < 1 ? 1 : (st > 127 ? 127 : st);
  uint32_t p1 = (uint32_t)st << 8;
  return (p1 & kMask0) | ((p1 & kMask1) << 16);
}

__device__ __forceinline__ uint32_t ctx_init_rates(uint32_t w) {
  // setLog2WindowSize, contexts.cpp:915-920
  uint32_t r0 = 2 + ((w >> 2) & 3);
  uint32_t r1 = 3 + r0 + (w & 3);
  return r0 | (r1 << 8);
}

__device__ __forceinline__ void ctx_store_init(CtxEntry *ctx, int qp, uint32_t init_id, int lane) {
  qp = qp < 0 ? 0 : (qp > 63 ? 63 : qp);  // CtxStore::init clips, contexts.cpp:1010
  for (int k = lane; k < kNumCtx; k += 64) {
    CtxEntry e;
    e.state = ctx_init_state(qp, c_init_tables[init_id * kNumCtx + k]);
    e.rates = ctx_init_rates(c_init_tables[3 * kNumCtx + k]);
    ctx[k] = e;
  }
}

// state() >> folded LPS multiplier, contexts.cpp:939-950.  All scalar.
__device__ __forceinline__ uint32_t state8(uint32_t st) { return (((st & 0xffffu) + (st >> 16)) >> 8) & 0xffu; }

__device__ __forceinline__ uint32_t lps_of(uint32_t q8, uint32_t range) {
  uint32_t q = (q8 & 0x80u) ? (q8 ^ 0xffu) : q8;
  return (((q >> 2) * (range >> 5)) >> 1) + 4;
}

// getRenormBitsLPS: m_RenormTable_32[LPS >> 3] == 8 - floor(log2(LPS)) for LPS in 4..255
__device__ __forceinline__ int renorm_bits_lps(uint32_t lps) { return __builtin_clz(lps) - 23; }

// update(bin), contexts.cpp:903-913, on the packed word
__device__ __forceinline__ uint32_t ctx_update(uint32_t st, uint32_t rates, uint32_t bin) {
  uint32_t r0 = rates & 0xffu, r1 = rates >> 8;
  uint32_t s0 = st & 0xffffu, s1 = st >> 16;
  s0 -= (s0 >> r0) & kMask0;
  s1 -= (s1 >> r1) & kMask1;
  if (bin) {
    s0 += (0x7fffu >> r0) & kMask0;
    s1 += (0x7fffu >> r1) & kMask1;
  }
  return s0 | (s1 << 16);
}

// ------------------------------------------------------------------------------------------
// ctx-init only kernel (parity tests of a2/a3 in SURVEY.md §8a)
__global__ __launch_bounds__(64) void ctx_init_kernel(uint32_t n_sub, const int32_t *qp, const uint32_t *init_id,
                                                      uint32_t *state, uint8_t *rate) {
  uint32_t s = blockIdx.x;
  if (s >= n_sub) return;
  int q = qp[s];
  q = q < 0 ? 0 : (q > 63 ? 63 : q);
  uint32_t id = init_id[s] & 3u;
  for (int k = threadIdx.x; k < kNumCtx; k += 64) {
    state[(size_t)s * kNumCtx + k] = ctx_init_state(q, c_init_tables[id * kNumCtx + k]);
    uint32_t r = ctx_init_rates(c_init_tables[3 * kNumCtx + k]);
    rate[(size_t)s * kNumCtx + k] = (uint8_t)(16 * (r & 0xff) + (r >> 8));  // m_rate layout
  }
}

// ------------------------------------------------------------------------------------------
// Output byte sink of one wave: 256-byte window kept in one VGPR (4 bytes per lane).
struct ByteSink {
  uint8_t *dst;       // substream output base (16-B aligned)
  uint32_t cap;       // capacity in bytes
  uint32_t pos;       // bytes produced so far (may run past cap: overflow)
  uint32_t cur;       // scalar word being assembled (little endian)
  uint32_t window;    // per-lane VGPR: word (pos >> 2) & 63 of the current 256-B window
};

__device__ __forceinline__ void sink_flush_window(ByteSink &s, uint32_t window_base, int lane) {
  // full 256-byte window at byte offset window_base
  uint32_t off = window_base + 4u * (uint32_t)lane;
  if (off + 4u <= s.cap) {
    *reinterpret_cast<uint32_t *>(s.dst + off) = s.window;
  } else {
    for (uint32_t b = 0; b < 4; b++)
      if (off + b < s.cap) s.dst[off + b] = (uint8_t)(s.window >> (8 * b));
  }
}

__device__ __forceinline__ void sink_put(ByteSink &s, uint32_t byte, int lane) {
  uint32_t p = s.pos;
  s.cur |= (byte & 0xffu) << (8u * (p & 3u));
  if ((p & 3u) == 3u) {
    s.window = ((uint32_t)lane == ((p >> 2) & 63u)) ? s.cur : s.window;  // v_cmp + v_cndmask
    s.cur = 0;
    if ((p & 255u) == 255u) sink_flush_window(s, p & ~255u, lane);
  }
  s.pos = p + 1;
}

__device__ __forceinline__ void sink_finish(ByteSink &s, int lane) {
  // store the tail of the current window (bytes [pos & ~255, pos))
  uint32_t p = s.pos;
  if (p & 3u) s.window = ((uint32_t)lane == ((p >> 2) & 63u)) ? s.cur : s.window;
  uint32_t base = p & ~255u;
  uint32_t off = base + 4u * (uint32_t)lane;
  for (uint32_t b = 0; b < 4; b++)
    if (off + b < p && off + b < s.cap) s.dst[off + b] = (uint8_t)(s.window >> (8 * b));
}

// ------------------------------------------------------------------------------------------
// Encoder arithmetic state (all wave-uniform)
struct EncState {
  uint32_t low, range, buffered_byte;
  int32_t num_buffered, bits_left;
};

// BinEncoderBase::writeOut, arith_codec.cpp:524-546
__device__ __forceinline__ void enc_write_out(EncState &e, ByteSink &s, int lane) {
  uint32_t lead = e.low >> (24 - e.bits_left);
  e.bits_left += 8;
  e.low &= 0xffffffffu >> e.bits_left;
  if (lead == 0xffu) {
    e.num_buffered++;
  } else if (e.num_buffered > 0) {
    uint32_t carry = lead >> 8;
    uint32_t byte = e.buffered_byte + carry;
    e.buffered_byte = lead & 0xffu;
    sink_put(s, byte, lane);
    byte = (0xffu + carry) & 0xffu;
    while (e.num_buffered > 1) {
      sink_put(s, byte, lane);
      e.num_buffered--;
    }
  } else {
    e.num_buffered = 1;
    e.buffered_byte = lead;
  }
}

// BinEncoderBase::finish, arith_codec.cpp:339-357, then (optionally) writeByteAlignment,
// bit_stream.cpp:152-155.  Returns the number of bits in the stream.
__device__ __forceinline__ uint32_t enc_finish(EncState &e, ByteSink &s, bool do_finish, bool align_rbsp, int lane) {
  uint32_t held = 0, nheld = 0;  // MSB-aligned partial byte
  if (do_finish) {
    if (e.low >> (32 - e.bits_left)) {
      sink_put(s, e.buffered_byte + 1, lane);
      while (e.num_buffered > 1) {
        sink_put(s, 0x00, lane);
        e.num_buffered--;
      }
      e.low -= 1u << (32 - e.bits_left);
    } else {
      if (e.num_buffered > 0) sink_put(s, e.buffered_byte, lane);
      while (e.num_buffered > 1) {
        sink_put(s, 0xff, lane);
        e.num_buffered--;
      }
    }
    // write(low >> 8, 24 - bitsLeft): 1..12 bits, MSB first
    uint32_t nb = (uint32_t)(24 - e.bits_left);
    uint32_t v = e.low >> 8;
    while (nb >= 8) {
      sink_put(s, (v >> (nb - 8)) & 0xffu, lane);
      nb -= 8;
    }
    nheld = nb;
    held = nb ? ((v & ((1u << nb) - 1u)) << (8 - nb)) : 0;
    if (align_rbsp) {
      held |= 1u << (7 - nheld);  // stop bit; the zero pad is already there
      sink_put(s, held, lane);
      held = 0;
      nheld = 0;
    }
  }
  uint32_t n_bits = s.pos * 8u + nheld;
  if (nheld) sink_put(s, held, lane);  // MSB-aligned partial byte follows the whole bytes
  sink_finish(s, lane);
  return n_bits;
}

// ------------------------------------------------------------------------------------------
// encode, v1 (wave-serial with in-register state forwarding)
__global__ __launch_bounds__(64) void encode_kernel_v1(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                       const uint16_t *__restrict__ records, uint8_t *__restrict__ bytes,
                                                       cabac_substream_result *__restrict__ results) {
  __shared__ CtxEntry ctx[kNumCtx + 5];
  const int lane = threadIdx.x;
  const uint32_t sub = blockIdx.x;
  if (sub >= n_sub) return;

  const cabac_substream_desc d = desc[sub];
  const uint32_t n = d.n_records;
  const uint16_t *rec = records + d.rec_offset;

  ctx_store_init(ctx, d.qp, d.init_id & 3u, lane);
  __syncthreads();

  EncState e;
  e.low = 0;
  e.range = 510;
  e.buffered_byte = 0xff;
  e.num_buffered = 0;
  e.bits_left = 23;  // start(), arith_codec.cpp:329-337
  ByteSink sink;
  sink.dst = bytes + d.byte_offset;
  sink.cap = d.byte_capacity;
  sink.pos = 0;
  sink.cur = 0;
  sink.window = 0;
  uint32_t bad = 0;

  uint32_t next_rec = (uint32_t)lane < n ? rec[lane] : 0;
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t cnt = (n - base) < 64u ? (n - base) : 64u;
    const uint32_t r = next_rec;
    {
      uint32_t nxt = base + 64u + (uint32_t)lane;
      next_rec = nxt < n ? rec[nxt] : 0;  // prefetch the next 64 records
    }
    const uint32_t id = r & CABAC_REC_ID_MASK;
    const bool active = (uint32_t)lane < cnt;
    const bool is_ctx = active && id < (uint32_t)kNumCtx;
    if (active && !is_ctx && id < CABAC_REC_ALIGN) bad = 1;
    CtxEntry ce = {0u, 0u};
    if (is_ctx) ce = ctx[id];
    uint32_t st_v = ce.state;
    // per-lane record word for the scalar walk: id | bin << 15 | r0 << 16 | r1 << 24
    const uint32_t info_v = (r & 0xffffu) | (ce.rates << 16);
    const uint32_t key_v = is_ctx ? id : 0xffffu;  // forwarding key

    for (uint32_t i = 0; i < cnt; i++) {
      const uint32_t info = __builtin_amdgcn_readlane(info_v, i);
      const uint32_t rid = info & CABAC_REC_ID_MASK;
      const uint32_t bin = (info >> 15) & 1u;
      int nb = 0;
      if (rid < (uint32_t)kNumCtx) {
        // TBinEncoder::encodeBin, arith_codec.cpp:553-582
        const uint32_t st = __builtin_amdgcn_readlane(st_v, i);
        const uint32_t q8 = state8(st);
        const uint32_t lps = lps_of(q8, e.range);
        e.range -= lps;
        if (bin != (q8 >> 7)) {
          nb = renorm_bits_lps(lps);
          e.low = (e.low + e.range) << nb;
          e.range = lps << nb;
        } else if (e.range < 256u) {
          nb = 1;
          e.low <<= 1;
          e.range <<= 1;
        }
        const uint32_t st_new = ctx_update(st, info >> 16, bin);
        st_v = (key_v == rid) ? st_new : st_v;  // forward to every lane holding this context
      } else if (rid == CABAC_REC_EP) {
        // encodeBinEP, arith_codec.cpp:389-399
        e.low <<= 1;
        if (bin) e.low += e.range;
        nb = 1;
      } else if (rid == CABAC_REC_TRM) {
        // encodeBinTrm, arith_codec.cpp:460-478
        e.range -= 2;
        if (bin) {
          e.low += e.range;
          e.low <<= 7;
          e.range = 2u << 7;
          nb = 7;
        } else if (e.range < 256u) {
          e.low <<= 1;
          e.range <<= 1;
          nb = 1;
        }
      } else if (rid == CABAC_REC_ALIGN) {
        e.range = 256;  // align(), arith_codec.cpp:480
      }
      e.bits_left -= nb;
      if (e.bits_left < 12) enc_write_out(e, sink, lane);
    }
    if (is_ctx) ctx[id].state = st_v;  // lanes of one context all hold its final state
  }

  const uint32_t n_bits = enc_finish(e, sink, (d.init_id & CABAC_SUB_FINISH) != 0,
                                     (d.init_id & CABAC_SUB_ALIGN_RBSP) != 0, lane);
  const uint64_t any_bad = __ballot(bad != 0);
  if (lane == 0) {
    cabac_substream_result res;
    res.n_bits = n_bits;
    res.flags = (sink.pos > sink.cap ? CABAC_RES_OVERFLOW : 0u) | (any_bad ? CABAC_RES_BAD_RECORD : 0u);
    results[sub] = res;
  }
}

// ------------------------------------------------------------------------------------------
// Input byte source of one wave: 256-byte window in one VGPR
struct ByteSource {
  const uint8_t *src;
  uint32_t cap;     // valid bytes
  uint32_t pos;     // next byte to read
  uint32_t window;  // per-lane VGPR: bytes [wbase + 4*lane, +4)
  uint32_t underrun;
};

__device__ __forceinline__ uint32_t source_load_window(const ByteSource &s, uint32_t wbase, int lane) {
  uint32_t off = wbase + 4u * (uint32_t)lane;
  uint32_t w = 0;
  if (off + 4u <= s.cap) {
    w = *reinterpret_cast<const uint32_t *>(s.src + off);
  } else {
    for (uint32_t b = 0; b < 4; b++)
      if (off + b < s.cap) w |= (uint32_t)s.src[off + b] << (8 * b);
  }
  return w;
}

// InputBitstream::readByte, bit_stream.cpp:268-274
__device__ __forceinline__ uint32_t source_get(ByteSource &s, int lane) {
  uint32_t p = s.pos;
  if (p >= s.cap) s.underrun = 1;
  uint32_t w = __builtin_amdgcn_readlane(s.window, (p >> 2) & 63u);
  uint32_t b = (w >> (8u * (p & 3u))) & 0xffu;
  s.pos = p + 1;
  if (((p + 1) & 255u) == 0u) s.window = source_load_window(s, p + 1, lane);
  return b;
}

// decode, v1
__global__ __launch_bounds__(64) void decode_kernel_v1(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                       const uint16_t *__restrict__ records,
                                                       const uint8_t *__restrict__ bytes, uint8_t *__restrict__ bins,
                                                       cabac_substream_result *__restrict__ results) {
  __shared__ CtxEntry ctx[kNumCtx + 5];
  const int lane = threadIdx.x;
  const uint32_t sub = blockIdx.x;
  if (sub >= n_sub) return;

  const cabac_substream_desc d = desc[sub];
  const uint32_t n = d.n_records;
  const uint16_t *rec = records + d.rec_offset;
  uint8_t *out = bins + d.rec_offset;

  ctx_store_init(ctx, d.qp, d.init_id & 3u, lane);
  __syncthreads();

  ByteSource src;
  src.src = bytes + d.byte_offset;
  src.cap = d.byte_capacity;
  src.pos = 0;
  src.underrun = 0;
  src.window = source_load_window(src, 0, lane);

  // BinDecoderBase::start, arith_codec.cpp:60-66
  uint32_t range = 510;
  uint32_t value = source_get(src, lane) << 8;
  value += source_get(src, lane);
  int32_t bits_needed = -8;
  uint32_t bad = 0;

  uint32_t next_rec = (uint32_t)lane < n ? rec[lane] : 0;
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t cnt = (n - base) < 64u ? (n - base) : 64u;
    const uint32_t r = next_rec;
    {
      uint32_t nxt = base + 64u + (uint32_t)lane;
      next_rec = nxt < n ? rec[nxt] : 0;
    }
    const uint32_t id = r & CABAC_REC_ID_MASK;
    const bool active = (uint32_t)lane < cnt;
    const bool is_ctx = active && id < (uint32_t)kNumCtx;
    if (active && !is_ctx && id < CABAC_REC_ALIGN) bad = 1;
    CtxEntry ce = {0u, 0u};
    if (is_ctx) ce = ctx[id];
    uint32_t st_v = ce.state;
    const uint32_t info_v = id | (ce.rates << 16);
    const uint32_t key_v = is_ctx ? id : 0xffffu;
    uint64_t bin_mask = 0;

    for (uint32_t i = 0; i < cnt; i++) {
      const uint32_t info = __builtin_amdgcn_readlane(info_v, i);
      const uint32_t rid = info & CABAC_REC_ID_MASK;
      uint32_t bin = 0;
      if (rid < (uint32_t)kNumCtx) {
        // TBinDecoder::decodeBin, arith_codec.cpp:242-277
        const uint32_t st = __builtin_amdgcn_readlane(st_v, i);
        const uint32_t q8 = state8(st);
        bin = q8 >> 7;
        const uint32_t lps = lps_of(q8, range);
        range -= lps;
        const uint32_t sr = range << 7;
        int nb = 0;
        if (value < sr) {
          if (range < 256u) {
            nb = 1;
            range <<= 1;
            value <<= 1;
          }
        } else {
          bin = 1u - bin;
          nb = renorm_bits_lps(lps);
          value = (value - sr) << nb;
          range = lps << nb;
        }
        bits_needed += nb;
        if (nb != 0 && bits_needed >= 0) {
          value += source_get(src, lane) << bits_needed;
          bits_needed -= 8;
        }
        const uint32_t st_new = ctx_update(st, info >> 16, bin);
        st_v = (key_v == rid) ? st_new : st_v;
      } else if (rid == CABAC_REC_EP) {
        // decodeBinEP, arith_codec.cpp:100-114
        value += value;
        if (++bits_needed >= 0) {
          value += source_get(src, lane);
          bits_needed = -8;
        }
        const uint32_t sr = range << 7;
        if (value >= sr) {
          value -= sr;
          bin = 1;
        }
      } else if (rid == CABAC_REC_TRM) {
        // decodeBinTrm, arith_codec.cpp:181-197
        range -= 2;
        const uint32_t sr = range << 7;
        if (value >= sr) {
          bin = 1;
        } else if (range < 256u) {
          range += range;
          value += value;
          if (++bits_needed == 0) {
            value += source_get(src, lane);
            bits_needed = -8;
          }
        }
      } else if (rid == CABAC_REC_ALIGN) {
        range = 256;
      }
      bin_mask |= (uint64_t)bin << i;
    }
    if (is_ctx) ctx[id].state = st_v;
    if (active) out[base + lane] = (uint8_t)((bin_mask >> lane) & 1u);
  }

  uint32_t flags = 0;
  if (d.init_id & CABAC_SUB_FINISH) {
    // BinDecoderBase::finish, arith_codec.cpp:68-73
    uint32_t ok = 0;
    if (src.pos >= 1 && src.pos <= src.cap) {
      uint32_t last = src.src[src.pos - 1];
      ok = ((last << (8 + bits_needed)) & 0xffu) == 0x80u;
    }
    if (!ok) flags |= CABAC_RES_BAD_STOP;
  }
  const uint64_t any_bad = __ballot(bad != 0);
  if (lane == 0) {
    cabac_substream_result res;
    res.n_bits = 8u * src.pos + (uint32_t)bits_needed;
    if (src.underrun) flags |= CABAC_RES_UNDERRUN;
    if (any_bad) flags |= CABAC_RES_BAD_RECORD;
    res.flags = flags;
    results[sub] = res;
  }
}

// ------------------------------------------------------------------------------------------
// launchers (called from cabac_capi.cpp)
hipError_t launch_ctx_init(hipStream_t st, uint32_t n_sub, const int32_t *qp, const uint32_t *init_id, uint32_t *state,
                           uint8_t *rate) {
  if (n_sub == 0) return hipSuccess;
  hipLaunchKernelGGL(ctx_init_kernel, dim3(n_sub), dim3(64), 0, st, n_sub, qp, init_id, state, rate);
  return hipGetLastError();
}

hipError_t launch_encode(hipStream_t st, int variant, uint32_t n_sub, const cabac_substream_desc *desc,
                         const uint16_t *records, uint8_t *bytes, cabac_substream_result *results) {
  if (n_sub == 0) return hipSuccess;
  (void)variant;
  hipLaunchKernelGGL(encode_kernel_v1, dim3(n_sub), dim3(64), 0, st, n_sub, desc, records, bytes, results);
  return hipGetLastError();
}

hipError_t launch_decode(hipStream_t st, int variant, uint32_t n_sub, const cabac_substream_desc *desc,
                         const uint16_t *records, const uint8_t *bytes, uint8_t *bins,
                         cabac_substream_result *results) {
  if (n_sub == 0) return hipSuccess;
  (void)variant;
  hipLaunchKernelGGL(decode_kernel_v1, dim3(n_sub), dim3(64), 0, st, n_sub, desc, records, bytes, bins, results);
  return hipGetLastError();
}

}  // namespace cabac
