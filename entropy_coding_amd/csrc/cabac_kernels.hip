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

#include "cabac_device.h"
#include "cabac_hip.h"
#include "cabac_kernels.h"

namespace cabac {

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
    const uint32_t r = next_rec;  // loaded one step ago
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
    {  // prefetch the next 64 records; placed after the uses of `r` so that hipcc's vmcnt(0) wait
       // in front of them does not also wait for this load
      const uint32_t nxt = base + 64u + (uint32_t)lane;
      next_rec = nxt < n ? rec[nxt] : 0;
    }

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
    const uint32_t r = next_rec;  // loaded one step ago
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
    {  // prefetch the next 64 records; placed after the uses of `r` so that hipcc's vmcnt(0) wait
       // in front of them does not also wait for this load
      const uint32_t nxt = base + 64u + (uint32_t)lane;
      next_rec = nxt < n ? rec[nxt] : 0;
    }

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
    if (!ok && !src.underrun) flags |= CABAC_RES_BAD_STOP;  // an underrun throws before finish() is reached
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

// ==========================================================================================
// v2 "lane-per-substream": every lane walks its own substream with ordinary per-lane (VALU)
// arithmetic; a workgroup is ONE wave of L <= 32 active lanes, L chosen at launch.  Why: v1's chain is
// ~50 *scalar* instructions per bin and a CU has a single scalar issue port shared by all its waves,
// so with 16 waves per CU (C4: 4 096 substreams) each wave gets one instruction per ~16 cycles.
// SIMT rules shape the code: different substreams take the LPS / bypass / terminate / byte-output
// paths at different times, so every record type goes through ONE branch-free sequence (selects,
// not branches); the only real branches are the rare ones (byte output, refill, align).
//   * bypass  == a context bin with LPS width t = 0 (range untouched) plus a 1-bit post shift;
//   * terminate == a context bin with t = 2 and "LPS" == bin: clz(2) - 23 = 7 gives exactly the
//     7-bit renormalisation and range = 2 << 7 of encodeBinTrm(1) (arith_codec.cpp:460-478).
// Per-lane context store in LDS: 381 words (379 + a dummy slot that absorbs the always-executed
// read/write of non-context records; odd stride spreads lanes over banks);
// word = state0[14:5] | rate bits[4:0] | state1 << 16, rate bits [1:0] = rate0 - 2, [4:2] = rate1 - 5
// (rate1 <= 9 is CHECKed by the reference, contexts.cpp:919).  The rate bits ride along untouched
// because every update term is masked.
__device__ __forceinline__ void lane_ctx_init(uint32_t *ctx, int qp_in, uint32_t iid) {
  const int qp = qp_in < 0 ? 0 : (qp_in > 63 ? 63 : qp_in);
  for (int k = 0; k < kNumCtx; k++)
    ctx[k] = ctx2_init(qp, c_init_tables[iid * kNumCtx + k], c_init_tables[3 * kNumCtx + k]);
  ctx[kDummySlot] = 0;
}

// per-lane byte sink: 4 bytes assembled in a register, stored as one dword
struct LaneSink {
  uint8_t *dst;
  uint32_t cap, pos, cur;
};

__device__ __forceinline__ void lane_put(LaneSink &s, uint32_t byte) {
  const uint32_t p = s.pos;
  s.cur |= (byte & 0xffu) << (8u * (p & 3u));
  if ((p & 3u) == 3u) {
    const uint32_t off = p & ~3u;
    if (off + 4u <= s.cap) {
      *reinterpret_cast<uint32_t *>(s.dst + off) = s.cur;
    } else {
      for (uint32_t b = 0; b < 4; b++)
        if (off + b < s.cap) s.dst[off + b] = (uint8_t)(s.cur >> (8 * b));
    }
    s.cur = 0;
  }
  s.pos = p + 1;
}

__device__ __forceinline__ void lane_sink_finish(LaneSink &s) {
  const uint32_t p = s.pos, off = p & ~3u;
  for (uint32_t b = 0; b < (p & 3u); b++)
    if (off + b < s.cap) s.dst[off + b] = (uint8_t)(s.cur >> (8 * b));
}

// writeOut, arith_codec.cpp:524-546 (per lane; selects except for the actual byte output)
__device__ __forceinline__ void lane_write_out(EncState &e, LaneSink &s) {
  const uint32_t lead = e.low >> (24 - e.bits_left);
  e.bits_left += 8;
  e.low &= 0xffffffffu >> e.bits_left;
  const bool is_ff = lead == 0xffu;
  const bool emit = !is_ff && e.num_buffered > 0;
  const uint32_t carry = lead >> 8;
  const uint32_t first = e.buffered_byte + carry;
  const int32_t fill_n = e.num_buffered - 1;
  e.buffered_byte = is_ff ? e.buffered_byte : (lead & 0xffu);
  e.num_buffered = is_ff ? e.num_buffered + 1 : 1;
  if (emit) {
    lane_put(s, first);
    for (int32_t k = 0; k < fill_n; k++) lane_put(s, 0xffu + carry);
  }
}

__device__ __forceinline__ void lane_encode_record(uint32_t r, uint32_t *ctx, EncState &e, LaneSink &sink, uint32_t &bad) {
  const uint32_t id = r & CABAC_REC_ID_MASK;
  const uint32_t bin = (r >> 15) & 1u;
  const bool is_ctx = id < (uint32_t)kNumCtx;
  const bool is_ep = id == CABAC_REC_EP;
  const bool is_trm = id == CABAC_REC_TRM;
  const uint32_t slot = is_ctx ? id : kDummySlot;
  const uint32_t st = ctx[slot];
  const uint32_t q8 = ctx2_q8(st);
  const uint32_t mps = q8 >> 7;
  const uint32_t k = is_ctx ? ctx2_k(q8) : 0u;
  const uint32_t c = is_ctx ? 4u : (is_trm ? 2u : 0u);
  const uint32_t t = (((e.range >> 5) * k) >> 1) + c;            // LPS width (getLPS, contexts.cpp:945-950)
  const bool lps_path = is_ctx ? (bin != mps) : (is_trm && bin);
  const uint32_t rm = e.range - t;
  const int nl = __builtin_clz(t | 1u) - 23;                     // getRenormBitsLPS; unused when t == 0
  const int nm = rm < 256u ? 1 : 0;
  const int n = lps_path ? nl : nm;
  e.low = (e.low + (lps_path ? rm : 0u)) << n;
  e.range = (lps_path ? t : rm) << n;
  // bypass: low = (low << 1) + bin * range (encodeBinEP, arith_codec.cpp:389-399)
  const uint32_t ep = is_ep ? 1u : 0u;
  e.low = (e.low << ep) + ((is_ep && bin) ? e.range : 0u);
  if (id == CABAC_REC_ALIGN) e.range = 256;                      // align(), :480
  bad |= (!is_ctx && id < CABAC_REC_ALIGN) ? 1u : 0u;
  ctx[slot] = ctx2_update(st, bin);
  e.bits_left -= n + (int)ep;
  if (e.bits_left < 12) lane_write_out(e, sink);
}

// Launch geometry of v2: 4 waves per workgroup (so that the workgroup's waves are dealt to the CU's four
// SIMDs), `lanes` active lanes per wave.
constexpr uint32_t kV2Waves = 4;

__global__ __launch_bounds__(256) void encode_kernel_v2(uint32_t n_sub, uint32_t lanes,
                                                        const cabac_substream_desc *__restrict__ desc,
                                                        const uint16_t *__restrict__ records, uint8_t *__restrict__ bytes,
                                                        cabac_substream_result *__restrict__ results) {
  extern __shared__ uint32_t lds[];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t sub = (blockIdx.x * kV2Waves + wave) * lanes + lane;
  if (lane >= lanes || sub >= n_sub) return;
  uint32_t *ctx = lds + (wave * lanes + lane) * kLaneStride;

  const cabac_substream_desc d = desc[sub];
  lane_ctx_init(ctx, d.qp, d.init_id & 3u);
  EncState e;
  e.low = 0;
  e.range = 510;
  e.buffered_byte = 0xff;
  e.num_buffered = 0;
  e.bits_left = 23;
  LaneSink sink;
  sink.dst = bytes + d.byte_offset;
  sink.cap = d.byte_capacity;
  sink.pos = 0;
  sink.cur = 0;
  uint32_t bad = 0;

  const uint16_t *rec = records + d.rec_offset;
  const uint32_t n = d.n_records;
  uint32_t i = 0;
  // head: single records until the pointer is 16-byte aligned
  while (i < n && (reinterpret_cast<uintptr_t>(rec + i) & 15u)) lane_encode_record(rec[i++], ctx, e, sink, bad);
  // body: 8 records per 16-byte load, next group prefetched while this one is coded
  if (i + 8 <= n) {
    uint4 nxt = *reinterpret_cast<const uint4 *>(rec + i);
    while (i + 8 <= n) {
      const uint4 cur = nxt;
      if (i + 16 <= n) nxt = *reinterpret_cast<const uint4 *>(rec + i + 8);
      uint32_t g0 = cur.x, g1 = cur.y, g2 = cur.z, g3 = cur.w;
#pragma unroll 1
      for (int k = 0; k < 8; k++) {  // rolled on purpose: keeps the loop body small
        lane_encode_record(g0 & 0xffffu, ctx, e, sink, bad);
        g0 = __builtin_amdgcn_alignbit(g1, g0, 16);
        g1 = __builtin_amdgcn_alignbit(g2, g1, 16);
        g2 = __builtin_amdgcn_alignbit(g3, g2, 16);
        g3 >>= 16;
      }
      i += 8;
    }
  }
  while (i < n) lane_encode_record(rec[i++], ctx, e, sink, bad);

  // finish(), arith_codec.cpp:339-357 (+ writeByteAlignment, bit_stream.cpp:152-155)
  uint32_t held = 0, nheld = 0;
  if (d.init_id & CABAC_SUB_FINISH) {
    if (e.low >> (32 - e.bits_left)) {
      lane_put(sink, e.buffered_byte + 1);
      while (e.num_buffered > 1) {
        lane_put(sink, 0x00);
        e.num_buffered--;
      }
      e.low -= 1u << (32 - e.bits_left);
    } else {
      if (e.num_buffered > 0) lane_put(sink, e.buffered_byte);
      while (e.num_buffered > 1) {
        lane_put(sink, 0xff);
        e.num_buffered--;
      }
    }
    uint32_t nbf = (uint32_t)(24 - e.bits_left);
    const uint32_t v = e.low >> 8;
    while (nbf >= 8) {
      lane_put(sink, (v >> (nbf - 8)) & 0xffu);
      nbf -= 8;
    }
    nheld = nbf;
    held = nbf ? ((v & ((1u << nbf) - 1u)) << (8 - nbf)) : 0;
    if (d.init_id & CABAC_SUB_ALIGN_RBSP) {
      held |= 1u << (7 - nheld);
      lane_put(sink, held);
      held = 0;
      nheld = 0;
    }
  }
  cabac_substream_result res;
  res.n_bits = sink.pos * 8u + nheld;
  if (nheld) lane_put(sink, held);
  lane_sink_finish(sink);
  res.flags = (sink.pos > sink.cap ? CABAC_RES_OVERFLOW : 0u) | (bad ? CABAC_RES_BAD_RECORD : 0u);
  results[sub] = res;
}

// ---- decode, v2 --------------------------------------------------------------------------
// The reference keeps a 16-bit `value` plus up to 8 prefetched bits and reads one byte each time
// `bitsNeeded` crosses zero (arith_codec.cpp:257-260).  Bits below the compared 9 never influence a
// decision before they are shifted up, so fetching them earlier is exactly equivalent; here the
// window is 64 bits — value in [62:47] (bit 63 is headroom: decodeBinEP doubles value before it
// compares, arith_codec.cpp:101), 9..47 valid look-ahead bits below — refilled 4 bytes at a time with
// one rare branch.  S = total bits shifted gives the reference's counters back:
// bytes read = 2 + S/8, bitsNeeded = S%8 - 8.
struct LaneWindow {
  const uint8_t *src;
  uint32_t cap;
  uint32_t hi, lo;   // the 64-bit window
  int32_t look;      // valid look-ahead bits below bit 47
  uint32_t rp;       // byte offset of the next refill
  uint32_t nxt;      // prefetched dword at rp, still little-endian (swapped when consumed, so that
                     // the load's latency hides behind ~36 bins instead of being waited for at once)
};

__device__ __forceinline__ void window_shift(LaneWindow &w, int n) {
  uint64_t v = ((uint64_t)w.hi << 32) | w.lo;
  v <<= n;
  w.hi = (uint32_t)(v >> 32);
  w.lo = (uint32_t)v;
  w.look -= n;
}

__global__ __launch_bounds__(256) void decode_kernel_v2(uint32_t n_sub, uint32_t lanes,
                                                        const cabac_substream_desc *__restrict__ desc,
                                                        const uint16_t *__restrict__ records,
                                                        const uint8_t *__restrict__ bytes, uint8_t *__restrict__ bins,
                                                        cabac_substream_result *__restrict__ results) {
  extern __shared__ uint32_t lds[];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t sub = (blockIdx.x * kV2Waves + wave) * lanes + lane;
  if (lane >= lanes || sub >= n_sub) return;
  uint32_t *ctx = lds + (wave * lanes + lane) * kLaneStride;

  const cabac_substream_desc d = desc[sub];
  lane_ctx_init(ctx, d.qp, d.init_id & 3u);

  LaneWindow w;
  w.src = bytes + d.byte_offset;
  w.cap = d.byte_capacity;
  {
    const uint32_t first = __builtin_bswap32(lane_load_le32(w.src, w.cap, 0));  // start(): value = first two bytes (arith_codec.cpp:60-66)
    w.hi = first >> 1;
    w.lo = first << 31;
  }
  w.look = 16;
  w.rp = 4;
  w.nxt = lane_load_le32(w.src, w.cap, 4);
  uint32_t range = 510;
  uint32_t shifts = 0;  // S
  uint32_t bad = 0;

  const uint16_t *rec = records + d.rec_offset;
  uint8_t *out = bins + d.rec_offset;
  const uint32_t n = d.n_records;

  auto decode_one = [&](uint32_t r) -> uint32_t {
    if (w.look <= 15) {  // rare: every 32 consumed bits
      const uint64_t add = (uint64_t)__builtin_bswap32(w.nxt) << (15 - w.look);
      w.hi |= (uint32_t)(add >> 32);
      w.lo |= (uint32_t)add;
      w.look += 32;
      w.rp += 4;
      w.nxt = lane_load_le32(w.src, w.cap, w.rp);
    }
    const uint32_t id = r & CABAC_REC_ID_MASK;
    const bool is_ctx = id < (uint32_t)kNumCtx;
    const bool is_ep = id == CABAC_REC_EP;
    const bool is_trm = id == CABAC_REC_TRM;
    const uint32_t slot = is_ctx ? id : kDummySlot;
    const uint32_t st = ctx[slot];
    const uint32_t q8 = ctx2_q8(st);
    const uint32_t mps = q8 >> 7;
    const uint32_t k = is_ctx ? ctx2_k(q8) : 0u;
    const uint32_t c = is_ctx ? 4u : (is_trm ? 2u : 0u);
    const int ep = is_ep ? 1 : 0;
    window_shift(w, ep);                                          // decodeBinEP doubles value first (:100-105)
    const uint32_t t = (((range >> 5) * k) >> 1) + c;
    const uint32_t rm = range - t;
    const uint32_t sr = rm << 22;                                 // (range << 7) aligned to window bits [62:47]
    const bool ge = w.hi >= sr;
    // value >= scaledRange: LPS for a context bin (:262-273), bin 1 for bypass (:109-112) / terminate (:184-185)
    const uint32_t bin = is_ctx ? (ge ? 1u - mps : mps) : (ge ? 1u : 0u);
    const int nl = __builtin_clz(t | 1u) - 23;
    const int nm = rm < 256u ? 1 : 0;
    const bool renorm_lps = ge && is_ctx;
    const int nsh = ge ? (is_ctx ? nl : 0) : nm;
    w.hi -= (ge && !is_trm) ? sr : 0u;                            // terminate bin 1 leaves value untouched
    range = (renorm_lps ? t : rm) << nsh;
    window_shift(w, nsh);
    shifts += (uint32_t)(nsh + ep);
    if (id == CABAC_REC_ALIGN) range = 256;
    bad |= (!is_ctx && id < CABAC_REC_ALIGN) ? 1u : 0u;
    ctx[slot] = ctx2_update(st, bin);
    return bin;
  };

  uint32_t i = 0;
  while (i < n && (reinterpret_cast<uintptr_t>(rec + i) & 15u)) {
    out[i] = (uint8_t)decode_one(rec[i]);
    i++;
  }
  if (i + 8 <= n) {
    uint4 nxt = *reinterpret_cast<const uint4 *>(rec + i);
    const bool out_aligned = (reinterpret_cast<uintptr_t>(out + i) & 3u) == 0;
    while (i + 8 <= n) {
      const uint4 cur = nxt;
      if (i + 16 <= n) nxt = *reinterpret_cast<const uint4 *>(rec + i + 8);
      uint32_t g0 = cur.x, g1 = cur.y, g2 = cur.z, g3 = cur.w;
      uint32_t lo = 0, hi = 0;  // 8 decoded bins, one byte each (bin k in byte k)
#pragma unroll 1
      for (int k = 0; k < 8; k++) {
        const uint32_t b = decode_one(g0 & 0xffffu);
        g0 = __builtin_amdgcn_alignbit(g1, g0, 16);
        g1 = __builtin_amdgcn_alignbit(g2, g1, 16);
        g2 = __builtin_amdgcn_alignbit(g3, g2, 16);
        g3 >>= 16;
        lo = __builtin_amdgcn_alignbit(hi, lo, 8);   // shift the 64-bit byte queue right by one byte
        hi = (hi >> 8) | (b << 24);
      }
      if (out_aligned) {
        reinterpret_cast<uint32_t *>(out + i)[0] = lo;
        reinterpret_cast<uint32_t *>(out + i)[1] = hi;
      } else {
        for (int k = 0; k < 4; k++) {
          out[i + k] = (uint8_t)(lo >> (8 * k));
          out[i + 4 + k] = (uint8_t)(hi >> (8 * k));
        }
      }
      i += 8;
    }
  }
  while (i < n) {
    out[i] = (uint8_t)decode_one(rec[i]);
    i++;
  }

  // the reference's counters from S (see the comment above LaneWindow)
  const uint32_t bytes_read = 2u + (shifts >> 3);
  const int32_t bits_needed = (int32_t)(shifts & 7u) - 8;
  uint32_t flags = 0;
  if (d.init_id & CABAC_SUB_FINISH) {
    // BinDecoderBase::finish, arith_codec.cpp:68-73
    uint32_t ok = 0;
    if (bytes_read <= w.cap) {
      const uint32_t last = w.src[bytes_read - 1];
      ok = ((last << (8 + bits_needed)) & 0xffu) == 0x80u;
    }
    if (!ok && bytes_read <= w.cap) flags |= CABAC_RES_BAD_STOP;  // an underrun throws before finish() is reached
  }
  if (bytes_read > w.cap) flags |= CABAC_RES_UNDERRUN;  // the reference throws "FIFO exceeded" at that read
  if (bad) flags |= CABAC_RES_BAD_RECORD;
  cabac_substream_result res;
  res.n_bits = 8u * bytes_read + (uint32_t)bits_needed;
  res.flags = flags;
  results[sub] = res;
}

// ==========================================================================================
// v3 "phased wave": one wavefront per substream like v1, restructured around the measured gfx950
// cost model (tools/ubench_*.hip): every instruction of a wave costs ~4 issue cycles, a CU retires
// about one scalar and one vector instruction per cycle in total, a VALU->SALU->VALU round trip is
// ~55 cycles and a taken branch ~35.  With 16 waves per CU (C4) the kernel is bound by the number of
// *scalar* instructions per bin, so everything that does not belong to the serial low/range chain is
// moved onto the 64 lanes:
//  encode: the context-state sequence of a 64-bin step does not depend on low/range at all (the bins
//    are known), so it is resolved in parallel first: a 9-bit match-any groups the lanes by ctxId,
//    each lane pulls the updated state from the previous lane of its group (ds_bpermute), one round
//    per repeat of a context inside the step; every lane then derives its own LPS factor / LPS-or-MPS
//    flag.  The scalar loop that remains is ~20 instructions per bin: range split, renormalisation,
//    low update, byte output.
//  decode: the bin is only known after the compare, so the update cannot run ahead; instead ALL lanes
//    apply the decoded bin to their own copy of the state and the lanes holding that ctxId keep the
//    result (vector work), and every lane re-derives its LPS factor, so the scalar chain only does the
//    interval arithmetic.  Scalar and vector instruction counts per bin end up about equal.
// Both use the packed context word of v2 (state0 | rate bits | state1 << 16).

__device__ __forceinline__ uint64_t match_any9(uint32_t key) {
  uint64_t m = ~0ull;
#pragma unroll
  for (int b = 0; b < 9; b++) {
    const bool bit = (key >> b) & 1u;
    const uint64_t bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return m;
}

// info word of one bin for the scalar encode loop
enum : uint32_t { kInfoLps = 0x100u, kInfoEp = 0x200u, kInfoEpOne = 0x400u, kInfoAlign = 0x800u };

template <bool kAlign>
__device__ __forceinline__ void enc3_step(uint32_t info, EncState &e, ByteSink &sink, int lane) {
  int nb;
  if (info & kInfoEp) {
    // encodeBinEP, arith_codec.cpp:389-399
    e.low = (e.low << 1) + ((info & kInfoEpOne) ? e.range : 0u);
    nb = 1;
  } else {
    // encodeBin / encodeBinTrm with LPS width t (arith_codec.cpp:553-582, :460-478)
    const uint32_t k = info & 31u, c = (info >> 5) & 7u;
    const uint32_t t = (((e.range >> 5) * k) >> 1) + c;
    const uint32_t rm = e.range - t;
    if (info & kInfoLps) {
      nb = __builtin_clz(t) - 23;
      e.low = (e.low + rm) << nb;
      e.range = t << nb;
    } else {
      nb = (int)((rm >> 8) ^ 1u);
      e.low <<= nb;
      e.range = rm << nb;
    }
    if (kAlign && (info & kInfoAlign)) e.range = 256;
  }
  e.bits_left -= nb;
  if (e.bits_left < 12) enc_write_out(e, sink, lane);
}

__global__ __launch_bounds__(64) void encode_kernel_v3(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                       const uint16_t *__restrict__ records, uint8_t *__restrict__ bytes,
                                                       cabac_substream_result *__restrict__ results) {
  __shared__ uint32_t ctx[kNumCtx + 5];
  const int lane = threadIdx.x;
  const uint32_t sub = blockIdx.x;
  if (sub >= n_sub) return;

  const cabac_substream_desc d = desc[sub];
  const uint32_t n = d.n_records;
  const uint16_t *rec = records + d.rec_offset;
  {
    const int qp = d.qp < 0 ? 0 : (d.qp > 63 ? 63 : d.qp);
    const uint32_t iid = d.init_id & 3u;
    for (int k = lane; k < kNumCtx; k += 64)
      ctx[k] = ctx2_init(qp, c_init_tables[iid * kNumCtx + k], c_init_tables[3 * kNumCtx + k]);
  }
  __syncthreads();

  EncState e;
  e.low = 0;
  e.range = 510;
  e.buffered_byte = 0xff;
  e.num_buffered = 0;
  e.bits_left = 23;
  ByteSink sink;
  sink.dst = bytes + d.byte_offset;
  sink.cap = d.byte_capacity;
  sink.pos = 0;
  sink.cur = 0;
  sink.window = 0;
  uint32_t bad = 0;
  const uint64_t lt_mask = (1ull << lane) - 1ull;

  uint32_t next_rec = (uint32_t)lane < n ? rec[lane] : 0;
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t cnt = (n - base) < 64u ? (n - base) : 64u;
    const uint32_t r = next_rec;  // loaded one step ago
    const bool active = (uint32_t)lane < cnt;
    const uint32_t id = active ? (r & CABAC_REC_ID_MASK) : CABAC_REC_ID_MASK;
    const uint32_t bin = (r >> 15) & 1u;
    const bool is_ctx = id < (uint32_t)kNumCtx;
    const bool is_ep = active && id == CABAC_REC_EP;
    const bool is_trm = active && id == CABAC_REC_TRM;
    const bool is_align = active && id == CABAC_REC_ALIGN;
    if (active && !is_ctx && id < CABAC_REC_ALIGN) bad = 1;

    // ---- phase A: the context state each bin sees (parallel over the 64 bins) ----------------
    const uint64_t same = match_any9(id);
    const uint64_t before = same & lt_mask;
    const uint32_t prev = 63u - (uint32_t)__builtin_clzll(before | 1ull);
    const bool is_last = (same & ~lt_mask & ~(1ull << lane)) == 0;
    uint32_t st = is_ctx ? ctx[id] : 0u;
    bool pending = is_ctx && before != 0;
    for (;;) {
      const uint64_t pend = __ballot(pending);
      if (pend == 0) break;
      const uint32_t post = ctx2_update(st, bin);
      const uint32_t pulled = __shfl(post, (int)prev);
      if (pending && !((pend >> prev) & 1ull)) {  // the previous bin of this context is settled
        st = pulled;
        pending = false;
      }
    }
    if (is_ctx && is_last) ctx[id] = ctx2_update(st, bin);
    const uint32_t q8 = ctx2_q8(st);
    const uint32_t mps = q8 >> 7;
    uint32_t info = 0;
    if (is_ctx) info = ctx2_k(q8) | (4u << 5) | ((bin ^ mps) ? kInfoLps : 0u);
    if (is_trm) info = (2u << 5) | (bin ? kInfoLps : 0u);
    if (is_ep) info = kInfoEp | (bin ? kInfoEpOne : 0u);
    if (is_align) info = kInfoAlign;
    {  // prefetch the next 64 records; placed after the uses of `r` so that hipcc's vmcnt(0) wait
       // in front of them does not also wait for this load
      const uint32_t nxt = base + 64u + (uint32_t)lane;
      next_rec = nxt < n ? rec[nxt] : 0;
    }
    const bool any_align = __ballot(is_align) != 0;

    // ---- phase B: the serial low / range chain (wave-uniform) -------------------------------
    if (!any_align && cnt == 64u) {
#pragma unroll 8
      for (uint32_t i = 0; i < 64u; i++) enc3_step<false>(__builtin_amdgcn_readlane(info, i), e, sink, lane);
    } else {
      for (uint32_t i = 0; i < cnt; i++) enc3_step<true>(__builtin_amdgcn_readlane(info, i), e, sink, lane);
    }
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

// ---- decode, v3 ----------------------------------------------------------------------------
enum : uint32_t { kDecEp = 0x8u, kDecCtx = 0x10u, kDecAlign = 0x20u };  // bits 2..0 = c, bits 16.. = ctxId

__device__ __forceinline__ uint32_t dec3_window_load(const uint8_t *src, uint32_t cap, uint32_t off) {
  // one big-endian dword per lane: lane l holds stream bytes [off + 4l, +4), first byte in bits 31..24
  return __builtin_bswap32(lane_load_le32(src, cap, off));
}

__global__ __launch_bounds__(64) void decode_kernel_v3(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                       const uint16_t *__restrict__ records,
                                                       const uint8_t *__restrict__ bytes, uint8_t *__restrict__ bins,
                                                       cabac_substream_result *__restrict__ results) {
  __shared__ uint32_t ctx[kNumCtx + 5];
  const int lane = threadIdx.x;
  const uint32_t sub = blockIdx.x;
  if (sub >= n_sub) return;

  const cabac_substream_desc d = desc[sub];
  const uint32_t n = d.n_records;
  const uint16_t *rec = records + d.rec_offset;
  uint8_t *out = bins + d.rec_offset;
  {
    const int qp = d.qp < 0 ? 0 : (d.qp > 63 ? 63 : d.qp);
    const uint32_t iid = d.init_id & 3u;
    for (int k = lane; k < kNumCtx; k += 64)
      ctx[k] = ctx2_init(qp, c_init_tables[iid * kNumCtx + k], c_init_tables[3 * kNumCtx + k]);
  }
  __syncthreads();

  const uint8_t *src = bytes + d.byte_offset;
  const uint32_t cap = d.byte_capacity;
  // input: 256-byte windows, 4 bytes per lane, the next window prefetched
  uint32_t win_cur = dec3_window_load(src, cap, 4u * (uint32_t)lane);
  uint32_t win_nxt = dec3_window_load(src, cap, 256u + 4u * (uint32_t)lane);
  // 64-bit decode window as in v2: value in [62:47], `look` valid bits below (scalar)
  uint32_t hi, lo;
  {
    const uint32_t first = __builtin_amdgcn_readlane(win_cur, 0);
    hi = first >> 1;
    lo = first << 31;
  }
  int32_t look = 16;
  uint32_t rp = 4;  // byte offset of the next refill dword
  uint32_t range = 510, shifts = 0, bad = 0;

  uint32_t next_rec = (uint32_t)lane < n ? rec[lane] : 0;
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t cnt = (n - base) < 64u ? (n - base) : 64u;
    const uint32_t r = next_rec;  // loaded one step ago
    const bool active = (uint32_t)lane < cnt;
    const uint32_t id = active ? (r & CABAC_REC_ID_MASK) : CABAC_REC_ID_MASK;
    const bool is_ctx = id < (uint32_t)kNumCtx;
    if (active && !is_ctx && id < CABAC_REC_ALIGN) bad = 1;
    uint32_t st_v = is_ctx ? ctx[id] : 0u;
    const uint32_t key_v = is_ctx ? id : 0xffffu;
    // per-lane constants of the update (the rates of a context never change)
    const uint32_t r0_v = (st_v & 3u) + 2u, r1_v = ((st_v >> 2) & 7u) + 5u;
    const uint32_t a_v = ((0x7fffu >> r0_v) & kMask0) | (((0x7fffu >> r1_v) & kMask1) << 16);
    uint32_t kq_v;  // k | mps << 5 of this lane's context as it stands
    {
      const uint32_t q8 = ctx2_q8(st_v);
      kq_v = is_ctx ? (ctx2_k(q8) | ((q8 >> 7) << 5)) : 0u;
    }
    uint32_t info_v = 0;
    if (is_ctx) info_v = 4u | kDecCtx | (id << 16);
    else if (active && id == CABAC_REC_TRM) info_v = 2u;
    else if (active && id == CABAC_REC_EP) info_v = kDecEp;
    else if (active && id == CABAC_REC_ALIGN) info_v = kDecAlign;
    {  // prefetch the next 64 records; placed after the uses of `r` so that hipcc's vmcnt(0) wait
       // in front of them does not also wait for this load
      const uint32_t nxt = base + 64u + (uint32_t)lane;
      next_rec = nxt < n ? rec[nxt] : 0;
    }
    uint32_t my_bin = 0;

    for (uint32_t i = 0; i < cnt; i++) {
      if (look <= 15) {  // refill 32 bits (every ~36 bins)
        const uint32_t w = __builtin_amdgcn_readlane(win_cur, (rp >> 2) & 63u);
        const uint64_t add = (uint64_t)w << (15 - look);
        hi |= (uint32_t)(add >> 32);
        lo |= (uint32_t)add;
        look += 32;
        rp += 4;
        if ((rp & 255u) == 0u) {
          win_cur = win_nxt;
          win_nxt = dec3_window_load(src, cap, rp + 256u + 4u * (uint32_t)lane);
        }
      }
      const uint32_t info = __builtin_amdgcn_readlane(info_v, i);
      uint32_t bin;
      if (info & kDecEp) {
        // decodeBinEP, arith_codec.cpp:100-114
        {
          const uint64_t v = (((uint64_t)hi << 32) | lo) << 1;
          hi = (uint32_t)(v >> 32);
          lo = (uint32_t)v;
        }
        const uint32_t sr = range << 22;
        bin = hi >= sr ? 1u : 0u;
        hi -= hi >= sr ? sr : 0u;
        shifts += 1;
        look -= 1;
      } else {
        const uint32_t kq = __builtin_amdgcn_readlane(kq_v, i);
        const uint32_t k = kq & 31u, mps = kq >> 5, c = info & 7u;
        const uint32_t t = (((range >> 5) * k) >> 1) + c;
        const uint32_t rm = range - t;
        const uint32_t sr = rm << 22;
        int nb;
        if (hi >= sr) {
          if (info & kDecCtx) {  // LPS path, arith_codec.cpp:262-273
            nb = __builtin_clz(t) - 23;
            hi -= sr;
            range = t << nb;
            bin = 1u - mps;
          } else {  // terminate bin 1, :184-185
            nb = 0;
            range = rm;
            bin = 1;
          }
        } else {  // MPS path :250-261 / terminate bin 0 :186-195
          nb = (int)((rm >> 8) ^ 1u);
          range = rm << nb;
          bin = mps;  // mps == 0 for non-context records
        }
        {
          const uint64_t v = (((uint64_t)hi << 32) | lo) << nb;
          hi = (uint32_t)(v >> 32);
          lo = (uint32_t)v;
        }
        shifts += (uint32_t)nb;
        look -= nb;
        if (info & kDecCtx) {
          // vector side: every lane applies the bin to its own copy; lanes of this context keep it
          const uint32_t s0 = st_v & kMask0, s1 = st_v >> 16;
          const uint32_t dlt = ((s0 >> r0_v) & kMask0) | (((s1 >> r1_v) & kMask1) << 16);
          const uint32_t upd = st_v - dlt + (bin ? a_v : 0u);
          st_v = (key_v == (info >> 16)) ? upd : st_v;
          const uint32_t q8 = ctx2_q8(st_v);
          kq_v = is_ctx ? (ctx2_k(q8) | ((q8 >> 7) << 5)) : 0u;
        }
        if (info & kDecAlign) range = 256;
      }
      my_bin = ((uint32_t)lane == i) ? bin : my_bin;  // vector side keeps lane i's bin (no scalar mask upkeep)
    }
    if (is_ctx) ctx[id] = st_v;
    if (active) out[base + lane] = (uint8_t)my_bin;
  }

  const uint32_t bytes_read = 2u + (shifts >> 3);
  const int32_t bits_needed = (int32_t)(shifts & 7u) - 8;
  uint32_t flags = 0;
  if (d.init_id & CABAC_SUB_FINISH) {
    uint32_t ok = 0;
    if (bytes_read <= cap) {
      const uint32_t last = src[bytes_read - 1];
      ok = ((last << (8 + bits_needed)) & 0xffu) == 0x80u;
    }
    if (!ok && bytes_read <= cap) flags |= CABAC_RES_BAD_STOP;  // an underrun throws before finish() is reached
  }
  if (bytes_read > cap) flags |= CABAC_RES_UNDERRUN;
  const uint64_t any_bad = __ballot(bad != 0);
  if (any_bad) flags |= CABAC_RES_BAD_RECORD;
  if (lane == 0) {
    cabac_substream_result res;
    res.n_bits = 8u * bytes_read + (uint32_t)bits_needed;
    res.flags = flags;
    results[sub] = res;
  }
}

static size_t v2_lds_bytes(uint32_t lanes) { return (size_t)kV2Waves * lanes * kLaneStride * sizeof(uint32_t); }

// lanes per wave for v2: smallest power of two that brings the grid down to <= ~1 wave per SIMD
static uint32_t v2_lanes(uint32_t n_sub, int variant) {
  uint32_t forced = (uint32_t)variant >> 8;
  if (forced >= 1 && forced <= 8) return forced;
  uint32_t l = 1;  // <= 8 lanes: 4 waves x 8 x 1 524 B of LDS stays under the 64 KB dynamic-LDS default
  while (l < 8 && (n_sub + l - 1) / l > 1024u) l <<= 1;
  return l;
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
                         const uint16_t *records, uint8_t *bytes, cabac_substream_result *results, uint32_t in_flight) {
  if (n_sub == 0) return hipSuccess;
  const int kind = variant & 0xff;
  // auto (measured, DESIGN.md §3, tools/enc_scaling.py): from 3 072 substreams the lane-serial encoder (v7: C4 0.64, C5 10.3 ms,
  // 16 384 substreams 1.90 ms against v6's 0.82 / 12.8 / 2.81); below, one unit per workgroup, the four-wave quad encoder (v6:
  // C2 0.61, C3 12.8 ms, 1 024 substreams 0.43, 2 048: 0.57 ms against v7's 0.78 / 16.5 / 0.65 / 0.64)
  if (kind == 7 || (kind == 0 && max(n_sub, in_flight) >= 3072u)) return launch_encode_v7(st, n_sub, desc, records, bytes, results, in_flight);
  if (kind == 6 || kind == 0) return launch_encode_v6(st, n_sub, desc, records, bytes, results, in_flight);
  if (kind == 5) return launch_encode_v5(st, n_sub, desc, records, bytes, results, in_flight);
  if (kind == 4) return launch_encode_v4(st, n_sub, desc, records, bytes, results);
  if (kind == 1) {
    hipLaunchKernelGGL(encode_kernel_v1, dim3(n_sub), dim3(64), 0, st, n_sub, desc, records, bytes, results);
  } else if (kind != 2) {
    hipLaunchKernelGGL(encode_kernel_v3, dim3(n_sub), dim3(64), 0, st, n_sub, desc, records, bytes, results);
  } else {
    const uint32_t l = v2_lanes(n_sub, variant);
    const uint32_t per_block = l * kV2Waves;
    hipLaunchKernelGGL(encode_kernel_v2, dim3((n_sub + per_block - 1) / per_block), dim3(64 * kV2Waves), v2_lds_bytes(l),
                       st, n_sub, l, desc, records, bytes, results);
  }
  return hipGetLastError();
}

hipError_t launch_decode(hipStream_t st, int variant, uint32_t n_sub, const cabac_substream_desc *desc,
                         const uint16_t *records, const uint8_t *bytes, uint8_t *bins,
                         cabac_substream_result *results, uint32_t in_flight) {
  if (n_sub == 0) return hipSuccess;
  const int kind = variant & 0xff;
  // auto: the quad decoder has the shortest per-substream chain at every batch size measured (C2: 10,
  // C3: 256, C4: 4 096 substreams), because it never crosses between the scalar and vector pipes
  if (kind == 4 || kind == 5 || kind == 6 || kind == 7 || kind == 0) return launch_decode_v4(st, n_sub, desc, records, bytes, bins, results, in_flight);
  if (kind == 1) {
    hipLaunchKernelGGL(decode_kernel_v1, dim3(n_sub), dim3(64), 0, st, n_sub, desc, records, bytes, bins, results);
  } else if (kind != 2) {
    hipLaunchKernelGGL(decode_kernel_v3, dim3(n_sub), dim3(64), 0, st, n_sub, desc, records, bytes, bins, results);
  } else {
    const uint32_t l = v2_lanes(n_sub, variant);
    const uint32_t per_block = l * kV2Waves;
    hipLaunchKernelGGL(decode_kernel_v2, dim3((n_sub + per_block - 1) / per_block), dim3(64 * kV2Waves), v2_lds_bytes(l),
                       st, n_sub, l, desc, records, bytes, bins, results);
  }
  return hipGetLastError();
}

}  // namespace cabac
