// Device-side building blocks shared by the kernel files: init tables, the probability model on the
// packed context word, small load helpers.  Reference lines are cited at each function.
#ifndef CABAC_DEVICE_H
#define CABAC_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cabac_ctx_tables.h"
#include "cabac_hip.h"

namespace cabac {

static __constant__ uint8_t c_init_tables[CABAC_CTX_TABLE_ROWS * CABAC_CTX_TABLE_COLS] = {CABAC_CTX_INIT_TABLE_VALUES};

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

constexpr int kLaneStride = 381;
constexpr uint32_t kDummySlot = 379;

__device__ __forceinline__ uint32_t ctx2_init(int qp, uint32_t init_value, uint32_t w) {
  uint32_t st = ctx_init_state(qp, init_value);
  uint32_t r = ctx_init_rates(w);
  uint32_t r0 = r & 0xffu, r1 = r >> 8;
  return st | (r0 - 2u) | ((r1 - 5u) << 2);
}

// q8 = state() of the packed word (contexts.cpp:939-941)
__device__ __forceinline__ uint32_t ctx2_q8(uint32_t st) { return (((st & kMask0) + (st >> 16)) >> 8) & 0xffu; }

// (q folded to 0..127) >> 2, contexts.cpp:945-949
__device__ __forceinline__ uint32_t ctx2_k(uint32_t q8) {
  const uint32_t x = (uint32_t)((int32_t)(q8 << 24) >> 31);  // 0 or ~0 from bit 7
  return ((q8 ^ x) >> 2) & 31u;
}

__device__ __forceinline__ uint32_t ctx2_update(uint32_t st, uint32_t bin) {
  const uint32_t r0 = (st & 3u) + 2u, r1 = ((st >> 2) & 7u) + 5u;
  const uint32_t s0 = st & kMask0, s1 = st >> 16;
  const uint32_t d = ((s0 >> r0) & kMask0) | (((s1 >> r1) & kMask1) << 16);
  const uint32_t a = ((0x7fffu >> r0) & kMask0) | (((0x7fffu >> r1) & kMask1) << 16);
  return st - d + (bin ? a : 0u);  // halves never borrow/carry into each other (15-bit estimators)
}

__device__ __forceinline__ uint32_t lane_load_le32(const uint8_t *src, uint32_t cap, uint32_t off) {
  uint32_t w = 0;
  if (off + 4u <= cap) {
    w = *reinterpret_cast<const uint32_t *>(src + off);
  } else {
    for (uint32_t b = 0; b < 4; b++)
      if (off + b < cap) w |= (uint32_t)src[off + b] << (8 * b);
  }
  return w;
}


}  // namespace cabac
#endif
