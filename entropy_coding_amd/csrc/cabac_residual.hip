// Residual binariser (encoder side): transform-block coefficients -> bin records.
// Restates for the device
//   CABACWriter::residual_coding            entropy_codec/cabac_writer.cpp:2424-2525
//   CABACWriter::ts_flag                    cabac_writer.cpp:2527-2534   (flag value 0 only)
//   CABACWriter::last_sig_coeff             cabac_writer.cpp:2639-2720
//   CABACWriter::residual_coding_subblock   cabac_writer.cpp:2722-2872
//   CoeffCodingContext                      common/context_modelling.hpp:71-244, context_modelling.cpp:7-106
//   grouped diagonal scan, g_log2SbbSize    common/rom.cpp:41-50, :72-92, :148-260
//   BinEncIf::encodeRemAbsEP                entropy_codec/arith_codec.cpp:426-458
//
// The reference walks a block serially; almost nothing in that walk is serial by nature.  Context choice reads
// a five-coefficient template of the INPUT (not of coded state), so every position knows its contexts on its
// own.  What couples positions is (a) the budget of context-coded bins, which only ever decreases — position p
// is context coded iff budget - (bins of the positions before it) >= 4, a prefix sum; (b) the dependent-
// quantisation state, a 4-state machine driven by coefficient parity; (c) where each position's bins land in
// the output, another prefix sum.  Layout: one block per 16-lane DPP row (four blocks per wave), lane = scan
// position inside the coefficient group, one loop iteration per coefficient group in coding order; row-wide
// sums come from ballots + popcounts, the output offsets from a 4-step DPP suffix scan.  The budget, the state
// and the write offset are carried from group to group as row-uniform values.
// Rows of one wave should run the same number of groups, so a pre-pass (class_hist / class_scatter) orders the
// blocks by their group count into a permutation held in library scratch; the main kernel takes its blocks
// from that list.  The dependent-quantisation state machine is linear over GF(2) — (s1, s0) -> (parity ^ s0, s1)
// — so the state on entry to every position is two masked popcounts of the group's parity bits, not a walk.
// Reads 4 B per coefficient (plus template re-reads that hit L1/L2), writes 2 B per bin: HBM-bound by
// construction; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cabac_device.h"
#include "cabac_hip.h"
#include "cabac_kernels.h"
#include "cabac_scan.h"

namespace cabac {

namespace {

template <int N>
__device__ __forceinline__ uint32_t row_shl(uint32_t v) {  // lane l <- lane l + N of the same row, 0 outside
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 | N, 0xf, 0xf, true);
}

template <int N>
__device__ __forceinline__ uint32_t row_shr(uint32_t v) {  // lane l <- lane l - N of the same row, 0 outside
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 | N, 0xf, 0xf, true);
}

// sum of v over the lanes below this one in its row (forward scan order: transform-skip blocks)
__device__ __forceinline__ uint32_t row_sum_below(uint32_t v) {
  uint32_t s = v;
  s += row_shr<1>(s);
  s += row_shr<2>(s);
  s += row_shr<4>(s);
  s += row_shr<8>(s);
  return s - v;
}

// sum of v over the lanes above this one in its row (the positions coded before it)
__device__ __forceinline__ uint32_t row_sum_above(uint32_t v) {
  uint32_t s = v;
  s += row_shl<1>(s);
  s += row_shl<2>(s);
  s += row_shl<4>(s);
  s += row_shl<8>(s);
  return s - v;
}

__device__ __forceinline__ uint32_t row_bits(bool pred, uint32_t row_shift) {  // this row's 16 ballot bits
  return (uint32_t)(__ballot(pred) >> row_shift) & 0xffffu;
}

struct EpCode {  // two bypass code words, MSB first
  uint32_t code1, len1, code2, len2;
};

// encodeRemAbsEP with cutoff 5 (COEF_REMAIN_BIN_REDUCTION), arith_codec.cpp:426-458
__device__ __forceinline__ EpCode rem_abs_code(uint32_t value, uint32_t rice, uint32_t max_log2) {
  EpCode s;
  const uint32_t cutoff = 5u;
  if (value < (cutoff << rice)) {
    s.len1 = (value >> rice) + 1u;
    s.code1 = (1u << s.len1) - 2u;
    s.code2 = value & ((1u << rice) - 1u);
    s.len2 = rice;
  } else {
    const uint32_t max_prefix = 32u - cutoff - max_log2;
    const uint32_t code = (value >> rice) - cutoff;
    uint32_t prefix_len, suffix_len;
    if (code >= ((1u << max_prefix) - 1u)) {
      prefix_len = max_prefix;
      suffix_len = max_log2;
    } else {
      prefix_len = 31u - (uint32_t)__builtin_clz(code + 1u);  // smallest n with code <= 2^(n+1) - 2
      suffix_len = prefix_len + rice + 1u;
    }
    s.len1 = prefix_len + cutoff;
    s.code1 = (1u << s.len1) - 1u;
    s.code2 = ((code - ((1u << prefix_len) - 1u)) << rice) | (value & ((1u << rice) - 1u));
    s.len2 = suffix_len;
  }
  return s;
}

constexpr uint32_t kRowsPerBlock = 16;  // 256 threads

}  // namespace

// g_log2SbbSize (rom.cpp:41-50): log2 of a coefficient group's width and height for a block of 2^lw x 2^lh
__device__ __forceinline__ void group_shape(uint32_t lw, uint32_t lh, uint32_t &cgw_l2, uint32_t &cgh_l2) {
  if (lw == 0u) { cgw_l2 = 0u; cgh_l2 = lh < 4u ? lh : 4u; }
  else if (lh == 0u) { cgw_l2 = lw < 4u ? lw : 4u; cgh_l2 = 0u; }
  else if (lw == 1u) { cgw_l2 = 1u; cgh_l2 = lh <= 2u ? 1u : 3u; }
  else if (lh == 1u) { cgh_l2 = 1u; cgw_l2 = lw <= 2u ? 1u : 3u; }
  else { cgw_l2 = 2u; cgh_l2 = 2u; }
}

// log2 of the number of coefficient groups of a block's coded region (0..6), 7 for a descriptor the kernel rejects.
// Class 0 is exactly "one group": those blocks take the lean walk (residual_rows<.., kSingle>).
__device__ __forceinline__ uint32_t size_class(const cabac_tu_desc &d) {
  const uint32_t lw = d.log2_width, lh = d.log2_height;
  if (lw > 6u || lh > 6u) return 7u;
  uint32_t cgw_l2, cgh_l2;
  group_shape(lw, lh, cgw_l2, cgh_l2);
  return ((lw < 5u ? lw : 5u) - cgw_l2) + ((lh < 5u ? lh : 5u) - cgh_l2);
}

constexpr uint32_t kClasses = 8;
// scratch layout (uint32): [0..7] blocks per class, [8..15] scatter cursors, [16] non-zero if any block is transform-skip
// coded, [24..] permutation (padded per class)
constexpr uint32_t kScratchHeader = 24;
constexpr uint32_t kScratchAnyTs = 16;

// Pre-pass: each thread looks at kSortItems descriptors and keeps its per-class counts as 4-bit fields of one word
// (a count is at most 8); counts meet through wave shuffles of 16-bit pairs, then one LDS add per wave and class,
// one global add per workgroup and class.
constexpr uint32_t kSortItems = 8;
constexpr uint32_t kSortSpan = 256u * kSortItems;  // descriptors per workgroup

__device__ __forceinline__ uint32_t class_of(const cabac_tu_desc *tus, uint32_t i) {
  // one 8-byte load of the descriptor's second half: log2_width, log2_height in its low bytes
  const uint64_t hi = reinterpret_cast<const uint64_t *>(tus)[2u * (uint64_t)i + 1u];
  cabac_tu_desc d{};
  d.log2_width = (uint8_t)hi;
  d.log2_height = (uint8_t)(hi >> 8);
  return size_class(d);
}

// per-thread nibble counts -> four words of two 16-bit counts (classes 2k and 2k+1)
__device__ __forceinline__ void widen_counts(uint32_t nib, uint32_t out[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) out[k] = ((nib >> (8 * k)) & 15u) | (((nib >> (8 * k + 4)) & 15u) << 16);
}

__global__ __launch_bounds__(256) void class_hist(uint32_t n_tu, const cabac_tu_desc *__restrict__ tus,
                                                   uint32_t *__restrict__ scratch) {
  __shared__ uint32_t h[kClasses];
  if (threadIdx.x < kClasses) h[threadIdx.x] = 0;
  __syncthreads();
  uint32_t nib = 0;
  bool ts = false;
#pragma unroll
  for (uint32_t k = 0; k < kSortItems; k++) {
    const uint32_t i = blockIdx.x * kSortSpan + k * 256u + threadIdx.x;
    if (i < n_tu) {
      nib += 1u << (4u * class_of(tus, i));
      ts = ts || (reinterpret_cast<const uint8_t *>(tus + i)[11] & CABAC_TU_TRANSFORM_SKIP);
    }
  }
  if (__ballot(ts) != 0ull && (threadIdx.x & 63u) == 0u) atomicOr(&scratch[kScratchAnyTs], 1u);
  uint32_t c[4];
  widen_counts(nib, c);
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) c[k] += (uint32_t)__shfl_xor((int)c[k], d);
  if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (c[k] & 0xffffu) atomicAdd(&h[2 * k], c[k] & 0xffffu);
      if (c[k] >> 16) atomicAdd(&h[2 * k + 1], c[k] >> 16);
    }
  }
  __syncthreads();
  if (threadIdx.x < kClasses && h[threadIdx.x]) atomicAdd(&scratch[threadIdx.x], h[threadIdx.x]);
}

__device__ __forceinline__ uint32_t class_base(const uint32_t *scratch, uint32_t cls) {  // start of a class, 16-aligned
  uint32_t base = 0;
  for (uint32_t k = kClasses; k-- > cls + 1u;) base += (scratch[k] + kRowsPerBlock - 1u) & ~(kRowsPerBlock - 1u);
  return base;  // largest blocks first: the long-running workgroups start first
}

__global__ __launch_bounds__(256) void class_scatter(uint32_t n_tu, const cabac_tu_desc *__restrict__ tus,
                                                      uint32_t *__restrict__ scratch) {
  __shared__ uint32_t wave_cnt[4][kClasses], start[kClasses];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t nib = 0, cls_all = 0;  // classes of this thread's descriptors, 3 bits each (7 = none / rejected)
#pragma unroll
  for (uint32_t k = 0; k < kSortItems; k++) {
    const uint32_t i = blockIdx.x * kSortSpan + k * 256u + threadIdx.x;
    uint32_t cls = 7u;
    if (i < n_tu) {
      cls = class_of(tus, i);
      nib += 1u << (4u * cls);
    }
    cls_all |= cls << (3u * k);
  }
  // exclusive prefix of the counts over the lanes of the wave, and the wave totals
  uint32_t c[4], incl[4];
  widen_counts(nib, c);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    incl[k] = c[k];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl[k], d);
      if ((int)lane >= d) incl[k] += up;
    }
  }
  if (lane == 63u) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      wave_cnt[wave][2 * k] = incl[k] & 0xffffu;
      wave_cnt[wave][2 * k + 1] = incl[k] >> 16;
    }
  }
  __syncthreads();
  if (threadIdx.x < kClasses) {
    const uint32_t total = wave_cnt[0][threadIdx.x] + wave_cnt[1][threadIdx.x] + wave_cnt[2][threadIdx.x] + wave_cnt[3][threadIdx.x];
    start[threadIdx.x] = class_base(scratch, threadIdx.x) + (total ? atomicAdd(&scratch[kClasses + threadIdx.x], total) : 0u);
  }
  __syncthreads();
  uint32_t next[kClasses];  // where this thread's next block of each class goes
#pragma unroll
  for (uint32_t q = 0; q < kClasses; q++) {
    uint32_t before = 0;
    for (uint32_t v = 0; v < wave; v++) before += wave_cnt[v][q];
    const uint32_t mine = (q & 1u) ? (incl[q >> 1] >> 16) - (c[q >> 1] >> 16) : (incl[q >> 1] & 0xffffu) - (c[q >> 1] & 0xffffu);
    next[q] = start[q] + before + mine;
  }
#pragma unroll
  for (uint32_t k = 0; k < kSortItems; k++) {
    const uint32_t i = blockIdx.x * kSortSpan + k * 256u + threadIdx.x;
    const uint32_t cls = (cls_all >> (3u * k)) & 7u;
    if (i < n_tu) {
      uint32_t slot = 0;
#pragma unroll
      for (uint32_t q = 0; q < kClasses; q++)
        if (q == cls) slot = next[q]++;
      scratch[kScratchHeader + slot] = i;
    }
  }
}

// kStageDw != 0: the blocks of this launch (up to kStageDw coded coefficients) are first copied into LDS row by row — every
// 128-byte line of the block is fetched once and used whole — and all later reads (scan-order walks, templates) go
// to LDS.  Read straight from memory in scan order such a block costs a line per 16 bytes used, and with every wave
// of an XCD holding four of them the lines do not survive in L2 between uses (measured: 10 x the block's bytes).
// Smaller blocks fit a line or two and are read directly.
// kTs: the transform-skip walk (residual_codingTS) for the blocks flagged CABAC_TU_TRANSFORM_SKIP, which the other
// variants leave alone: a launch of its own, so that its registers and code do not weigh on the regular walk.
// kSingle: every block of the workgroup is ONE coefficient group (class 0: 4 x 4 and smaller, 2 x 8, 1 x 16): no group
// scan, no group flags, the coefficient is loaded once, and the template of a position never leaves the block — it is read
// from the row's own lanes (five ds_bpermute of a packed contribution word) instead of five loads.
// C: how the coefficients lie in memory — int32_t (the reference's TCoeff) or int16_t (what a block whose dynamic range is 15 bits
// needs: half the bytes over PCIe for the host-pointer path, cabac_hip_encode_batch_residual16)
template <bool kWrite, uint32_t kStageDw, bool kTs, bool kSingle = false, class C = int32_t>
__device__ __forceinline__ void residual_rows(uint32_t wg_index, int32_t *stage, uint32_t n_tu, const cabac_tu_desc *__restrict__ tus,
                                              const C *__restrict__ coeff_all, const uint64_t *__restrict__ rec_offset,
                                              uint32_t *__restrict__ n_records, uint32_t *__restrict__ info_out,
                                              uint16_t *__restrict__ records, const uint32_t *__restrict__ perm) {
  const uint32_t lane = threadIdx.x & 63u, l = lane & 15u, row_shift = lane & 48u;
  const uint32_t tu_idx = perm[wg_index * kRowsPerBlock + (threadIdx.x >> 4)];  // 0xFFFFFFFF: padding
  bool live = tu_idx < n_tu;

  // ---- geometry (row-uniform) -------------------------------------------------------------------
  uint32_t lw = 0, lh = 0, chroma = 0, flags = 0, max_log2 = 15;
  const C *coeff = coeff_all;
  if (live) {
    const cabac_tu_desc d = tus[tu_idx];
    lw = d.log2_width;
    lh = d.log2_height;
    chroma = d.channel;
    flags = d.flags;
    max_log2 = d.max_log2_tr_range ? d.max_log2_tr_range : 15u;
    coeff = coeff_all + d.coeff_offset;
  }
  const bool mine = live && (((flags & CABAC_TU_TRANSFORM_SKIP) != 0u) == kTs);  // the other launch's rows are not touched
  live = mine;
  const bool bad = live && (lw > 6u || lh > 6u || chroma > 1u || max_log2 > 20u ||
                            ((flags & CABAC_TU_TRANSFORM_SKIP) && (lw > 5u || lh > 5u)));  // TS blocks are at most 32 x 32
  if (bad) lw = lh = chroma = 0;
  live = live && !bad;
  const uint32_t w = 1u << lw, h = 1u << lh;
  uint32_t cgw_l2, cgh_l2;
  group_shape(lw, lh, cgw_l2, cgh_l2);
  const uint32_t cg_l2 = cgw_l2 + cgh_l2, cg_size = 1u << cg_l2;
  const uint32_t we = w < 32u ? w : 32u, he = h < 32u ? h : 32u;
  const uint32_t lwg = (31u - (uint32_t)__builtin_clz(we)) - cgw_l2, lhg = (31u - (uint32_t)__builtin_clz(he)) - cgh_l2;
  const uint32_t wg = kSingle ? 1u : 1u << lwg, hg = kSingle ? 1u : 1u << lhg, n_cg = live ? wg * hg : 0u;
  const uint32_t in_cg = c_diag.in_cg[cgw_l2][cgh_l2][l & (cg_size - 1u)];
  const uint32_t ix = in_cg & 15u, iy = in_cg >> 4;
  const uint8_t *grid = c_diag.grid[lwg][lhg];
  const bool lane_in_cg = l < cg_size;

  const uint32_t lwe = 31u - (uint32_t)__builtin_clz(we);
  constexpr bool kStage = kStageDw != 0u;  // LDS dwords per row: 1024 (blocks of 512-1024 coefficients), 256, or none
  const uint32_t stage_base = (threadIdx.x >> 4) * kStageDw;
  if (kStage) {
    const uint32_t total = live ? min(we * he, kStageDw) : 0u;
    uint32_t tmax = total;
    tmax = max(tmax, (uint32_t)__shfl_xor((int)tmax, 16));
    tmax = max(tmax, (uint32_t)__shfl_xor((int)tmax, 32));
    tmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)tmax);
    for (uint32_t i = l; i < tmax; i += 64u) {
      int32_t v[4];
#pragma unroll
      for (uint32_t u = 0; u < 4u; u++) {
        const uint32_t idx = i + 16u * u;
        v[u] = idx < total ? coeff[((idx >> lwe) << lw) + (idx & (we - 1u))] : 0;
      }
#pragma unroll
      for (uint32_t u = 0; u < 4u; u++) {
        const uint32_t idx = i + 16u * u;
        if (idx < total) stage[stage_base + idx] = v[u];
      }
    }
  }
  // coefficient (x, y) of this row's block; within the coded region only
  auto coef_at = [&](uint32_t x, uint32_t y) -> int32_t {
    return kStage ? stage[stage_base + (y << lwe) + x] : coeff[(y << lw) + x];
  };

  // SBT / MTS zero-out (CABAC_TU_SBT_ZERO_OUT; cabac_writer.cpp:2660-2667, :2507-2516, unit.cpp:465-479): a 32-wide (32-tall)
  // luma block is coded as if only its left (upper) 16 columns (rows) existed
  const bool zo = !kTs && !kSingle && live && (flags & CABAC_TU_SBT_ZERO_OUT) && chroma == 0u && w <= 32u && h <= 32u;
  const uint32_t zo_w = (zo && w == 32u) ? 16u : we, zo_h = (zo && h == 32u) ? 16u : he;
  auto zeroed_out = [&](uint32_t gpos) { return (((gpos & 15u) << cgw_l2) >= zo_w) || (((gpos >> 4) << cgh_l2) >= zo_h); };
  uint64_t zo_groups = 0;  // by scan index: groups the walk passes over without a flag
  // ---- sweep 1: which groups hold a coefficient, and the last significant position ----------------------
  int last = -1;
  uint64_t coded = 0;    // by scan index of the group
  uint64_t sig_map = 0;  // by raster position in the group grid: bit gy * wg + gx
  int32_t c_single = 0;  // kSingle: the lane's coefficient, loaded here once
  if constexpr (kSingle) {
    if (live && lane_in_cg) c_single = coeff[(iy << lw) + ix];
    const uint32_t nz = row_bits(c_single != 0, row_shift);
    if (nz) {
      last = (int)(31u - (uint32_t)__builtin_clz(nz));
      coded = 1ull;
      sig_map = 1ull;
    }
  } else {
    int top = (int)n_cg - 1;  // the rows of a wave normally share n_cg (class order)
    top = max(top, __shfl_xor(top, 16));
    top = max(top, __shfl_xor(top, 32));
    top = __builtin_amdgcn_readfirstlane(top);
    for (int k = top; k >= 0; k--) {
      const bool on = k < (int)n_cg;
      int32_t c = 0;
      uint32_t gpos = 0;
      if (on) gpos = grid[k];
      const bool out_of_play = on && zeroed_out(gpos);
      if (out_of_play) zo_groups |= 1ull << k;
      if (on && !out_of_play && lane_in_cg) c = coef_at(((gpos & 15u) << cgw_l2) + ix, ((gpos >> 4) << cgh_l2) + iy);
      const uint32_t nz = row_bits(c != 0, row_shift);
      if (nz) {
        if (last < 0) last = (int)(((uint32_t)k << cg_l2) + (31u - (uint32_t)__builtin_clz(nz)));
        coded |= 1ull << k;
        sig_map |= 1ull << ((gpos >> 4) * wg + (gpos & 15u));
      }
    }
  }
  const bool empty = live && last < 0;
  live = live && !empty;

  uint16_t *out = (kWrite && live) ? records + rec_offset[tu_idx] : nullptr;
  uint32_t off = 0;  // records produced so far (row-uniform)
  uint32_t info = live ? (uint32_t)last : (bad ? CABAC_TU_INFO_BAD_DESC : empty ? CABAC_TU_INFO_EMPTY : 0u);

  // ---- ts_flag and the last position ------------------------------------------------------------------
  const bool is_ts = kTs && live;
  if (live && (flags & CABAC_TU_TS_FLAG)) {
    if (kWrite && l == 0u) out[0] = (uint16_t)((is_ts ? CABAC_REC_BIN : 0u) | CABAC_CTX_TRANSFORM_SKIP_FLAG(chroma));
    off = 1;
  }
  if (!kTs && live) {
    const uint32_t lcg = (uint32_t)last >> cg_l2;
    const uint32_t lgp = kSingle ? 0u : grid[lcg];
    const uint32_t lin = c_diag.in_cg[cgw_l2][cgh_l2][(uint32_t)last & (cg_size - 1u)];
    const uint32_t px = ((lgp & 15u) << cgw_l2) + (lin & 15u), py = ((lgp >> 4) << cgh_l2) + (lin >> 4);
    const uint32_t luma_off_x = lw < 3u ? 0u : lw == 3u ? 3u : lw == 4u ? 6u : lw == 5u ? 10u : 15u;
    const uint32_t luma_off_y = lh < 3u ? 0u : lh == 3u ? 3u : lh == 4u ? 6u : lh == 5u ? 10u : 15u;
    const uint32_t off_x = chroma ? 0u : luma_off_x, off_y = chroma ? 0u : luma_off_y;
    const uint32_t sh_x = chroma ? min(w >> 3, 2u) : (lw + 1u) >> 2, sh_y = chroma ? min(h >> 3, 2u) : (lh + 1u) >> 2;
    const uint32_t gix = group_idx(px), giy = group_idx(py);
    const uint32_t nx = gix + (gix < group_idx(zo_w - 1u) ? 1u : 0u), ny = giy + (giy < group_idx(zo_h - 1u) ? 1u : 0u);
    const uint32_t sx = gix > 3u ? (gix - 2u) >> 1 : 0u, sy = giy > 3u ? (giy - 2u) >> 1 : 0u;
    if (kWrite) {
      if (l < nx) out[off + l] = (uint16_t)((l < gix ? CABAC_REC_BIN : 0u) | (CABAC_CTX_LAST_X(chroma) + off_x + (l >> sh_x)));
      if (l < ny) out[off + nx + l] = (uint16_t)((l < giy ? CABAC_REC_BIN : 0u) | (CABAC_CTX_LAST_Y(chroma) + off_y + (l >> sh_y)));
      if (l < sx) out[off + nx + ny + l] = (uint16_t)(((((px - min_in_group(gix)) >> (sx - 1u - l)) & 1u) ? CABAC_REC_BIN : 0u) | CABAC_REC_EP);
      if (l < sy) out[off + nx + ny + sx + l] = (uint16_t)(((((py - min_in_group(giy)) >> (sy - 1u - l)) & 1u) ? CABAC_REC_BIN : 0u) | CABAC_REC_EP);
    }
    off += nx + ny + sx + sy;
  }

  // ---- sweep 2: the coefficient groups in coding order ----------------------------------------------
  const bool dq = !kTs && live && (flags & CABAC_TU_DEP_QUANT);  // state transitions 32040 (cabac_writer.cpp:2482), else state 0
  int budget = (int)((zo_w * zo_h * 28u) >> 4);          // cabac_writer.cpp:2485-2489 (the area after the zero-out)
  uint32_t state = 0;
  const int last_cg = live ? (last >> cg_l2) : -1;
  // Only groups that hold a coefficient (and group 0) are walked; the empty ones in between cost one group flag
  // each, written sixteen at a time.  `todo` = the groups this row still has to walk, by scan index.
  uint64_t todo = (!kTs && live) ? ((coded | 1ull) & ((2ull << last_cg) - 1ull)) : 0ull;
  int prev_cg = last_cg + 1;

  while (__ballot(todo != 0ull) != 0ull) {
    const bool row_on = todo != 0ull;
    const int cg = (kSingle || !row_on) ? 0 : 63 - __builtin_clzll(todo);
    if constexpr (kSingle) todo = 0ull;  // the one group of the block
    else todo &= ~(1ull << cg);
    // coded_sub_block_flag (cabac_writer.cpp:2733-2743) of the empty groups passed over, then of this group
    const uint32_t gap = (kSingle || !row_on) ? 0u : (uint32_t)(prev_cg - 1 - cg);
    uint32_t gap_flags = gap;  // flags actually coded: the zeroed-out groups among those passed over have none
    if (!kSingle && __ballot(zo && gap != 0u) != 0ull) {
      uint32_t kept = 0;
      for (uint32_t base = 0; __ballot(base < gap) != 0ull; base += 16u) {
        const uint32_t j = base + l;
        const bool in_gap = j < gap;
        const uint32_t sp = in_gap ? grid[prev_cg - 1 - (int)j] : 0u;
        const bool keep = in_gap && !zeroed_out(sp);
        const uint32_t m_keep = row_bits(keep, row_shift);
        if (kWrite && keep) {
          const uint32_t sx_ = sp & 15u, sy_ = sp >> 4, sb = sy_ * wg + sx_;
          const uint32_t right = sx_ + 1u < wg ? (uint32_t)(sig_map >> (sb + 1u)) & 1u : 0u;
          const uint32_t below = sy_ + 1u < hg ? (uint32_t)(sig_map >> (sb + wg)) & 1u : 0u;
          out[off + kept + (uint32_t)__builtin_popcount(m_keep & ((1u << l) - 1u))] = (uint16_t)(CABAC_CTX_SIG_COEFF_GROUP(chroma) + (right | below));
        }
        kept += (uint32_t)__builtin_popcount(m_keep);
      }
      gap_flags = zo ? kept : gap;
    }
    if (kWrite && !kSingle && __ballot(!zo && gap != 0u) != 0ull) {
      for (uint32_t base = 0; __ballot(!zo && base < gap) != 0ull; base += 16u) {
        const uint32_t j = base + l;
        if (!zo && j < gap) {
          const uint32_t sp = grid[prev_cg - 1 - (int)j];
          const uint32_t sx_ = sp & 15u, sy_ = sp >> 4, sb = sy_ * wg + sx_;
          const uint32_t right = sx_ + 1u < wg ? (uint32_t)(sig_map >> (sb + 1u)) & 1u : 0u;
          const uint32_t below = sy_ + 1u < hg ? (uint32_t)(sig_map >> (sb + wg)) & 1u : 0u;
          out[off + j] = (uint16_t)(CABAC_CTX_SIG_COEFF_GROUP(chroma) + (right | below));
        }
      }
    }
    off += gap_flags;
    prev_cg = row_on ? cg : prev_cg;
    const uint32_t gpos = (kSingle || !row_on) ? 0u : grid[cg];
    const uint32_t gx = gpos & 15u, gy = gpos >> 4;
    const uint32_t gbit = gy * wg + gx;
    const bool coded_group = (coded >> cg) & 1ull;
    if (row_on && cg != last_cg && cg != 0) {
      if (kWrite && l == 0u) {
        const uint32_t right = gx + 1u < wg ? (uint32_t)(sig_map >> (gbit + 1u)) & 1u : 0u;
        const uint32_t below = gy + 1u < hg ? (uint32_t)(sig_map >> (gbit + wg)) & 1u : 0u;
        out[off] = (uint16_t)(CABAC_REC_BIN | (CABAC_CTX_SIG_COEFF_GROUP(chroma) + (right | below)));
      }
      off += 1u;
    }
    const bool walk = row_on;  // residual_coding_subblock goes past its early return for these groups only
    if (walk && chroma == 0u && coded_group && (gx > 3u || gy > 3u)) info |= CABAC_TU_INFO_MTS_VIOLATION;
    const int lo = cg << cg_l2;
    const int first = cg == last_cg ? last : lo + (int)cg_size - 1;
    const int infer = cg == last_cg ? last : (cg != 0 ? lo : -1);
    const int pos = lo + (int)l;
    const bool in_range = row_on && lane_in_cg && pos <= first;
    const uint32_t x = (gx << cgw_l2) + ix, y = (gy << cgh_l2) + iy, diag = x + y;

    // the coefficient; its template is fetched below, once it is known that somebody codes this group
    int32_t c = 0;
    if constexpr (kSingle) c = in_range ? c_single : 0;
    else if (in_range) c = coef_at(x, y);
    const uint32_t a = (uint32_t)(c < 0 ? -c : c);
    const bool nzero = c != 0;
    const uint32_t m_nz = row_bits(nzero, row_shift);

    const bool act = walk && in_range;

    // template of the position (sigCtxIdAbs / templateAbsSum, context_modelling.hpp:71-117, :152-176): five
    // neighbours to the right and below, absent ones count as zero.  Loads are unconditional from clamped
    // addresses (no exec juggling); an absent neighbour is zeroed afterwards.
    // The count-only pass needs the template only for escape codes: a level of 4 or more, or a position the
    // budget (at most 64 context bins go per group) may no longer reach.
    int sum_abs = 0, sum_clip = 0, n_tmpl = 0;
    const bool want_tmpl = kWrite || __ballot(act && (a >= 4u || budget < 68)) != 0ull;
    if constexpr (kSingle) {
      if (want_tmpl) {  // wave-uniform: every lane takes part in the exchange
        // this position's contribution to a template that holds it: min(|c|, 63) (what rice_of can tell apart), the clipped
        // level of sigCtxIdAbs, non-zero — 9 + 5 + 3 bits, wide enough for a sum of five
        const uint32_t a6 = a < 63u ? a : 63u;
        const uint32_t mine_word = act ? (a6 | (min(a, 4u + (a & 1u)) << 9) | ((a != 0u ? 1u : 0u) << 14)) : 0u;
        const uint32_t nb = c_diag.nbr[cgw_l2][cgh_l2][l & (cg_size - 1u)];
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t q = 0; q < 5u; q++) {
          const uint32_t at = (nb >> (5u * q)) & 31u;
          const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane & 48u) | (at & 15u)) << 2), (int)mine_word);
          sum += at < 16u ? v : 0u;
        }
        sum_abs = (int)(sum & 511u);
        sum_clip = (int)((sum >> 9) & 31u);
        n_tmpl = (int)(sum >> 14);
      }
    } else if (want_tmpl && act) {
      // the coded region is the block, or its top-left 32 x 32: what lies outside is zero by construction of the
      // stream (rom.cpp:218-226) and is not read
      const bool x1 = x + 1u < we, x2 = x + 2u < we, y1 = y + 1u < he, y2 = y + 2u < he;
      const uint32_t xa = x + (x1 ? 1u : 0u), xb = x + (x2 ? 2u : 0u), ya = y + (y1 ? 1u : 0u), yb = y + (y2 ? 2u : 0u);
      const int32_t v0 = coef_at(xa, y), v1 = coef_at(xb, y), v2 = coef_at(xa, ya), v3 = coef_at(x, ya), v4 = coef_at(x, yb);
      auto add = [&](int32_t v, bool present) {
        int a = v < 0 ? -v : v;
        a = present ? a : 0;
        sum_abs += a;
        sum_clip += min(a, 4 + (a & 1));
        n_tmpl += a != 0;
      };
      add(v0, x1);
      add(v1, x2);
      add(v2, x1 && y1);
      add(v3, y1);
      add(v4, y2);
    }

    // which bins exist, and how far the context-bin budget reaches
    const uint32_t above_mask = ~0u << (l + 1u);  // positions coded before this one
    const bool sig_coded = act && !(pos == infer && (m_nz & above_mask) == 0u);
    const uint32_t m_sig = row_bits(sig_coded, row_shift);
    const uint32_t m_gt1 = row_bits(a > 1u, row_shift);
    const uint32_t spent_before = (uint32_t)__builtin_popcount(m_sig & above_mask) + (uint32_t)__builtin_popcount(m_nz & above_mask) +
                                  2u * (uint32_t)__builtin_popcount(m_gt1 & above_mask);
    const bool ctx_mode = act && (budget - (int)spent_before >= 4);
    const uint32_t m_ctx = row_bits(ctx_mode, row_shift);
    const uint32_t n_ctx_bins = (uint32_t)__builtin_popcount(m_sig & m_ctx) + (uint32_t)__builtin_popcount(m_nz & m_ctx) +
                                2u * (uint32_t)__builtin_popcount(m_gt1 & m_ctx);

    // Dependent-quantisation state on entry to each position (cabac_writer.cpp:2787, :2837).  The transition
    // (s1, s0) -> (parity ^ s0, s1) makes s1 after t steps the XOR of every second parity before it, so with the
    // positions top .. 0 of the group coded in that order, position l (step t = top - l) enters with
    //   s1 = xor of parity[l+1], parity[l+3], ... ^ (t odd ? s0 : s1 at group entry),   s0 = the same one step earlier.
    uint32_t my_state = 0;
    {
      const uint32_t m_par = row_bits(dq && act && (a & 1u), row_shift);
      const uint32_t top = (uint32_t)(first - lo), t = top - l;
      const uint32_t s1 = state >> 1, s0 = state & 1u;
      const uint32_t e1 = (uint32_t)__builtin_popcount((m_par >> (l + 1u)) & 0x5555u) & 1u;
      const uint32_t e0 = (uint32_t)__builtin_popcount((m_par >> (l + 2u)) & 0x5555u) & 1u;
      my_state = ((e1 ^ ((t & 1u) ? s0 : s1)) << 1) | (e0 ^ ((t & 1u) ? s1 : s0));
      if (walk) {  // state after the top + 1 steps of this group
        const uint32_t x1 = (uint32_t)__builtin_popcount(m_par & 0x5555u) & 1u, x0 = (uint32_t)__builtin_popcount(m_par & 0xAAAAu) & 1u;
        const uint32_t n = top + 1u;
        state = ((x1 ^ ((n & 1u) ? s0 : s1)) << 1) | (x0 ^ ((n & 1u) ? s1 : s0));
      }
      if (!dq) my_state = 0;
    }

    // pass 2 (remainder of a context-coded level) or pass 3 (whole level in bypass mode): one escape code per position
    uint32_t ep_value = 0, ep_rice = 0;
    bool has_ep = false;
    if (ctx_mode) {
      has_ep = a >= 4u;
      ep_value = (a - 4u) >> 1;
      ep_rice = rice_of(sum_abs, 4);
    } else if (act) {
      has_ep = true;
      ep_rice = rice_of(sum_abs, 0);
      const uint32_t pos0 = (my_state < 2u ? 1u : 2u) << ep_rice;
      ep_value = a == 0u ? pos0 : (a <= pos0 ? a - 1u : a);
    }
    EpCode ep = {0, 0, 0, 0};
    if (has_ep) ep = rem_abs_code(ep_value, ep_rice, max_log2);
    const uint32_t n23 = ep.len1 + ep.len2;
    const uint32_t before23 = row_sum_above(n23);
    const uint32_t total23 = (uint32_t)__shfl((int)(before23 + n23), (int)(lane & 48u));  // lane 0 of the row sees all

    // signs (cabac_writer.cpp:2860-2871)
    const uint32_t n_nz = (uint32_t)__builtin_popcount(m_nz);
    uint32_t n_signs = n_nz;
    bool hidden = false;
    if (n_nz && (flags & CABAC_TU_SIGN_HIDING)) {
      const uint32_t hi_nz = 31u - (uint32_t)__builtin_clz(m_nz), lo_nz = (uint32_t)__builtin_ctz(m_nz);
      if (hi_nz - lo_nz >= 4u) {
        n_signs--;
        hidden = l == lo_nz;
      }
    }

    if (kWrite && act) {
      if (ctx_mode) {
        // (written as arithmetic on compare results: as nested ?: / if chains hipcc builds an exec-mask region per level)
        const uint32_t luma = chroma ^ 1u;
        const uint32_t ofs = min((uint32_t)(sum_clip + 1) >> 1, 3u) + 4u * ((uint32_t)(diag < 2u) + (luma & (uint32_t)(diag < 5u)));
        const uint32_t set = chroma + 2u * (my_state - (my_state > 1u ? 1u : my_state));  // SigFlag[chType + 2 * max(0, state - 1)]
        const uint32_t sig_base = (uint32_t)(0x8E827A6E665Aull >> (8u * set)) & 0xffu;    // 90, 102, 110, 122, 130, 142
        // ctxOffsetAbs, context_modelling.hpp:131-143: 1 + min(sumClip - numPos, 4) + 5 * (luma: (d < 10) + (d < 3) + (d < 1); chroma: (d < 1))
        const uint32_t steps = (uint32_t)(diag == 0u) + luma * ((uint32_t)(diag < 3u) + (uint32_t)(diag < 10u));
        const uint32_t aofs = (pos != last ? 1u : 0u) * ((uint32_t)min(sum_clip - n_tmpl, 4) + 1u + 5u * steps);
        const uint32_t gt1_base = 214u + 21u * chroma, par_base = 150u + 21u * chroma, gt2_base = 182u + 21u * chroma;  // GtxFlag(2 + ch), ParFlag(ch), GtxFlag(ch)
        uint16_t *o1 = out + (off + spent_before);  // every position above a context-coded one is context coded
#ifndef CABAC_EXP_NO_CTX
        if (sig_coded) *o1++ = (uint16_t)((nzero ? CABAC_REC_BIN : 0u) | (sig_base + ofs));
        if (nzero) {
          *o1++ = (uint16_t)((a > 1u ? CABAC_REC_BIN : 0u) | (gt1_base + aofs));
          if (a > 1u) {
            const uint32_t rem = a - 2u;
            *o1++ = (uint16_t)(((rem & 1u) ? CABAC_REC_BIN : 0u) | (par_base + aofs));
            *o1 = (uint16_t)(((rem >> 1) ? CABAC_REC_BIN : 0u) | (gt2_base + aofs));
          }
        }
#endif
      }
      uint16_t *o2 = out + (off + n_ctx_bins + before23);  // (one 64-bit add: the sum in 32 bits first)
#ifndef CABAC_EXP_NO_EP
      for (uint32_t j = 0; j < ep.len1; j++) o2[j] = (uint16_t)((((ep.code1 >> (ep.len1 - 1u - j)) & 1u) ? CABAC_REC_BIN : 0u) | CABAC_REC_EP);
      o2 += ep.len1;
      for (uint32_t j = 0; j < ep.len2; j++) o2[j] = (uint16_t)((((ep.code2 >> (ep.len2 - 1u - j)) & 1u) ? CABAC_REC_BIN : 0u) | CABAC_REC_EP);
#endif
      if (nzero && !hidden)
        out[off + n_ctx_bins + total23 + (uint32_t)__builtin_popcount(m_nz & above_mask)] = (uint16_t)((c < 0 ? CABAC_REC_BIN : 0u) | CABAC_REC_EP);
    }
    if (walk) {
      off += n_ctx_bins + total23 + n_signs;
      budget -= (int)n_ctx_bins;
    }
  }

  // ---- transform-skip blocks: residual_codingTS / residual_coding_subblockTS (cabac_writer.cpp:2874-3046) ----------
  // Forward scan order, neighbours are the left and the upper sample, signs are context coded in the first pass, the
  // level is mapped through its neighbours (deriveModCoeff, context_modelling.hpp:344-364).  Budget: 7/4 context bins
  // per sample for the block, checked before every position of pass 1 (sig, sign, >1, parity) and of pass 2 (up to
  // four greater-than flags): both are prefix sums over the lanes below.
  if (kTs && __ballot(is_ts) != 0ull) {
    const bool bdpcm = (flags & CABAC_TU_BDPCM) != 0u;
    int tbudget = (int)((w * h * 7u) >> 2);
    uint32_t top = is_ts ? n_cg : 0u;
    top = max(top, (uint32_t)__shfl_xor((int)top, 16));
    top = max(top, (uint32_t)__shfl_xor((int)top, 32));
    top = (uint32_t)__builtin_amdgcn_readfirstlane((int)top);
    const uint32_t below_mask = (1u << l) - 1u;
    for (uint32_t cg = 0; cg < top; cg++) {
      const bool row_on = is_ts && cg < n_cg;
      const uint32_t gpos = row_on ? grid[cg] : 0u;
      const uint32_t gx = gpos & 15u, gy = gpos >> 4, gbit = gy * wg + gx;
      const bool coded_group = (coded >> cg) & 1ull;
      const bool others = (coded & ((1ull << cg) - 1ull)) != 0ull;  // a significant group before this one
      const bool has_flag = row_on && (cg != n_cg - 1u || others);  // cabac_writer.cpp:2933
      if (has_flag) {
        if (kWrite && l == 0u) {
          const uint32_t left = gx > 0u ? (uint32_t)(sig_map >> (gbit - 1u)) & 1u : 0u;
          const uint32_t above = gy > 0u ? (uint32_t)(sig_map >> (gbit - wg)) & 1u : 0u;
          out[off] = (uint16_t)((coded_group ? CABAC_REC_BIN : 0u) | (CABAC_CTX_TS_SIG_COEFF_GROUP + left + above));
        }
        off += 1u;
      }
      const bool walk = row_on && (coded_group || !has_flag);
      if (__ballot(walk) == 0ull) continue;
      const bool act = walk && lane_in_cg;
      const uint32_t x = (gx << cgw_l2) + ix, y = (gy << cgh_l2) + iy;
      int32_t c = 0, vl = 0, va = 0;
      if (act) {
        c = coef_at(x, y);
        vl = coef_at(x > 0u ? x - 1u : x, y);
        va = coef_at(x, y > 0u ? y - 1u : y);
        vl = x > 0u ? vl : 0;
        va = y > 0u ? va : 0;
      }
      const uint32_t a = (uint32_t)(c < 0 ? -c : c);
      const bool nzero = c != 0;
      const uint32_t n_nb = (vl != 0 ? 1u : 0u) + (va != 0 ? 1u : 0u);
      uint32_t mod = a;  // deriveModCoeff
      if (!bdpcm && a != 0u) {
        const uint32_t pred = max((uint32_t)(vl < 0 ? -vl : vl), (uint32_t)(va < 0 ? -va : va));
        mod = a == pred ? 1u : (a < pred ? a + 1u : a);
      }
      // pass 1
      const uint32_t m_nz = row_bits(nzero, row_shift);
      const bool sig_coded = act && !(l == cg_size - 1u && (m_nz & below_mask) == 0u);
      const uint32_t m_sig = row_bits(sig_coded, row_shift);
      const uint32_t m_g1 = row_bits(nzero && mod > 1u, row_shift);
      const uint32_t spent1 = (uint32_t)__builtin_popcount(m_sig & below_mask) + 2u * (uint32_t)__builtin_popcount(m_nz & below_mask) +
                              (uint32_t)__builtin_popcount(m_g1 & below_mask);
      const bool pass1 = act && (tbudget - (int)spent1 >= 4);
      const uint32_t m_p1 = row_bits(pass1, row_shift);
      const uint32_t n1 = (uint32_t)__builtin_popcount(m_sig & m_p1) + 2u * (uint32_t)__builtin_popcount(m_nz & m_p1) +
                          (uint32_t)__builtin_popcount(m_g1 & m_p1);
      const int last1 = m_p1 ? 31 - __builtin_clz(m_p1) : -1;
      // pass 2
      const uint32_t cost2 = mod >= 2u ? min(4u, mod >> 1) : 0u;
      const uint32_t m_c0 = row_bits(act && (cost2 & 1u), row_shift), m_c1 = row_bits(act && (cost2 & 2u), row_shift),
                     m_c2 = row_bits(act && (cost2 & 4u), row_shift);
      const uint32_t spent2 = (uint32_t)__builtin_popcount(m_c0 & below_mask) + 2u * (uint32_t)__builtin_popcount(m_c1 & below_mask) +
                              4u * (uint32_t)__builtin_popcount(m_c2 & below_mask);
      const bool pass2 = act && (tbudget - (int)n1 - (int)spent2 >= 4);
      const uint32_t m_p2 = row_bits(pass2, row_shift);
      const uint32_t n2 = (uint32_t)__builtin_popcount(m_c0 & m_p2) + 2u * (uint32_t)__builtin_popcount(m_c1 & m_p2) +
                          4u * (uint32_t)__builtin_popcount(m_c2 & m_p2);
      const int last2 = m_p2 ? 31 - __builtin_clz(m_p2) : -1;
      // pass 3
      const uint32_t cut = (int)l <= last2 ? 10u : ((int)l <= last1 ? 2u : 0u);
      const uint32_t lvl = cut ? mod : a;
      const bool has_rem = act && lvl >= cut;
      const bool ep_sign = has_rem && lvl != 0u && (int)l > last1;
      EpCode ep = {0, 0, 0, 0};
      if (has_rem) ep = rem_abs_code((int)l <= last1 ? (lvl - cut) >> 1 : lvl, 1u, max_log2);
      const uint32_t n3 = ep.len1 + ep.len2 + (ep_sign ? 1u : 0u);
      const uint32_t before3 = row_sum_below(n3);
      const uint32_t total3 = (uint32_t)__shfl((int)(before3 + n3), (int)(lane | 15u));
      if (kWrite && act) {
        if (pass1) {
          uint16_t *o1 = out + (off + spent1);
          if (sig_coded) *o1++ = (uint16_t)((nzero ? CABAC_REC_BIN : 0u) | (CABAC_CTX_TS_SIG_FLAG + n_nb));
          if (nzero) {
            const int sl = (vl > 0) - (vl < 0), sa = (va > 0) - (va < 0);  // signCtxIdAbsTS, context_modelling.hpp:293-317
            uint32_t sctx = ((sl == 0 && sa == 0) || sl * sa < 0) ? 0u : (sl >= 0 && sa >= 0) ? 1u : 2u;
            sctx += bdpcm ? 3u : 0u;
            *o1++ = (uint16_t)((c < 0 ? CABAC_REC_BIN : 0u) | (CABAC_CTX_TS_RESIDUAL_SIGN + sctx));
            *o1++ = (uint16_t)((mod > 1u ? CABAC_REC_BIN : 0u) | (CABAC_CTX_TS_LRG1_FLAG + (bdpcm ? 3u : n_nb)));
            if (mod > 1u) *o1 = (uint16_t)((((mod - 2u) & 1u) ? CABAC_REC_BIN : 0u) | CABAC_CTX_TS_PAR_FLAG);
          }
        }
        if (pass2) {
          uint16_t *o2 = out + (off + n1 + spent2);
          for (uint32_t k = 1; k <= cost2; k++)  // greater-than-(2k+1) flags, contexts TsGtxFlag(1..4)
            o2[k - 1u] = (uint16_t)((mod >= 2u * k + 2u ? CABAC_REC_BIN : 0u) | (CABAC_CTX_TS_GTX_FLAG + k));
        }
        uint16_t *o3 = out + (off + n1 + n2 + before3);
        for (uint32_t j = 0; j < ep.len1; j++) o3[j] = (uint16_t)((((ep.code1 >> (ep.len1 - 1u - j)) & 1u) ? CABAC_REC_BIN : 0u) | CABAC_REC_EP);
        o3 += ep.len1;
        for (uint32_t j = 0; j < ep.len2; j++) o3[j] = (uint16_t)((((ep.code2 >> (ep.len2 - 1u - j)) & 1u) ? CABAC_REC_BIN : 0u) | CABAC_REC_EP);
        if (ep_sign) o3[ep.len2] = (uint16_t)((c < 0 ? CABAC_REC_BIN : 0u) | CABAC_REC_EP);
      }
      if (walk) {
        off += n1 + n2 + total3;
        tbudget -= (int)(n1 + n2);
      }
    }
  }

  if (mine && l == 0u) {
    n_records[tu_idx] = off;
    if (info_out) info_out[tu_idx] = info;
  }
}

// The block order starts with the classes 7 (rejected), 6, 5 — the rows of the launch staged with 4 KB per row — then
// class 4 (256 coded coefficients: 1 KB per row), then the rest, read directly.  The staged launches are small grids
// whose workgroups take their range in turn; the direct launch has one workgroup per 16 rows, and those that fall into
// a staged range leave at once.
template <bool kWrite, uint32_t kStageDw, class C>
__global__ __launch_bounds__(256) void residual_kernel(uint32_t n_tu, const cabac_tu_desc *__restrict__ tus,
                                                        const C *__restrict__ coeff_all,
                                                        const uint64_t *__restrict__ rec_offset,
                                                        uint32_t *__restrict__ n_records,
                                                        uint32_t *__restrict__ info_out,
                                                        uint16_t *__restrict__ records,
                                                        const uint32_t *__restrict__ perm,
                                                        const uint32_t *__restrict__ class_count) {
  __shared__ int32_t stage[kStageDw ? kRowsPerBlock * kStageDw : 1u];
  const uint32_t r = kRowsPerBlock - 1u;
  const uint32_t big_wgs = (((class_count[7] + r) & ~r) + ((class_count[6] + r) & ~r) + ((class_count[5] + r) & ~r)) / kRowsPerBlock;
  const uint32_t mid_wgs = ((class_count[4] + r) & ~r) / kRowsPerBlock;
  if (kStageDw == 1024u) {
    for (uint32_t wg = blockIdx.x; wg < big_wgs; wg += gridDim.x)
      residual_rows<kWrite, 1024u, false, false, C>(wg, stage, n_tu, tus, coeff_all, rec_offset, n_records, info_out, records, perm);
  } else if (kStageDw == 256u) {
    for (uint32_t wg = big_wgs + blockIdx.x; wg < big_wgs + mid_wgs; wg += gridDim.x)
      residual_rows<kWrite, 256u, false, false, C>(wg, stage, n_tu, tus, coeff_all, rec_offset, n_records, info_out, records, perm);
  } else if (blockIdx.x >= big_wgs + mid_wgs) {
    // class 0 — one group per block — comes last in the order and takes the lean walk
    const uint32_t single_from = big_wgs + mid_wgs + (((class_count[3] + r) & ~r) + ((class_count[2] + r) & ~r) + ((class_count[1] + r) & ~r)) / kRowsPerBlock;
    if (blockIdx.x >= single_from)
      residual_rows<kWrite, 0u, false, true, C>(blockIdx.x, stage, n_tu, tus, coeff_all, rec_offset, n_records, info_out, records, perm);
    else
      residual_rows<kWrite, 0u, false, false, C>(blockIdx.x, stage, n_tu, tus, coeff_all, rec_offset, n_records, info_out, records, perm);
  }
}

// Transform-skip blocks: a small grid that leaves at once when the batch holds none (the ordering pre-pass notes it),
// and otherwise goes over all row groups, taking the flagged blocks.
template <bool kWrite, class C>
__global__ __launch_bounds__(256) void residual_ts_kernel(uint32_t n_tu, uint32_t n_wg, const cabac_tu_desc *__restrict__ tus,
                                                           const C *__restrict__ coeff_all,
                                                           const uint64_t *__restrict__ rec_offset,
                                                           uint32_t *__restrict__ n_records, uint32_t *__restrict__ info_out,
                                                           uint16_t *__restrict__ records, const uint32_t *__restrict__ perm,
                                                           const uint32_t *__restrict__ header) {
  __shared__ int32_t stage[1];
  if (header[kScratchAnyTs] == 0u) return;
  for (uint32_t wg = blockIdx.x; wg < n_wg; wg += gridDim.x)
    residual_rows<kWrite, 0u, true, false, C>(wg, stage, n_tu, tus, coeff_all, rec_offset, n_records, info_out, records, perm);
}

// (the residual parser lives in cabac_residual_parse.hip)

size_t residual_scratch_bytes(uint32_t n_tu) {
  return sizeof(uint32_t) * (kScratchHeader + (size_t)n_tu + kClasses * kRowsPerBlock);
}

template <bool kWrite, class C>
static hipError_t launch_residual_passes(hipStream_t st, uint32_t n_tu, const cabac_tu_desc *tus, const C *coeff,
                                         const uint64_t *rec_offset, uint32_t *n_records, uint32_t *info, uint16_t *records,
                                         uint32_t *s32, uint32_t rows) {
  // (The three launches side by side on three streams, the staged ones holding few waves per CU, were measured: 1.53 ms
  // against 1.45 one after the other for the records pass — each of them is bound by instruction issue, not by occupancy.)
  const dim3 grid(rows / kRowsPerBlock);
  const dim3 grid_staged(grid.x < 1024u ? grid.x : 1024u), grid_mid(grid.x < 8192u ? grid.x : 8192u);
  const uint32_t *order = s32 + kScratchHeader;
  hipLaunchKernelGGL((residual_kernel<kWrite, 1024u, C>), grid_staged, dim3(256), 0, st, n_tu, tus, coeff, rec_offset, n_records, info,
                     records, order, s32);
  hipLaunchKernelGGL((residual_kernel<kWrite, 256u, C>), grid_mid, dim3(256), 0, st, n_tu, tus, coeff, rec_offset, n_records, info,
                     records, order, s32);
  hipLaunchKernelGGL((residual_kernel<kWrite, 0u, C>), grid, dim3(256), 0, st, n_tu, tus, coeff, rec_offset, n_records, info,
                     records, order, s32);
  hipLaunchKernelGGL((residual_ts_kernel<kWrite, C>), grid_staged, dim3(256), 0, st, n_tu, grid.x, tus, coeff, rec_offset, n_records,
                     info, records, order, s32);
  return hipGetLastError();
}

hipError_t launch_residual(hipStream_t st, uint32_t n_tu, const cabac_tu_desc *tus, const void *coeff, int coeff_bytes,
                           const uint64_t *rec_offset, uint32_t *n_records, uint32_t *info, uint16_t *records,
                           void *scratch, bool order_ready) {
  if (n_tu == 0) return hipSuccess;
  if (coeff_bytes != 4 && coeff_bytes != 2) return hipErrorInvalidValue;
  // blocks ordered by group count: [counts | cursors | permutation, 0xFFFFFFFF where a class is padded to 16 rows]
  uint32_t *s32 = static_cast<uint32_t *>(scratch);
  const uint32_t rows = n_tu + kClasses * kRowsPerBlock;  // upper bound of the padded list
  if (!order_ready) {  // any complete permutation of the n_tu blocks is correct; this one balances the waves
    hipError_t e = hipMemsetAsync(s32, 0, sizeof(uint32_t) * kScratchHeader, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(s32 + kScratchHeader, 0xff, sizeof(uint32_t) * rows, st);
    if (e != hipSuccess) return e;
    const uint32_t g = (n_tu + kSortSpan - 1u) / kSortSpan;
    hipLaunchKernelGGL(class_hist, dim3(g), dim3(256), 0, st, n_tu, tus, s32);
    hipLaunchKernelGGL(class_scatter, dim3(g), dim3(256), 0, st, n_tu, tus, s32);
  }
  if (coeff_bytes == 2) {
    const int16_t *c16 = static_cast<const int16_t *>(coeff);
    return records ? launch_residual_passes<true, int16_t>(st, n_tu, tus, c16, rec_offset, n_records, info, records, s32, rows)
                   : launch_residual_passes<false, int16_t>(st, n_tu, tus, c16, rec_offset, n_records, info, records, s32, rows);
  }
  const int32_t *c32 = static_cast<const int32_t *>(coeff);
  return records ? launch_residual_passes<true, int32_t>(st, n_tu, tus, c32, rec_offset, n_records, info, records, s32, rows)
                 : launch_residual_passes<false, int32_t>(st, n_tu, tus, c32, rec_offset, n_records, info, records, s32, rows);
}

}  // namespace cabac
