// Residual parser (decoder side): bytes -> transform-block coefficients, every context derived
// on the device from the coefficients decoded so far.  Restates
//   CABACReader::residual_coding             entropy_codec/cabac_reader.cpp:2647-2735
//   CABACReader::ts_flag                     cabac_reader.cpp:2737-2752
//   CABACReader::last_sig_coeff              cabac_reader.cpp:2865-2938
//   CABACReader::residual_coding_subblock    cabac_reader.cpp:2946-3128
//   CABACReader::residual_codingTS / residual_coding_subblockTS   cabac_reader.cpp:3130-3339
//   CoeffCodingContext                       common/context_modelling.hpp:71-244, :268-384, context_modelling.cpp:7-106
//   BinDecoderBase / TBinDecoder             entropy_codec/arith_codec.cpp:60-78, :100-197, :242-277
//
// The walk is serial by nature: the context of a bin is chosen from the values of the bins before it.  What can be had
// is a short instruction stream per bin and the other waves of the SIMD to fill its gaps, so: ONE substream per wave
// with uniform control flow (real scalar branches, no exec regions), four waves per workgroup dealt to the CU's four
// SIMDs, ~7 KB of LDS per wave (16 waves per CU).  Inside the wave the lanes hold what the serial walk keeps looking up:
//   * lane i of the first 16 lanes is scan position i of the current coefficient group: its level, its position in the
//     block, and the running five-sample template sums of its position (sum of |level|, the clipped sum, the count of
//     non-zero neighbours).  A decoded level is added to exactly the lanes whose template contains its position (one
//     masked add from a per-shape bit table), so the context of the next position is ONE v_readlane away instead of five
//     dependent LDS reads;
//   * the input is held 256 bytes at a time, one byte-swapped dword per lane, the next 256 bytes already loaded: the
//     arithmetic decoder's refill is a v_readlane, no memory access near the chain;
//   * signs, the sign-hiding parity rule and the write-out of a group are lane-parallel (ballots and popcounts).
// The arithmetic decoder keeps (value, look-ahead) in a 64-bit window as the quad decoder does (cabac_kernels_v4.hip):
// value in bits 62..47, refilled 16 bits at a time while fewer than 32 look-ahead bits are valid — a check per syntax
// element (a position's four flags shift at most 24 bits), not per bin.  Its arithmetic is branch-free vector code on
// wave-uniform values (the context state comes out of LDS, i.e. out of a VGPR); only the decoded bin crosses to the
// scalar side, once per bin.  Runs of bypass bins (Rice prefixes / suffixes, sign patterns, last-position suffixes) are
// decoded up to 15 at a time with one division: n bypass bins are the n-bit quotient of the value by the range
// (arith_codec.cpp:100-151 decodes them one by one and eight at a time, to the same effect).
//
// Algorithmic bytes: B_in read, 4 B per coded coefficient written, 16 B per block descriptor: HBM traffic is a few
// percent of the roof; the limit is the serial chain (instruction issue), as for the bin decoder.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "cabac_device.h"
#include "cabac_hip.h"
#include "cabac_kernels.h"
#include "cabac_scan.h"

namespace cabac {

#ifdef CABAC_PARSE_PROFILE  // build with CABAC_EXTRA_FLAGS=-DCABAC_PARSE_PROFILE: where wave 0 of workgroup 0 spends its cycles
__device__ unsigned long long g_parse_prof[16];
__device__ unsigned long long g_parse_wave[3 * 8192];  // per substream: start, end (memtime), HW_ID | XCC_ID << 32
#define PP_TICK(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PP_ADD(slot, t0, t1) \
  if (blockIdx.x == 0 && threadIdx.x == 0) g_parse_prof[slot] += (t1) - (t0)
#else
#define PP_TICK(var)
#define PP_ADD(slot, t0, t1)
#endif

namespace {

// The parser only ever touches the residual-coding contexts: SigCoeffGroup .. LastY (ids 86..291), TransformSkipFlag
// (310, 311) and the transform-skip sets (357..378): 230 slots of 16 bytes {state, shifts, adds, -} per wave.
constexpr uint32_t kCtxSlots = 232;
constexpr uint32_t kBlkWords = 1024;   // coded region of a block, pitch = its coded width
__host__ __device__ constexpr uint32_t slot_of(uint32_t id) { return id < 292u ? id - 86u : id < 312u ? id - 310u + 206u : id - 357u + 208u; }
__device__ __forceinline__ uint32_t id_of_slot(uint32_t s) { return s < 206u ? s + 86u : s < 208u ? s - 206u + 310u : s - 208u + 357u; }
#define SL_A(id) ((id) - 86u)            /* ids 86..291 */
#define SL_TS(id) ((id) - 357u + 208u)   /* ids 357..378 */
#define SL_TS_FLAG(ch) (206u + (ch))

// Per coefficient-group shape: which scan positions of the group lie in the five-sample template of each position
// (context_modelling.hpp:71-117: right, right+1, below-right, below, below+1), and for transform-skip blocks the
// in-group index of the left / upper neighbour (0xFF: outside the group).
struct ShapeLut {
  uint16_t in_template[5][5][16];  // bit q: position q's sample is in the template of this position
  uint8_t left[5][5][16], above[5][5][16];
};

constexpr ShapeLut make_shapes() {
  ShapeLut t{};
  for (int a = 0; a < 5; a++)
    for (int b = 0; a + b < 5; b++) {
      uint8_t pos[16] = {};
      fill_diag(pos, 1 << a, 1 << b);
      const int n = 1 << (a + b);
      for (int i = 0; i < n; i++) {
        const int x = pos[i] & 15, y = pos[i] >> 4;
        uint16_t m = 0;
        uint8_t l = 0xff, u = 0xff;
        for (int q = 0; q < n; q++) {
          const int dx = (pos[q] & 15) - x, dy = (pos[q] >> 4) - y;
          if ((dx == 1 && dy == 0) || (dx == 2 && dy == 0) || (dx == 1 && dy == 1) || (dx == 0 && dy == 1) || (dx == 0 && dy == 2))
            m = (uint16_t)(m | (1u << q));
          if (dx == -1 && dy == 0) l = (uint8_t)q;
          if (dx == 0 && dy == -1) u = (uint8_t)q;
        }
        t.in_template[a][b][i] = m;
        t.left[a][b][i] = l;
        t.above[a][b][i] = u;
      }
    }
  return t;
}

__constant__ ShapeLut c_shapes = make_shapes();

// the tables the walk indexes with lane-varying or data-dependent indices, copied to LDS once per workgroup
struct LdsTables {
  uint8_t grid[4][4][64];
  uint8_t in_cg[5][5][16];
  uint16_t in_template[5][5][16];
  uint8_t left[5][5][16], above[5][5][16];
};

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }

// ---------------------------------------------------------------------------------------------------------------
// arithmetic decoder

template <bool kS>
struct PDecT {
  // kS: this wave keeps the decoder on the SCALAR unit (see pd_bin)
  static constexpr bool kScalar = kS;
  uint32_t hi, lo;   // window (wave-uniform, kept in vector registers): value in bits 62..47
  uint32_t range;    //   "
  int32_t look;      //   "   valid look-ahead bits below bit 47
  uint32_t rp;       // scalar: byte offset of the next unread 16-bit unit
  uint32_t in_cur, in_nxt;  // per lane: dword `lane` of the 256-byte block that holds rp / of the block after it, byte-swapped
  const uint8_t *src_safe;
  uint32_t cap, last_dword, lane;
};

template <class D>
__device__ __forceinline__ uint32_t pd_load_block(const D &d, uint32_t blk) {
  const uint32_t off = blk * 256u + 4u * d.lane;
  const uint32_t w = *reinterpret_cast<const uint32_t *>(d.src_safe + min(off, d.last_dword));
  return off < d.cap ? __builtin_bswap32(w) : 0u;  // past the end of the substream the window is fed zeros
}

// Append 16-bit units while fewer than 32 look-ahead bits are valid (at most two: look >= 0 here).  Afterwards 32 bits
// can be consumed before the next check.
template <class D>
__device__ __forceinline__ void pd_check(D &d) {
  int32_t look = (int32_t)rfl((uint32_t)d.look);
  while (__builtin_expect(look < 32, 0)) {
    const uint32_t dw = rl(d.in_cur, (d.rp >> 2) & 63u);
    const uint32_t unit = (d.rp & 2u) ? (dw & 0xffffu) : (dw >> 16);
    const uint64_t add = (uint64_t)unit << (31 - look);
    d.hi |= (uint32_t)(add >> 32);
    d.lo |= (uint32_t)add;
    look += 16;
    d.rp += 2u;
    if ((d.rp & 255u) == 0u) {
      d.in_cur = d.in_nxt;
      d.in_nxt = pd_load_block(d, (d.rp >> 8) + 1u);
    }
  }
  d.look = look;
}

// decodeBin, arith_codec.cpp:242-277, with BinProbModel_Std::getLPS / update (contexts.cpp:903-913, :939-954).
// ctx[slot] = {s0 | s1 << 16 (rate bits cleared), shift0 | shift1 << 16, add0 | add1 << 16, -}.  Branch-free: the one
// data-dependent decision of the walk is taken on the returned bin.  The walk is bound by instruction issue, so this is
// written for count: 31 vector instructions, two LDS accesses and the one v_readfirstlane.
template <class D>
__device__ __forceinline__ uint32_t pd_bin(D &d, uint4 *ctx, uint32_t slot) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const uint4 e = ctx[slot];
  // A SIMD has a vector and a scalar pipe that run side by side — for DIFFERENT waves (tools/ubench_mix.hip: four waves of
  // vector code 7.9 ns per instruction and wave, four of scalar code 7.1, two and two 4.2).  The decision arithmetic of a bin
  // is the same ~20 instructions on wave-uniform values either way, so every other wave of a workgroup runs it on the
  // scalar pipe: the state word crosses over first (one v_readfirstlane) and hipcc, seeing uniform values, picks s_
  // instructions; the other waves keep it in vector registers and only the decoded bin crosses.
  const uint32_t st = D::kScalar ? rfl(e.x) : e.x;
  const uint32_t sum = (st & 0xffffu) + (st >> 16);
  const uint32_t sx = (uint32_t)((int32_t)(sum << 16) >> 31);  // 0 / ~0 from the MPS bit
  const uint32_t k = ((sum >> 10) ^ sx) & 31u;
  const uint32_t t = (__umul24(d.range >> 5, k) + 8u) >> 1;     // ((range >> 5) * k >> 1) + 4
  const uint32_t rm = d.range - t;
  uint32_t ev;                                                  // value - scaledRange
  if (D::kScalar) ev = d.hi - (rm << 22);
  else asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(ev) : "v"(rm), "v"(0xffc00000u), "v"(d.hi));  // rm * -2^22 + value: one instruction
  // renormalisation, one rule for both paths: the chosen sub-range shifted up to [256, 511] — LPS by getRenormBitsLPS =
  // clz(t) - 23; MPS by one bit iff rm < 256, which is clz(rm) - 23 because rm >= 128 (an LPS width is at most 15.5 / 32
  // of the range plus 4)
  uint32_t bin, x;
  if (D::kScalar) {
    const bool mps = (int32_t)ev < 0;
    bin = (mps ? sx : ~sx) & 1u;
    x = mps ? rm : t;
  } else {
    uint32_t ngem;  // ~0: value < scaledRange (MPS).  Through asm: as (int)ev >> 31 hipcc turns every use back into v_cmp + v_cndmask pairs
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(ngem) : "v"(ev));
    bin = ~(ngem ^ sx) & 1u;
    x = (rm & ngem) | (t & ~ngem);
  }
  const uint32_t nsh = (uint32_t)__builtin_clz(x) - 23u;
  d.hi = min(d.hi, ev);   // ev wraps above 2^31 exactly when value < scaledRange
  d.range = x << nsh;
  const uint64_t v = (((uint64_t)d.hi << 32) | d.lo) << nsh;
  d.hi = (uint32_t)(v >> 32);
  d.lo = (uint32_t)v;
  d.look -= (int32_t)nsh;
  // update(bin) on both 15-bit estimators at once (vector pipe in both kinds of wave: there is no packed scalar math)
  const u16x2 st2 = __builtin_bit_cast(u16x2, e.x);
  const u16x2 dlt2 = (st2 >> __builtin_bit_cast(u16x2, e.y)) & __builtin_bit_cast(u16x2, (kMask1 << 16) | kMask0);
  const uint32_t rest = __builtin_bit_cast(uint32_t, (u16x2)(st2 - dlt2));
  uint32_t upd;
  if (D::kScalar) asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(upd) : "v"(e.z), "s"(bin), "v"(rest));
  else asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(upd) : "v"(e.z), "v"(bin), "v"(rest));
  ctx[slot].x = upd;
  return D::kScalar ? bin : rfl(bin);
}

// decodeBinEP, arith_codec.cpp:100-114: the doubling of value is folded into the comparison (against scaledRange / 2)
template <class D>
__device__ __forceinline__ uint32_t pd_ep(D &d) {
  const uint32_t ev = d.hi - (d.range << 21);
  const uint32_t ngem = (uint32_t)((int32_t)ev >> 31);
  d.hi = (d.hi & ngem) | (ev & ~ngem);
  const uint64_t v = (((uint64_t)d.hi << 32) | d.lo) << 1;
  d.hi = (uint32_t)(v >> 32);
  d.lo = (uint32_t)v;
  d.look -= 1;
  return rfl(~ngem & 1u);
}

// The next n (1..15) bypass bins as one number, MSB first, WITHOUT consuming them: they are the quotient of the value
// extended by n stream bits by the scaled range (what n rounds of decodeBinEP compute bit by bit).  look >= n.
template <class D>
__device__ __forceinline__ uint32_t pd_ep_peek(const D &d, uint32_t n) {
  // value (16 bits) and n more bits, over 128 * range: floor(floor(V / 128) / range), V / 128 < 2^24 exactly a float
  const uint32_t a = d.hi >> (22u - n);
  const float fr = (float)d.range;
  uint32_t q = (uint32_t)((float)a * __builtin_amdgcn_rcpf(fr));
  int32_t r = (int32_t)(a - q * d.range);
  if (r < 0) q -= 1u;
  else if (r >= (int32_t)d.range) q += 1u;
  return rfl(q);
}

// consume k of the bins just peeked (bins = their values, MSB first)
template <class D>
__device__ __forceinline__ void pd_ep_take(D &d, uint32_t bins, uint32_t k) {
  const uint64_t v = (((uint64_t)d.hi << 32) | d.lo) << k;
  d.hi = (uint32_t)(v >> 32) - ((bins * d.range) << 22);
  d.lo = (uint32_t)v;
  d.look -= (int32_t)k;
}

// decodeBinsEP(n), n <= 32 (arith_codec.cpp:116-151)
template <class D>
__device__ __forceinline__ uint32_t pd_bins_ep(D &d, uint32_t n) {
  uint32_t out = 0;
  while (n != 0u) {
    const uint32_t k = n < 15u ? n : 15u;
    pd_check(d);
    const uint32_t b = pd_ep_peek(d, k);
    pd_ep_take(d, b, k);
    out = (out << k) | b;
    n -= k;
  }
  return out;
}

// decodeRemAbsEP with cutoff 5 (arith_codec.cpp:153-179): unary prefix of at most 32 - maxLog2 ones, then the suffix
template <class D>
__device__ __forceinline__ uint32_t pd_rem_abs(D &d, uint32_t rice, uint32_t max_log2) {
  const uint32_t cutoff = 5u, max_prefix = 32u - max_log2;
  uint32_t prefix = 0;
  for (;;) {
    pd_check(d);
    const uint32_t want = min(15u, max_prefix - prefix);
    if (want == 0u) break;
    const uint32_t b = pd_ep_peek(d, want);
    const uint32_t ones = (uint32_t)__builtin_clz(~(b << (32u - want)));  // leading ones of the `want` bins (want < 32)
    if (ones < want) {
      pd_ep_take(d, b >> (want - ones - 1u), ones + 1u);  // the ones and the terminating zero
      prefix += ones;
      break;
    }
    pd_ep_take(d, b, want);
    prefix += want;
  }
  uint32_t length = rice, offset;
  if (prefix < cutoff) {
    offset = prefix << rice;
  } else {
    offset = ((1u << (prefix - cutoff)) + cutoff - 1u) << rice;
    length += prefix == max_prefix ? max_log2 - rice : prefix - cutoff;
  }
  return offset + pd_bins_ep(d, length);
}

// ---------------------------------------------------------------------------------------------------------------
struct BlockGeom {  // scalars of the block being parsed
  uint32_t lw, lh, chroma, fl, max_log2;
  uint32_t we, he, lwe;                       // coded region (rom.cpp:218-226) and its pitch
  uint32_t cgw_l2, cgh_l2, cg_l2, cg_size;    // coefficient group (g_log2SbbSize, rom.cpp:41-50)
  uint32_t lwg, lhg, wg, hg;                  // group grid
};

__device__ __forceinline__ void geom_init(BlockGeom &g, uint32_t lw, uint32_t lh) {
  g.lw = lw;
  g.lh = lh;
  const uint32_t w = 1u << lw, h = 1u << lh;
  g.we = w < 32u ? w : 32u;
  g.he = h < 32u ? h : 32u;
  g.lwe = lw < 5u ? lw : 5u;
  if (lw == 0u) { g.cgw_l2 = 0u; g.cgh_l2 = lh < 4u ? lh : 4u; }
  else if (lh == 0u) { g.cgw_l2 = lw < 4u ? lw : 4u; g.cgh_l2 = 0u; }
  else if (lw == 1u) { g.cgw_l2 = 1u; g.cgh_l2 = lh <= 2u ? 1u : 3u; }
  else if (lh == 1u) { g.cgh_l2 = 1u; g.cgw_l2 = lw <= 2u ? 1u : 3u; }
  else { g.cgw_l2 = 2u; g.cgh_l2 = 2u; }
  g.cg_l2 = g.cgw_l2 + g.cgh_l2;
  g.cg_size = 1u << g.cg_l2;
  g.lwg = g.lwe - g.cgw_l2;
  g.lhg = (lh < 5u ? lh : 5u) - g.cgh_l2;
  g.wg = 1u << g.lwg;
  g.hg = 1u << g.lhg;
}

// |value| of the block sample at (xx, yy), 0 outside the coded region
__device__ __forceinline__ uint32_t blk_abs(const int32_t *blk, const BlockGeom &g, uint32_t xx, uint32_t yy) {
  const bool in = xx < g.we && yy < g.he;
  const int32_t v = blk[in ? (yy << g.lwe) + xx : 0u];
  return in ? (uint32_t)(v < 0 ? -v : v) : 0u;
}
__device__ __forceinline__ int32_t blk_val(const int32_t *blk, const BlockGeom &g, int32_t xx, int32_t yy) {
  const bool in = xx >= 0 && yy >= 0;
  const int32_t v = blk[in ? ((uint32_t)yy << g.lwe) + (uint32_t)xx : 0u];
  return in ? v : 0;
}

// SigFlag context sets by (state > 1 ? state - 1 : 0) * 2 + chroma: 90, 102, 110, 122, 130, 142
__device__ __forceinline__ uint32_t sig_set_base(uint32_t set) { return (uint32_t)((0x8E827A6E665Aull >> (8u * set)) & 0xffu); }

// ---- regular residual coding: one block (after ts_flag) -----------------------------------------------------------
// kZo: the block is coded with the SBT / MTS zero-out.  A variant of its own, because its three extra scalars (the reduced
// extents and the flag) are live across the whole walk: compiled into the common walk they push it over the scalar register
// file and hipcc moves uniform values of the chain to vector registers (static count 1 842 -> 2 340 vector instructions,
// 5.91 -> 6.13 ms on the bench's tiles, none of which uses the zero-out).
template <class D, bool kZo>
__device__ __forceinline__ uint32_t parse_regular(D &d, uint4 *ctx, int32_t *blk, const LdsTables &tab, const BlockGeom &g,
                                                  uint32_t lane) {
  const uint32_t chroma = g.chroma, j = lane & 15u;
  PP_TICK(p0);
  // ---- last significant position (cabac_reader.cpp:2865-2938)
  const uint32_t luma_off_x = g.lw < 3u ? 0u : g.lw == 3u ? 3u : g.lw == 4u ? 6u : g.lw == 5u ? 10u : 15u;
  const uint32_t luma_off_y = g.lh < 3u ? 0u : g.lh == 3u ? 3u : g.lh == 4u ? 6u : g.lh == 5u ? 10u : 15u;
  const uint32_t off_x = chroma ? 0u : luma_off_x, off_y = chroma ? 0u : luma_off_y;
  const uint32_t sh_x = chroma ? min((1u << g.lw) >> 3, 2u) : (g.lw + 1u) >> 2, sh_y = chroma ? min((1u << g.lh) >> 3, 2u) : (g.lh + 1u) >> 2;
  // SBT / MTS zero-out (CABAC_TU_SBT_ZERO_OUT; cabac_reader.cpp:2880-2891, :2718-2727, unit.cpp:465-479)
  const uint32_t zo_w = (kZo && g.lw == 5u) ? 16u : g.we, zo_h = (kZo && g.lh == 5u) ? 16u : g.he;
  const uint32_t max_x = group_idx(zo_w - 1u), max_y = group_idx(zo_h - 1u);
  uint32_t px = 0, py = 0;
  for (; px < max_x; px++) {
    pd_check(d);
    if (!pd_bin(d, ctx, SL_A(CABAC_CTX_LAST_X(chroma)) + off_x + (px >> sh_x))) break;
  }
  for (; py < max_y; py++) {
    pd_check(d);
    if (!pd_bin(d, ctx, SL_A(CABAC_CTX_LAST_Y(chroma)) + off_y + (py >> sh_y))) break;
  }
  if (px > 3u) px = min_in_group(px) + pd_bins_ep(d, (px - 2u) >> 1);
  if (py > 3u) py = min_in_group(py) + pd_bins_ep(d, (py - 2u) >> 1);
  px = min(px, g.we - 1u);  // a corrupt stream must not lead outside the block
  py = min(py, g.he - 1u);
  // its scan position: the group and the place in the group, found by the lanes
  const uint32_t ip = tab.in_cg[g.cgw_l2][g.cgh_l2][j];
  const uint32_t ix = ip & 15u, iy = ip >> 4;
  const uint32_t gl = tab.grid[g.lwg][g.lhg][lane];
  const uint32_t want_g = (px >> g.cgw_l2) | ((py >> g.cgh_l2) << 4);
  const uint32_t want_i = (px & ((1u << g.cgw_l2) - 1u)) | ((py & ((1u << g.cgh_l2) - 1u)) << 4);
  const uint64_t mg = __ballot(gl == want_g && lane < g.wg * g.hg), mi = __ballot(ip == want_i && lane < g.cg_size);
  const uint32_t last_cg = mg ? (uint32_t)__builtin_ctzll(mg) : 0u, last_i = mi ? (uint32_t)__builtin_ctzll(mi) : 0u;
  uint32_t info = (last_cg << g.cg_l2) + last_i;

  const uint32_t in_tmpl = tab.in_template[g.cgw_l2][g.cgh_l2][j];
  // dependent-quantisation state (cabac_reader.cpp:2699-2700: table 32040): kept as sq = 8 * state, the bit offset of the
  // state's SigFlag context slot in a lane's packed word; the transition table holds the next sq for (state, parity)
  const uint64_t trans8 = (g.fl & CABAC_TU_DEP_QUANT) ? 0x0818180800101000ull : 0ull;  // byte [2 * state + parity] = 8 * next state: 0,2 | 2,0 | 1,3 | 3,1
  // SigFlag slots by state (sets: state 0, 1 -> 0; 2 -> 1; 3 -> 2; luma sets 0, 2, 4 / chroma 1, 3, 5), packed one byte each
  const uint32_t sb0 = SL_A(CABAC_CTX_SIG_FLAG(0)), sb1 = SL_A(CABAC_CTX_SIG_FLAG(1)), sb2 = SL_A(CABAC_CTX_SIG_FLAG(2)),
                 sb3 = SL_A(CABAC_CTX_SIG_FLAG(3)), sb4 = SL_A(CABAC_CTX_SIG_FLAG(4)), sb5 = SL_A(CABAC_CTX_SIG_FLAG(5));
  const uint32_t sig_bases = chroma ? (sb1 | sb1 << 8 | sb3 << 16 | sb5 << 24) : (sb0 | sb0 << 8 | sb2 << 16 | sb4 << 24);
  const uint32_t gt1_base = SL_A(CABAC_CTX_GTX_FLAG(2u + chroma)), par_base = SL_A(CABAC_CTX_PAR_FLAG(chroma)),
                 gt2_base = SL_A(CABAC_CTX_GTX_FLAG(chroma));
  uint32_t sq = 0;
  int32_t budget = (int32_t)((zo_w * zo_h * 28u) >> 4);
  uint64_t sig_map = 0;
  PP_TICK(p1);
  PP_ADD(1, p0, p1);
  const uint32_t pitch = 1u << g.lwe;
  for (int32_t cg = (int32_t)last_cg; cg >= 0; cg--) {
    PP_TICK(c0);
    const uint32_t gp = rl(gl, (uint32_t)cg), gx = gp & 15u, gy = gp >> 4, gbit = gy * g.wg + gx;
    if (kZo && ((gx << g.cgw_l2) >= zo_w || (gy << g.cgh_l2) >= zo_h)) continue;  // zeroed out: nothing is coded for this group
    bool sig = cg == (int32_t)last_cg || cg == 0;
    if (!sig) {  // coded_sub_block_flag (cabac_reader.cpp:2965-2975)
      const uint32_t right = gx + 1u < g.wg ? (uint32_t)(sig_map >> (gbit + 1u)) & 1u : 0u;
      const uint32_t below = gy + 1u < g.hg ? (uint32_t)(sig_map >> (gbit + g.wg)) & 1u : 0u;
      pd_check(d);
      sig = pd_bin(d, ctx, SL_A(CABAC_CTX_SIG_COEFF_GROUP(chroma)) + (right | below)) != 0u;
    }
    PP_TICK(c1);
    PP_ADD(2, c0, c1);
    if (!sig) continue;
    sig_map |= 1ull << gbit;
    if (!chroma && (gx > 3u || gy > 3u)) info |= CABAC_TU_INFO_MTS_VIOLATION;  // cabac_reader.cpp:2729-2732
    // This lane's position and what its contexts need (context_modelling.hpp:71-143): the template over the groups
    // decoded before (samples of this group still read as zero) — sc the clipped sum, dd = sc minus the count of non-zero
    // neighbours, t_abs the plain sum — and the diagonal class offsets.
    const uint32_t x = (gx << g.cgw_l2) + ix, y = (gy << g.cgh_l2) + iy, diag = x + y;
    uint32_t t_abs, sc, dd, sig_cls, abs_cls1;
    {
      const uint32_t at = (y << g.lwe) + x;
      const bool x1 = x + 1u < g.we, x2 = x + 2u < g.we, y1 = y + 1u < g.he, y2 = y + 2u < g.he;
      auto mag = [&](bool in, uint32_t idx) {  // unconditional read from a clamped address: no exec region per sample
        const uint32_t m = in ? ~0u : 0u;
        const int32_t v = blk[idx & m];
        return (uint32_t)(v < 0 ? -v : v) & m;
      };
      const uint32_t a0 = mag(x1, at + 1u), a1 = mag(x2, at + 2u), a2 = mag(x1 && y1, at + pitch + 1u), a3 = mag(y1, at + pitch),
                     a4 = mag(y2, at + 2u * pitch);
      t_abs = a0 + a1 + a2 + a3 + a4;
      auto clip = [](uint32_t a) { return min(a, 4u + (a & 1u)); };
      sc = clip(a0) + clip(a1) + clip(a2) + clip(a3) + clip(a4);
      dd = sc - ((uint32_t)(a0 != 0u) + (uint32_t)(a1 != 0u) + (uint32_t)(a2 != 0u) + (uint32_t)(a3 != 0u) + (uint32_t)(a4 != 0u));
      sig_cls = diag < 2u ? 4u : 0u;
      if (!chroma) sig_cls += diag < 5u ? 4u : 0u;
      abs_cls1 = 1u;
      if (diag == 0u) abs_cls1 += chroma ? 5u : 15u;
      else if (!chroma) abs_cls1 += diag < 3u ? 10u : diag < 10u ? 5u : 0u;
    }
    // per lane, kept up to date by the lanes themselves: the SigFlag slot of this position for each state (one byte each)
    // and the offset of its gt1 / parity / gt2 contexts — the serial walk reads them with one v_readlane each
    auto sig_slots = [&]() {
      const uint32_t o = min((sc + 1u) >> 1, 3u) + sig_cls;
      return sig_bases + __builtin_amdgcn_perm(o, o, 0u);  // the offset in all four bytes
    };
    uint32_t sig4 = sig_slots(), aofs_v = min(dd, 4u) + abs_cls1;
    const bool is_last_cg = cg == (int32_t)last_cg;
    const int32_t first_i = is_last_cg ? (int32_t)last_i : (int32_t)g.cg_size - 1;
    uint32_t nz_mask = 0, g2_mask = 0, lev = 0;
    int32_t i = first_i;
    PP_TICK(c2);
    PP_ADD(3, c1, c2);
    // ---- pass 1: sig / gt1 / parity / gt2 while the budget of context-coded bins lasts (cabac_reader.cpp:3007-3063)
    // The first position of the group's walk is special (the last significant position of the block: significant by
    // definition, context offset 0) and so is its last (significance inferred when nothing before it was significant and
    // the group flag was coded): both are peeled off the loop's common path by `special`.
    // kInfer: the significance of this position is not coded if nothing before it in the group was significant
    // (:3011-3013); kLast: the block's last significant position — significant by definition, context offset 0.
    auto position = [&](auto infer, auto at_last) {
      pd_check(d);
      uint32_t sf = 1u;
      if (!(infer.value && nz_mask == 0u)) {
        sf = pd_bin(d, ctx, (rl(sig4, (uint32_t)i) >> sq) & 0xffu);
        budget--;
      }
      uint32_t par2 = 0;  // 2 * parity of the level
      if (sf) {
        const uint32_t aofs = at_last.value ? 0u : rl(aofs_v, (uint32_t)i);
        uint32_t level = 1u;
        budget--;
        if (pd_bin(d, ctx, gt1_base + aofs)) {
          const uint32_t par = pd_bin(d, ctx, par_base + aofs);
          const uint32_t g2 = pd_bin(d, ctx, gt2_base + aofs);
          budget -= 2;
          level = 2u + par + (g2 << 1);
          g2_mask |= g2 << i;
        }
        par2 = (level & 1u) << 1;
        nz_mask |= 1u << i;
        lev = j == (uint32_t)i ? level : lev;
        // into the templates that hold this position, and their context offsets again
        const uint32_t hit = 0u - ((in_tmpl >> i) & 1u);
        sc += hit & level;
        dd += hit & (level - 1u);
        t_abs += hit & level;
        sig4 = sig_slots();
        aofs_v = min(dd, 4u) + abs_cls1;
      }
      sq = (uint32_t)(trans8 >> (2u * sq + 4u * par2)) & 0xffu;
      i--;
    };
    typedef std::integral_constant<bool, true> yes_t;
    typedef std::integral_constant<bool, false> no_t;
    if (is_last_cg) {
      if (budget >= 4) position(yes_t(), yes_t());  // the last significant position itself
      while (i >= 0 && budget >= 4) position(no_t(), no_t());
    } else {
      const int32_t stop = cg != 0 ? 1 : 0;          // position 0 of a group other than the DC group may be inferred
      while (i >= stop && budget >= 4) position(no_t(), no_t());
      if (i == 0 && budget >= 4) position(yes_t(), no_t());
    }
    const int32_t bypass_i = i;  // positions bypass_i .. 0 are coded without contexts
    PP_TICK(c3);
    PP_ADD(4, c2, c3);
#ifdef CABAC_PARSE_PROFILE
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      g_parse_prof[10] += (unsigned)(first_i - i);                 // positions visited in pass 1
      g_parse_prof[11] += 1;                                       // coded groups
    }
#endif
    // ---- pass 2: remainders of the context-coded levels (:3065-3077)
    for (uint32_t m = g2_mask; m != 0u;) {
      const uint32_t q = 31u - (uint32_t)__builtin_clz(m);
      m &= ~(1u << q);
      const uint32_t rem2 = pd_rem_abs(d, rice_of((int)rl(t_abs, q), 4), g.max_log2) << 1;
      lev += j == q ? rem2 : 0u;
      t_abs += (0u - ((in_tmpl >> q) & 1u)) & rem2;
    }
    // ---- pass 3: whole levels in bypass mode (:3079-3098)
    for (int32_t q = bypass_i; q >= 0; q--) {
      const uint32_t rice = rice_of((int)rl(t_abs, (uint32_t)q), 0);
      const uint32_t pos0 = (sq < 16u ? 1u : 2u) << rice;
      const uint32_t rem = pd_rem_abs(d, rice, g.max_log2);
      const uint32_t v = rem == pos0 ? 0u : (rem < pos0 ? rem + 1u : rem);
      sq = (uint32_t)(trans8 >> (2u * sq + 8u * (v & 1u))) & 0xffu;
      if (v) {
        nz_mask |= 1u << q;
        lev = j == (uint32_t)q ? v : lev;
        t_abs += (0u - ((in_tmpl >> q) & 1u)) & v;
      }
    }
    PP_TICK(c4);
    PP_ADD(5, c3, c4);
    // ---- signs; with sign-data hiding the lowest position's sign is the parity of the level sum (:3100-3126)
    const uint32_t n_nz = (uint32_t)__builtin_popcount(nz_mask);
    const int32_t first_nz = nz_mask ? (int32_t)__builtin_ctz(nz_mask) : first_i;
    const int32_t last_nz = nz_mask ? 31 - (int32_t)__builtin_clz(nz_mask) : -1;
    const bool hide = (g.fl & CABAC_TU_SIGN_HIDING) && last_nz - first_nz >= 4;
    const uint32_t n_signs = hide ? n_nz - 1u : n_nz;
    const uint32_t pattern = pd_bins_ep(d, n_signs);
    {
      const bool mine = lane < g.cg_size && ((nz_mask >> j) & 1u);
      const uint32_t odd = (uint32_t)__builtin_popcountll(__ballot(mine && (lev & 1u))) & 1u;
      const uint32_t k = (uint32_t)__builtin_popcount(nz_mask >> (j + 1u));  // non-zero positions decoded before this one
      const uint32_t neg = k < n_signs ? (pattern >> (n_signs - 1u - k)) & 1u : odd;
      if (mine) blk[(y << g.lwe) + x] = neg ? -(int32_t)lev : (int32_t)lev;
    }
    PP_TICK(c5);
    PP_ADD(6, c4, c5);
  }
  return info;
}

// ---- transform-skip residual coding: one block (cabac_reader.cpp:3130-3339) ------------------------------------------
template <class D>
__device__ __forceinline__ void parse_ts(D &d, uint4 *ctx, int32_t *blk, const LdsTables &tab, const BlockGeom &g, uint32_t lane) {
  const uint32_t j = lane & 15u;
  const bool bdpcm = (g.fl & CABAC_TU_BDPCM) != 0u;
  const uint32_t ip = tab.in_cg[g.cgw_l2][g.cgh_l2][j];
  const uint32_t ix = ip & 15u, iy = ip >> 4;
  const uint32_t gl = tab.grid[g.lwg][g.lhg][lane];
  const uint32_t idx_l = tab.left[g.cgw_l2][g.cgh_l2][j], idx_a = tab.above[g.cgw_l2][g.cgh_l2][j];
  const uint32_t n_cg = g.wg * g.hg;
  int32_t budget = (int32_t)((g.we * g.he * 7u) >> 2);
  uint64_t sig_map = 0;
  for (uint32_t cg = 0; cg < n_cg; cg++) {
    const uint32_t gp = rl(gl, cg), gx = gp & 15u, gy = gp >> 4, gbit = gy * g.wg + gx;
    bool sig = cg == n_cg - 1u && sig_map == 0ull;
    if (!sig) {
      const uint32_t left = gx > 0u ? (uint32_t)(sig_map >> (gbit - 1u)) & 1u : 0u;
      const uint32_t above = gy > 0u ? (uint32_t)(sig_map >> (gbit - g.wg)) & 1u : 0u;
      pd_check(d);
      sig = pd_bin(d, ctx, SL_TS(CABAC_CTX_TS_SIG_COEFF_GROUP) + left + above) != 0u;
    }
    if (!sig) continue;
    sig_map |= 1ull << gbit;
    // lane = forward scan position in the group; its left / upper samples as they stand (decoded groups: final values)
    const int32_t x = (int32_t)((gx << g.cgw_l2) + ix), y = (int32_t)((gy << g.cgh_l2) + iy);
    int32_t nb_l = blk_val(blk, g, x - 1, y), nb_a = blk_val(blk, g, x, y - 1), val = 0;
    uint32_t nz_mask = 0, sign_pattern = 0, n_nz = 0;
    int32_t last1 = -1, last2 = -1;
    const int32_t hi_i = (int32_t)g.cg_size - 1;
    auto put = [&](uint32_t i, int32_t v) {  // position i of the group becomes v: its own lane and the lanes beside / below it
      val = j == i ? v : val;
      nb_l = idx_l == i ? v : nb_l;
      nb_a = idx_a == i ? v : nb_a;
    };
    for (int32_t i = 0; i <= hi_i && budget >= 4; i++) {  // pass 1: sig, sign, greater-1, parity (:3222-3271)
      pd_check(d);
      const int32_t left = (int32_t)rl((uint32_t)nb_l, (uint32_t)i), above = (int32_t)rl((uint32_t)nb_a, (uint32_t)i);
      const uint32_t n_nb = (uint32_t)(left != 0) + (uint32_t)(above != 0);
      uint32_t sf = (n_nz == 0u && i == hi_i) ? 1u : 0u;
      if (!sf) {
        sf = pd_bin(d, ctx, SL_TS(CABAC_CTX_TS_SIG_FLAG) + n_nb);
        budget--;
      }
      if (sf) {
        const int32_t sl = (left > 0) - (left < 0), sa = (above > 0) - (above < 0);
        uint32_t sctx = ((sl == 0 && sa == 0) || sl * sa < 0) ? 0u : (sl >= 0 && sa >= 0) ? 1u : 2u;
        if (bdpcm) sctx += 3u;
        const uint32_t sign = pd_bin(d, ctx, SL_TS(CABAC_CTX_TS_RESIDUAL_SIGN) + sctx);
        const uint32_t g1 = pd_bin(d, ctx, SL_TS(CABAC_CTX_TS_LRG1_FLAG) + (bdpcm ? 3u : n_nb));
        budget -= 2;
        uint32_t par = 0;
        if (g1) {
          par = pd_bin(d, ctx, SL_TS(CABAC_CTX_TS_PAR_FLAG));
          budget--;
        }
        sign_pattern |= sign << n_nz;
        n_nz++;
        nz_mask |= 1u << i;
        const int32_t m = (int32_t)(1u + par + g1);
        put((uint32_t)i, sign ? -m : m);
      }
      last1 = i;
    }
    for (int32_t i = 0; i <= hi_i && budget >= 4; i++) {  // pass 2: greater-than-3/5/7/9 flags (:3276-3297)
      int32_t c = (int32_t)rl((uint32_t)val, (uint32_t)i);
      c = c < 0 ? -c : c;
      for (int32_t cut = 2; cut <= 8; cut += 2)
        if (c >= cut) {
          pd_check(d);
          c += (int32_t)(pd_bin(d, ctx, SL_TS(CABAC_CTX_TS_GTX_FLAG) + (uint32_t)(cut >> 1)) << 1);
          budget--;
        }
      put((uint32_t)i, c);
      last2 = i;
    }
    for (int32_t i = 0; i <= hi_i; i++) {  // pass 3: remainders, bypass levels with their signs, un-mapping (:3299-3329)
      int32_t c = (int32_t)rl((uint32_t)val, (uint32_t)i);
      c = c < 0 ? -c : c;
      const int32_t cut = i <= last2 ? 10 : i <= last1 ? 2 : 0;
      if (c >= cut) {
        const int32_t rem = (int32_t)pd_rem_abs(d, 1u, g.max_log2);
        c += i <= last1 ? rem << 1 : rem;
        if (c != 0 && i > last1) {
          pd_check(d);
          sign_pattern |= pd_ep(d) << n_nz;
          n_nz++;
          nz_mask |= 1u << i;
        }
      }
      if (!bdpcm && cut != 0 && c > 0) {  // decDeriveModCoeff, context_modelling.hpp:367-384
        int32_t left = (int32_t)rl((uint32_t)nb_l, (uint32_t)i), above = (int32_t)rl((uint32_t)nb_a, (uint32_t)i);
        left = left < 0 ? -left : left;
        above = above < 0 ? -above : above;
        const int32_t pred = left > above ? left : above;
        c = (c == 1 && pred > 0) ? pred : c - (c <= pred ? 1 : 0);
      }
      put((uint32_t)i, c);
    }
    {  // signs in the order the positions became non-zero: ascending scan position
      const bool mine = lane < g.cg_size && ((nz_mask >> j) & 1u);
      const uint32_t k = (uint32_t)__builtin_popcount(nz_mask & ((1u << j) - 1u));
      const bool neg = ((sign_pattern >> k) & 1u) != 0u;
      if (mine) blk[((uint32_t)y << g.lwe) + (uint32_t)x] = neg ? -val : val;
    }
  }
}

}  // namespace

// C: the type the coefficients are stored as — int32_t (the reference's TCoeff) or int16_t (streams of 15-bit dynamic range: half
// the bytes back over PCIe for the host-pointer path; a level that does not fit sets CABAC_RES_RANGE and is stored truncated)
template <int W, class C>
__global__ __launch_bounds__(64 * W) void residual_parse_kernel(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                                  const uint8_t *__restrict__ bytes,
                                                                  const uint32_t *__restrict__ tile_first,
                                                                  const cabac_tu_desc *__restrict__ tus, C *__restrict__ coeff_all,
                                                                  uint32_t *__restrict__ tu_info,
                                                                  cabac_substream_result *__restrict__ results) {
  __shared__ uint4 ctx_all[W * kCtxSlots];
  __shared__ int32_t blk_all[W * kBlkWords];
  __shared__ LdsTables tab;
  // Placement: the walk is bound by what the waves of a CU share (instruction issue, the scalar unit), so the waves must
  // be spread evenly: with 33 KB of LDS per workgroup at most four fit on a CU, and a 4 096-substream batch is exactly
  // four per CU (measured: 16 waves on every CU, 4 on every SIMD).
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  {  // scan / shape tables into LDS
    uint8_t *dst = reinterpret_cast<uint8_t *>(&tab);
    const uint8_t *g0 = &c_diag.grid[0][0][0], *g1 = &c_diag.in_cg[0][0][0];
    for (uint32_t k = threadIdx.x; k < sizeof(tab.grid); k += 64u * W) dst[offsetof(LdsTables, grid) + k] = g0[k];
    for (uint32_t k = threadIdx.x; k < sizeof(tab.in_cg); k += 64u * W) dst[offsetof(LdsTables, in_cg) + k] = g1[k];
    const uint16_t *t0 = &c_shapes.in_template[0][0][0];
    uint16_t *t1 = &tab.in_template[0][0][0];
    for (uint32_t k = threadIdx.x; k < 5u * 5u * 16u; k += 64u * W) {
      t1[k] = t0[k];
      (&tab.left[0][0][0])[k] = (&c_shapes.left[0][0][0])[k];
      (&tab.above[0][0][0])[k] = (&c_shapes.above[0][0][0])[k];
    }
  }
  const uint32_t sub = rfl(blockIdx.x * W + wave);
  const bool live = sub < n_sub;
  uint4 *ctx = ctx_all + wave * kCtxSlots;
  int32_t *blk = blk_all + wave * kBlkWords;
  const cabac_substream_desc dsc = desc[live ? sub : 0];
  {
    const int qp = dsc.qp < 0 ? 0 : (dsc.qp > 63 ? 63 : dsc.qp);
    const uint32_t iid = dsc.init_id & 3u;
    for (uint32_t k = lane; k < 230u; k += 64u) {
      const uint32_t id = id_of_slot(k);
      const uint32_t packed = ctx2_init(qp, c_init_tables[iid * kNumCtx + id], c_init_tables[3 * kNumCtx + id]);
      const uint32_t r0 = (packed & 3u) + 2u, r1 = ((packed >> 2) & 7u) + 5u;
      ctx[k] = make_uint4(packed & ~31u, r0 | (r1 << 16), ((0x7fffu >> r0) & kMask0) | (((0x7fffu >> r1) & kMask1) << 16), 0u);
    }
    for (uint32_t k = lane; k < kBlkWords; k += 64u) blk[k] = 0;
  }
  __syncthreads();
  if (!live) return;

  // every other wave SLOT of a SIMD keeps its decoder on the scalar pipe (pd_bin): whatever the placement, the waves that
  // share a SIMD split about evenly
#ifndef CABAC_PARSE_SCALAR_WAVES
#define CABAC_PARSE_SCALAR_WAVES 0   // 0: none, 1: every other wave slot, 2: all.  Measured on the bench tiles: 6.30 / 6.73 / - ms
#endif
  const bool scalar_wave = CABAC_PARSE_SCALAR_WAVES == 2 ||
                           (CABAC_PARSE_SCALAR_WAVES == 1 && (__builtin_amdgcn_s_getreg((4) | (0 << 6) | (3 << 11)) & 1u) != 0u);   // HW_ID.wave_id[3:0]
  auto walk = [&](auto kind) {
  PDecT<decltype(kind)::value> d;
  d.lane = lane;
  d.cap = dsc.byte_capacity;
  const uint8_t *src = bytes + dsc.byte_offset;
  d.src_safe = d.cap != 0u ? src : reinterpret_cast<const uint8_t *>(desc);
  d.last_dword = d.cap != 0u ? ((d.cap - 1u) & ~3u) : 0u;
  d.in_cur = pd_load_block(d, 0u);
  d.in_nxt = pd_load_block(d, 1u);
  {  // start(), arith_codec.cpp:60-66
    const uint32_t first = rl(d.in_cur, 0u);
    d.hi = first >> 1;
    d.lo = first << 31;
  }
  d.look = 16;
  d.rp = 4u;
  d.range = 510u;
  uint32_t flags_out = 0;

  const uint32_t t_end = tile_first[sub + 1];
  PP_TICK(k0);
  for (uint32_t t = tile_first[sub]; t < t_end; t++) {
    PP_TICK(b0);
    const cabac_tu_desc tu = tus[t];
    BlockGeom g;
    const uint32_t lw = tu.log2_width, lh = tu.log2_height;
    g.chroma = tu.channel;
    g.fl = tu.flags;
    g.max_log2 = tu.max_log2_tr_range ? tu.max_log2_tr_range : 15u;
    if (lw > 6u || lh > 6u || g.chroma > 1u || g.max_log2 > 20u || g.max_log2 < 15u) {
      flags_out |= CABAC_RES_BAD_RECORD;  // a block this parser does not cover: stop here
      break;
    }
    geom_init(g, lw, lh);
    // ts_flag, cabac_reader.cpp:2737-2752: in the stream where transform skip is allowed, else as the descriptor says
    uint32_t ts = (g.fl & CABAC_TU_TRANSFORM_SKIP) ? 1u : 0u;
    if (g.fl & CABAC_TU_TS_FLAG) {
      pd_check(d);
      ts = pd_bin(d, ctx, SL_TS_FLAG(g.chroma));
    }
    if (ts && (lw > 5u || lh > 5u)) {
      flags_out |= CABAC_RES_BAD_RECORD;
      break;
    }
    uint32_t info = CABAC_TU_INFO_TS;
    PP_TICK(b1);
    PP_ADD(0, b0, b1);
    if (ts) parse_ts(d, ctx, blk, tab, g, lane);
    else if ((g.fl & CABAC_TU_SBT_ZERO_OUT) && g.chroma == 0u && g.lw <= 5u && g.lh <= 5u) info = parse_regular<decltype(d), true>(d, ctx, blk, tab, g, lane);
    else info = parse_regular<decltype(d), false>(d, ctx, blk, tab, g, lane);
    if (tu_info && lane == 0u) tu_info[t] = info;
    PP_TICK(b2);
    // the finished block goes out row by row (all lanes), the LDS copy is cleared for the next block
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    C *out = coeff_all + tu.coeff_offset;
    uint32_t outside = 0;  // int16_t: a level beyond the type
    for (uint32_t i = lane; i < g.we * g.he; i += 64u) {
      const int32_t v = blk[i];
      out[((i >> g.lwe) << lw) + (i & (g.we - 1u))] = (C)v;
      if (sizeof(C) == 2) outside |= ((uint32_t)v + 32768u) >> 16;
      blk[i] = 0;
    }
    if (sizeof(C) == 2 && __ballot(outside != 0u) != 0ull) flags_out |= CABAC_RES_RANGE;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PP_TICK(b3);
    PP_ADD(7, b2, b3);
#ifdef CABAC_PARSE_PROFILE
    if (blockIdx.x == 0 && threadIdx.x == 0) g_parse_prof[12] += 1;  // blocks
#endif
  }
  PP_TICK(k1);
  PP_ADD(8, k0, k1);
#ifdef CABAC_PARSE_PROFILE
  if (lane == 0u) {  // spread of the waves' walk times, and when (after the first wave's start) the last one ended
    atomicMin(&g_parse_prof[13], k1 - k0);
    atomicMax(&g_parse_prof[14], k1 - k0);
    atomicAdd(&g_parse_prof[15], k1 - k0);
    if (sub < 8192u) {
      g_parse_wave[3 * sub] = k0;
      g_parse_wave[3 * sub + 1] = k1;
      g_parse_wave[3 * sub + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |
                                  ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32);
    }
  }
#endif

  // encodeBinTrm(1) closes the substream (cabac_writer.cpp:104-107); decodeBinTrm, arith_codec.cpp:181-197
  uint32_t trm = 1;
  const bool finish = (dsc.init_id & CABAC_SUB_FINISH) && !flags_out;
  if (finish) {
    pd_check(d);
    const uint32_t range = rfl(d.range) - 2u, hi = rfl(d.hi);
    trm = hi >= (range << 22) ? 1u : 0u;
    if (!trm && range < 256u) {
      const uint64_t v = (((uint64_t)d.hi << 32) | d.lo) << 1;
      d.hi = (uint32_t)(v >> 32);
      d.lo = (uint32_t)v;
      d.look -= 1;
    }
  }
  // the reference's counters (arith_codec.cpp:257-260) from the bits consumed: everything moved into the window
  // (8 * rp) minus value (16) minus look-ahead
  const uint32_t shifts = 8u * d.rp - 16u - rfl((uint32_t)d.look);
  const uint32_t bytes_read = 2u + (shifts >> 3);
  const int32_t bits_needed = (int32_t)(shifts & 7u) - 8;
  if (bytes_read > d.cap) {
    flags_out |= CABAC_RES_UNDERRUN;  // an underrun throws before finish() is reached
  } else if (finish) {
    // finish(), arith_codec.cpp:68-73: the last byte read holds the stop bit where the decoder stands
    const uint32_t lastb = src[bytes_read - 1u];
    const bool stop_ok = ((lastb << (8 + bits_needed)) & 0xffu) == 0x80u;
    if (!(trm && stop_ok)) flags_out |= CABAC_RES_BAD_STOP;
  }
  if (lane == 0u) {
    cabac_substream_result r;
    r.n_bits = 8u * bytes_read + (uint32_t)bits_needed;
    r.flags = flags_out;
    results[sub] = r;
  }
  };
  if (scalar_wave) walk(std::true_type{});
  else walk(std::false_type{});
}

#ifdef CABAC_PARSE_PROFILE
hipError_t debug_read_parse_waves(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_parse_wave), sizeof(unsigned long long) * 3 * 8192);
}
hipError_t debug_read_parse_prof(unsigned long long *out) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_parse_prof), sizeof(unsigned long long) * 16);
  unsigned long long zero[16] = {};
  zero[13] = ~0ull;
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_parse_prof), zero, sizeof zero);
  return e;
}
#endif

template <class C>
static hipError_t launch_residual_parse_as(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint8_t *bytes,
                                           const uint32_t *tile_first, const cabac_tu_desc *tus, C *coeff, uint32_t *tu_info,
                                           cabac_substream_result *results) {
  // four waves per workgroup, one per SIMD of the CU; small batches spread single waves over the chip
  if (n_sub >= 1024u)
    hipLaunchKernelGGL((residual_parse_kernel<4, C>), dim3((n_sub + 3u) / 4u), dim3(256), 0, st, n_sub, desc, bytes, tile_first, tus,
                       coeff, tu_info, results);
  else
    hipLaunchKernelGGL((residual_parse_kernel<1, C>), dim3(n_sub), dim3(64), 0, st, n_sub, desc, bytes, tile_first, tus, coeff, tu_info,
                       results);
  return hipGetLastError();
}

hipError_t launch_residual_parse(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint8_t *bytes,
                                 const uint32_t *tile_first, const cabac_tu_desc *tus, void *coeff, int coeff_bytes, uint32_t *tu_info,
                                 cabac_substream_result *results) {
  if (n_sub == 0) return hipSuccess;
  if (coeff_bytes == 2) return launch_residual_parse_as(st, n_sub, desc, bytes, tile_first, tus, static_cast<int16_t *>(coeff), tu_info, results);
  if (coeff_bytes == 4) return launch_residual_parse_as(st, n_sub, desc, bytes, tile_first, tus, static_cast<int32_t *>(coeff), tu_info, results);
  return hipErrorInvalidValue;
}

}  // namespace cabac
