// v4 "quad" kernels: FOUR substreams per wavefront, 16 lanes each, serial chain on the VECTOR pipe.
//
// Measured cost model of gfx950 (tools/ubench_*.hip): a wave issues one instruction per
// ~4.4 cycles, scalar or vector, dependent or not; a SIMD executes one wave-wide integer instruction per
// ~2.2 cycles (two waves per SIMD both run at full speed); a scalar instruction that consumes a vector
// result stalls the wave ~55 cycles.  With one substream per wave (v1, v3) every instruction of the serial
// low/range chain advances ONE bin.  Here the four 16-lane rows of a wave carry four different substreams and
// the chain is written with per-lane (row-uniform) vector arithmetic, so one instruction advances FOUR bins;
// the record of step i is delivered to its row with a DPP row broadcast (v_mov_b32_dpp row_newbcast:i — one
// instruction, no LDS, no scalar round trip).  Everything row-divergent is mask arithmetic; the only
// branches are the rare ones (output of whole 16-bit units, input refill).
//
//  encode: per 16-bin step  (a) quad_resolve: the context state each of the 16 bins of a row sees (match-any
//          on ctxId, then every lane applies the earlier bins of its context itself);  (b) 16 unrolled
//          chain steps of 23 vector instructions for the 4 rows together.  encode_kernel_v4 does both in one
//          wave, encode_kernel_v5 in two (context wave / chain wave, see there).
//          Byte output: `low` is kept as the exact code value and 16 bits are peeled off whenever 16
//          have accumulated; the delayed carry of arith_codec.cpp:524-546 (buffered byte + count of
//          outstanding 0xFF) is the same algorithm in base 2^16 (buffered unit + count of outstanding
//          0xFFFF).  The emitted stream is the identical number: the top S+1 bits of low >> 8 with
//          carries resolved (S = bits shifted), which is what finish() (:339-357) leaves.
//  decode: a 64-bit look-ahead window per row; after each bin every lane applies it to its own
//          copy of the context state and the lanes of that row holding the same ctxId keep it.
//  estimate: BitEstimator_Std on the same records — quad_resolve plus a table lookup, no chain.
//
// Layout: row r of wave w codes substream 4w + r.  Rows whose substreams are shorter idle at the end,
// so batches should group substreams of similar length (cabac_hip.h: order is the caller's).
#include <cstdlib>

#include "cabac_device.h"
#include "cabac_kernels.h"

namespace cabac {

constexpr uint32_t kQuadSubs = 4;        // substreams (rows) per wave
constexpr uint32_t kQuadCtxStride = 380; // LDS words per row context store (379 + pad)

template <int I>
__device__ __forceinline__ uint32_t row_bcast(uint32_t v) {
  // lane I of every 16-lane row -> all lanes of that row (DPP_ROW_NEWBCAST0 = 0x150, gfx90a+)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + I, 0xf, 0xf, true);  // bound_ctrl: no old-value init
}

// lane I of every group of L lanes -> all lanes of that group: L = 16 the DPP row broadcast above, L = 4 a DPP quad
// permutation (quad_perm:[I,I,I,I])
template <int L, int I>
__device__ __forceinline__ uint32_t group_bcast(uint32_t v) {
  if constexpr (L == 64) {  // one substream per wave: a scalar — handed back in a vector register, so that the chain stays the
    uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)v, I);  // vector code of the other geometries (as scalar code it is longer)
    asm volatile("" : "+v"(r));
    return r;
  }
  else if constexpr (L == 16) return row_bcast<I>(v);
  else return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, I | (I << 2) | (I << 4) | (I << 6), 0xf, 0xf, true);
}

// Lanes (of `lanes`) that hold the same BITS-bit key as this lane.  One ballot per key bit; each lane keeps
// the lanes that agree with it on that bit: m &= ~(ballot ^ mybit), with the lane's bit spread to a 0 / ~0
// word so that the whole round is four vector instructions (bfe, cmp, 2 x bitop3) and no scalar one.
template <int BITS>
__device__ __forceinline__ uint64_t match_any_bits(uint32_t key, uint64_t lanes) {
  uint32_t mlo = (uint32_t)lanes, mhi = (uint32_t)(lanes >> 32);
#pragma unroll
  for (int b = 0; b < BITS; b++) {
    const uint32_t mine = (uint32_t)((int32_t)(key << (31 - b)) >> 31);
    const uint64_t bal = __ballot(mine != 0);
    mlo &= ~((uint32_t)bal ^ mine);
    mhi &= ~((uint32_t)(bal >> 32) ^ mine);
  }
  return ((uint64_t)mhi << 32) | mlo;
}

// 0 / ~0 from one bit of x
template <int BIT>
__device__ __forceinline__ uint32_t bit_mask(uint32_t x) {
  return (uint32_t)((int32_t)(x << (31 - BIT)) >> 31);
}
// (a & m) | (b & ~m)  — v_bfi_b32
__device__ __forceinline__ uint32_t sel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
__device__ __forceinline__ uint32_t neg_mask(uint32_t x) { return (uint32_t)((int32_t)x >> 31); }  // ~0 if bit 31 set

// sixteen consecutive LDS words into registers, four at a time (16-byte reads)
__device__ __forceinline__ void lds_read4(const uint32_t *p, int q, uint32_t (&dst)[16]) {
  const uint4 v = reinterpret_cast<const uint4 *>(p)[q];
  dst[4 * q] = v.x;
  dst[4 * q + 1] = v.y;
  dst[4 * q + 2] = v.z;
  dst[4 * q + 3] = v.w;
}
__device__ __forceinline__ void lds_read16(const uint32_t *p, uint32_t (&dst)[16]) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint4 v = q[i];
    dst[4 * i] = v.x;
    dst[4 * i + 1] = v.y;
    dst[4 * i + 2] = v.z;
    dst[4 * i + 3] = v.w;
  }
}


// ---------------------------------------------------------------------------------------------
// encode

struct QuadEnc {       // row-uniform values
  uint64_t low;        // exact code value: 9 + pend (+1 carry) bits; pend <= 15 + 4*7 between flush checks
  uint32_t range;
  int32_t pend;        // bits shifted since the last 16-bit unit was peeled off (< 16 between steps)
  uint32_t buf;        // buffered unit
  int32_t nbuf;        // buffered unit + outstanding 0xFFFF units
  uint32_t pos;        // bytes stored so far
  uint8_t *dst;
  uint32_t cap;
};

__device__ __forceinline__ void quad_put_byte(QuadEnc &e, uint32_t byte, bool writer) {
  if (writer && e.pos < e.cap) e.dst[e.pos] = (uint8_t)byte;
  e.pos += 1;
}

__device__ __forceinline__ void quad_put16(QuadEnc &e, uint32_t unit, bool writer) {
  // big-endian 16-bit unit at byte offset pos (pos is even until finish)
  if (writer) {
    if (e.pos + 2u <= e.cap) {
      *reinterpret_cast<uint16_t *>(e.dst + e.pos) = (uint16_t)(((unit & 0xffu) << 8) | ((unit >> 8) & 0xffu));
    } else if (e.pos < e.cap) {
      e.dst[e.pos] = (uint8_t)(unit >> 8);
    }
  }
  e.pos += 2;
}

// one 16-bit unit with the carry that arrived on top of it (lead = carry << 16 | unit): delayed carry as
// writeOut (arith_codec.cpp:524-546), in base 2^16
__device__ __forceinline__ void quad_flush_lead(QuadEnc &e, uint32_t lead, bool writer) {
  const bool is_ff = lead == 0xffffu;
  const bool emit = !is_ff && e.nbuf > 0;
  const uint32_t carry = lead >> 16;
  const uint32_t first = e.buf + carry;
  const int32_t fill_n = e.nbuf - 1;
  e.buf = is_ff ? e.buf : (lead & 0xffffu);
  e.nbuf = is_ff ? e.nbuf + 1 : 1;
  if (emit) {
    quad_put16(e, first, writer);
    for (int32_t k = 0; k < fill_n; k++) quad_put16(e, 0xffffu + carry, writer);
  }
}

// peel 16 bits (and the carry above them) off the top of low
__device__ __forceinline__ void quad_flush16(QuadEnc &e, bool writer) {
  const uint32_t sh = (uint32_t)(9 + e.pend - 16);
  const uint32_t lead = (uint32_t)(e.low >> sh);  // carry + 16 bits
  e.low &= (1ull << sh) - 1ull;
  e.pend -= 16;
  quad_flush_lead(e, lead, writer);
}

// Where the chain wave of the two-wave encoder (v5) leaves its output: every 4th bin it posts (low, pend) to LDS
// and keeps only the pend % 16 bits that are not yet a whole unit; the context wave peels the units off the
// posted value one step later (quad_flush16) — the carry / output stage is serial per unit and full of
// per-row conditions, and on the chain wave it cost 30 % of the kernel.
struct QuadPost {
  uint32_t *lo, *hi, *pend;  // [check 0..3][row 0..3] of the current step's slot, already offset to this row
};

// Per-lane fields of one bin for the chain (phase A fills them; all zero = no-op step)
struct QuadEncInfo {
  uint32_t k;     // LPS factor (state folded to 0..127) >> 2; 0 for bypass / terminate
  uint32_t c2;    // 2 * constant term of the LPS width: 8 context bin, 4 terminate bin, 0 bypass
  uint32_t lpsm;  // ~0 if this bin takes the LPS path (context: bin != mps; terminate: bin == 1)
  uint32_t ep;    // 1 for a bypass bin
  uint32_t pem;   // ~0 for a bypass bin with value 1
  uint32_t alm;   // ~0 for an align() record
};

// One chain step for the four rows.  The fields reach the row by DPP row broadcasts; masks are used
// with and/bfi (no compares, no exec regions): the wave is alone on its SIMD and every VALU->SALU->EXEC
// round trip would sit on the critical path.
template <int I, bool kAlign, bool kPost>
__device__ __forceinline__ void quad_enc_step(const QuadEncInfo &f, QuadEnc &e, bool writer, const QuadPost &post) {
  const uint32_t k = row_bcast<I>(f.k), c2 = row_bcast<I>(f.c2), lpsm = row_bcast<I>(f.lpsm), ep = row_bcast<I>(f.ep);
  const uint32_t t = (__umul24((e.range >> 5) & 15u, k) + c2) >> 1;  // LPS width: ((r>>5)*k>>1) + c
  const uint32_t rm = e.range - t;
  const uint32_t nl = (uint32_t)(__builtin_clz(t) - 23);  // getRenormBitsLPS (contexts.cpp:952-954); masked out when t == 0
  const uint32_t nm = (rm >> 8) ^ 1u;                     // rm < 512: 1 iff rm < 256
  const uint32_t nb = sel(lpsm, nl, nm);
  e.range = sel(lpsm, t, rm) << nb;
  // low = ((low + (LPS ? rm : 0)) << nb), then for a bypass bin (low << 1) + bin * range (arith_codec.cpp:389-399,
  // :553-582).  The two additions exclude each other, so both go after ONE shift: a single 64-bit shift and
  // add per bin keep the 64-bit dependency chain short.
  const uint32_t term = ((rm & lpsm) << nb) | (e.range & row_bcast<I>(f.pem));
  e.low = (e.low << (nb + ep)) + term;
  if (kAlign) e.range = sel(row_bcast<I>(f.alm), 256u, e.range);
  e.pend += (int32_t)(nb + ep);
  // output check only every 4th bin: 4 bins shift at most 28 bits, which the 64-bit low absorbs, and the
  // three steps in between stay in one basic block, so hipcc can overlap their independent parts
  if ((I & 3) == 3) {
    if (kPost) {
      constexpr int kCheck = (I >> 2) * kQuadSubs;
      if (writer) {
        post.lo[kCheck] = (uint32_t)e.low;
        post.hi[kCheck] = (uint32_t)(e.low >> 32);
        post.pend[kCheck] = (uint32_t)e.pend;
      }
      // whole units, and the carry above them, now belong to the post; without a whole unit nothing is cut
      // (a carry bit above the 9 + pend bits must stay until it can be posted on top of a unit)
      const uint32_t keep = (uint32_t)e.pend & 15u;
      const uint32_t width = e.pend >= 16 ? 9u + keep : 63u;
      e.low &= ~(~0ull << width);
      e.pend = (int32_t)keep;
    } else {
      while (__builtin_expect(e.pend >= 16, 0)) quad_flush16(e, writer);
    }
  }
}

template <bool kAlign, bool kPost>
__device__ __forceinline__ void quad_enc_steps(const QuadEncInfo &f, QuadEnc &e, bool writer, const QuadPost &post) {
#define QSTEP(I) quad_enc_step<I, kAlign, kPost>(f, e, writer, post)
  QSTEP(0); QSTEP(1); QSTEP(2); QSTEP(3); QSTEP(4); QSTEP(5); QSTEP(6); QSTEP(7);
  QSTEP(8); QSTEP(9); QSTEP(10); QSTEP(11); QSTEP(12); QSTEP(13); QSTEP(14); QSTEP(15);
#undef QSTEP
}

// Phase (a) of one 16-bin step for the four rows: the context state each bin sees, resolved in parallel
// (see v3), the LDS context store brought up to date, and the bin's chain fields packed into one word:
//   bits 4..0 k | bits 8..5 2c | bit 9 LPS path | bit 10 bypass | bit 11 bypass bin 1 | bit 12 align
struct QuadRecord {  // one bin record of a 16-bin step, with the context state it sees
  uint32_t id, bin, st;
  uint32_t ctxm, epm, trmm, alnm;  // 0 / ~0: what kind of record it is (all zero: none — past the end of the substream)
};

// Written with 0 / ~0 masks throughout (round 2): as booleans the record kinds became lane masks in SGPRs, and the
// s_and_b64 / s_and_saveexec that combined them each waited ~55 cycles for a vector compare — six to eight times per
// step in the context wave, which every encoder since v5 waits for.
constexpr uint32_t kMatchWords = 1024;  // per wave: which lanes of a row hold which record id, 16 bits per row and id, two rows per word
// `match` (optional, LDS, all zero between calls): the lanes of a row with the same id found through LDS instead of with nine
// ballots — every lane ORs its bit into the word of its id, reads the word back and clears it: three LDS instructions and
// one round trip instead of 36 vector instructions that write scalar registers (measured alone, tools/ubench_ctx.hip:
// the nine-round match-any is 160 of quad_phase_a's 350 ns)
// `rates` (kRateTab, LDS, 512 entries): the packed shift amounts and addends of the two estimators of every context, looked
// up instead of derived from the state word's rate bits with ten vector instructions (they depend on the context only:
// the fourth row of the init table; the estimator, which may start from given rates, derives them)
template <uint32_t kLowestSpecial = CABAC_REC_ALIGN, bool kLds = false, bool kRateTab = false>  // ids from kLowestSpecial up to 0x1FF are not "bad" (the estimator has two more)
__device__ __forceinline__ QuadRecord quad_resolve(uint32_t r, bool active, uint32_t lane, uint32_t row, uint32_t *rctx,
                                                   uint32_t &bad, uint32_t *match = nullptr, const uint2 *rates = nullptr) {
  QuadRecord q;
  uint32_t actm = active ? ~0u : 0u;
  asm volatile("" : "+v"(actm));   // opaque: hipcc otherwise turns the mask arithmetic below back into lane-mask booleans
  const uint32_t id = sel(actm, r & CABAC_REC_ID_MASK, CABAC_REC_ID_MASK);
  const uint32_t bin = (r >> 15) & 1u;
  const uint32_t ctxm = neg_mask(id - (uint32_t)kNumCtx);                         // id < 379 (an inactive lane's id is 0x1FF)
  const uint32_t epm = actm & neg_mask((id ^ CABAC_REC_EP) - 1u);
  const uint32_t trmm = actm & neg_mask((id ^ CABAC_REC_TRM) - 1u);
  const uint32_t alnm = actm & neg_mask((id ^ CABAC_REC_ALIGN) - 1u);
  bad |= actm & ~ctxm & neg_mask(id - kLowestSpecial) & 1u;
  const uint32_t j = lane & 15u;
  uint32_t same;
  if (kLds) {   // (a template parameter, not `match != nullptr`: an LDS address may be 0, so that test survives to run time)
    uint32_t *word = match + ((row >> 1) << 9) + id;
    const uint32_t shift = ((row & 1u) << 4);
    atomicOr(word, 1u << (shift + j));
    asm volatile("" ::: "memory");                 // one wave: LDS executes its instructions in order
    same = (*word >> shift) & 0xffffu;             // (not through a volatile pointer: that becomes a system-scope flat load)
    asm volatile("" ::: "memory");
    *word = 0u;
  } else {
    same = (uint32_t)(match_any_bits<9>(id, 0xffffull << (16u * row)) >> (16u * row)) & 0xffffu;
  }
  const uint32_t binmask = (uint32_t)(__ballot(bin != 0) >> (16u * row)) & 0xffffu;  // the bins of this row
  const uint32_t before = same & ((1u << j) - 1u);      // earlier bins of the same context in this step
  const uint32_t lastm = neg_mask(((same >> j) ^ 1u) - 1u);  // no later one
  const uint32_t stored = rctx[min(id, (uint32_t)kNumCtx)];  // unconditional load (slot kNumCtx: the row's pad word)
  uint32_t st = stored & ctxm;
  // The state this bin sees is the stored one updated by those earlier bins, oldest first.  Their values are
  // known (encoder), so every lane walks its own `before` set — no hand-over between lanes, hence no LDS
  // round trip per repetition of a context.  update(), contexts.cpp:903-913, on both 15-bit estimators at
  // once with packed 16-bit math as in the decoder (the rates of a context never change).
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  u16x2 rate2, add2;
  if (kRateTab) {
    const uint2 ra = rates[id];
    rate2 = __builtin_bit_cast(u16x2, ra.x);
    add2 = __builtin_bit_cast(u16x2, ra.y);
  } else {
    const uint32_t r0 = (st & 3u) + 2u, r1 = ((st >> 2) & 7u) + 5u;
    rate2 = __builtin_bit_cast(u16x2, r0 | (r1 << 16));
    add2 = __builtin_bit_cast(u16x2, ((0x7fffu >> r0) & kMask0) | (((0x7fffu >> r1) & kMask1) << 16));
  }
  const u16x2 mask2 = __builtin_bit_cast(u16x2, (kMask1 << 16) | kMask0);
  auto updated = [&](uint32_t s, uint32_t b) {
    const u16x2 s2 = __builtin_bit_cast(u16x2, s);
    const u16x2 b2 = __builtin_bit_cast(u16x2, b | (b << 16));
    return __builtin_bit_cast(uint32_t, (u16x2)(add2 * b2 + (s2 - ((s2 >> rate2) & mask2))));
  };
  uint32_t todo = before & ctxm;
  // How many repetitions the wave needs is asked right here (three ballots on todo with its lowest one / two bits
  // removed), long before the answers are branched on: a round is ten vector instructions for all lanes, and most
  // steps need none or one.
  const uint32_t after1 = todo & (todo - 1u), after2 = after1 & (after1 - 1u);
  uint64_t any1 = __ballot(todo != 0u), any2 = __ballot(after1 != 0u), more = __ballot(after2 != 0u);
  asm volatile("" : "+s"(any1), "+s"(any2), "+s"(more));
  auto one_round = [&]() {
    const uint32_t valid = (uint32_t)((int32_t)(0u - todo) >> 31);
    const uint32_t which = (uint32_t)__builtin_ctz(todo | 0x10000u);
    const uint32_t b = (binmask >> which) & 1u;
    st = sel(valid, updated(st, b), st);
    todo &= todo - 1u;
  };
  if (any1 != 0) {
    one_round();
    if (any2 != 0) {
      one_round();
      if (__builtin_expect(more != 0, 0)) {
        for (int round = 2; round < 16; round++) {
          one_round();
          if (__ballot(todo != 0) == 0) break;
        }
      }
    }
  }
  rctx[sel(ctxm & lastm, id, (uint32_t)kNumCtx)] = updated(st, bin);  // others write the pad word
  q.id = id;
  q.bin = bin;
  q.st = st;
  q.ctxm = ctxm;
  q.epm = epm;
  q.trmm = trmm;
  q.alnm = alnm;
  return q;
}

template <bool kLds = false>
__device__ __forceinline__ uint32_t quad_phase_a(uint32_t r, bool active, uint32_t lane, uint32_t row, uint32_t *rctx,
                                                 uint32_t &bad, uint32_t *match = nullptr, const uint2 *rates = nullptr) {
  const QuadRecord q = quad_resolve<CABAC_REC_ALIGN, kLds, kLds>(r, active, lane, row, rctx, bad, match, rates);
  const uint32_t bin = q.bin;
  const uint32_t q8 = ctx2_q8(q.st);
  const uint32_t mps = q8 >> 7;
  // inactive lanes: a no-op step (t = 0, no shift); the kinds exclude each other
  return ((ctx2_k(q8) | (8u << 5) | ((bin ^ mps) << 9)) & q.ctxm) |
         (((4u << 5) | (bin << 9)) & q.trmm) |          // terminate == LPS width 2 (arith_codec.cpp:460-478)
         (((1u << 10) | (bin << 11)) & q.epm) |
         ((1u << 12) & q.alnm);
}

// The same as separate words for the lane-serial encoder (v7), which parks them in LDS field by field: no packing here and
// no unpacking there.
struct QuadFields {
  uint32_t k, c2, lpsm, lp9, ep, alm;
};
template <bool kLds = false>
__device__ __forceinline__ QuadFields quad_phase_fields(uint32_t r, bool active, uint32_t lane, uint32_t row, uint32_t *rctx,
                                                        uint32_t &bad, uint32_t *match = nullptr, const uint2 *rates = nullptr) {
  const QuadRecord q = quad_resolve<CABAC_REC_ALIGN, kLds, kLds>(r, active, lane, row, rctx, bad, match, rates);
  const uint32_t sum = (q.st & kMask0) + (q.st >> 16);            // state(), contexts.cpp:939-950
  const uint32_t sx = (uint32_t)((int32_t)(sum << 16) >> 31);     // 0 / ~0: the MPS
  const uint32_t binm = 0u - q.bin;                               // 0 / ~0: the bin
  QuadFields f;
  f.k = ((sum >> 10) ^ sx) & 31u & q.ctxm;
  f.c2 = (8u & q.ctxm) | (4u & q.trmm);                           // terminate == LPS width 2 (arith_codec.cpp:460-478)
  f.lpsm = ((binm ^ sx) & q.ctxm) | (binm & q.trmm);              // the LPS path: bin != MPS, or a terminate bin 1
  f.lp9 = (f.lpsm | (binm & q.epm)) & 0x1ffu;                     // the bin adds its MPS sub-range to low: LPS, or a bypass bin 1
  f.ep = q.epm & 1u;
  f.alm = q.alnm;
  return f;
}

__device__ __forceinline__ QuadEncInfo quad_unpack(uint32_t info) {
  QuadEncInfo f;
  f.k = info & 31u;
  f.c2 = (info >> 5) & 15u;
  f.lpsm = bit_mask<9>(info);
  f.ep = (info >> 10) & 1u;
  f.pem = bit_mask<11>(info);
  f.alm = bit_mask<12>(info);
  return f;
}

// finish(), arith_codec.cpp:339-357, on the exact code value (+ writeByteAlignment, bit_stream.cpp:152-155)
// probe (CABAC_SUB_PROBE): no flush; the answer of getNumWrittenBits() (arith_codec.cpp:482-485) instead — every shift so far
// has either left as a stored unit (8 * pos), waits as the buffered unit or an outstanding 0xFFFF (16 * nbuf), or is
// still inside low (pend)
__device__ __forceinline__ uint32_t quad_enc_finish(QuadEnc &e, bool align_rbsp, bool writer, bool probe = false) {
  if (probe) return 8u * e.pos + 16u * (uint32_t)e.nbuf + (uint32_t)e.pend;
  const uint32_t total = (uint32_t)(9 + e.pend);
  if ((e.low >> total) & 1ull) {
    quad_put16(e, e.buf + 1u, writer);
    for (int32_t k = 1; k < e.nbuf; k++) quad_put16(e, 0x0000u, writer);
    e.low -= 1ull << total;
  } else {
    if (e.nbuf > 0) quad_put16(e, e.buf, writer);
    for (int32_t k = 1; k < e.nbuf; k++) quad_put16(e, 0xffffu, writer);
  }
  uint32_t nb = (uint32_t)(e.pend + 1);  // write(low >> 8, 24 - bitsLeft)
  const uint32_t v = (uint32_t)(e.low >> 8);
  while (nb >= 8) {
    quad_put_byte(e, (v >> (nb - 8)) & 0xffu, writer);
    nb -= 8;
  }
  uint32_t held = nb ? ((v & ((1u << nb) - 1u)) << (8 - nb)) : 0u;
  if (align_rbsp) {
    held |= 1u << (7 - nb);
    quad_put_byte(e, held, writer);
    held = 0;
    nb = 0;
  }
  const uint32_t n_bits = e.pos * 8u + nb;
  if (nb) quad_put_byte(e, held, writer);
  return n_bits;
}

// the table quad_resolve<.., kRateTab> reads: per record id {shift0 | shift1 << 16, add0 | add1 << 16} (0 for ids that are no context)
__device__ __forceinline__ void quad_rate_tab_init(uint2 *tab, uint32_t tid, uint32_t n_threads) {
  for (uint32_t id = tid; id < 512u; id += n_threads) {
    uint2 v = make_uint2(0u, 0u);
    if (id < (uint32_t)kNumCtx) {
      const uint32_t rates = ctx2_init(0, c_init_tables[id], c_init_tables[3 * kNumCtx + id]) & 31u;
      const uint32_t r0 = (rates & 3u) + 2u, r1 = ((rates >> 2) & 7u) + 5u;
      v.x = r0 | (r1 << 16);
      v.y = ((0x7fffu >> r0) & kMask0) | (((0x7fffu >> r1) & kMask1) << 16);
    }
    tab[id] = v;
  }
}

__device__ __forceinline__ void quad_ctx_init(uint32_t *rctx, int qp_in, uint32_t iid, uint32_t j) {
  const int qp = qp_in < 0 ? 0 : (qp_in > 63 ? 63 : qp_in);
  for (uint32_t k = j; k < (uint32_t)kNumCtx; k += 16)
    rctx[k] = ctx2_init(qp, c_init_tables[iid * kNumCtx + k], c_init_tables[3 * kNumCtx + k]);
}

// encode, one wave per 4 substreams: phase (a) and the chain alternate in the same wave
__global__ __launch_bounds__(64) void encode_kernel_v4(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                       const uint16_t *__restrict__ records, uint8_t *__restrict__ bytes,
                                                       cabac_substream_result *__restrict__ results) {
  __shared__ uint32_t ctx[kQuadSubs * kQuadCtxStride];
  const uint32_t lane = threadIdx.x, row = lane >> 4, j = lane & 15u;
  const uint32_t sub = blockIdx.x * kQuadSubs + row;
  const bool live = sub < n_sub;  // rows beyond the batch idle with n = 0
  const cabac_substream_desc d = desc[live ? sub : 0];
  const uint32_t n = live ? d.n_records : 0u;
  const uint16_t *rec = records + d.rec_offset;
  uint32_t *rctx = ctx + row * kQuadCtxStride;
  quad_ctx_init(rctx, d.qp, d.init_id & 3u, j);
  __syncthreads();

  QuadEnc e;
  e.low = 0;
  e.range = 510;  // start(), arith_codec.cpp:329-337
  e.pend = 0;
  e.buf = 0;
  e.nbuf = 0;
  e.pos = 0;
  e.dst = bytes + d.byte_offset;
  e.cap = live ? d.byte_capacity : 0u;
  const bool writer = live && j == 0;
  uint32_t bad = 0;

  uint32_t next_rec = j < n ? rec[j] : 0;
  for (uint32_t base = 0; __ballot(base < n) != 0; base += 16) {
    const uint32_t r = next_rec;  // loaded one step ago
    const uint32_t info = quad_phase_a(r, base + j < n, lane, row, rctx, bad);
    const QuadEncInfo f = quad_unpack(info);
    // prefetch the next 16 records of each row now: the load completes under the serial chain.  (Issued
    // any earlier, hipcc's s_waitcnt vmcnt(0) in front of the first use of `r` would wait for it too.)
    {
      const uint32_t nxt = base + 16u + j;
      next_rec = nxt < n ? rec[nxt] : 0;
    }
    // (b) the serial chain, 4 rows at once
    if (__ballot(info >> 12) == 0) quad_enc_steps<false, false>(f, e, writer, QuadPost());
    else quad_enc_steps<true, false>(f, e, writer, QuadPost());
  }

  const uint32_t n_bits = live ? quad_enc_finish(e, (d.init_id & CABAC_SUB_ALIGN_RBSP) != 0, writer, (d.init_id & CABAC_SUB_PROBE) != 0) : 0u;
  const uint64_t bad_mask = __ballot(bad != 0);  // row-wide OR of the bad-record flag
  const bool row_bad = ((bad_mask >> (row * 16u)) & 0xffffull) != 0;
  if (writer) {
    cabac_substream_result res;
    res.n_bits = n_bits;
    res.flags = (e.pos > e.cap ? CABAC_RES_OVERFLOW : 0u) | (row_bad ? CABAC_RES_BAD_RECORD : 0u);
    results[sub] = res;
  }
}

// ---- output stage of the two-wave encoder (runs on the context wave) ---------------------------------
// The chain wave posts (low, pend) after bins 3, 7, 11, 15 of a step.  A post with pend >= 16 carries
// pend / 16 whole 16-bit units (at most two) and possibly a carry above them; the units of one step are at
// most eight per row.  quad_flush16 is the exact, serial rule for one unit.  Almost always every row is in
// the plain state "one unit buffered, no 0xFFFF run, no new unit equals 0xFFFF, room in the buffer", and
// then the rule degenerates to "store the previous unit plus the carry that arrived with this one": that
// is done for all units of the four rows at once, lane k of a row taking unit k, through a small LDS list.
// Anything else (first unit of a substream, 0xFFFF runs, a full buffer) takes the serial path.
constexpr uint32_t kUnitSlots = 12;  // 8 units + spare + a dump slot for lanes that have nothing to list
constexpr uint32_t kUnitDump = 11;

template <int N>
__device__ __forceinline__ uint32_t row_shr(uint32_t v) {  // lane j gets lane j - N of its row, 0 below
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, true);
}

// Everything below is written with 0 / ~0 masks in VGPRs (neg_mask, sel): boolean expressions become lane masks in SGPRs,
// and every scalar instruction that consumes a vector result stalls the wave for ~55 cycles.

struct QuadUnits {
  uint32_t m;            // units of this step in the row (0..8)
  uint64_t odd_rows;     // lanes of rows that are not in the plain state (before looking at the units themselves);
                         // kept as the raw mask: a scalar compare right behind the v_cmp would wait for it
  uint64_t store_lanes;  // lanes that will store a unit
};

// Part 1: lanes 0..3 of a row list the units of posts 0..3 in stream order.  kHalfRows: a substream has EIGHT lanes (two
// substreams per 16-lane DPP row, j = lane & 7) — eight is the most units a step can produce, and lanes 4..7 of a group
// count nothing, so the row shifts never carry a count across the two groups of a row.
template <bool kHalfRows = false>
__device__ __forceinline__ QuadUnits quad_list_units(const QuadEnc &e, const uint32_t *plo, const uint32_t *phi,
                                                     const uint32_t *ppend, uint32_t row, uint32_t j, bool live,
                                                     uint32_t *list, uint32_t lane = 0) {
  const uint32_t c = (j & 3u) * kQuadSubs + row;  // lanes >= 4 read along, and count nothing
  const uint64_t low = ((uint64_t)phi[c] << 32) | plo[c];
  const uint32_t pend = ppend[c];
  const uint32_t nun = (pend >> 4) & neg_mask(j - 4u);  // 0, 1 or 2 units
  const uint64_t v = low >> (9u + (pend & 15u));         // carry, then the units
  uint32_t incl = nun + row_shr<1>(nun);
  incl += row_shr<2>(incl);
  const uint32_t idx = incl - nun;
  const uint32_t first = (uint32_t)(v >> (((nun - 1u) & 1u) << 4));  // carry + first unit (nun >= 1)
  const uint32_t second = (uint32_t)v & 0xffffu;                      // second unit (nun == 2)
  const uint32_t has1 = neg_mask(0u - nun);                           // nun >= 1
  const uint32_t has2 = neg_mask(1u - nun);                           // nun == 2
  list[sel(has1, idx, kUnitDump)] = first;
  list[sel(has2, idx + 1u, kUnitDump)] = second;
  QuadUnits u;
  if (kHalfRows) u.m = (lane & 8u) ? row_bcast<11>(incl) : row_bcast<3>(incl);
  else u.m = row_bcast<3>(incl);
  const uint32_t room = e.cap - e.pos - 2u * u.m;                     // negative: the buffer would overflow
  const uint32_t rowodd = neg_mask(0u - u.m) & (neg_mask(0u - ((uint32_t)e.nbuf ^ 1u)) | neg_mask(room));
  u.odd_rows = __ballot(live && rowodd != 0);   // (rows without a substream never store: nothing to be careful about)
  u.store_lanes = __ballot(live && j < u.m);
  return u;
}

// the serial path: the listed units of one step, one by one
__device__ __forceinline__ void quad_flush_list(QuadEnc &e, uint32_t m, const uint32_t *list, bool writer) {
  for (uint32_t k = 0; k < 8u && __ballot(k < m) != 0; k++) {
    if (k < m) quad_flush_lead(e, list[k], writer);
  }
}

// Part 2: emit the listed units (one step later, so that neither the list nor the masks are waited for).
__device__ __forceinline__ void quad_emit_units(QuadEnc &e, const QuadUnits &u, uint32_t j, const uint32_t *list,
                                                bool writer) {
  if (u.odd_rows != 0) return quad_flush_list(e, u.m, list, writer);
  const uint32_t act = neg_mask(j - u.m);  // lane j holds unit j
  const uint32_t jnz = neg_mask(0u - j);
  const uint32_t before = list[sel(act & jnz, j - 1u, kUnitDump)];
  const uint32_t mine = list[sel(act, j, kUnitDump)];
  const uint32_t last = list[sel(neg_mask(0u - u.m), u.m - 1u, kUnitDump)];
  const uint32_t x = (mine & 0x1ffffu) ^ 0xffffu;  // 0: no carry, unit 0xFFFF — starts or extends a run
  if (__ballot((act & neg_mask(x - 1u)) != 0) != 0) return quad_flush_list(e, u.m, list, writer);
  const uint32_t prev = sel(jnz, before & 0xffffu, e.buf);
  const uint32_t unit = prev + (mine >> 16);  // buf + carry (writeOut, arith_codec.cpp:524-546, in base 2^16)
  const uint32_t be = ((unit & 0xffu) << 8) | ((unit >> 8) & 0xffu);
  uint8_t *addr = e.dst + e.pos + 2u * j;
  uint64_t saved;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tglobal_store_short %2, %3, off\n\ts_mov_b64 exec, %0"
               : "=&s"(saved) : "s"(u.store_lanes), "v"(addr), "v"(be) : "memory");
  e.buf = sel(neg_mask(0u - u.m), last & 0xffffu, e.buf);
  e.pos += 2u * u.m;
}

// encode, THREE waves per 4 substreams ("v5"): a context wave runs phase (a) one step ahead and owns the LDS
// context stores; a chain wave runs the range / low recurrence and nothing else; an output wave turns what the
// chain wave posts into bytes.  They never wait on each other's memory: the hand-offs are one packed word per bin
// through a double-buffered LDS mailbox (context -> chain), the posted (low, pend) of every 4th bin (chain ->
// output, see QuadPost) and one workgroup barrier per 16-bin step.  A wave issues at most one instruction per
// ~4.4 cycles while a SIMD executes one per ~2.2 (tools/ubench_ilp.hip): the chain wave (~420 instructions per
// step, raised priority) sets the pace, and the other two (~170 and ~100) fit into the issue slots it cannot use.
// All units of a workgroup run the same number of steps (the longest substream's) so that the barrier counts
// match; surplus steps are no-ops.
#ifdef CABAC_V5_PROFILE  // tools/ubench_v5.hip: where the two waves of workgroup 0 spend their cycles
__device__ unsigned long long g_v5_prof[16];
#define V5_TICK(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define V5_ADD(slot, t0, t1) \
  if (blockIdx.x == 0 && lane == 0) g_v5_prof[slot] += (t1) - (t0)
#else
#define V5_TICK(var)
#define V5_ADD(slot, t0, t1)
#endif

// (encode_kernel_v5, the three-wave encoder described above, was retired in round 3: v6 and v7 below grew out of it and it
// was never dispatched after them; its pieces that they share — QuadPost, quad_list_units, quad_emit_units — stay.)

// ---------------------------------------------------------------------------------------------
// v6: the three-wave encoder with `low` taken out of the chain wave.
//
// In v5 the chain wave — the wave that sets the pace — spends 9 of its 23 instructions per bin on the code value: the
// addend, the 64-bit shift-and-add, the count of pending bits.  None of that feeds back into the range recurrence.  Here
// the chain wave only runs the range recurrence (14 instructions per bin) and leaves, per bin, the twelve bits the code
// value needs — rm (the MPS sub-range, nine bits) and the renormalisation shift (three) — in eight row-uniform registers
// it posts once per 16-bin step.  The output wave rebuilds the code value from them, exactly, with the lanes:
//   low' = (low << sh) + term per bin  (arith_codec.cpp:389-399, :553-582)   =>   over a step
//   new  = (acc << S) + sum_I term_I << (sh_{I+1} + .. + sh_15),   S = sh_0 + .. + sh_15,
// a sum of sixteen shifted 16-bit terms into a number of up to 138 bits: lane I forms (term_I, sh_I) of its bin, and a
// four-level tree over the row (row_shr 1, 2, 4, 8) combines neighbouring segments — (left << S_right) + right — in 64,
// 64, 96 and 160 bits; the remainder carried from the step before (9 bits, the bits of the unfinished 16-bit unit and a
// carry) enters as part of lane 0's term.  All carries inside `new` are resolved by the exact additions; the one carry
// that can leave its top lands on the buffered unit, as in writeOut (arith_codec.cpp:524-546).  The whole 16-bit units
// are then cut off the top by the lanes (lane k takes unit k) and go through v5's emission (quad_emit_units).
struct QuadRngCap {
  uint32_t w[8];  // bins 2k (low half) and 2k + 1 (high half): rm | shift << 9
};

template <int I, bool kAlign>
__device__ __forceinline__ void quad_rng_step(const QuadEncInfo &f, uint32_t &range, QuadRngCap &cap) {
  const uint32_t k = row_bcast<I>(f.k), c2 = row_bcast<I>(f.c2), lpsm = row_bcast<I>(f.lpsm);
  const uint32_t t = (__umul24((range >> 5) & 15u, k) + c2) >> 1;  // LPS width: ((r>>5)*k>>1) + c
  const uint32_t rm = range - t;
  // One renormalisation rule for both paths: the new range is the chosen sub-range shifted up to [256, 511].  LPS:
  // clz(t) - 23 is getRenormBitsLPS (contexts.cpp:952-954; 7 for the terminate bin's t = 2).  MPS: rm >= 128 always
  // (an LPS width is at most 15.5 / 32 of the range plus 4), so the shift is 1 iff rm < 256 (arith_codec.cpp:389-399).
  const uint32_t x = sel(lpsm, t, rm);
  const uint32_t nb = (uint32_t)(__builtin_clz(x) - 23);
  range = x << nb;
  if (kAlign) range = sel(row_bcast<I>(f.alm), 256u, range);
  const uint32_t w = rm | (nb << 9);
  if (I & 1) cap.w[I >> 1] |= w << 16;
  else cap.w[I >> 1] = w;
}

// the same with the row-uniform fields of the 16 bins in registers (read back from LDS, where lane I parked bin I's: an
// LDS read of one address by all lanes of a row is the broadcast — three v_mov_b32_dpp per bin less, see QuadDecRow)
struct QuadEncRow {
  uint32_t k[16], c2[16], lpsm[16];
};
template <int I, bool kAlign>
__device__ __forceinline__ void quad_rng_step_row(const QuadEncRow &u, const QuadEncInfo &f, uint32_t &range, QuadRngCap &cap) {
  const uint32_t t = (__umul24((range >> 5) & 15u, u.k[I]) + u.c2[I]) >> 1;  // LPS width: ((r>>5)*k>>1) + c
  const uint32_t rm = range - t;
  const uint32_t x = sel(u.lpsm[I], t, rm);
  const uint32_t nb = (uint32_t)(__builtin_clz(x) - 23);
  range = x << nb;
  if (kAlign) range = sel(row_bcast<I>(f.alm), 256u, range);
  const uint32_t w = rm | (nb << 9);
  if (I & 1) cap.w[I >> 1] |= w << 16;
  else cap.w[I >> 1] = w;
}
template <bool kAlign>
__device__ __forceinline__ void quad_rng_steps_row(const QuadEncRow &u, const QuadEncInfo &f, uint32_t &range, QuadRngCap &cap) {
#define QSTEP(I) quad_rng_step_row<I, kAlign>(u, f, range, cap)
  QSTEP(0); QSTEP(1); QSTEP(2); QSTEP(3); QSTEP(4); QSTEP(5); QSTEP(6); QSTEP(7);
  QSTEP(8); QSTEP(9); QSTEP(10); QSTEP(11); QSTEP(12); QSTEP(13); QSTEP(14); QSTEP(15);
#undef QSTEP
}

template <bool kAlign>
__device__ __forceinline__ void quad_rng_steps(const QuadEncInfo &f, uint32_t &range, QuadRngCap &cap) {
#define QSTEP(I) quad_rng_step<I, kAlign>(f, range, cap)
  QSTEP(0); QSTEP(1); QSTEP(2); QSTEP(3); QSTEP(4); QSTEP(5); QSTEP(6); QSTEP(7);
  QSTEP(8); QSTEP(9); QSTEP(10); QSTEP(11); QSTEP(12); QSTEP(13); QSTEP(14); QSTEP(15);
#undef QSTEP
}

__device__ __forceinline__ uint64_t shl64(uint32_t lo, uint32_t hi, uint32_t n) { return (((uint64_t)hi << 32) | lo) << n; }

// The code value of one step for the four rows (see above), in two parts.  Part 1 (chain wave, which has the time):
// lane I picks the twelve bits of its bin out of the row-uniform capture registers, forms (term, shift) and combines four
// bins at a time: lanes 4m + 3 end up with (value < 2^56, shift) of bins 4m .. 4m + 3.
// this lane's twelve bits (rm | shift << 9 of bin j) out of the row-uniform capture registers
__device__ __forceinline__ uint32_t quad_cap_mine(const QuadRngCap &cap, uint32_t j) {
  const uint32_t h = j >> 1;
  const uint32_t a0 = (h & 1u) ? cap.w[1] : cap.w[0], a1 = (h & 1u) ? cap.w[3] : cap.w[2], a2 = (h & 1u) ? cap.w[5] : cap.w[4],
                 a3 = (h & 1u) ? cap.w[7] : cap.w[6];
  const uint32_t b0 = (h & 2u) ? a1 : a0, b1 = (h & 2u) ? a3 : a2;
  const uint32_t pair = (h & 4u) ? b1 : b0;
  return ((j & 1u) ? pair >> 16 : pair) & 0xfffu;
}

// (term, shift) of the lane's bin and the first two tree levels: lanes 4m + 3 end up with (value < 2^56, shift) of bins 4m .. 4m + 3
__device__ __forceinline__ void quad_low_quads12(uint32_t w12, uint32_t info, uint32_t &v0, uint32_t &v1, uint32_t &s) {
  const uint32_t rm = w12 & 0x1ffu, nb = (w12 >> 9) & 7u;
  const uint32_t lpsm = bit_mask<9>(info), pem = bit_mask<11>(info), ep = (info >> 10) & 1u;
  const uint32_t term = ((rm & lpsm) << nb) | (rm & pem);  // what the bin adds to low after its shift
  s = nb + ep;                                             // ... and the shift itself
  uint64_t v = term;
  {  // bins (2m, 2m + 1) in lane 2m + 1
    const uint32_t p0 = row_shr<1>(term), ps = row_shr<1>(s);
    v = shl64(p0, 0u, s) + v;                              // < 2^24
    s += ps;
  }
  v0 = (uint32_t)v;
  {  // four bins in lane 4m + 3
    const uint32_t p0 = row_shr<2>(v0), ps = row_shr<2>(s);
    v = shl64(p0, 0u, s) + v;                              // < 2^39
    s += ps;
  }
  v0 = (uint32_t)v;
  v1 = (uint32_t)(v >> 32);
}

__device__ __forceinline__ void quad_low_quads(const QuadRngCap &cap, uint32_t info, uint32_t j, uint32_t &v0, uint32_t &v1,
                                               uint32_t &s) {
  quad_low_quads12(quad_cap_mine(cap, j), info, v0, v1, s);
}

// Part 2 (low wave): the four quads and the row's remainder `acc` (9 + rem bits and a possible carry above) -> the five
// words of `new`, in every lane of the row, and S.
__device__ __forceinline__ uint32_t quad_low_join(uint32_t v0, uint32_t v1, uint32_t s, uint32_t acc, uint32_t j, uint32_t (&out)[5]) {
  // the remainder rides on the first quad: (acc << S_quad0) + V_quad0, acc < 2^25, S_quad0 <= 28
  {
    const uint64_t a = shl64(acc & neg_mask(j - 4u), 0u, s) + ((((uint64_t)v1) << 32) | v0);  // lanes 0..3; < 2^54
    v0 = (uint32_t)a;
    v1 = (uint32_t)(a >> 32);
  }
  // eight bins in lane 8m + 7, three words
  uint32_t v2;
  {
    const uint32_t p0 = row_shr<4>(v0), p1 = row_shr<4>(v1), ps = row_shr<4>(s);  // s <= 28 here
    const uint64_t lo = shl64(p0, p1, s), hi = shl64(p1, 0u, s);                  // bits 0..63 and 32..95 of p << s
    const uint64_t a = (lo & 0xffffffffull) + v0;
    const uint64_t b = (lo >> 32) + v1 + (a >> 32);
    v0 = (uint32_t)a;
    v1 = (uint32_t)b;
    v2 = (uint32_t)(hi >> 32) + (uint32_t)(b >> 32);
    s += ps;
  }
  // the sixteen bins in lane 15, five words
  uint32_t w0, w1, w2, w3, w4;
  {
    const uint32_t p0 = row_shr<8>(v0), p1 = row_shr<8>(v1), p2 = row_shr<8>(v2), ps = row_shr<8>(s);  // s <= 56 here
    const uint32_t wide = neg_mask(31u - s);  // ~0: shift by 32 or more
    const uint32_t y0 = p0 & ~wide, y1 = sel(wide, p0, p1), y2 = sel(wide, p1, p2), y3 = p2 & wide, b = s & 31u;
    const uint64_t f01 = shl64(y0, y1, b), f12 = shl64(y1, y2, b), f23 = shl64(y2, y3, b), f34 = shl64(y3, 0u, b);
    const uint64_t a0 = (f01 & 0xffffffffull) + v0;
    const uint64_t a1 = (f01 >> 32) + v1 + (a0 >> 32);
    const uint64_t a2 = (f12 >> 32) + v2 + (a1 >> 32);
    const uint64_t a3 = (f23 >> 32) + (a2 >> 32);
    w0 = (uint32_t)a0;
    w1 = (uint32_t)a1;
    w2 = (uint32_t)a2;
    w3 = (uint32_t)a3;
    w4 = (uint32_t)(f34 >> 32) + (uint32_t)(a3 >> 32);
    s += ps;
  }
  out[0] = row_bcast<15>(w0);
  out[1] = row_bcast<15>(w1);
  out[2] = row_bcast<15>(w2);
  out[3] = row_bcast<15>(w3);
  out[4] = row_bcast<15>(w4);
  return row_bcast<15>(s);
}

#ifndef CABAC_V6_PRIO_CTX
#define CABAC_V6_PRIO_CTX 0
#define CABAC_V6_PRIO_CHAIN 3
#define CABAC_V6_PRIO_LOW 0
#define CABAC_V6_PRIO_EMIT 0
#endif
// Four waves per unit of four substreams: context wave (v5's), chain wave (range only), low wave (the code value of a
// step -> its 16-bit units) and emit wave (v5's emission: delayed carry, 0xFFFF runs, byte stores).  Measured with the
// code value and the emission in ONE output wave, that wave became the longest (its tree and its LDS round trips in a row:
// 1.02 ms against v5's 0.98); apart, each fits beside the chain wave.  One workgroup barrier per 16-bin step; step k is
// coded by the chain wave in iteration k, turned into units in iteration k + 1 and written out in iteration k + 2.
template <int U, int kSync>
__global__ __launch_bounds__(256 * U) void encode_kernel_v6(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                            const uint16_t *__restrict__ records,
                                                            uint8_t *__restrict__ bytes,
                                                            cabac_substream_result *__restrict__ results) {
  // kSync 16-bin steps per workgroup barrier.  Measured (tools/ubench_v5.hip, per-role probes): every role waits ~130 ns
  // at each barrier on top of its own work, whoever arrives last.  With one unit per CU (small batches) two steps per
  // barrier are 8 % faster (C2 0.81 -> 0.75 ms, C3 17.2 -> 15.9 ms); with four units per CU the SIMDs are the limit and
  // one step per barrier is (C4 0.80 against 0.84 ms).  All hand-offs are rings of 2 kSync steps: in iteration i the
  // context wave prepares steps kSync (i + 1) + h, the chain wave runs kSync i + h, the low wave turns kSync (i - 1) + h
  // into units and the emit wave stores those of kSync (i - 2) + h (h < kSync).  Steps past the longest substream are
  // no-ops (no active record, no shift, no unit).
  __shared__ uint32_t ctx_all[U * kQuadSubs * kQuadCtxStride];
  constexpr uint32_t kRing = 2 * kSync;         // steps in flight between two neighbouring roles
  constexpr bool kTreeInChain = U == 4;         // where the first two levels of the code-value tree run (see the chain wave)
  __shared__ __attribute__((aligned(16))) uint32_t mail_all[U][kRing][4][64];   // context -> chain: per bin the packed word, k, 2c, the LPS mask
  __shared__ uint32_t quad_post[U][kRing][3][64];  // chain -> low: per lane (value low, value high, shift) of its four-bin segment
  __shared__ uint32_t unit_list[U][kRing][kQuadSubs][kUnitSlots];  // low -> emit: the units of a step, first with its carry
  __shared__ uint32_t unit_count[U][kRing][kQuadSubs];
  __shared__ uint32_t fin_acc[U][kQuadSubs], fin_rem[U][kQuadSubs];
  __shared__ uint32_t match_all[U][kMatchWords];   // the context waves' same-id bitmaps (quad_resolve)
  __shared__ uint2 rate_tab[512];
  __shared__ uint32_t bad_rows[U];
  __shared__ uint32_t wg_max_n;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, row = lane >> 4, j = lane & 15u;
  const uint32_t unit = wave % U, role = wave / U;  // role 0: context wave, 1: chain wave, 2: low wave, 3: emit wave
  quad_rate_tab_init(rate_tab, threadIdx.x, 256u * U);
  // the same-id lanes of the context wave through LDS (quad_resolve): C4 0.85 -> 0.81 ms, C3 16.3 -> 15.8 ms
  constexpr bool kLdsMatch = true;
  for (uint32_t k = threadIdx.x; k < U * kMatchWords; k += 256u * U) (&match_all[0][0])[k] = 0u;
  const uint32_t sub = (blockIdx.x * U + unit) * kQuadSubs + row;
  const bool live = sub < n_sub;
  const cabac_substream_desc d = desc[live ? sub : 0];
  const uint32_t n = live ? d.n_records : 0u;
  uint32_t (*mail)[4][64] = mail_all[unit];

  if (threadIdx.x == 0) wg_max_n = 0;
  __syncthreads();
  atomicMax(&wg_max_n, n);
  __syncthreads();
  const uint32_t max_n = wg_max_n;
  const uint32_t n_iters = (max_n + 16u * kSync - 1u) / (16u * kSync);
  // every role: one barrier before the loop, one per iteration, two after it

  if (role == 0) {
    // ---- context wave ------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(CABAC_V6_PRIO_CTX);
    const uint16_t *rec = records + d.rec_offset;
    uint32_t *rctx = ctx_all + (unit * kQuadSubs + row) * kQuadCtxStride;
    quad_ctx_init(rctx, d.qp, d.init_id & 3u, j);
    uint32_t bad = 0;
    const uint16_t *rec_safe = n != 0 ? rec : reinterpret_cast<const uint16_t *>(desc);
    const uint32_t last_rec = n != 0 ? n - 1u : 0u;
    auto rec_of = [&](uint32_t step) { return (uint32_t)rec_safe[min(16u * step + j, last_rec)]; };
    auto phase = [&](uint32_t step, uint32_t r) {
      const uint32_t info = quad_phase_a<kLdsMatch>(r, 16u * step + j < n, lane, row, rctx, bad, match_all[unit], rate_tab);
      uint32_t (*m)[64] = mail[step & (kRing - 1u)];
      m[0][lane] = info;
      m[1][lane] = info & 31u;
      m[2][lane] = (info >> 5) & 15u;
      m[3][lane] = bit_mask<9>(info);
    };
    // records are fetched four steps ahead: a step is shorter than a trip to HBM (measured: with one step of lead this
    // wave waited ~1 400 cycles per step for its load and was what every other wave of the unit waited for)
    uint32_t ahead[4];  // the records of the next four steps
#pragma unroll
    for (uint32_t h = 0; h < kSync; h++) phase(h, rec_of(h));
#pragma unroll
    for (uint32_t h = 0; h < 4; h++) ahead[h] = rec_of(kSync + h);
    __syncthreads();
    for (uint32_t it = 0; it < n_iters; it++) {
      const uint32_t first = kSync * (it + 1u);
      uint32_t fresh[kSync];
#pragma unroll
      for (uint32_t h = 0; h < kSync; h++) fresh[h] = rec_of(first + 4u + h);
      V5_TICK(t2);
#pragma unroll
      for (uint32_t h = 0; h < kSync; h++) phase(first + h, ahead[h]);
#pragma unroll
      for (uint32_t h = 0; h + kSync < 4; h++) ahead[h] = ahead[h + kSync];
#pragma unroll
      for (uint32_t h = 0; h < kSync; h++) ahead[4 - kSync + h] = fresh[h];
      V5_TICK(t3);
      __syncthreads();
      V5_TICK(t4);
      if (unit == 0) V5_ADD(2, t2, t3);  // phase (a)
      if (unit == 0) V5_ADD(3, t3, t4);  // waiting at the barrier
    }
    const uint64_t bad_mask = __ballot(bad != 0);
    if (lane == 0) {
      uint32_t rows = 0;
      for (uint32_t k = 0; k < 4; k++) rows |= ((bad_mask >> (16u * k)) & 0xffffull) ? (1u << k) : 0u;
      bad_rows[unit] = rows;
    }
    __syncthreads();
    __syncthreads();
  } else if (role == 1) {
    // ---- chain wave: the range recurrence and nothing else -----------------------------------------------
    __builtin_amdgcn_s_setprio(CABAC_V6_PRIO_CHAIN);
    uint32_t range = 510;  // start(), arith_codec.cpp:329-337
    __syncthreads();
    for (uint32_t it = 0; it < n_iters; it++) {
      V5_TICK(t0);
#pragma unroll
      for (uint32_t h = 0; h < kSync; h++) {
        const uint32_t slot = (kSync * it + h) & (kRing - 1u);
        const uint32_t info = mail[slot][0][lane];
        QuadEncRow u;
#pragma unroll
        for (int q = 0; q < 4; q++) {   // in bin order, so that the first four bins only wait for the first three reads
          lds_read4(&mail[slot][1][row * 16u], q, u.k);
          lds_read4(&mail[slot][2][row * 16u], q, u.c2);
          lds_read4(&mail[slot][3][row * 16u], q, u.lpsm);
        }
        QuadEncInfo f;
        f.alm = bit_mask<12>(info);
        QuadRngCap cap;
        if (__ballot(info >> 12) == 0) quad_rng_steps_row<false>(u, f, range, cap);
        else quad_rng_steps_row<true>(u, f, range, cap);
        if (kTreeInChain) {
          uint32_t q0, q1, qs;  // the first two levels of the code-value tree are done here, where there is time
          quad_low_quads(cap, info, j, q0, q1, qs);
          quad_post[unit][slot][0][lane] = q0;
          quad_post[unit][slot][1][lane] = q1;
          quad_post[unit][slot][2][lane] = qs;
        } else {                // one unit per workgroup: every wave has a SIMD to itself and this one is the longest
          quad_post[unit][slot][0][lane] = quad_cap_mine(cap, j);
          quad_post[unit][slot][1][lane] = info;
        }
      }
      V5_TICK(t1);
      __syncthreads();
      V5_TICK(t2);
      if (unit == 0) V5_ADD(4, t0, t1);   // chain + posting
      if (unit == 0) V5_ADD(5, t1, t2);   // waiting at the barrier
    }
    __syncthreads();
    __syncthreads();
  } else if (role == 2) {
    // ---- low wave: the code values of the steps of the iteration before and their whole units ------------------
    __builtin_amdgcn_s_setprio(CABAC_V6_PRIO_LOW);
    uint32_t acc = 0, rem = 0;  // row-uniform: the low 9 + rem bits of the code value (and a carry above), rem < 16
    auto list_step = [&](uint32_t step) {
      const uint32_t slot = step & (kRing - 1u);
      uint32_t w[5];
      uint32_t q0 = quad_post[unit][slot][0][lane], q1 = quad_post[unit][slot][1][lane], qs;
      if (kTreeInChain) qs = quad_post[unit][slot][2][lane];
      else quad_low_quads12(q0, q1, q0, q1, qs);
      const uint32_t s_total = quad_low_join(q0, q1, qs, acc, j, w);
      const uint32_t pendn = rem + s_total, m = pendn >> 4;  // whole units in this step: at most 7
      rem = pendn & 15u;
      const uint32_t base_off = 9u + rem;
      // unit k (k = 0 first in the stream) sits at bit base_off + 16 * (m - 1 - k); the first one with the carry above it
      const uint32_t o = base_off + 16u * ((m - 1u - min(j, m - 1u)) & 7u);
      const uint32_t wi = o >> 5;  // 0..3
      const uint32_t lo = wi >= 2u ? (wi == 3u ? w[3] : w[2]) : (wi == 1u ? w[1] : w[0]);
      const uint32_t hi = wi >= 2u ? (wi == 3u ? w[4] : w[3]) : (wi == 1u ? w[2] : w[1]);
      const uint32_t lead = (uint32_t)((((uint64_t)hi << 32) | lo) >> (o & 31u)) & (j == 0u ? 0x1ffffu : 0xffffu);
      acc = m != 0u ? w[0] & ((1u << base_off) - 1u) : w[0];
      unit_list[unit][slot][row][j < m ? j : kUnitDump] = lead;
      if (j == 0u) unit_count[unit][slot][row] = m;
    };
    __syncthreads();
    for (uint32_t it = 0; it < n_iters; it++) {
      V5_TICK(t0);
      if (it != 0) {
#pragma unroll
        for (uint32_t h = 0; h < kSync; h++) list_step(kSync * (it - 1u) + h);
      }
      V5_TICK(t1);
      __syncthreads();
      V5_TICK(t2);
      if (unit == 0) V5_ADD(1, t0, t1);   // low wave: code value + units
      if (unit == 0) V5_ADD(7, t1, t2);   // its barrier wait
    }
    if (n_iters != 0) {
#pragma unroll
      for (uint32_t h = 0; h < kSync; h++) list_step(kSync * (n_iters - 1u) + h);
    }
    if (j == 0u) {
      fin_acc[unit][row] = acc;
      fin_rem[unit][row] = rem;
    }
    __syncthreads();
    __syncthreads();
  } else {
    // ---- emit wave: the units of the steps two iterations back ------------------------------------------------
    __builtin_amdgcn_s_setprio(CABAC_V6_PRIO_EMIT);
    QuadEnc e;
    e.low = 0;
    e.range = 0;
    e.pend = 0;
    e.buf = 0;
    e.nbuf = 0;
    e.pos = 0;
    e.dst = bytes + d.byte_offset;
    e.cap = live ? d.byte_capacity : 0u;
    const bool writer = live && j == 0;
    auto emit_step = [&](uint32_t step) {
      const uint32_t slot = step & (kRing - 1u);
      const uint32_t *list = unit_list[unit][slot][row];
      QuadUnits u;
      u.m = unit_count[unit][slot][row];
      const uint32_t room = e.cap - e.pos - 2u * u.m;  // negative: the buffer would overflow
      const uint32_t rowodd = neg_mask(0u - u.m) & (neg_mask(0u - ((uint32_t)e.nbuf ^ 1u)) | neg_mask(room));
      u.odd_rows = __ballot(rowodd != 0);
      u.store_lanes = __ballot(live && j < u.m);
      quad_emit_units(e, u, j, list, writer);
    };
    __syncthreads();
    for (uint32_t it = 0; it < n_iters; it++) {
      V5_TICK(t0);
      if (it >= 2u) {
#pragma unroll
        for (uint32_t h = 0; h < kSync; h++) emit_step(kSync * (it - 2u) + h);
      }
      V5_TICK(t1);
      __syncthreads();
      V5_TICK(t2);
      if (unit == 0) V5_ADD(0, t0, t1);   // emit wave
      if (unit == 0) V5_ADD(12, t1, t2);  // its barrier wait
    }
    if (n_iters >= 2u) {  // listed in the last iteration
#pragma unroll
      for (uint32_t h = 0; h < kSync; h++) emit_step(kSync * (n_iters - 2u) + h);
    }
    __syncthreads();
    if (n_iters != 0) {   // listed after the loop
#pragma unroll
      for (uint32_t h = 0; h < kSync; h++) emit_step(kSync * (n_iters - 1u) + h);
    }
    __syncthreads();
    e.low = fin_acc[unit][row];
    e.pend = (int32_t)fin_rem[unit][row];
    const uint32_t n_bits = live ? quad_enc_finish(e, (d.init_id & CABAC_SUB_ALIGN_RBSP) != 0, writer, (d.init_id & CABAC_SUB_PROBE) != 0) : 0u;
    if (writer) {
      cabac_substream_result res;
      res.n_bits = n_bits;
      res.flags = (e.pos > e.cap ? CABAC_RES_OVERFLOW : 0u) | (((bad_rows[unit] >> row) & 1u) ? CABAC_RES_BAD_RECORD : 0u);
      results[sub] = res;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// v7: the encoder with its serial parts turned by 90 degrees.
//
// v6 runs the range recurrence of FOUR substreams per chain wave, sixteen lanes per substream all computing the same
// value, and a SIMD executes the four waves of a unit (~740 instructions per 16-bin step) — measured, v6 is bound by
// that total, not by any one wave.  What needs the lanes of a row is only the context wave's work (which earlier bins of
// the step share my context).  Everything downstream of it is a per-substream recurrence with nothing to share, so here
// it runs one substream per LANE: per workgroup of S = 4U substreams
//   U context waves   (v6's, four substreams each, lane = bin): per bin the fields of the chain and of the code value,
//                      written to LDS field by field, [substream][bin];
//   1 chain wave       lane = substream: reads its sixteen bins of each field with 16-byte reads and runs the range
//                      recurrence — 9 instructions per bin for ALL substreams of the workgroup instead of 13 per four;
//   1 output wave      lane = substream: the code value (low = (low << s) + term, 64-bit), its 16-bit units, the
//                      delayed carry and the byte stores, exactly v4's per-row code, one step behind the chain.
// One workgroup barrier per step; fields live in a ring of four steps (context waves one step ahead, output one behind).
constexpr uint32_t kV7Pad = 20;  // words per substream and field: 16 bins + 4, so that neighbouring lanes' 16-byte reads spread over the banks
enum : uint32_t { kV7K = 0, kV7C2, kV7Lpsm, kV7Lp9, kV7Ep, kV7Alm, kV7Fields };

template <bool kAlign>
__device__ __forceinline__ uint32_t lane_rng_step(uint32_t k, uint32_t c2, uint32_t lpsm, uint32_t alm, uint32_t &range) {
  const uint32_t t = (__umul24((range >> 5) & 15u, k) + c2) >> 1;  // LPS width: ((r>>5)*k>>1) + c
  const uint32_t rm = range - t;
  const uint32_t x = sel(lpsm, t, rm);                              // see quad_rng_step
  const uint32_t nb = (uint32_t)(__builtin_clz(x) - 23);
  range = x << nb;
  if (kAlign) range = sel(alm, 256u, range);
  return rm | (nb << 9);
}

template <int U>
__global__ __launch_bounds__(64 * (U + 2 + (U + 1) / 2)) void encode_kernel_v7(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                                      const uint16_t *__restrict__ records,
                                                                      uint8_t *__restrict__ bytes,
                                                                      cabac_substream_result *__restrict__ results) {
  constexpr uint32_t S = U * kQuadSubs;
  __shared__ uint32_t ctx_all[S * kQuadCtxStride];
  __shared__ __attribute__((aligned(16))) uint32_t fld[4][kV7Fields][S * kV7Pad];  // context waves -> chain / low wave
  // chain -> low: rm | shift << 9 per bin; one row per substream and one that all lanes past the last substream share (they
  // mirror it: same values) — LDS is what decides whether two workgroups fit on a CU (76 KB each)
  __shared__ __attribute__((aligned(16))) uint32_t wpost[2][(S + 1) * kV7Pad];
  __shared__ uint32_t post_lo[U][2][4 * kQuadSubs], post_hi[U][2][4 * kQuadSubs], post_pend[U][2][4 * kQuadSubs];  // low -> output (v5's posts)
  __shared__ uint32_t fin_lo[S], fin_hi[S], fin_pend[S];
  __shared__ uint32_t unit_list[U][kQuadSubs][kUnitSlots];
  __shared__ uint32_t match_all[U][kMatchWords];   // the context waves' same-id bitmaps (quad_resolve)
  __shared__ uint2 rate_tab[512];
  __shared__ uint32_t bad_rows[U];
  __shared__ uint32_t wg_max_n;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  quad_rate_tab_init(rate_tab, threadIdx.x, blockDim.x);
  // same-id lanes of the context waves through LDS (C4 0.81 -> 0.65 ms)
  constexpr bool kLdsMatch = true;
  for (uint32_t k = threadIdx.x; k < U * kMatchWords; k += blockDim.x) (&match_all[0][0])[k] = 0u;
  // Roles by wave number.  Waves are dealt to the CU's four SIMDs in turn and a SIMD's vector pipe is what bounds this
  // kernel (SQ counters: 318 vector instructions per SIMD and step x 2 ns = the step), so the roles are ordered to load
  // the four SIMDs evenly.  For U = 4, eight waves: context 0, context 1, chain, low | context 2, context 3, output 0,
  // output 1 — two context waves on SIMD 0 and on SIMD 1, chain + output on SIMD 2, low + output on SIMD 3 (~300 each;
  // with the four context waves first, SIMD 0 had context + chain = 340).  For U = 1: context, chain, low, output.
  // Context waves: lane = (row, bin) of one unit's four substreams; chain and low wave: lane = substream; output waves:
  // EIGHT lanes per substream (eight substreams = two units per wave).
  const uint32_t role_slot = U == 4 ? wave : (wave == 0 ? 0u : wave + 1u);   // U = 1: 0 context, 2 chain, 3 low, 4+ output
  const bool is_ctx = role_slot < 2u || (U == 4 && (role_slot == 4u || role_slot == 5u));
  const bool is_chain = role_slot == 2u, is_low = role_slot == 3u;
  const bool is_out = !is_ctx && !is_chain && !is_low;
  const uint32_t ctx_unit = role_slot < 2u ? role_slot : role_slot - 2u;          // context waves 0, 1 | 4, 5 -> units 0 .. 3
  const uint32_t out_index = U == 4 ? role_slot - 6u : 0u;
  // the substream of this thread
  uint32_t row, j, unit, local;
  bool in_range = true;
  if (is_ctx) {
    row = lane >> 4;
    j = lane & 15u;
    unit = ctx_unit;
    local = unit * kQuadSubs + row;
  } else if (is_out) {
    const uint32_t g = out_index * 8u + (lane >> 3);   // substream of the workgroup
    in_range = g < S;
    local = min(g, S - 1u);
    unit = local >> 2;
    row = local & 3u;
    j = lane & 7u;
  } else {
    row = lane >> 4;
    j = lane & 15u;
    unit = 0;
    in_range = lane < S;
    local = min(lane, S - 1u);
  }
  const uint32_t sub = blockIdx.x * S + local;
  const bool live = sub < n_sub && in_range;
  const cabac_substream_desc d = desc[sub < n_sub ? sub : 0];
  const uint32_t n = sub < n_sub ? d.n_records : 0u;

  if (threadIdx.x == 0) wg_max_n = 0;
  __syncthreads();
  atomicMax(&wg_max_n, n);
  __syncthreads();
  const uint32_t max_n = wg_max_n;
  const uint32_t n_steps = (max_n + 15u) >> 4;
  // every role: one barrier before the loop, one per step, two after it.  In iteration k the context waves prepare step
  // k + 1, the chain wave runs step k, the low wave step k - 1, the output waves list the units of step k - 2 and store
  // those of step k - 3.

  if (is_ctx) {
    const uint16_t *rec = records + d.rec_offset;
    uint32_t *rctx = ctx_all + local * kQuadCtxStride;
    quad_ctx_init(rctx, d.qp, d.init_id & 3u, j);
    uint32_t bad = 0;
    const uint16_t *rec_safe = n != 0 ? rec : reinterpret_cast<const uint16_t *>(desc);
    const uint32_t last_rec = n != 0 ? n - 1u : 0u;
    const uint32_t at = local * kV7Pad + j;
    auto post = [&](uint32_t slot, const QuadFields &f) {
      fld[slot][kV7K][at] = f.k;
      fld[slot][kV7C2][at] = f.c2;
      fld[slot][kV7Lpsm][at] = f.lpsm;
      fld[slot][kV7Lp9][at] = f.lp9;
      fld[slot][kV7Ep][at] = f.ep;
      fld[slot][kV7Alm][at] = f.alm;
    };
    const uint32_t cur_rec = rec_safe[min(j, last_rec)];
    uint32_t next_rec = rec_safe[min(16u + j, last_rec)];
    uint32_t ahead1 = rec_safe[min(32u + j, last_rec)], ahead2 = rec_safe[min(48u + j, last_rec)], ahead3 = rec_safe[min(64u + j, last_rec)];
    post(0, quad_phase_fields<kLdsMatch>(cur_rec, j < n, lane, row, rctx, bad, match_all[unit], rate_tab));  // step 0
    __syncthreads();
    for (uint32_t k = 0; k < n_steps; k++) {
      const uint32_t base = 16u * k;
      const uint32_t r = next_rec;
      next_rec = ahead1;
      ahead1 = ahead2;
      ahead2 = ahead3;
      ahead3 = rec_safe[min(base + 80u + j, last_rec)];
      V5_TICK(t0);
      post((k + 1u) & 3u, quad_phase_fields<kLdsMatch>(r, base + 16u + j < n, lane, row, rctx, bad, match_all[unit], rate_tab));
      V5_TICK(t1);
      __syncthreads();
      V5_TICK(t2);
      if (unit == 0) V5_ADD(0, t0, t1);
      if (unit == 0) V5_ADD(1, t1, t2);
    }
    const uint64_t bad_mask = __ballot(bad != 0);
    if (lane == 0) {
      uint32_t rows = 0;
      for (uint32_t q = 0; q < 4; q++) rows |= ((bad_mask >> (16u * q)) & 0xffffull) ? (1u << q) : 0u;
      bad_rows[unit] = rows;
    }
    __syncthreads();
    __syncthreads();
  } else if (is_chain) {
    // ---- chain wave: lane = substream ------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(3);
    uint32_t range = 510;  // start(), arith_codec.cpp:329-337
    __syncthreads();
    for (uint32_t k = 0; k < n_steps; k++) {
      V5_TICK(t0);
      const uint32_t slot = k & 3u;
      uint32_t kk[16], c2[16], lm[16], w[16];
      // in bin order, so that the first four bins only wait for the first three reads
#pragma unroll
      for (int q = 0; q < 4; q++) {
        lds_read4(&fld[slot][kV7K][local * kV7Pad], q, kk);
        lds_read4(&fld[slot][kV7C2][local * kV7Pad], q, c2);
        lds_read4(&fld[slot][kV7Lpsm][local * kV7Pad], q, lm);
      }
      // always with the align() handling (one select per bin): asking whether the step has such a record would cost the
      // context waves — which everything waits for — a ballot and a store per step
      uint32_t al[16];
      lds_read16(&fld[slot][kV7Alm][local * kV7Pad], al);
#pragma unroll
      for (int i = 0; i < 16; i++) w[i] = lane_rng_step<true>(kk[i], c2[i], lm[i], al[i], range);
      uint4 *dst = reinterpret_cast<uint4 *>(&wpost[k & 1u][min(lane, S) * kV7Pad]);
#pragma unroll
      for (int i = 0; i < 4; i++) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
      V5_TICK(t1);
      __syncthreads();
      V5_TICK(t2);
      V5_ADD(2, t0, t1);
      V5_ADD(3, t1, t2);
    }
    __syncthreads();
    __syncthreads();
  } else if (is_low) {
    // ---- low wave: lane = substream, the code value one step behind the chain; every 4th bin it posts (low, pend) and
    // keeps only the pend % 16 bits that are not yet a whole unit (v5's chain wave did the same, see quad_enc_step) ----
    uint64_t low = 0;
    uint32_t pend = 0;
    const uint32_t pu = local >> 2, pr = local & 3u;
    auto low_step = [&](uint32_t k) {
      const uint32_t slot = k & 3u;
      uint32_t w[16], lp9[16], ep[16];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        lds_read4(&wpost[k & 1u][min(lane, S) * kV7Pad], q, w);
        lds_read4(&fld[slot][kV7Lp9][local * kV7Pad], q, lp9);
        lds_read4(&fld[slot][kV7Ep][local * kV7Pad], q, ep);
      }
#pragma unroll
      for (int i = 0; i < 16; i++) {
        // low = (low + (LPS ? rm : 0)) << shift for a context / terminate bin, (low << 1) + (bin ? range : 0) for a bypass
        // bin (arith_codec.cpp:389-399, :426-478); a bypass bin does not renormalise (nb = 0), so both are (rm << nb)
        const uint32_t nb = w[i] >> 9;
        const uint32_t term = (w[i] & lp9[i]) << nb;
        const uint32_t sh = nb + ep[i];
        low = (low << sh) + term;
        pend += sh;
        if ((i & 3) == 3) {
          const uint32_t c = (uint32_t)(i >> 2) * kQuadSubs + pr;
          // (no `if (lane < S)`: the lanes past the last substream mirror it — same fields, same values — and an `if`
          // is a compare the scalar unit waits for, four times per step)
          post_lo[pu][k & 1u][c] = (uint32_t)low;
          post_hi[pu][k & 1u][c] = (uint32_t)(low >> 32);
          post_pend[pu][k & 1u][c] = pend;
          // whole units, and the carry above them, now belong to the post; without a whole unit nothing is cut
          const uint32_t keep = pend & 15u;
          const uint32_t width = pend >= 16u ? 9u + keep : 63u;
          low &= ~(~0ull << width);
          pend = keep;
        }
      }
    };
    __syncthreads();
    for (uint32_t k = 0; k < n_steps; k++) {
      V5_TICK(t0);
      if (k != 0) low_step(k - 1u);
      V5_TICK(t1);
      __syncthreads();
      V5_TICK(t2);
      V5_ADD(4, t0, t1);
      V5_ADD(5, t1, t2);
    }
    if (n_steps != 0) low_step(n_steps - 1u);
    fin_lo[local] = (uint32_t)low;
    fin_hi[local] = (uint32_t)(low >> 32);
    fin_pend[local] = pend;
    __syncthreads();
    __syncthreads();
  } else {
    // ---- output waves: v5's (quad_list_units / quad_emit_units) with eight lanes per substream, two steps behind the chain ----
    QuadEnc e;
    e.low = 0;
    e.range = 0;
    e.pend = 0;
    e.buf = 0;
    e.nbuf = 0;
    e.pos = 0;
    e.dst = bytes + d.byte_offset;
    e.cap = live ? d.byte_capacity : 0u;
    const bool writer = live && j == 0;
    uint32_t *list = unit_list[unit][row];   // (lanes past the last substream of the workgroup share the last one's; they never store)
    QuadUnits units;
    units.m = 0;
    units.odd_rows = 0;
    units.store_lanes = 0;
    bool listed = false;
    auto list_step = [&](uint32_t k) {
      units = quad_list_units<true>(e, post_lo[unit][k & 1u], post_hi[unit][k & 1u], post_pend[unit][k & 1u], row, j, live, list, lane);
    };
    __syncthreads();
    for (uint32_t k = 0; k < n_steps; k++) {
      V5_TICK(t0);
      if (listed) quad_emit_units(e, units, j, list, writer);
      listed = k >= 2u;
      if (listed) list_step(k - 2u);
      V5_TICK(t1);
      __syncthreads();
      V5_TICK(t2);
      if (unit == 0) V5_ADD(6, t0, t1);
      if (unit == 0) V5_ADD(7, t1, t2);
    }
    if (listed) quad_emit_units(e, units, j, list, writer);
    if (n_steps >= 2u) {
      list_step(n_steps - 2u);
      quad_emit_units(e, units, j, list, writer);
    }
    __syncthreads();  // the low wave has posted its last step and what it still holds; bad_rows is written
    if (n_steps != 0) {
      list_step(n_steps - 1u);
      quad_emit_units(e, units, j, list, writer);
    }
    e.low = ((uint64_t)fin_hi[local] << 32) | fin_lo[local];
    e.pend = (int32_t)fin_pend[local];
    const uint32_t n_bits = live ? quad_enc_finish(e, (d.init_id & CABAC_SUB_ALIGN_RBSP) != 0, writer, (d.init_id & CABAC_SUB_PROBE) != 0) : 0u;
    if (writer) {
      cabac_substream_result res;
      res.n_bits = n_bits;
      res.flags = (e.pos > e.cap ? CABAC_RES_OVERFLOW : 0u) | (((bad_rows[unit] >> row) & 1u) ? CABAC_RES_BAD_RECORD : 0u);
      results[sub] = res;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// decode

struct QuadDec {   // row-uniform values
  uint32_t hi, lo;   // 64-bit window: value in [62:47] (see v2)
  int32_t look;
  uint32_t range;
  uint32_t rp;       // byte offset of the next unread 16-bit unit
  uint32_t q_hi, q_lo;  // the two ring dwords from the one that holds that unit (big-endian: first unit on top),
                     // fetched by the previous check
  uint32_t *ring;    // this row's input ring in LDS: 64 byte-swapped dwords (+ dword 64 = a copy of dword 0)
  uint32_t filled;   // bytes of the substream staged into the ring so far (a multiple of 64)
  uint32_t pf_data, pf_off, pf_mask;  // a 64-byte block on its way: this lane's dword, its byte offset, wanted or not
  const uint8_t *src;
  uint32_t cap;
  const uint8_t *src_safe;  // src, or any readable address for an empty substream
  uint32_t last_dword;      // offset of the last dword that holds a byte of the substream
};

// Per-lane fields of one record for the decode chain.  A lane past the end of its substream carries the
// fields of "nothing" (c2 = 0, ep = 0, ctxm = 0): t = 0, value < range keeps the comparison false, and the
// step changes nothing.
struct QuadDecInfo {
  uint32_t c2;    // 2 * constant term of the LPS width: 8 context bin, 4 terminate bin, 0 bypass / none
  uint32_t ep;    // 1 for a bypass bin
  uint32_t srmul; // -2^(22 - ep) (as a signed 24-bit factor): the bypass doubling of value (arith_codec.cpp:100-105) is done by comparing against
                  // scaledRange / 2 and shifting afterwards, together with the renormalisation
  uint32_t ctxm;  // ~0 for a context-coded bin
  uint32_t ntrm;  // 0 for a terminate bin, ~0 otherwise            (kSpecial steps only)
  uint32_t alm;   // ~0 for an align() record                        (kSpecial steps only)
  uint32_t key;   // ctxId of a context bin; a value unique in the row otherwise (so that it matches no other lane)
};

// Input.  The substream is staged into an LDS ring in 64-byte blocks — one dword per lane, byte-swapped so that
// the first of its two 16-bit units is on top — by quad_dec_stage_load / _store below (every 4th step, a block
// whenever fewer than 128 bytes are staged ahead; 4 steps consume at most 48).  A check takes its units from
// (q_hi, q_lo), two ring dwords read by the PREVIOUS check, and reads the two dwords for the next one: no global
// memory and no wait anywhere near the chain.
//
// Append one 16-bit unit to the window of every row that has fewer than 32 valid look-ahead bits: mask
// arithmetic for all rows at once, no branch (the callers branch on scalar masks computed a step earlier).
template <bool kSecond>
__device__ __forceinline__ void quad_dec_refill(QuadDec &w, uint32_t units) {  // units: the next two, first on top
  // look is 2..47 here, so "fewer than 32" is bit 5 clear
  const uint32_t take = ~(uint32_t)((int32_t)(w.look << 26) >> 31);
  const uint32_t unit = (kSecond ? (units & 0xffffu) : (units >> 16)) & take;
  const uint64_t add = (uint64_t)unit << ((31 - w.look) & 63);
  w.hi |= (uint32_t)(add >> 32);
  w.lo |= (uint32_t)add;
  w.look += (int32_t)(16u & take);
  w.rp += 2u & take;
}

// one check: up to two units per row, then the ring dwords for the next check
__device__ __forceinline__ void quad_dec_check(QuadDec &w, bool second) {
  const uint64_t q = (((uint64_t)w.q_hi << 32) | w.q_lo) << ((w.rp & 2u) << 3);  // rp odd unit: skip the first one
  const uint32_t units = (uint32_t)(q >> 32);
  quad_dec_refill<false>(w, units);
  if (second) quad_dec_refill<true>(w, units);  // look >= 2 here, so two units always reach 32
  const uint32_t at = (w.rp >> 2) & 63u;
  w.q_hi = w.ring[at];
  w.q_lo = w.ring[at + 1u];  // at == 63: dword 64 mirrors dword 0
}

// staging, part 1 (a step whose number is 0 mod 4): request the next block (one dword per lane of the substream: 64 bytes
// with 16 lanes, 16 with 4 — four steps of L bins consume at most 3 L bytes) if it is wanted
template <int L = 16>
__device__ __forceinline__ void quad_dec_stage_load(QuadDec &w, uint32_t j) {
  constexpr uint32_t kLanes = L > 16 ? 16u : (uint32_t)L;  // a block is at most 64 bytes (the ring holds 256)
  const uint32_t wanted = neg_mask(w.filled - w.rp - 128u);  // fewer than 128 bytes ahead (filled >= rp always)
  w.pf_mask = L > 16 ? wanted & neg_mask(j - kLanes) : wanted;  // (with 64 lanes per substream the first 16 fetch)
  w.pf_off = w.filled + 4u * (j & (kLanes - 1u));
  w.pf_data = *reinterpret_cast<const uint32_t *>(w.src_safe + min(w.pf_off, w.last_dword));
  w.filled += (4u * kLanes) & wanted;
}
// part 2 (the step after): into the ring; past the end of the substream the window is fed zeros
__device__ __forceinline__ void quad_dec_stage_store(QuadDec &w) {
  const uint32_t v = __builtin_bswap32(w.pf_data) & neg_mask(w.pf_off - w.cap);
  const uint32_t at = (w.pf_off >> 2) & 63u;
  w.ring[sel(w.pf_mask, at, 65u)] = v;                                  // dword 65: nobody reads it
  w.ring[sel(w.pf_mask & neg_mask(at - 1u), 64u, 65u)] = v;             // at == 0: also the mirror
}

constexpr uint32_t kRingStride = 66;  // 64 ring dwords + the mirror of dword 0 + a dump word

// One decode step for the four rows.  Written with bit masks instead of ?: on purpose: on a lone wave
// every exec-mask region hipcc builds out of a conditional costs a VALU->SALU->EXEC round trip.  The cost
// of a step is its instruction count: a wave issues one instruction per ~4.4 cycles whether or not it
// depends on the previous one (tools/ubench_ilp.hip), so nothing is gained by shortening dependency
// chains and everything by dropping instructions.
// Per-lane state: st_v, the packed context word of the lane's own record (0 for a non-context record).
// kSpecial: the 16 records contain a terminate or an align record (rare) — the common variant leaves their
// handling out.
// The row-uniform fields of the 16 records of a step, one register per bin: the lanes compute them (lane I for bin I),
// park them in LDS and every lane reads its row's sixteen back with four 16-byte reads per field — an LDS read of one
// address by all lanes of a row IS the broadcast, and it replaces one v_mov_b32_dpp per bin and field (the consumers are
// VOP3 / VOPC / SDWA encodings that cannot take a DPP operand themselves).
template <int L = 16>
struct QuadDecRow {   // L = 64: the fields of sixteen bins at a time (quad_dec_steps fetches four times a step)
  static constexpr int kN = L > 16 ? 16 : L;
  uint32_t c2[kN], ctxm[kN], srmul[kN], ep[kN], key[kN];
  // fields: this wave's parking area, [5][64] words — c2, ctxm, key, srmul, ep of the lanes' records; at: the first lane
  __device__ __forceinline__ void fetch(const uint32_t *fields, uint32_t at) {
    auto one = [&](uint32_t which, uint32_t (&dst)[kN]) {
      const uint4 *p = reinterpret_cast<const uint4 *>(fields + which * 64u + at);
#pragma unroll
      for (int q = 0; q < kN / 4; q++) {
        const uint4 v = p[q];
        dst[4 * q] = v.x;
        dst[4 * q + 1] = v.y;
        dst[4 * q + 2] = v.z;
        dst[4 * q + 3] = v.w;
      }
    };
    one(0, c2);
    one(1, ctxm);
    one(2, key);
    one(3, srmul);
    one(4, ep);
  }
};

template <int I, bool kSpecial, int L = 16>
__device__ __forceinline__ void quad_dec_step(const QuadDecInfo &f, const QuadDecRow<L> &u, uint32_t r0_v, uint32_t a_v, uint32_t &st_v,
                                              uint32_t (&bits)[2], QuadDec &w) {
  constexpr int J = I % QuadDecRow<L>::kN;  // where bin I's fields sit in u
  // Input check only every 4th bin (4 bins consume at most 24 bits): 16-bit units are appended while fewer than
  // 32 look-ahead bits are valid.  The question is asked here, the answer acted upon at the END of this step: a
  // branch right behind the compare would stall ~55 cycles, and the step in between cannot be hurt — a decision
  // only depends on the nine bits of value that are compared with the range (window bits 62..54), and the
  // look-ahead is never short of them by more than this one step's 6 bits below bit 47.
  uint64_t refill = 0, refill2 = 0;
  if ((I & 3) == 0) {
    refill = __ballot(w.look <= 31);
    refill2 = __ballot(w.look <= 21);  // 15 + the 6 bits this step can consume: a second unit may be needed
    asm volatile("" : "+s"(refill), "+s"(refill2));
  }
  // the state of this bin's context, from the lane that holds the record; state() / getLPS, contexts.cpp:939-950
  uint32_t sum;  // the two estimators added: the low half carries no rate bits here (see the kernel)
  if constexpr (L == 64) {
    // one substream per wave: the state crosses as a scalar and is consumed by the SDWA add at once; its result is a vector
    // register, so that the chain stays the vector code of the other geometries (hipcc makes scalar code of a uniform chain,
    // and that is longer: 157 instructions per four bins against 145)
    const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)st_v, I);
    asm("v_add_u32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(sum) : "s"(st));
  } else {
    const uint32_t st = group_bcast<L, I>(st_v);
    sum = (st & 0xffffu) + (st >> 16);
  }
  const uint32_t sx = (uint32_t)((int32_t)(sum << 16) >> 31);  // 0 / ~0 from the MPS bit (bit 15)
  const uint32_t k = ((sum >> 10) ^ sx) & 31u;
  const uint32_t t = (__umul24(w.range >> 5, k) + u.c2[J]) >> 1;
  const uint32_t rm = w.range - t;
  // scaledRange at the window's scale is 2^22 (2^21 for a bypass bin) * rm.  (Round 1 had the scale as a DPP operand of a
  // multiplication, because v_lshlrev_b32 with DPP on its shift-amount operand returned wrong results on gfx950 — bisected
  // with the parity tests.)
  // value - scaledRange in ONE instruction: the record's field is MINUS the scale, a signed 24-bit factor (v_mad_i32_i24)
  uint32_t e;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(e) : "v"(rm), "v"(u.srmul[J]), "v"(w.hi));
  // 0: value >= scaledRange (LPS / bin 1), ~0: MPS / bin 0.  Through asm so that hipcc sees an opaque mask: written
  // as (int)e >> 31 it turns every use back into v_cmp + v_cndmask pairs, two instructions where a v_bfi /
  // v_bitop3 on the mask is one.
  uint32_t ngem;
  asm("v_ashrrev_i32 %0, 31, %1" : "=v"(ngem) : "v"(e));
  const uint32_t bin = ~(ngem ^ sx) & 1u;                        // LPS ? !mps : mps; sx is the MPS as a mask (0 if st == 0)
  const uint32_t gc = u.ctxm[J] & ~ngem;
  // One renormalisation rule for both paths: the chosen sub-range shifted up to [256, 511].  LPS (context bins only):
  // clz(t) - 23 is getRenormBitsLPS.  Otherwise rm >= 128 (an LPS width is at most 15.5 / 32 of the range plus 4), so
  // clz(rm) - 23 is 1 iff rm < 256 — the one-bit MPS renormalisation (arith_codec.cpp:60-73) — and 0 for a bypass bin
  // (rm = range) and for the no-op steps past the end.
  const uint32_t x = sel(gc, t, rm);
  uint32_t nsh = (uint32_t)__builtin_clz(x) - 23u;
  uint32_t keep = ngem;
  if (kSpecial) {
    keep |= ~group_bcast<L, I>(f.ntrm);                               // terminate bin 1 leaves value untouched (:184-185)
    nsh &= ngem | group_bcast<L, I>(f.ntrm);                          // ... and does not renormalise
  }
  w.hi = sel(keep, w.hi, e);
  w.range = x << nsh;
  if (kSpecial) {
    w.range = sel(group_bcast<L, I>(f.alm), 256u, w.range);
    // After a terminate bin 1 nothing but finish() follows; keep range >= 256 so that the no-op steps past the
    // end of the substream (t = 0, rm = range) never look like a renormalising MPS step.
    w.range |= 256u & ~(ngem | group_bcast<L, I>(f.ntrm));
  }
  {
    // (as DPP operands of their VOP2 consumers srmul and ep would cost no move, but hipcc then pads every such consumer
    // with s_nops — it applies the DPP read-after-write hazard to all of its operands: 122 s_nops per step against 38)
    const uint32_t tot = nsh + u.ep[J];  // the renormalisation shift and the bypass bit
    const uint64_t v = (((uint64_t)w.hi << 32) | w.lo) << tot;
    w.hi = (uint32_t)(v >> 32);
    w.lo = (uint32_t)v;
    w.look -= (int32_t)tot;
  }
  bits[I >> 5] |= bin << (I & 31);
  if constexpr (L == 64) asm volatile("" : "+v"(bits[I >> 5]));  // now: left to itself hipcc keeps all 64 shifted bins and ORs them at the end
  // every lane applies the bin to its own copy of the state; the lanes of this row that hold the
  // same ctxId keep it (update(), contexts.cpp:903-913).
  // Both 15-bit estimators at once with packed 16-bit math: the halves never borrow or carry into each
  // other, and the rate bits below bit 5 ride along untouched.
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const u16x2 st2 = __builtin_bit_cast(u16x2, st_v);
  const u16x2 dlt2 = (st2 >> __builtin_bit_cast(u16x2, r0_v)) & __builtin_bit_cast(u16x2, (kMask1 << 16) | kMask0);
  const uint32_t rest = __builtin_bit_cast(uint32_t, (u16x2)(st2 - dlt2));
  uint32_t upd;  // both halves: rest + a * bin, the low half of `bin` feeding both lanes of the packed mad (op_sel_hi)
  asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(upd) : "v"(a_v), "v"(bin), "v"(rest));
  asm volatile("" : "+v"(upd));   // keep the update unconditional: hipcc would otherwise wrap it in an exec
  st_v = (f.key == u.key[J]) ? upd : st_v;
  asm volatile("" : "+v"(st_v));  // region (SALU round trip + branch per bin)
  if ((I & 3) == 0 && refill != 0) {
    quad_dec_check(w, refill2 != 0);
  }
}

template <bool kSpecial, int L = 16>
__device__ __forceinline__ void quad_dec_steps(const QuadDecInfo &f, QuadDecRow<L> &u, const uint32_t *fields, uint32_t r0_v, uint32_t a_v,
                                               uint32_t &st_v, uint32_t (&bits)[2], QuadDec &w) {
#define QSTEP(U, I) quad_dec_step<I, kSpecial, L>(f, U, r0_v, a_v, st_v, bits, w)
#define QSTEP16(U, B)                                                                                                     \
  QSTEP(U, B + 0); QSTEP(U, B + 1); QSTEP(U, B + 2); QSTEP(U, B + 3); QSTEP(U, B + 4); QSTEP(U, B + 5); QSTEP(U, B + 6);   \
  QSTEP(U, B + 7); QSTEP(U, B + 8); QSTEP(U, B + 9); QSTEP(U, B + 10); QSTEP(U, B + 11); QSTEP(U, B + 12); QSTEP(U, B + 13); \
  QSTEP(U, B + 14); QSTEP(U, B + 15)
  if constexpr (L == 64) {
    // sixty-four bins of ONE substream, the fields of sixteen at a time (a second set of eighty registers to fetch the next
    // sixteen under the chain does not fit: 256 vector registers are addressable, the rest is reached through moves)
    QSTEP16(u, 0);
    u.fetch(fields, 16u);
    QSTEP16(u, 16);
    u.fetch(fields, 32u);
    QSTEP16(u, 32);
    u.fetch(fields, 48u);
    QSTEP16(u, 48);
  } else if constexpr (L == 16) {
    QSTEP16(u, 0);
  } else {
    QSTEP(u, 0); QSTEP(u, 1); QSTEP(u, 2); QSTEP(u, 3);
  }
#undef QSTEP16
#undef QSTEP
}

// W independent waves per workgroup: with W = 4 a workgroup's waves are dealt to the CU's four SIMDs, which
// pins "one chain wave per SIMD" instead of leaving it to where the dispatcher happens to put
// single-wave workgroups (measured: the same decode kernel ran 2.06 ms or 3.3 ms depending on the geometry
// of the kernel launched before it).
// L lanes per substream, 64 / L substreams per wave, L bins per step.  L = 16 is the quad decoder; L = 4 ("hex": sixteen
// substreams per wave, the state of bin I of a step broadcast inside its quad of lanes by a DPP quad permutation) trades a
// longer chain per bin — the per-step work is spread over 4 bins instead of 16 — for four times the substreams per
// instruction: the geometry for batches that offer more than one quad wave per SIMD (launch_decode_v4).
// select (may be null): the launch runs only if *select == want — the dispatch launches both geometries behind
// decode_select_kernel, which looks at the batch on the device (the descriptors of a *_device call are not on the host).
template <int W, int L = 16>
__global__ __launch_bounds__(64 * W) void decode_kernel_v4(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                           const uint16_t *__restrict__ records,
                                                           const uint8_t *__restrict__ bytes, uint8_t *__restrict__ bins,
                                                           cabac_substream_result *__restrict__ results,
                                                           const uint32_t *__restrict__ select, uint32_t want) {
  if (select != nullptr && *select != want) return;
  // decode keeps the two window sizes of a context (they never change) apart from its state word, whose low
  // five bits are then zero: state() is one SDWA add of the two halves, without masking
  constexpr uint32_t kSubs = 64u / L;  // substreams per wave
  __shared__ uint32_t ctx_all[W * kSubs * kQuadCtxStride];
  // what a record id means for the chain, looked up instead of computed (one 32-byte entry per id, the same for every
  // substream: the window sizes of a context do not depend on QP or slice type): {c2, srmul, ctxm, ep | r0_v, a_v, ntrm, alm}
  __shared__ __attribute__((aligned(16))) uint32_t rec_tab[512][8];
  __shared__ uint32_t ring_all[W * kSubs * kRingStride];
  __shared__ uint32_t field_all[W][5][64];  // the record fields of a step on their way from lane I to the row (QuadDecRow)
  const uint32_t wave = threadIdx.x >> 6;
  uint32_t *ctx = ctx_all + wave * (kSubs * kQuadCtxStride);
  const uint32_t lane = threadIdx.x & 63u, row = lane / L, j = lane % L;
  const uint32_t sub = (blockIdx.x * W + wave) * kSubs + row;
  const bool live = sub < n_sub;
  const cabac_substream_desc d = desc[live ? sub : 0];
  const uint32_t n = live ? d.n_records : 0u;
  const uint16_t *rec = records + d.rec_offset;
  uint8_t *out = bins + d.rec_offset;
  uint32_t *rctx = ctx + row * kQuadCtxStride;
  {
    const int qp = d.qp < 0 ? 0 : (d.qp > 63 ? 63 : d.qp);
    const uint32_t iid = d.init_id & 3u;
    for (uint32_t k = j; k < (uint32_t)kNumCtx; k += L) {
      const uint32_t packed = ctx2_init(qp, c_init_tables[iid * kNumCtx + k], c_init_tables[3 * kNumCtx + k]);
      rctx[k] = packed & ~31u;
    }
  }
  for (uint32_t id = threadIdx.x; id < 512u; id += 64u * W) {
    const uint32_t ctxm = id < (uint32_t)kNumCtx ? ~0u : 0u;
    const uint32_t trm_m = id == CABAC_REC_TRM ? ~0u : 0u, aln_m = id == CABAC_REC_ALIGN ? ~0u : 0u;
    const uint32_t ep = id == CABAC_REC_EP ? 1u : 0u;
    uint32_t r0_v = 0, a_v = 0;
    if (ctxm) {
      const uint32_t rates = ctx2_init(0, c_init_tables[id], c_init_tables[3 * kNumCtx + id]) & 31u;
      const uint32_t r0 = (rates & 3u) + 2u, r1 = ((rates >> 2) & 7u) + 5u;
      a_v = ((0x7fffu >> r0) & kMask0) | (((0x7fffu >> r1) & kMask1) << 16);
      r0_v = r0 | (r1 << 16);  // packed shift amounts for the 2 x 16-bit update
    }
    rec_tab[id][0] = (8u & ctxm) | (4u & trm_m);  // 2 * constant term of the LPS width
    rec_tab[id][1] = 0u - (0x400000u >> ep);      // -2^(22 - ep), see quad_dec_step
    rec_tab[id][2] = ctxm;
    rec_tab[id][3] = ep;
    rec_tab[id][4] = r0_v;
    rec_tab[id][5] = a_v;
    rec_tab[id][6] = ~trm_m;
    rec_tab[id][7] = aln_m;
  }
  __syncthreads();

  QuadDec w;
  w.src = bytes + d.byte_offset;
  w.cap = live ? d.byte_capacity : 0u;
  w.src_safe = w.cap != 0 ? w.src : reinterpret_cast<const uint8_t *>(desc);
  w.last_dword = w.cap != 0 ? ((w.cap - 1u) & ~3u) : 0u;
  {
    // start(), arith_codec.cpp:60-66.  Input is read as aligned dwords that contain at least one valid byte
    // (cabac_hip.h: the bytes buffer is readable up to the next multiple of 4 past every substream).
    const uint32_t first = __builtin_bswap32(w.cap ? *reinterpret_cast<const uint32_t *>(w.src) : 0u);
    w.hi = first >> 1;
    w.lo = first << 31;
  }
  w.look = 16;
  // the first 192 bytes go into the ring at once (blocks of 4 L bytes); the window itself started from bytes 0..3
  w.ring = ring_all + (wave * kSubs + row) * kRingStride;
  constexpr uint32_t kStageLanes = L > 16 ? 16u : (uint32_t)L;  // lanes that fetch a block (quad_dec_stage_load)
  for (uint32_t blk = 0; blk < 192u / (4u * kStageLanes); blk++) {
    w.filled = 4u * kStageLanes * blk;
    w.rp = 0;  // "wanted"
    quad_dec_stage_load<L>(w, j);
    w.pf_mask = L > 16 ? neg_mask(j - kStageLanes) : ~0u;
    quad_dec_stage_store(w);
  }
  w.filled = 192;
  w.rp = 4;
  w.pf_mask = 0;
  w.q_hi = w.ring[1];
  w.q_lo = w.ring[2];
  w.range = 510;
  uint32_t bad = 0;

  // Loop bound and loads without lane conditions: an exec region or a branch on a vector compare makes the
  // scalar unit wait for the vector result (~55 cycles each, per step).  The longest row's length is made
  // scalar once; loads past the end of a row read a valid address and are ignored (`active`).
  uint32_t n_wave = n;
#pragma unroll
  for (int d = L; d < 64; d <<= 1) n_wave = max(n_wave, (uint32_t)__shfl_xor((int)n_wave, d));
  const uint32_t max_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_wave);
  const uint16_t *rec_safe = n != 0 ? rec : reinterpret_cast<const uint16_t *>(desc);
  const uint32_t last_rec = n != 0 ? n - 1u : 0u;
  // Records are requested three steps ahead.  What a step needs besides the bins — the meaning of its record ids (rec_tab),
  // the fields of the chain parked in LDS and read back per row (QuadDecRow), the choice of the step variant, the context
  // states — is prepared in two stages off the top of the step: the table rows are requested at the START of the step
  // before (they arrive during its chain), the rest at its END (the context store has just been written back).
  uint32_t rec1 = rec_safe[min(j, last_rec)], rec2 = rec_safe[min(L + j, last_rec)], rec3 = rec_safe[min(2u * L + j, last_rec)];
  uint32_t prev_bin = 0, prev_idx = ~0u;  // the bins of the previous step, not yet stored
  uint64_t prev_lanes = 0;                 // ... and the lanes that have one
  uint32_t nxt_id, nxt_actm;
  uint4 nxt_a, nxt_b;
  auto request = [&](uint32_t base) {  // stage 1 for the step at `base`: its ids and their table rows
    uint32_t r = rec1;               // loaded two steps ago
    rec1 = rec2;
    rec2 = rec3;
    rec3 = rec_safe[min(base + 3u * L + j, last_rec)];
    nxt_actm = neg_mask(base + j - n);                              // ~0: a record of this substream
    nxt_id = sel(nxt_actm, r & CABAC_REC_ID_MASK, 0x1f0u);         // past the end: an id that is nothing
    const uint4 *row4 = reinterpret_cast<const uint4 *>(rec_tab[nxt_id]);
    nxt_a = row4[0];
    nxt_b = row4[1];
  };
  uint32_t cur_id, cur_stored, cur_ctxm, cur_r0v, cur_av;
  uint64_t cur_special;
  QuadDecInfo f;
  QuadDecRow<L> u;
  auto prepare = [&]() {              // stage 2 for the step requested last
    const uint32_t id = nxt_id, ctxm = nxt_a.z;
    cur_id = id;
    cur_ctxm = ctxm;
    cur_r0v = nxt_b.x;
    cur_av = nxt_b.y;
    cur_stored = rctx[min(id, (uint32_t)kNumCtx)];                 // slot kNumCtx is the row's pad word
    f.c2 = nxt_a.x;
    f.srmul = nxt_a.y;
    f.ctxm = ctxm;
    f.ep = nxt_a.w;
    f.ntrm = nxt_b.z;
    f.alm = nxt_b.w;
    cur_special = __ballot((~f.ntrm | f.alm) != 0);
    bad |= nxt_actm & ~ctxm & neg_mask(id - CABAC_REC_ALIGN);
    f.key = sel(ctxm, id, 0x200u + j);
    // one wave writes and reads: LDS executes a wave's instructions in order, only the compiler has to keep it
    field_all[wave][0][lane] = f.c2;
    field_all[wave][1][lane] = f.ctxm;
    field_all[wave][2][lane] = f.key;
    field_all[wave][3][lane] = f.srmul;
    field_all[wave][4][lane] = f.ep;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    u.fetch(&field_all[wave][0][0], row * L);
  };
  request(0);
  prepare();
  for (uint32_t base = 0; base < max_n; base += L) {
    V5_TICK(t0);
    V5_TICK(t1);
    // The bins of the previous step are stored only now, after the wait at the end of that step: loads and stores share
    // one in-order counter, so a store issued before a wait would add its whole latency to it (the same goes for the
    // input block requested a step ago: into the ring with it before anything new is issued)
    // (L = 64: a step consumes up to 48 bytes, so a 64-byte block is requested in EVERY step and stored in the next)
    if (L == 64 ? base != 0u : (base & (3u * L)) == L) quad_dec_stage_store(w);     // steps 1, 5, 9, ...
    {  // under the lane mask asked for at the end of the step before: an `if` here is a compare the scalar unit waits for
      uint64_t saved;
      asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tglobal_store_byte %2, %3, off\n\ts_mov_b64 exec, %0"
                   : "=&s"(saved) : "s"(prev_lanes), "v"(out + min(prev_idx, last_rec)), "v"(prev_bin) : "memory");
    }
    if (L == 64 || (base & (3u * L)) == 0u) quad_dec_stage_load<L>(w, j);    // steps 0, 4, 8, ...: request a block of input
    const uint32_t id = cur_id, ctxm = cur_ctxm;
    // asked long ago, needed now (the choice of the step variant): the branch finds the answer waiting
    uint64_t special = cur_special;
    asm volatile("" : "+s"(special));
    uint32_t st_v = cur_stored & ctxm;
    const uint32_t a_v = cur_av, r0_v = cur_r0v;
    request(base + L);                                  // the ids and table rows of the next step
    uint32_t bits[2] = {0u, 0u};  // row-uniform: bit I = the bin of record base + I
    V5_TICK(t2);
    if (special == 0) quad_dec_steps<false, L>(f, u, &field_all[wave][0][0], r0_v, a_v, st_v, bits, w);
    else quad_dec_steps<true, L>(f, u, &field_all[wave][0][0], r0_v, a_v, st_v, bits, w);
    V5_TICK(t3);
    const uint32_t my_bin = ((L > 32 && j >= 32u ? bits[1] : bits[0]) >> (j & 31u)) & 1u;
    rctx[sel(ctxm, id, (uint32_t)kNumCtx)] = st_v;  // a lane without a context writes the pad word
    prev_bin = my_bin;
    prev_idx = base + j;
    prev_lanes = __ballot(prev_idx < n);
    prepare();                                      // the next step's context states and record fields
    V5_TICK(t4);
    if (wave == 0) {
      V5_ADD(8, t0, t1);   // waiting for the record
      V5_ADD(9, t1, t2);   // prologue
      V5_ADD(10, t2, t3);  // 16 chain steps incl. refills
      V5_ADD(11, t3, t4);  // epilogue
    }
  }
  if (prev_idx < n) out[prev_idx] = (uint8_t)prev_bin;

  // bits shifted so far: everything moved into the window (8 * rp) minus value (16) minus look-ahead
  const uint32_t shifts = 8u * w.rp - 16u - (uint32_t)w.look;
  const uint32_t bytes_read = 2u + (shifts >> 3);  // the reference's counters (arith_codec.cpp:257-260)
  const int32_t bits_needed = (int32_t)(shifts & 7u) - 8;
  uint32_t flags = 0;
  if (live && (d.init_id & CABAC_SUB_FINISH)) {
    uint32_t ok = 0;
    if (bytes_read <= w.cap) {
      const uint32_t last = w.src[bytes_read - 1];
      ok = ((last << (8 + bits_needed)) & 0xffu) == 0x80u;
    }
    if (!ok && bytes_read <= w.cap) flags |= CABAC_RES_BAD_STOP;  // an underrun throws before finish() is reached
  }
  if (bytes_read > w.cap) flags |= CABAC_RES_UNDERRUN;
  const uint64_t bad_mask = __ballot(bad != 0);
  if ((bad_mask >> (row * L)) & (L == 64 ? ~0ull : (1ull << (L & 63)) - 1ull)) flags |= CABAC_RES_BAD_RECORD;
  if (live && j == 0) {
    cabac_substream_result res;
    res.n_bits = 8u * bytes_read + (uint32_t)bits_needed;
    res.flags = flags;
    results[sub] = res;
  }
}

// ---------------------------------------------------------------------------------------------
// bit estimator: BitEstimator_Std, arith_codec.cpp:603-711
//
// The cost of a bin string in 1/32768 bit: a context bin costs m_binFracBits[state][bin] and updates its
// context (estFracBitsUpdate, contexts.cpp:922-925), a bypass bin 1 bit, a terminate bin a constant, and
// align() rounds the running total up to a whole bit.  There is no low / range recurrence: the only serial
// part is the context state each bin sees, which quad_resolve gives for 16 bins of four substreams at once.
// Every lane adds up its own bins; the sixteen partial sums of a row meet at the end — and at an align
// record, which needs the running total in order (rare: that step is then summed serially).
static __constant__ uint32_t c_frac_bits[512] = {CABAC_FRAC_BITS_TABLE_VALUES};

__device__ __forceinline__ uint64_t row_sum64(uint64_t v) {  // sum over the 16 lanes of a row, in every lane
  for (int d = 1; d < 16; d <<= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d);
    v += ((uint64_t)hi << 32) | lo;
  }
  return v;
}

template <int W>
__global__ __launch_bounds__(64 * W) void estimate_kernel(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                          const uint16_t *__restrict__ records,
                                                          uint64_t *__restrict__ frac_bits, uint32_t *__restrict__ flags,
                                                          const uint32_t *__restrict__ start_state,
                                                          const uint8_t *__restrict__ start_rate,
                                                          const uint32_t *__restrict__ start_set) {
  __shared__ uint32_t ctx_all[W * kQuadSubs * kQuadCtxStride];
  __shared__ uint32_t frac[512];
  __shared__ uint32_t match_all[W][kMatchWords];   // same-id lanes through LDS (quad_resolve: 347 -> 291 ns per step alone)
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, row = lane >> 4, j = lane & 15u;
  for (uint32_t k = threadIdx.x; k < W * kMatchWords; k += 64u * W) (&match_all[0][0])[k] = 0u;
  const uint32_t sub = (blockIdx.x * W + wave) * kQuadSubs + row;
  const bool live = sub < n_sub;
  const cabac_substream_desc d = desc[live ? sub : 0];
  const uint32_t n = live ? d.n_records : 0u;
  const uint16_t *rec = records + d.rec_offset;
  uint32_t *rctx = ctx_all + (wave * kQuadSubs + row) * kQuadCtxStride;
  if (start_state == nullptr) {
    quad_ctx_init(rctx, d.qp, d.init_id & 3u, j);  // reset(qp, initId), arith_codec.cpp:623-626
  } else {
    // contexts assigned from another coder's (contexts.hpp:254): set start_set[sub] of the given states, in the
    // format of cabac_hip_ctx_init_device (m_state[0] | m_state[1] << 16, m_rate = 16 * rate0 + rate1)
    const uint64_t set = live ? start_set[sub] : 0u;
    for (uint32_t k = j; k < (uint32_t)kNumCtx; k += 16) {
      const uint32_t st = start_state[set * kNumCtx + k], rt = start_rate[set * kNumCtx + k];
      rctx[k] = (st & kMask0) | (st & 0xffff0000u) | (((rt >> 4) - 2u) & 3u) | ((((rt & 15u) - 5u) & 7u) << 2);
    }
  }
  for (uint32_t k = threadIdx.x; k < 512u; k += 64u * W) frac[k] = c_frac_bits[k];
  __syncthreads();

  uint32_t n_wave = n;
  n_wave = max(n_wave, (uint32_t)__shfl_xor((int)n_wave, 16));
  n_wave = max(n_wave, (uint32_t)__shfl_xor((int)n_wave, 32));
  const uint32_t max_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_wave);
  const uint16_t *rec_safe = n != 0 ? rec : reinterpret_cast<const uint16_t *>(desc);
  const uint32_t last_rec = n != 0 ? n - 1u : 0u;
  uint32_t next_rec = rec_safe[min(j, last_rec)];
  uint32_t bad = 0;
  uint64_t acc = 0;  // this lane's share of the row's total
  for (uint32_t base = 0; base < max_n; base += 16) {
    const uint32_t r = next_rec;
    next_rec = rec_safe[min(base + 16u + j, last_rec)];
    const QuadRecord q = quad_resolve<CABAC_REC_EST_RESTART, true>(r, base + j < n, lane, row, rctx, bad, match_all[wave]);
    const bool active = base + j < n;
    const bool zero = active && q.id == CABAC_REC_EST_RESETBITS, whole = active && q.id == CABAC_REC_EST_RESTART;
    uint32_t cost = 0;
    if (q.ctxm) cost = frac[2u * ctx2_q8(q.st) + q.bin];
    if (q.epm) cost = 1u << 15;                       // estFracBitsEP, contexts.cpp:880-882
    if (q.trmm) cost = q.bin ? 0x3bfbbu : 0x0010cu;    // estFracBitsTrm, contexts.cpp:931-933
    const uint32_t special = q.alnm ? 1u : zero ? 2u : whole ? 3u : 0u;  // records that need the total in order
    if (__builtin_expect(__ballot(special != 0) != 0, 0)) {
      uint64_t total = row_sum64(acc);  // everything before this step
      for (int k = 0; k < 16; k++) {
        total += (uint32_t)__shfl((int)cost, (int)(row * 16u + k));
        const int what = __shfl((int)special, (int)(row * 16u + k));
        if (what == 1) total = (total + 0x7fffull) & ~0x7fffull;  // align(), arith_codec.cpp:679-684
        if (what == 2) total = 0;                                   // resetBits() / start(), :615, :628
        if (what == 3) total &= ~0x7fffull;                         // restart(), :619-621
      }
      acc = j == 0 ? total : 0ull;
    } else {
      acc += cost;
    }
  }
  const uint64_t total = row_sum64(acc);
  const uint64_t bad_mask = __ballot(bad != 0);
  if (live && j == 0) {
    frac_bits[sub] = total;
    if (flags) flags[sub] = ((bad_mask >> (row * 16u)) & 0xffffull) ? CABAC_RES_BAD_RECORD : 0u;
  }
}

hipError_t launch_estimate(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                           uint64_t *frac_bits, uint32_t *flags, const uint32_t *start_state, const uint8_t *start_rate,
                           const uint32_t *start_set) {
  if (n_sub == 0) return hipSuccess;
  const uint32_t waves = (n_sub + kQuadSubs - 1) / kQuadSubs;
  // four-wave workgroups pin one wave per SIMD (see decode_kernel_v4); small batches spread single waves
  if (waves >= 1024u)
    hipLaunchKernelGGL(estimate_kernel<4>, dim3((waves + 3) / 4), dim3(256), 0, st, n_sub, desc, records, frac_bits, flags,
                       start_state, start_rate, start_set);
  else
    hipLaunchKernelGGL(estimate_kernel<1>, dim3(waves), dim3(64), 0, st, n_sub, desc, records, frac_bits, flags, start_state,
                       start_rate, start_set);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
hipError_t launch_encode_v4(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                            uint8_t *bytes, cabac_substream_result *results) {
  hipLaunchKernelGGL(encode_kernel_v4, dim3((n_sub + kQuadSubs - 1) / kQuadSubs), dim3(64), 0, st, n_sub, desc, records,
                     bytes, results);
  return hipGetLastError();
}

hipError_t launch_encode_v6(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                            uint8_t *bytes, cabac_substream_result *results, uint32_t in_flight) {
  const uint32_t units = (n_sub + kQuadSubs - 1) / kQuadSubs;
  const uint32_t units_on_chip = (max(n_sub, in_flight) + kQuadSubs - 1) / kQuadSubs;
  if (units_on_chip >= 1024u) hipLaunchKernelGGL((encode_kernel_v6<4, 1>), dim3((units + 3) / 4), dim3(1024), 0, st, n_sub, desc, records, bytes, results);
  else hipLaunchKernelGGL((encode_kernel_v6<1, 2>), dim3(units), dim3(256), 0, st, n_sub, desc, records, bytes, results);
  return hipGetLastError();
}

hipError_t launch_encode_v7(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                            uint8_t *bytes, cabac_substream_result *results, uint32_t in_flight) {
  // sixteen substreams per workgroup once there are enough of them to give every CU one; four below that
  if (max(n_sub, in_flight) >= 1024u) hipLaunchKernelGGL(encode_kernel_v7<4>, dim3((n_sub + 15) / 16), dim3(512), 0, st, n_sub, desc, records, bytes, results);
  else hipLaunchKernelGGL(encode_kernel_v7<1>, dim3((n_sub + 3) / 4), dim3(256), 0, st, n_sub, desc, records, bytes, results);
  return hipGetLastError();
}

// How many substreams of the longest one's length the batch is worth: sum of n_records / max n_records.  The sixteen-per-wave
// geometry pays when the chip is offered more than two quad waves per SIMD of EQUAL work; a batch whose time is that of a
// few long substreams (BASELINE config C5: 4 096 long ones among 8 192) is better off with the shorter chain per bin of the
// quad decoder.  *select = 1: sixteen per wave, 0: four.
__global__ __launch_bounds__(1024) void decode_select_kernel(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                             uint32_t threshold, uint32_t *__restrict__ select) {
  __shared__ unsigned long long sum_w[16];
  __shared__ uint32_t max_w[16];
  unsigned long long sum = 0;
  uint32_t mx = 0;
  for (uint32_t s = threadIdx.x; s < n_sub; s += 1024u) {
    const uint32_t n = desc[s].n_records;
    sum += n;
    mx = max(mx, n);
  }
  for (int d = 1; d < 64; d <<= 1) {
    sum += __shfl_xor(sum, d);
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, d));
  }
  if ((threadIdx.x & 63u) == 0) {
    sum_w[threadIdx.x >> 6] = sum;
    max_w[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; k++) {
      sum += sum_w[k];
      mx = max(mx, max_w[k]);
    }
    *select = (mx != 0 && sum >= (unsigned long long)threshold * mx) ? 1u : 0u;
  }
}

// Between 1 024 and 3 072 substreams: is the batch's time that of at most 1 024 long substreams that come first (a share of a
// longest-first sharded batch: BASELINE config C5 on four GPUs is 1 024 long + 1 024 short ones)?  Then one substream per wave
// (*select = 2: the long ones get a SIMD each, the short ones pass through beside them), else four (0).  "Long": more than a
// sixteenth of the longest.  They must come first because workgroups are placed in order: two long waves on one SIMD would
// cost more than the geometry saves.
__global__ __launch_bounds__(1024) void decode_select_solo_kernel(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                                  uint32_t max_long, uint32_t *__restrict__ select) {
  __shared__ uint32_t red[16];
  __shared__ uint32_t longest;
  auto block_max = [&](uint32_t v) {
    for (int d = 1; d < 64; d <<= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t m = 0;
    for (int k = 0; k < 16; k++) m = max(m, red[k]);
    return m;
  };
  uint32_t mx = 0;
  for (uint32_t s = threadIdx.x; s < n_sub; s += 1024u) mx = max(mx, desc[s].n_records);
  mx = block_max(mx);
  if (threadIdx.x == 0) longest = mx;
  __syncthreads();
  const uint32_t cut = longest >> 4;
  uint32_t past = 0;  // 1 + the index of the last long substream
  for (uint32_t s = threadIdx.x; s < n_sub; s += 1024u)
    if (desc[s].n_records > cut) past = s + 1u;
  past = block_max(past);
  // all long ones among the first max_long: then there are at most max_long of them
  if (threadIdx.x == 0) *select = (longest != 0u && past <= max_long) ? 2u : 0u;
}

constexpr uint32_t kSoloUpTo = 1024;
constexpr uint32_t kSoloAskUpTo = 3072;  // ... and up to here the batch is looked at on the device (below the headline's 4 096)  // substreams in flight up to which each gets a wave (and a SIMD) of its own
constexpr uint32_t kHexFrom = 9216;  // measured (tools/batch_scaling.py): 8 192 equal substreams 2.12 ms quad / 2.23 ms hex, 12 288: 3.14 / 2.23

hipError_t launch_decode_v4(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                            const uint8_t *bytes, uint8_t *bins, cabac_substream_result *results, uint32_t in_flight, int lanes_per_sub,
                            uint32_t *select) {
  if (lanes_per_sub == 0) {  // auto
    // up to one substream per SIMD: a wave each, 64 bins a step (16 ... 1 024 C4-type substreams 1.13-1.16 ms against 1.30)
    if (max(n_sub, in_flight) <= kSoloUpTo) return launch_decode_v4(st, n_sub, desc, records, bytes, bins, results, in_flight, 64);
    if (select != nullptr && max(n_sub, in_flight) <= kSoloAskUpTo && in_flight <= n_sub) {  // (not for chunks of a bigger batch)
      hipLaunchKernelGGL(decode_select_solo_kernel, dim3(1), dim3(1024), 0, st, n_sub, desc, kSoloUpTo, select);
      hipLaunchKernelGGL((decode_kernel_v4<4, 64>), dim3((n_sub + 3u) / 4u), dim3(256), 0, st, n_sub, desc, records, bytes, bins, results, select, 2u);
      hipLaunchKernelGGL((decode_kernel_v4<4, 16>), dim3((n_sub + 15u) / 16u), dim3(256), 0, st, n_sub, desc, records, bytes, bins, results, select, 0u);
      return hipGetLastError();
    }
    if (max(n_sub, in_flight) < kHexFrom || select == nullptr) return launch_decode_v4(st, n_sub, desc, records, bytes, bins, results, in_flight, 16);
    // enough substreams for the sixteen-per-wave geometry IF they are about equally long: asked on the device, both
    // geometries launched, the one not chosen returns at once
    // (a chunk of a bigger batch in flight is judged by its share of the threshold)
    const uint32_t thr = (uint32_t)((uint64_t)kHexFrom * n_sub / max(n_sub, in_flight));
    hipLaunchKernelGGL(decode_select_kernel, dim3(1), dim3(1024), 0, st, n_sub, desc, thr, select);
    hipLaunchKernelGGL((decode_kernel_v4<4, 4>), dim3((n_sub + 63u) / 64u), dim3(256), 0, st, n_sub, desc, records, bytes, bins, results, select, 1u);
    hipLaunchKernelGGL((decode_kernel_v4<4, 16>), dim3((n_sub + 15u) / 16u), dim3(256), 0, st, n_sub, desc, records, bytes, bins, results, select, 0u);
    return hipGetLastError();
  }
  const uint32_t waves = (n_sub + kQuadSubs - 1) / kQuadSubs;
  if (lanes_per_sub == 64) {  // one substream per wave, 64 bins per step: the per-step work around the chain is spread over 64 bins
    if (max(n_sub, in_flight) >= 1024u)
      hipLaunchKernelGGL((decode_kernel_v4<4, 64>), dim3((n_sub + 3u) / 4u), dim3(256), 0, st, n_sub, desc, records, bytes, bins, results,
                         (const uint32_t *)nullptr, 0u);
    else
      hipLaunchKernelGGL((decode_kernel_v4<1, 64>), dim3(n_sub), dim3(64), 0, st, n_sub, desc, records, bytes, bins, results,
                         (const uint32_t *)nullptr, 0u);
  } else if (lanes_per_sub == 4) {  // sixteen substreams per wave, 64 per workgroup (one workgroup's LDS fills most of a CU)
    hipLaunchKernelGGL((decode_kernel_v4<4, 4>), dim3((n_sub + 63u) / 64u), dim3(256), 0, st, n_sub, desc, records, bytes, bins, results,
                       (const uint32_t *)nullptr, 0u);
  } else if ((max(n_sub, in_flight) + kQuadSubs - 1) / kQuadSubs >= 1024u) {
    hipLaunchKernelGGL(decode_kernel_v4<4>, dim3((waves + 3) / 4), dim3(256), 0, st, n_sub, desc, records, bytes, bins,
                       results, (const uint32_t *)nullptr, 0u);
  } else {
    hipLaunchKernelGGL(decode_kernel_v4<1>, dim3(waves), dim3(64), 0, st, n_sub, desc, records, bytes, bins, results,
                       (const uint32_t *)nullptr, 0u);
  }
  return hipGetLastError();
}

}  // namespace cabac
