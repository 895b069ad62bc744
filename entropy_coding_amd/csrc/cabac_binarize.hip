// Device binariser: syntax-element records -> bin records (cabac_hip.h, "Syntax-element record").
// Restates the value -> bin-string helpers of the reference:
//   BinEncIf::encodeBinsEP / encodeRemAbsEP          entropy_codec/arith_codec.cpp:401-458
//   CABACWriter::unary_max_symbol / unary_max_eqprob / exp_golomb_eqprob   cabac_writer.cpp:3072-3118
//   CABACWriter::xWriteTruncBinCode                                          cabac_writer.cpp:854-882
// Every helper emits at most one context-coded unary run or two bypass code words, so a syntax
// element is reduced to {n_bins, (code1,len1), (code2,len2) | unary run}; bins are then written by
// OUTPUT position: a workgroup scans the bin counts of a tile of 256 elements, and thread o of the
// tile's output range finds its element by binary search in LDS and extracts its bin.  Reads are 8 B
// per element and writes 2 B per bin, both coalesced: the kernel is HBM-bound by construction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cabac_hip.h"
#include "cabac_rem_abs.hpp"
#include "cabac_kernels.h"

namespace cabac {

struct SeCode {
  uint32_t n;           // number of bins
  uint32_t kind;        // CABAC_SE_*
  uint32_t code1, len1; // first bypass code word (MSB first)
  uint32_t code2, len2; // second bypass code word
  uint32_t a, b, c;     // kind-specific (ctx ids, symbol)
};

__device__ __forceinline__ uint32_t floor_log2_u32(uint32_t x) { return 31u - (uint32_t)__builtin_clz(x | 1u); }

__device__ __forceinline__ SeCode se_decode(uint32_t w0, uint32_t value) {
  SeCode s;
  s.kind = w0 & 15u;
  s.n = 0;
  s.code1 = s.len1 = s.code2 = s.len2 = 0;
  s.a = s.b = s.c = 0;
  switch (s.kind) {
  case CABAC_SE_CTX_BIN:
    s.n = 1;
    s.a = (w0 >> 4) & 0x1ffu;
    s.b = value & 1u;
    break;
  case CABAC_SE_EP_BINS:
    s.len1 = (w0 >> 4) & 63u;
    s.code1 = value;
    s.n = s.len1;
    break;
  case CABAC_SE_REM_ABS: {  // arith_codec.cpp:426-458 (the code word: host/cabac_rem_abs.hpp)
    const cabac_code::RemAbsCode c = cabac_code::rem_abs_code(value, (w0 >> 4) & 31u, (w0 >> 9) & 31u, (w0 >> 14) & 63u);
    s.len1 = c.ones + c.stop;                    // the run and its separator as one field: ones, then a 0
    s.code1 = ((1u << c.ones) - 1u) << c.stop;   // (ones + stop <= 32: 1u << 32 does not occur, the longest run is 32 - maxLog2)
    s.len2 = c.tail_bits;
    s.code2 = c.tail;
    s.n = s.len1 + s.len2;
    break;
  }
  case CABAC_SE_TRM:
    s.n = 1;
    s.b = value & 1u;
    break;
  case CABAC_SE_UNARY_MAX: {  // cabac_writer.cpp:3072-3081
    s.a = (w0 >> 4) & 0x1ffu;
    s.b = (w0 >> 13) & 0x1ffu;
    const uint32_t mx = (w0 >> 22) & 0xffu;
    s.c = value;
    s.n = value + 1 < mx ? value + 1 : mx;
    break;
  }
  case CABAC_SE_UNARY_EP: {  // cabac_writer.cpp:3083-3101
    const uint32_t mx = (w0 >> 4) & 63u;
    if (mx != 0) {
      const uint32_t ones = value;  // `symbol` ones, then a zero if symbol < maxSymbol
      const uint32_t last = mx > value ? 1u : 0u;
      s.len1 = ones + last;
      s.code1 = (ones >= 32u ? 0xffffffffu : ((1u << ones) - 1u)) << last;
      s.n = s.len1;
    }
    break;
  }
  case CABAC_SE_EXP_GOLOMB: {  // cabac_writer.cpp:3103-3118
    uint32_t count = (w0 >> 4) & 31u, symbol = value, bins = 0, nb = 0;
    while (symbol >= (1u << count)) {
      bins = (bins << 1) + 1;
      nb++;
      symbol -= 1u << count;
      count++;
    }
    s.code1 = bins << 1;
    s.len1 = nb + 1;
    s.code2 = symbol;
    s.len2 = count;
    s.n = s.len1 + s.len2;
    break;
  }
  case CABAC_SE_TRUNC_BIN: {  // cabac_writer.cpp:854-882 (g_tbMax[k] == floor(log2 k))
    const uint32_t mx = w0 >> 4;
    const uint32_t thresh = floor_log2_u32(mx), val = 1u << thresh, b = mx - val;
    if (value < val - b) {
      s.code1 = value;
      s.len1 = thresh;
    } else {
      s.code1 = value + val - b;
      s.len1 = thresh + 1;
    }
    s.n = s.len1;
    break;
  }
  case CABAC_SE_ALIGN: s.n = 1; break;
  default: break;
  }
  return s;
}

__device__ __forceinline__ uint16_t se_bin(const SeCode &s, uint32_t idx) {
  switch (s.kind) {
  case CABAC_SE_CTX_BIN: return (uint16_t)(s.a | (s.b ? CABAC_REC_BIN : 0u));
  case CABAC_SE_TRM: return (uint16_t)(CABAC_REC_TRM | (s.b ? CABAC_REC_BIN : 0u));
  case CABAC_SE_ALIGN: return (uint16_t)CABAC_REC_ALIGN;
  case CABAC_SE_UNARY_MAX:
    return (uint16_t)((idx == 0 ? s.a : s.b) | ((s.c > idx) ? CABAC_REC_BIN : 0u));
  default: {  // one or two bypass code words, MSB first
    uint32_t bit;
    if (idx < s.len1) bit = (s.code1 >> (s.len1 - 1 - idx)) & 1u;
    else bit = (s.code2 >> (s.len2 - 1 - (idx - s.len1))) & 1u;
    return (uint16_t)(CABAC_REC_EP | (bit ? CABAC_REC_BIN : 0u));
  }
  }
}

constexpr int kBzThreads = 256;

__global__ __launch_bounds__(kBzThreads) void binarize_kernel(uint32_t n_sub, const uint64_t *__restrict__ se_offset,
                                                              const uint32_t *__restrict__ se,
                                                              const uint64_t *__restrict__ rec_offset,
                                                              uint32_t *__restrict__ n_records,
                                                              uint16_t *__restrict__ records) {
  __shared__ uint32_t scan[kBzThreads + 1];  // exclusive bin offsets of the tile's elements
  __shared__ uint32_t w0s[kBzThreads], vals[kBzThreads];
  __shared__ uint32_t wave_sum[kBzThreads / 64];
  const uint32_t sub = blockIdx.x;
  if (sub >= n_sub) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t se_begin = se_offset[sub], se_end = se_offset[sub + 1];
  uint16_t *out = records ? records + rec_offset[sub] : nullptr;
  uint32_t produced = 0;  // bins written by earlier tiles (uniform)

  for (uint64_t tile = se_begin; tile < se_end; tile += kBzThreads) {
    const uint64_t e = tile + tid;
    uint32_t w0 = 0xfu, value = 0;  // kind 15: no bins
    if (e < se_end) {
      const uint2 rec = *reinterpret_cast<const uint2 *>(se + 2 * e);
      w0 = rec.x;
      value = rec.y;
    }
    const uint32_t cnt = se_decode(w0, value).n;
    // exclusive scan of cnt over the 256 threads: wave scan + 4 wave totals
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(incl, d);
      if ((int)lane >= d) incl += up;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kBzThreads / 64; k++) {
      if (k < (int)wave) wave_base += wave_sum[k];
      total += wave_sum[k];
    }
    scan[tid] = wave_base + incl - cnt;
    w0s[tid] = w0;
    vals[tid] = value;
    if (tid == 0) scan[kBzThreads] = total;
    __syncthreads();
    if (out) {
      // output-position loop: thread o finds its element (largest i with scan[i] <= o) and its bin
      for (uint32_t o = tid; o < total; o += kBzThreads) {
        uint32_t lo = 0, hi = kBzThreads;  // invariant: scan[lo] <= o < scan[hi]
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (scan[mid] <= o) lo = mid;
          else hi = mid;
        }
        const SeCode s = se_decode(w0s[lo], vals[lo]);
        out[produced + o] = se_bin(s, o - scan[lo]);
      }
    }
    produced += total;
    __syncthreads();
  }
  if (tid == 0) n_records[sub] = produced;
}

hipError_t launch_binarize(hipStream_t st, uint32_t n_sub, const uint64_t *se_offset, const uint32_t *se,
                           const uint64_t *rec_offset, uint32_t *n_records, uint16_t *records) {
  if (n_sub == 0) return hipSuccess;
  hipLaunchKernelGGL(binarize_kernel, dim3(n_sub), dim3(kBzThreads), 0, st, n_sub, se_offset, se, rec_offset, n_records,
                     records);
  return hipGetLastError();
}

}  // namespace cabac
