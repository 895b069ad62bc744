// The code word of BinEncIf::encodeRemAbsEP (reference arith_codec.cpp:426-458; its length alone: :653-677; read back by
// decodeRemAbsEP, :153-179), stated once for the host recorders, the bit estimator's recorder, the replay decoder's planner
// and the device binariser (csrc/cabac_binarize.hip).
//
// The value splits into a quotient q = value >> rice and `rice` remainder bits.  Below the cutoff the quotient is coded in
// unary (Golomb-Rice).  From the cutoff on, the unary run goes on past `cutoff` ones by one more 1 per exp-Golomb step: step
// s covers 2^s quotients and is followed by s more bits that say which of them; when the run has reached
// 32 - cutoff - maxLog2TrDynamicRange extra ones it stops growing, no separating 0 is written, and the rest of the
// quotient goes out in maxLog2TrDynamicRange bits.  Every bit is a bypass bin.
#ifndef CABAC_REM_ABS_HPP
#define CABAC_REM_ABS_HPP
#include <stdint.h>

#if defined(__HIPCC__)
#define CABAC_HD __host__ __device__
#else
#define CABAC_HD
#endif

namespace cabac_code {

struct RemAbsCode {
  uint32_t ones;       // bypass bins of value 1 the code word starts with
  uint32_t stop;       // 1: a bypass bin of value 0 follows the run
  uint32_t tail_bits;  // then tail_bits bypass bins: `tail`, most significant bit first
  uint32_t tail;
  CABAC_HD uint32_t length() const { return ones + stop + tail_bits; }
};

CABAC_HD inline RemAbsCode rem_abs_code(uint32_t value, uint32_t rice, uint32_t cutoff, uint32_t max_log2_range) {
  const uint32_t q = value >> rice, rem = value & ((1u << rice) - 1u);
  RemAbsCode c;
  if (q < cutoff) {
    c.ones = q;
    c.stop = 1;
    c.tail_bits = rice;
    c.tail = rem;
    return c;
  }
  const uint32_t longest = 32u - cutoff - max_log2_range;  // exp-Golomb steps the run may take
  uint32_t left = q - cutoff, step = 0;
  while (step < longest && left >= (1u << step)) left -= 1u << step++;
  c.ones = cutoff + step;
  c.stop = step < longest ? 1u : 0u;
  c.tail_bits = step < longest ? step + rice : max_log2_range;
  c.tail = (left << rice) | rem;
  return c;
}

}  // namespace cabac_code
#endif
