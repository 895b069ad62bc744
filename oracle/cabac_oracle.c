/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See cabac_oracle.h for the status header.
 *
 * Plain-C restatement of the reference's CABAC bin codec (VVC clause 9.3.4), written in this
 * repository's own idiom (flat POD state, SoA context store, one bit-packer).  Every function
 * cites the reference lines it follows; paths are relative to /root/reference/src.
 */
#include "cabac_oracle.h"

#include <stdlib.h>
#include <string.h>

#include "cabac_ctx_tables.h"
#include "cabac_hip.h"

/* ---------------------------------------------------------------- tables */
static const uint8_t k_init_flat[CABAC_CTX_TABLE_ROWS * CABAC_CTX_TABLE_COLS] = {CABAC_CTX_INIT_TABLE_VALUES};
#define k_init(row, k) k_init_flat[(row) * CABAC_CTX_TABLE_COLS + (k)]

/* common/contexts.hpp:12-21 */
enum { MASK_0 = 0x7FE0, MASK_1 = 0x7FFE };

/* ---------------------------------------------------------------- context model */
typedef struct {
  uint16_t s0[ORC_NUM_CTX];
  uint16_t s1[ORC_NUM_CTX];
  uint8_t rate[ORC_NUM_CTX];
} ctx_store;

/* BinProbModel_Std::init, common/contexts.cpp:893-901;
 * setLog2WindowSize, :915-920; CtxStore::init (QP clip 0..63), :996-1015 */
static void ctx_store_init(ctx_store *c, int qp, int init_id) {
  if (qp < 0) qp = 0;
  if (qp > 63) qp = 63;
  for (int k = 0; k < ORC_NUM_CTX; k++) {
    int iv = k_init(init_id, k);
    int slope = (iv >> 3) - 4;
    int offset = (iv & 7) * 18 + 1;
    int st = ((slope * (qp - 16)) >> 1) + offset;
    if (st < 1) st = 1;
    if (st > 127) st = 127;
    int p1 = st << 8;
    c->s0[k] = (uint16_t)(p1 & MASK_0);
    c->s1[k] = (uint16_t)(p1 & MASK_1);
    int w = k_init(3, k);
    int r0 = 2 + ((w >> 2) & 3);
    int r1 = 3 + r0 + (w & 3);
    c->rate[k] = (uint8_t)(16 * r0 + r1);
  }
}

/* state(): contexts.cpp:939-941 — 8-bit truncation is semantic */
static inline unsigned ctx_state8(const ctx_store *c, unsigned k) {
  return ((unsigned)(c->s0[k] + c->s1[k]) >> 8) & 0xff;
}

/* getLPS(range): contexts.cpp:945-950 (returns uint8_t) */
static inline unsigned ctx_lps(unsigned state8, unsigned range) {
  unsigned q = state8;
  if (q & 0x80) q ^= 0xff;
  return ((((q >> 2) * (range >> 5)) >> 1) + 4) & 0xff;
}

/* update(bin): contexts.cpp:903-913 */
static inline void ctx_update(ctx_store *c, unsigned k, unsigned bin) {
  int r0 = c->rate[k] >> 4, r1 = c->rate[k] & 15;
  uint16_t a = c->s0[k], b = c->s1[k];
  a = (uint16_t)(a - ((a >> r0) & MASK_0));
  b = (uint16_t)(b - ((b >> r1) & MASK_1));
  if (bin) {
    a = (uint16_t)(a + ((0x7fffu >> r0) & MASK_0));
    b = (uint16_t)(b + ((0x7fffu >> r1) & MASK_1));
  }
  c->s0[k] = a;
  c->s1[k] = b;
}

/* getRenormBitsLPS: contexts.cpp:952-954 with m_RenormTable_32 (:787-789) */
static const uint8_t k_renorm[32] = {6, 5, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2,
                                     1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

void orc_ctx_init(int qp, int init_id, uint16_t *s0, uint16_t *s1, uint8_t *rate) {
  ctx_store c;
  ctx_store_init(&c, qp, init_id);
  memcpy(s0, c.s0, sizeof c.s0);
  memcpy(s1, c.s1, sizeof c.s1);
  memcpy(rate, c.rate, sizeof c.rate);
}

void orc_ctx_trace(int qp, int init_id, int ctx_id, const uint8_t *bins, int n, unsigned range,
                   uint8_t *state8, uint8_t *lps, uint16_t *s0_after, uint16_t *s1_after) {
  ctx_store c;
  ctx_store_init(&c, qp, init_id);
  for (int i = 0; i < n; i++) {
    unsigned q = ctx_state8(&c, (unsigned)ctx_id);
    state8[i] = (uint8_t)q;
    lps[i] = (uint8_t)ctx_lps(q, range);
    ctx_update(&c, (unsigned)ctx_id, bins[i]);
    s0_after[i] = c.s0[ctx_id];
    s1_after[i] = c.s1[ctx_id];
  }
}

/* ---------------------------------------------------------------- output bit packer
 * Observable behaviour of OutputBitstream::write / writeAlignZero / writeByteAlignment
 * (common/bit_stream.cpp:70-117, :125-132, :152-155): MSB-first packing, whole bytes to
 * the FIFO, < 8 bits held MSB-aligned. */
typedef struct {
  uint8_t *buf;
  long cap, n;    /* whole bytes */
  unsigned held;  /* MSB-aligned held bits */
  unsigned nheld; /* 0..7 */
  int overflow;
} bit_sink;

static void sink_put(bit_sink *s, uint32_t bits, unsigned nbits) {
  for (int i = (int)nbits - 1; i >= 0; i--) {
    s->held |= ((bits >> i) & 1u) << (7 - s->nheld);
    if (++s->nheld == 8) {
      if (s->n < s->cap) s->buf[s->n] = (uint8_t)s->held;
      else s->overflow = 1;
      s->n++;
      s->held = 0;
      s->nheld = 0;
    }
  }
}

static inline void sink_byte(bit_sink *s, unsigned b) { sink_put(s, b & 0xff, 8); }

/* ---------------------------------------------------------------- bin encoder */
typedef struct {
  ctx_store ctx;
  uint32_t low, range, buffered_byte;
  int32_t num_buffered, bits_left;
  uint32_t n_ctx, n_ep, n_trm; /* BinCounter, entropy_codec/arith_codec.cpp:281-316 */
  bit_sink *out;
} bin_enc;

/* BinEncoderBase::start, arith_codec.cpp:329-337 */
static void enc_start(bin_enc *e) {
  e->low = 0;
  e->range = 510;
  e->buffered_byte = 0xff;
  e->num_buffered = 0;
  e->bits_left = 23;
  e->n_ctx = e->n_ep = e->n_trm = 0;
}

/* BinEncoderBase::writeOut, arith_codec.cpp:524-546 */
static void enc_write_out(bin_enc *e) {
  unsigned lead = e->low >> (24 - e->bits_left);
  e->bits_left += 8;
  e->low &= 0xffffffffu >> e->bits_left;
  if (lead == 0xff) {
    e->num_buffered++;
  } else if (e->num_buffered > 0) {
    unsigned carry = lead >> 8;
    unsigned byte = e->buffered_byte + carry;
    e->buffered_byte = lead & 0xff;
    sink_byte(e->out, byte);
    byte = (0xff + carry) & 0xff;
    while (e->num_buffered > 1) {
      sink_byte(e->out, byte);
      e->num_buffered--;
    }
  } else {
    e->num_buffered = 1;
    e->buffered_byte = lead;
  }
}

/* TBinEncoder::encodeBin, arith_codec.cpp:553-582 */
static void enc_bin(bin_enc *e, unsigned bin, unsigned ctx_id) {
  e->n_ctx++;
  unsigned q = ctx_state8(&e->ctx, ctx_id);
  unsigned lps = ctx_lps(q, e->range);
  e->range -= lps;
  if (bin != (q >> 7)) {
    int nb = k_renorm[lps >> 3];
    e->bits_left -= nb;
    e->low += e->range;
    e->low <<= nb;
    e->range = lps << nb;
    if (e->bits_left < 12) enc_write_out(e);
  } else if (e->range < 256) {
    e->bits_left -= 1;
    e->low <<= 1;
    e->range <<= 1;
    if (e->bits_left < 12) enc_write_out(e);
  }
  ctx_update(&e->ctx, ctx_id, bin);
}

/* encodeBinEP, arith_codec.cpp:389-399 */
static void enc_ep(bin_enc *e, unsigned bin) {
  e->n_ep++;
  e->low <<= 1;
  if (bin) e->low += e->range;
  e->bits_left--;
  if (e->bits_left < 12) enc_write_out(e);
}

/* encodeBinsEP incl. the range==256 branch (encodeAlignedBinsEP), arith_codec.cpp:401-424,
 * :491-522.  Follows the reference's 8-at-a-time arithmetic literally (so that a `bins`
 * value with stray high bits behaves as it does there). */
static void enc_bins_ep(bin_enc *e, unsigned bins, unsigned n) {
  e->n_ep += n;
  if (e->range == 256) {
    unsigned rem = n;
    while (rem > 0) {
      unsigned k = rem < 8 ? rem : 8;
      unsigned mask = (1u << k) - 1;
      unsigned nb = (bins >> (rem - k)) & mask;
      e->low = (e->low << k) + (nb << 8);
      rem -= k;
      e->bits_left -= (int)k;
      if (e->bits_left < 12) enc_write_out(e);
    }
    return;
  }
  while (n > 8) {
    n -= 8;
    unsigned pattern = bins >> n;
    e->low <<= 8;
    e->low += e->range * pattern;
    bins -= pattern << n;
    e->bits_left -= 8;
    if (e->bits_left < 12) enc_write_out(e);
  }
  e->low <<= n;
  e->low += e->range * bins;
  e->bits_left -= (int)n;
  if (e->bits_left < 12) enc_write_out(e);
}

/* encodeRemAbsEP, arith_codec.cpp:426-458 */
static void enc_rem_abs(bin_enc *e, unsigned v, unsigned rice, unsigned cutoff, int max_log2) {
  const unsigned threshold = cutoff << rice;
  if (v < threshold) {
    const unsigned mask = (1u << rice) - 1;
    const unsigned length = (v >> rice) + 1;
    enc_bins_ep(e, (1u << length) - 2, length);
    enc_bins_ep(e, v & mask, rice);
  } else {
    const unsigned max_prefix = 32 - cutoff - (unsigned)max_log2;
    unsigned prefix_len = 0, suffix_len;
    unsigned code = (v >> rice) - cutoff;
    if (code >= ((1u << max_prefix) - 1)) {
      prefix_len = max_prefix;
      suffix_len = (unsigned)max_log2;
    } else {
      while (code > ((2u << prefix_len) - 2u)) prefix_len++;
      suffix_len = prefix_len + rice + 1;
    }
    const unsigned total_prefix = prefix_len + cutoff;
    const unsigned mask = (1u << rice) - 1;
    const unsigned prefix = (1u << total_prefix) - 1;
    const unsigned suffix = ((code - ((1u << prefix_len) - 1)) << rice) | (v & mask);
    enc_bins_ep(e, prefix, total_prefix);
    enc_bins_ep(e, suffix, suffix_len);
  }
}

/* encodeBinTrm, arith_codec.cpp:460-478 */
static void enc_trm(bin_enc *e, unsigned bin) {
  e->n_trm++;
  e->range -= 2;
  if (bin) {
    e->low += e->range;
    e->low <<= 7;
    e->range = 2 << 7;
    e->bits_left -= 7;
  } else if (e->range >= 256) {
    return;
  } else {
    e->low <<= 1;
    e->range <<= 1;
    e->bits_left--;
  }
  if (e->bits_left < 12) enc_write_out(e);
}

/* finish, arith_codec.cpp:339-357 */
static void enc_finish(bin_enc *e) {
  if (e->low >> (32 - e->bits_left)) {
    sink_byte(e->out, e->buffered_byte + 1);
    while (e->num_buffered > 1) {
      sink_byte(e->out, 0x00);
      e->num_buffered--;
    }
    e->low -= 1u << (32 - e->bits_left);
  } else {
    if (e->num_buffered > 0) sink_byte(e->out, e->buffered_byte);
    while (e->num_buffered > 1) {
      sink_byte(e->out, 0xff);
      e->num_buffered--;
    }
  }
  sink_put(e->out, e->low >> 8, (unsigned)(24 - e->bits_left));
}

/* ---------------------------------------------------------------- binarisation helpers */
static unsigned floor_log2(unsigned x) {
  unsigned r = 0;
  while (x >>= 1) r++;
  return r;
}

/* CABACWriter::unary_max_symbol, entropy_codec/cabac_writer.cpp:3072-3081 */
static int bz_unary_max(bin_enc *e, unsigned symbol, unsigned c0, unsigned cn, unsigned max_symbol) {
  if (symbol > max_symbol) return -2;
  unsigned total = symbol + 1 < max_symbol ? symbol + 1 : max_symbol;
  for (unsigned k = 0; k < total; k++) enc_bin(e, symbol > k, k == 0 ? c0 : cn);
  return 0;
}

/* unary_max_eqprob, cabac_writer.cpp:3083-3101 */
static int bz_unary_ep(bin_enc *e, unsigned symbol, unsigned max_symbol) {
  if (max_symbol == 0) return 0;
  int code_last = max_symbol > symbol;
  unsigned bins = 0, n = 0;
  while (symbol--) {
    bins = (bins << 1) + 1;
    n++;
  }
  if (code_last) {
    bins <<= 1;
    n++;
  }
  if (n > 32) return -2;
  enc_bins_ep(e, bins, n);
  return 0;
}

/* exp_golomb_eqprob, cabac_writer.cpp:3103-3118 */
static void bz_exp_golomb(bin_enc *e, unsigned symbol, unsigned count) {
  unsigned bins = 0, n = 0;
  while (symbol >= (1u << count)) {
    bins = (bins << 1) + 1;
    n++;
    symbol -= 1u << count;
    count++;
  }
  bins <<= 1;
  n++;
  enc_bins_ep(e, bins, n);
  enc_bins_ep(e, symbol, count);
}

/* xWriteTruncBinCode, cabac_writer.cpp:854-882 (g_tbMax[k] == floor(log2 k), rom.hpp:43-54) */
static void bz_trunc_bin(bin_enc *e, unsigned symbol, unsigned max_symbol) {
  unsigned thresh = floor_log2(max_symbol);
  unsigned val = 1u << thresh;
  unsigned b = max_symbol - val;
  if (symbol < val - b) {
    enc_bins_ep(e, symbol, thresh);
  } else {
    symbol += val - b;
    enc_bins_ep(e, symbol, thresh + 1);
  }
}

/* ---------------------------------------------------------------- encode drivers */
static long finish_stream(bin_enc *e, bit_sink *s, int flags, uint32_t *n_bits) {
  if (flags & 4) { /* probe: BinEncoderBase::getNumWrittenBits, arith_codec.cpp:482-485; nothing is flushed */
    *n_bits = (uint32_t)(s->n * 8 + s->nheld) + 8u * (uint32_t)e->num_buffered + 23u - (uint32_t)e->bits_left;
    return 0;
  }
  if (flags & 1) enc_finish(e);
  if (flags & 2) { /* writeByteAlignment, bit_stream.cpp:152-155 */
    sink_put(s, 1, 1);
    if (s->nheld) sink_put(s, 0, 8 - s->nheld);
  }
  *n_bits = (uint32_t)(s->n * 8 + s->nheld);
  long total = s->n + (s->nheld ? 1 : 0);
  if (s->overflow || total > s->cap) return -3;
  if (s->nheld) s->buf[s->n] = (uint8_t)s->held;
  return total;
}

long orc_encode_ops(const uint32_t *ops, long n_ops, int qp, int init_id, int flags, uint8_t *out,
                    long cap, uint32_t *n_bits, uint32_t *n_bins_out) {
  bit_sink s = {out, cap, 0, 0, 0, 0};
  bin_enc e;
  e.out = &s;
  ctx_store_init(&e.ctx, qp, init_id);
  enc_start(&e);
  for (long i = 0; i < n_ops; i++) {
    const uint32_t *o = ops + 4 * i;
    int rc = 0;
    switch (o[0]) {
    case ORC_OP_ENC_BIN:
      if (o[2] >= ORC_NUM_CTX) return -2;
      enc_bin(&e, o[1], o[2]);
      break;
    case ORC_OP_ENC_EP: enc_ep(&e, o[1]); break;
    case ORC_OP_ENC_BINS_EP: enc_bins_ep(&e, o[1], o[2]); break;
    case ORC_OP_ENC_REM_ABS: enc_rem_abs(&e, o[1], o[2], o[3] & 0xff, (int)(o[3] >> 8)); break;
    case ORC_OP_ENC_TRM: enc_trm(&e, o[1]); break;
    case ORC_OP_ALIGN: e.range = 256; break; /* align(), arith_codec.cpp:480 */
    case ORC_OP_UNARY_MAX: rc = bz_unary_max(&e, o[1], o[2] & 0xffff, o[2] >> 16, o[3]); break;
    case ORC_OP_UNARY_EP: rc = bz_unary_ep(&e, o[1], o[2]); break;
    case ORC_OP_EXP_GOLOMB: bz_exp_golomb(&e, o[1], o[2]); break;
    case ORC_OP_TRUNC_BIN: bz_trunc_bin(&e, o[1], o[2]); break;
    default: return -2;
    }
    if (rc) return rc;
  }
  if (n_bins_out) {
    n_bins_out[0] = e.n_ctx;
    n_bins_out[1] = e.n_ep;
    n_bins_out[2] = e.n_trm;
  }
  return finish_stream(&e, &s, flags, n_bits);
}

static int encode_record_run(bin_enc *e, const uint16_t *rec, long n) {
  for (long i = 0; i < n; i++) {
    unsigned id = rec[i] & CABAC_REC_ID_MASK, bin = rec[i] >> 15;
    if (id < ORC_NUM_CTX) enc_bin(e, bin, id);
    else if (id == CABAC_REC_EP) enc_ep(e, bin);
    else if (id == CABAC_REC_TRM) enc_trm(e, bin);
    else if (id == CABAC_REC_ALIGN) e->range = 256;
    else return -2;
  }
  return 0;
}

long orc_encode_records(const uint16_t *rec, long n, int qp, int init_id, int flags, uint8_t *out,
                        long cap, uint32_t *n_bits) {
  bit_sink s = {out, cap, 0, 0, 0, 0};
  bin_enc e;
  e.out = &s;
  ctx_store_init(&e.ctx, qp, init_id);
  enc_start(&e);
  if (encode_record_run(&e, rec, n)) return -2;
  return finish_stream(&e, &s, flags, n_bits);
}

/* ---------------------------------------------------------------- ops -> records */
typedef struct {
  uint16_t *rec;
  long cap, n;
} rec_sink;

static void rs_put(rec_sink *r, unsigned id, unsigned bin) {
  if (r->rec && r->n < r->cap) r->rec[r->n] = (uint16_t)(id | (bin ? CABAC_REC_BIN : 0));
  r->n++;
}
static void rs_bins_ep(rec_sink *r, unsigned bins, unsigned n) {
  for (int i = (int)n - 1; i >= 0; i--) rs_put(r, CABAC_REC_EP, (bins >> i) & 1);
}

/* encodeRemAbsEP as bin records, arith_codec.cpp:426-458 */
static void rs_rem_abs(rec_sink *r, unsigned v, unsigned rice, unsigned cutoff, int max_log2) {
  if (v < (cutoff << rice)) {
    unsigned length = (v >> rice) + 1;
    rs_bins_ep(r, (1u << length) - 2, length);
    rs_bins_ep(r, v & ((1u << rice) - 1), rice);
  } else {
    unsigned max_prefix = 32 - cutoff - (unsigned)max_log2, pl = 0, sl;
    unsigned code = (v >> rice) - cutoff;
    if (code >= ((1u << max_prefix) - 1)) {
      pl = max_prefix;
      sl = (unsigned)max_log2;
    } else {
      while (code > ((2u << pl) - 2u)) pl++;
      sl = pl + rice + 1;
    }
    rs_bins_ep(r, (1u << (pl + cutoff)) - 1, pl + cutoff);
    rs_bins_ep(r, ((code - ((1u << pl) - 1)) << rice) | (v & ((1u << rice) - 1)), sl);
  }
}

long orc_ops_to_records(const uint32_t *ops, long n_ops, uint16_t *rec, long cap) {
  rec_sink r = {rec, cap, 0};
  for (long i = 0; i < n_ops; i++) {
    const uint32_t *o = ops + 4 * i;
    switch (o[0]) {
    case ORC_OP_ENC_BIN: rs_put(&r, o[2], o[1]); break;
    case ORC_OP_ENC_EP: rs_put(&r, CABAC_REC_EP, o[1]); break;
    case ORC_OP_ENC_BINS_EP: rs_bins_ep(&r, o[1], o[2]); break;
    case ORC_OP_ENC_REM_ABS: rs_rem_abs(&r, o[1], o[2], o[3] & 0xff, (int)(o[3] >> 8)); break;
    case ORC_OP_ENC_TRM: rs_put(&r, CABAC_REC_TRM, o[1]); break;
    case ORC_OP_ALIGN: rs_put(&r, CABAC_REC_ALIGN, 0); break;
    case ORC_OP_UNARY_MAX: { /* cabac_writer.cpp:3072-3081 */
      unsigned symbol = o[1], c0 = o[2] & 0xffff, cn = o[2] >> 16, mx = o[3];
      if (symbol > mx) return -2;
      unsigned total = symbol + 1 < mx ? symbol + 1 : mx;
      for (unsigned k = 0; k < total; k++) rs_put(&r, k == 0 ? c0 : cn, symbol > k);
      break;
    }
    case ORC_OP_UNARY_EP: { /* cabac_writer.cpp:3083-3101 */
      unsigned symbol = o[1], mx = o[2];
      if (mx == 0) break;
      for (unsigned k = 0; k < symbol; k++) rs_put(&r, CABAC_REC_EP, 1);
      if (mx > symbol) rs_put(&r, CABAC_REC_EP, 0);
      break;
    }
    case ORC_OP_EXP_GOLOMB: { /* cabac_writer.cpp:3103-3118 */
      unsigned symbol = o[1], count = o[2];
      while (symbol >= (1u << count)) {
        rs_put(&r, CABAC_REC_EP, 1);
        symbol -= 1u << count;
        count++;
      }
      rs_put(&r, CABAC_REC_EP, 0);
      rs_bins_ep(&r, symbol, count);
      break;
    }
    case ORC_OP_TRUNC_BIN: { /* cabac_writer.cpp:854-882 */
      unsigned symbol = o[1], mx = o[2];
      unsigned thresh = floor_log2(mx), val = 1u << thresh, b = mx - val;
      if (symbol < val - b) rs_bins_ep(&r, symbol, thresh);
      else rs_bins_ep(&r, symbol + val - b, thresh + 1);
      break;
    }
    default: return -2;
    }
  }
  return r.n;
}

/* ---------------------------------------------------------------- bin decoder */
typedef struct {
  ctx_store ctx;
  uint32_t range, value;
  int32_t bits_needed;
  const uint8_t *in;
  long n_in, idx;
  int underrun;
} bin_dec;

/* InputBitstream::readByte, common/bit_stream.cpp:268-274 (CHECK "FIFO exceeded") */
static inline unsigned dec_read_byte(bin_dec *d) {
  if (d->idx >= d->n_in) {
    d->underrun = 1;
    d->idx++;
    return 0;
  }
  return d->in[d->idx++];
}

/* BinDecoderBase::start, arith_codec.cpp:60-66 */
static void dec_start(bin_dec *d) {
  d->range = 510;
  unsigned b0 = dec_read_byte(d);
  unsigned b1 = dec_read_byte(d);
  d->value = (b0 << 8) + b1;
  d->bits_needed = -8;
}

/* TBinDecoder::decodeBin, arith_codec.cpp:242-277 */
static unsigned dec_bin(bin_dec *d, unsigned ctx_id) {
  unsigned q = ctx_state8(&d->ctx, ctx_id);
  unsigned bin = q >> 7;
  unsigned lps = ctx_lps(q, d->range);
  d->range -= lps;
  uint32_t sr = d->range << 7;
  if (d->value < sr) {
    if (d->range < 256) {
      d->range <<= 1;
      d->value <<= 1;
      d->bits_needed += 1;
      if (d->bits_needed >= 0) {
        d->value += dec_read_byte(d) << d->bits_needed;
        d->bits_needed -= 8;
      }
    }
  } else {
    bin = 1 - bin;
    int nb = k_renorm[lps >> 3];
    d->value -= sr;
    d->value <<= nb;
    d->range = lps << nb;
    d->bits_needed += nb;
    if (d->bits_needed >= 0) {
      d->value += dec_read_byte(d) << d->bits_needed;
      d->bits_needed -= 8;
    }
  }
  ctx_update(&d->ctx, ctx_id, bin);
  return bin;
}

/* decodeBinEP, arith_codec.cpp:100-114 */
static unsigned dec_ep(bin_dec *d) {
  d->value += d->value;
  if (++d->bits_needed >= 0) {
    d->value += dec_read_byte(d);
    d->bits_needed = -8;
  }
  unsigned sr = d->range << 7;
  if (d->value >= sr) {
    d->value -= sr;
    return 1;
  }
  return 0;
}

/* decodeBinsEP (+ decodeAlignedBinsEP), arith_codec.cpp:116-151, :205-235 */
static unsigned dec_bins_ep(bin_dec *d, unsigned n) {
  unsigned bins = 0;
  if (d->range == 256) {
    unsigned rem = n;
    while (rem > 0) {
      unsigned k = rem < 8 ? rem : 8;
      unsigned mask = (1u << k) - 1;
      unsigned nb = (d->value >> (15 - k)) & mask;
      bins = (bins << k) | nb;
      d->value = (d->value << k) & 0x7FFF;
      rem -= k;
      d->bits_needed += (int)k;
      if (d->bits_needed >= 0) {
        d->value |= dec_read_byte(d) << d->bits_needed;
        d->bits_needed -= 8;
      }
    }
    return bins;
  }
  unsigned rem = n;
  while (rem > 8) {
    d->value = (d->value << 8) + (dec_read_byte(d) << (8 + d->bits_needed));
    unsigned sr = d->range << 15;
    for (int i = 0; i < 8; i++) {
      bins += bins;
      sr >>= 1;
      if (d->value >= sr) {
        bins++;
        d->value -= sr;
      }
    }
    rem -= 8;
  }
  d->bits_needed += (int)rem;
  d->value <<= rem;
  if (d->bits_needed >= 0) {
    d->value += dec_read_byte(d) << d->bits_needed;
    d->bits_needed -= 8;
  }
  unsigned sr = d->range << (rem + 7);
  for (unsigned i = 0; i < rem; i++) {
    bins += bins;
    sr >>= 1;
    if (d->value >= sr) {
      bins++;
      d->value -= sr;
    }
  }
  return bins;
}

/* decodeRemAbsEP, arith_codec.cpp:153-179 */
static unsigned dec_rem_abs(bin_dec *d, unsigned rice, unsigned cutoff, int max_log2) {
  unsigned prefix = 0;
  {
    const unsigned max_prefix = 32 - (unsigned)max_log2;
    unsigned cw;
    do {
      prefix++;
      cw = dec_ep(d);
    } while (cw && prefix < max_prefix);
    prefix -= 1 - cw;
  }
  unsigned length = rice, offset;
  if (prefix < cutoff) {
    offset = prefix << rice;
  } else {
    offset = ((1u << (prefix - cutoff)) + cutoff - 1) << rice;
    length += (prefix == (32u - (unsigned)max_log2)) ? (unsigned)max_log2 - rice : prefix - cutoff;
  }
  return offset + dec_bins_ep(d, length);
}

/* decodeBinTrm, arith_codec.cpp:181-197 */
static unsigned dec_trm(bin_dec *d) {
  d->range -= 2;
  unsigned sr = d->range << 7;
  if (d->value >= sr) return 1;
  if (d->range < 256) {
    d->range += d->range;
    d->value += d->value;
    if (++d->bits_needed == 0) {
      d->value += dec_read_byte(d);
      d->bits_needed = -8;
    }
  }
  return 0;
}

/* BinDecoderBase::finish, arith_codec.cpp:68-73 (+ peekPreviousByte, bit_stream.cpp:276-279) */
static int dec_finish(const bin_dec *d) {
  if (d->idx == 0 || d->idx > d->n_in) return -5;
  unsigned last = d->in[d->idx - 1];
  if (((last << (8 + d->bits_needed)) & 0xff) != 0x80) return -5;
  return 0;
}

int orc_decode_records(const uint16_t *rec, long n, int qp, int init_id, int flags,
                       const uint8_t *in, long n_in, uint8_t *bins, uint32_t *n_bits_read) {
  bin_dec d;
  memset(&d, 0, sizeof d);
  d.in = in;
  d.n_in = n_in;
  ctx_store_init(&d.ctx, qp, init_id);
  dec_start(&d);
  for (long i = 0; i < n; i++) {
    unsigned id = rec[i] & CABAC_REC_ID_MASK;
    if (id < ORC_NUM_CTX) bins[i] = (uint8_t)dec_bin(&d, id);
    else if (id == CABAC_REC_EP) bins[i] = (uint8_t)dec_ep(&d);
    else if (id == CABAC_REC_TRM) bins[i] = (uint8_t)dec_trm(&d);
    else if (id == CABAC_REC_ALIGN) { d.range = 256; bins[i] = 0; }
    else return -2;
    if (d.underrun) return -4;
  }
  if (n_bits_read) *n_bits_read = (uint32_t)(8 * d.idx + d.bits_needed);
  if (d.underrun) return -4;
  if (flags & 1) return dec_finish(&d);
  return 0;
}

/* reader-side binarisation twins: cabac_reader.cpp:3349-3379, :1162-1186 */
static unsigned bz_read_unary_max(bin_dec *d, unsigned c0, unsigned cn, unsigned mx) {
  unsigned ones = 0;
  while (ones < mx && dec_bin(d, ones == 0 ? c0 : cn) == 1) ++ones;
  return ones;
}
static unsigned bz_read_unary_ep(bin_dec *d, unsigned mx) {
  for (unsigned k = 0; k < mx; k++)
    if (!dec_ep(d)) return k;
  return mx;
}
static unsigned bz_read_exp_golomb(bin_dec *d, unsigned count) {
  unsigned symbol = 0, bit = 1;
  while (bit) {
    bit = dec_ep(d);
    symbol += bit << count++;
  }
  if (--count) symbol += dec_bins_ep(d, count);
  return symbol;
}
static unsigned bz_read_trunc_bin(bin_dec *d, unsigned mx) {
  unsigned thresh = floor_log2(mx), val = 1u << thresh, b = mx - val;
  unsigned symbol = dec_bins_ep(d, thresh);
  if (symbol >= val - b) {
    unsigned alt = dec_ep(d);
    symbol = (symbol << 1) + alt - (val - b);
  }
  return symbol;
}

int orc_decode_ops(const uint32_t *ops, long n_ops, int qp, int init_id, int flags,
                   const uint8_t *in, long n_in, uint32_t *values) {
  bin_dec d;
  memset(&d, 0, sizeof d);
  d.in = in;
  d.n_in = n_in;
  ctx_store_init(&d.ctx, qp, init_id);
  dec_start(&d);
  for (long i = 0; i < n_ops; i++) {
    const uint32_t *o = ops + 4 * i;
    uint32_t v = 0;
    switch (o[0]) {
    case ORC_OP_ENC_BIN:
      if (o[2] >= ORC_NUM_CTX) return -2;
      v = dec_bin(&d, o[2]);
      break;
    case ORC_OP_ENC_EP: v = dec_ep(&d); break;
    case ORC_OP_ENC_BINS_EP: v = dec_bins_ep(&d, o[2]); break;
    case ORC_OP_ENC_REM_ABS: v = dec_rem_abs(&d, o[2], o[3] & 0xff, (int)(o[3] >> 8)); break;
    case ORC_OP_ENC_TRM: v = dec_trm(&d); break;
    case ORC_OP_ALIGN: d.range = 256; break;
    case ORC_OP_UNARY_MAX: v = bz_read_unary_max(&d, o[2] & 0xffff, o[2] >> 16, o[3]); break;
    case ORC_OP_UNARY_EP: v = bz_read_unary_ep(&d, o[2]); break;
    case ORC_OP_EXP_GOLOMB: v = bz_read_exp_golomb(&d, o[2]); break;
    case ORC_OP_TRUNC_BIN: v = bz_read_trunc_bin(&d, o[2]); break;
    default: return -2;
    }
    values[i] = v;
    if (d.underrun) return -4;
  }
  if (flags & 1) return dec_finish(&d);
  return 0;
}

/* ---------------------------------------------------------------- batch helpers */
void orc_encode_batch(const void *desc_, uint32_t first, uint32_t count, const uint16_t *records,
                      uint8_t *bytes, uint32_t *results) {
  const cabac_substream_desc *desc = (const cabac_substream_desc *)desc_;
  for (uint32_t s = first; s < first + count; s++) {
    const cabac_substream_desc *d = &desc[s];
    int flags = ((d->init_id & CABAC_SUB_FINISH) ? 1 : 0) | ((d->init_id & CABAC_SUB_ALIGN_RBSP) ? 2 : 0) |
                ((d->init_id & CABAC_SUB_PROBE) ? 4 : 0);
    uint32_t nbits = 0;
    long rc = orc_encode_records(records + d->rec_offset, d->n_records, d->qp, (int)(d->init_id & 3),
                                 flags, bytes + d->byte_offset, d->byte_capacity, &nbits);
    results[2 * s] = nbits;
    results[2 * s + 1] = rc == -3 ? CABAC_RES_OVERFLOW : rc == -2 ? CABAC_RES_BAD_RECORD : 0;
  }
}

void orc_decode_batch(const void *desc_, uint32_t first, uint32_t count, const uint16_t *records,
                      const uint8_t *bytes, uint8_t *bins, uint32_t *results) {
  const cabac_substream_desc *desc = (const cabac_substream_desc *)desc_;
  for (uint32_t s = first; s < first + count; s++) {
    const cabac_substream_desc *d = &desc[s];
    uint32_t nbits = 0;
    int rc = orc_decode_records(records + d->rec_offset, d->n_records, d->qp, (int)(d->init_id & 3),
                                (d->init_id & CABAC_SUB_FINISH) ? 1 : 0, bytes + d->byte_offset,
                                d->byte_capacity, bins + d->rec_offset, &nbits);
    results[2 * s] = nbits;
    results[2 * s + 1] = rc == -4   ? CABAC_RES_UNDERRUN
                         : rc == -5 ? CABAC_RES_BAD_STOP
                         : rc == -2 ? CABAC_RES_BAD_RECORD
                                    : 0;
  }
}

/* CPU baseline of bench.py, all host cores: substreams [first, first + count) are encoded and decoded back (bins
 * checked) by n_threads POSIX threads taking substreams off a shared counter.  out = {bins coded, mismatching or failed
 * substreams, sum over threads of encode ns, of decode ns}; returns the wall time in ns. */
#include <pthread.h>
#include <stdatomic.h>
#include <time.h>
typedef struct {
  const cabac_substream_desc *desc;
  const uint16_t *records;
  uint32_t first, count;
  atomic_uint next;
  atomic_ullong bins, bad, enc_ns, dec_ns;
} mt_job;

static uint64_t now_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

static void *mt_worker(void *arg) {
  mt_job *j = (mt_job *)arg;
  uint8_t *buf = NULL, *bins = NULL;
  size_t buf_cap = 0, bins_cap = 0;
  for (;;) {
    const uint32_t k = atomic_fetch_add(&j->next, 1u);
    if (k >= j->count) break;
    const cabac_substream_desc *d = &j->desc[j->first + k];
    const uint16_t *rec = j->records + d->rec_offset;
    const size_t need = (size_t)d->n_records + 64;
    if (need > buf_cap) { free(buf); buf = (uint8_t *)malloc(buf_cap = need); }
    if (need > bins_cap) { free(bins); bins = (uint8_t *)malloc(bins_cap = need); }
    uint32_t nbits = 0, nread = 0;
    const uint64_t t0 = now_ns();
    const long nb = orc_encode_records(rec, d->n_records, d->qp, (int)(d->init_id & 3), 3, buf, (long)buf_cap, &nbits);
    const uint64_t t1 = now_ns();
    int rc = nb < 0 ? -1 : orc_decode_records(rec, d->n_records, d->qp, (int)(d->init_id & 3), 1, buf, nb, bins, &nread);
    const uint64_t t2 = now_ns();
    for (uint32_t i = 0; rc == 0 && i < d->n_records; i++)
      if (bins[i] != (rec[i] >> 15)) rc = -1;
    atomic_fetch_add(&j->bins, d->n_records);
    atomic_fetch_add(&j->enc_ns, t1 - t0);
    atomic_fetch_add(&j->dec_ns, t2 - t1);
    if (rc) atomic_fetch_add(&j->bad, 1);
  }
  free(buf);
  free(bins);
  return NULL;
}

uint64_t orc_roundtrip_mt(const void *desc, uint32_t first, uint32_t count, const uint16_t *records, int n_threads,
                          uint64_t *out) {
  mt_job j;
  j.desc = (const cabac_substream_desc *)desc;
  j.records = records;
  j.first = first;
  j.count = count;
  atomic_init(&j.next, 0u);
  atomic_init(&j.bins, 0ull);
  atomic_init(&j.bad, 0ull);
  atomic_init(&j.enc_ns, 0ull);
  atomic_init(&j.dec_ns, 0ull);
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  pthread_t th[256];
  const uint64_t t0 = now_ns();
  for (int t = 1; t < n_threads; t++) pthread_create(&th[t], NULL, mt_worker, &j);
  mt_worker(&j);
  for (int t = 1; t < n_threads; t++) pthread_join(th[t], NULL);
  const uint64_t wall = now_ns() - t0;
  out[0] = atomic_load(&j.bins);
  out[1] = atomic_load(&j.bad);
  out[2] = atomic_load(&j.enc_ns);
  out[3] = atomic_load(&j.dec_ns);
  return wall;
}

/* Digests for bench.py's whole-batch hash: FNV-1a 64 over a substream's coded bytes, then over the four bytes of its bit
 * count.  orc_digest_mt: of the bytes this restatement produces (finish() + writeByteAlignment()), substreams [first, first +
 * count) on n_threads threads — the stand-in for oracle/_ref's ref_digest_mt where the reference is not built;
 * orc_digest_slots: of bytes somebody else produced (the device), substream s at bytes + desc[s].byte_offset with
 * results[s].n_bits bits.  Returns the number of substreams that failed to encode / 0. */
static uint64_t fnv_bytes(const uint8_t *p, long n, uint32_t nbits) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (long i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
  for (int i = 0; i < 4; i++) h = (h ^ ((nbits >> (8 * i)) & 0xffu)) * 0x100000001b3ull;
  return h;
}

typedef struct {
  const cabac_substream_desc *desc;
  const uint16_t *records;
  uint32_t first, count;
  atomic_uint next;
  atomic_ullong bad;
  uint64_t *digests;
} digest_job;

static void *digest_worker(void *arg) {
  digest_job *j = (digest_job *)arg;
  uint8_t *buf = NULL;
  long cap = 0;
  for (;;) {
    const uint32_t k = atomic_fetch_add(&j->next, 1u);
    if (k >= j->count) break;
    const cabac_substream_desc *d = &j->desc[j->first + k];
    if ((long)d->n_records + 64 > cap) {
      cap = (long)d->n_records + 64;
      buf = (uint8_t *)realloc(buf, (size_t)cap);
    }
    uint32_t nbits = 0;
    const long nb = orc_encode_records(j->records + d->rec_offset, d->n_records, d->qp, (int)(d->init_id & 3), 3, buf, cap, &nbits);
    j->digests[k] = fnv_bytes(buf, nb > 0 ? nb : 0, nbits);
    if (nb < 0) atomic_fetch_add(&j->bad, 1ull);
  }
  free(buf);
  return NULL;
}

uint64_t orc_digest_mt(const void *desc, uint32_t first, uint32_t count, const uint16_t *records, int n_threads, uint64_t *digests) {
  digest_job j;
  j.desc = (const cabac_substream_desc *)desc;
  j.records = records;
  j.first = first;
  j.count = count;
  j.digests = digests;
  atomic_init(&j.next, 0u);
  atomic_init(&j.bad, 0ull);
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  pthread_t th[256];
  for (int t = 1; t < n_threads; t++) pthread_create(&th[t], NULL, digest_worker, &j);
  digest_worker(&j);
  for (int t = 1; t < n_threads; t++) pthread_join(th[t], NULL);
  return atomic_load(&j.bad);
}

void orc_digest_slots(const void *desc_, const uint32_t *results, uint32_t count, const uint8_t *bytes, uint64_t *digests) {
  const cabac_substream_desc *desc = (const cabac_substream_desc *)desc_;
  for (uint32_t s = 0; s < count; s++) {
    const uint32_t nbits = results[2 * s];
    digests[s] = fnv_bytes(bytes + desc[s].byte_offset, (long)((nbits + 7) / 8), nbits);
  }
}

/* ---------------------------------------------------------------- start-code emulation count
 * OutputBitstream::countStartCodeEmulations, common/bit_stream.cpp:157-181: greedy scan for
 * 00 00 {00,01,02,03}; search_n(found, end - 1, 2, 0) keeps the zero pair inside [0, n-1); after a hit the
 * scan resumes at the third byte. */
int orc_count_emulations(const uint8_t *p, long n) {
  int cnt = 0;
  long it = 0;
  while (it < n) {
    long found = it;
    for (;;) {
      /* found = search_n(found, end - 1, 2, 0): first q in [found, n-1) with p[q] == p[q+1] == 0 and
       * q + 1 < n - 1; the end of the range (n - 1) when there is none */
      long q = found;
      const long last = n - 1;
      while (q + 1 < last && !(p[q] == 0 && p[q + 1] == 0)) q++;
      if (!(q + 1 < last)) q = last;
      found = q + 1; /* found++ : second zero byte, or end() when not found */
      if (found == n) break;
      if (p[++found] <= 3) break; /* third byte */
    }
    it = found;
    if (found != n) cnt++;
  }
  return cnt;
}

/* ---------------------------------------------------------------- bit estimator
 * BitEstimatorBase / TBitEstimator, entropy_codec/arith_codec.cpp:603-711, with
 * BinProbModel_Std::estFracBitsUpdate / estFracBits / estFracBitsTrm (common/contexts.cpp:922-937) and
 * BinProbModelBase::estFracBitsEP (:880-884): the cost of a bin string in 1/32768 bit. */
static const uint32_t k_frac_bits[512] = {CABAC_FRAC_BITS_TABLE_VALUES};
enum { ORC_SCALE_BITS = 15 };

static int estimate_record_run(ctx_store *c, const uint16_t *rec, long n, uint64_t *acc) {
  uint64_t b = *acc;
  for (long i = 0; i < n; i++) {
    const unsigned id = rec[i] & CABAC_REC_ID_MASK, bin = rec[i] >> 15;
    if (id < ORC_NUM_CTX) {
      b += k_frac_bits[2 * ctx_state8(c, id) + bin]; /* estFracBitsUpdate, contexts.cpp:922-925 */
      ctx_update(c, id, bin);
    } else if (id == CABAC_REC_EP) {
      b += 1u << ORC_SCALE_BITS; /* encodeBinEP, arith_codec.cpp:636-638 */
    } else if (id == CABAC_REC_TRM) {
      b += bin ? 0x3bfbbu : 0x0010cu; /* estFracBitsTrm, contexts.cpp:931-933 */
    } else if (id == CABAC_REC_ALIGN) {
      const uint64_t add = (1u << ORC_SCALE_BITS) - 1; /* align(), arith_codec.cpp:679-684 */
      b = (b + add) & ~add;
    } else if (id == CABAC_REC_EST_RESETBITS) {
      b = 0; /* resetBits() / start(), arith_codec.cpp:615, :628 */
    } else if (id == CABAC_REC_EST_RESTART) {
      b = (b >> ORC_SCALE_BITS) << ORC_SCALE_BITS; /* restart(), :619-621 */
    } else {
      return -2;
    }
  }
  *acc = b;
  return 0;
}

int orc_estimate_records(const uint16_t *rec, long n, int qp, int init_id, uint64_t *frac_bits) {
  ctx_store c;
  ctx_store_init(&c, qp, init_id);
  *frac_bits = 0; /* reset(), arith_codec.cpp:623-626 */
  return estimate_record_run(&c, rec, n, frac_bits);
}

/* Started from given context states (the estimator's contexts assigned from another coder's, contexts.hpp:254,
 * then resetBits()): s0 / s1 / rate as orc_ctx_init delivers them. */
int orc_estimate_records_from(const uint16_t *rec, long n, const uint16_t *s0, const uint16_t *s1,
                              const uint8_t *rate, uint64_t *frac_bits) {
  ctx_store c;
  memcpy(c.s0, s0, sizeof c.s0);
  memcpy(c.s1, s1, sizeof c.s1);
  memcpy(c.rate, rate, sizeof c.rate);
  *frac_bits = 0;
  return estimate_record_run(&c, rec, n, frac_bits);
}

/* ops: the binarisation helpers emit the same bins as for the encoder; encodeBinsEP / encodeRemAbsEP cost
 * 1 bit per bypass bin (arith_codec.cpp:640-677), which is what their expansion into EP records gives. */
int orc_estimate_ops(const uint32_t *ops, long n_ops, int qp, int init_id, uint64_t *frac_bits) {
  const long n = orc_ops_to_records(ops, n_ops, 0, 0);
  if (n < 0) return (int)n;
  uint16_t *rec = (uint16_t *)malloc((size_t)(n ? n : 1) * sizeof(uint16_t));
  if (!rec) return -1;
  orc_ops_to_records(ops, n_ops, rec, n);
  const int rc = orc_estimate_records(rec, n, qp, init_id, frac_bits);
  free(rec);
  return rc;
}

void orc_estimate_batch(const void *desc_, uint32_t first, uint32_t count, const uint16_t *records,
                        uint64_t *frac_bits, uint32_t *flags) {
  const cabac_substream_desc *desc = (const cabac_substream_desc *)desc_;
  for (uint32_t s = first; s < first + count; s++) {
    const cabac_substream_desc *d = &desc[s];
    const int rc = orc_estimate_records(records + d->rec_offset, d->n_records, d->qp, (int)(d->init_id & 3),
                                        &frac_bits[s]);
    flags[s] = rc == -2 ? CABAC_RES_BAD_RECORD : 0;
  }
}

/* ---------------------------------------------------------------- residual coding
 * Restates CABACWriter::residual_coding (cabac_writer.cpp:2424-2525), ts_flag (:2527-2534), last_sig_coeff
 * (:2639-2720), residual_coding_subblock (:2722-2872), CoeffCodingContext (context_modelling.hpp:71-244,
 * context_modelling.cpp:7-106) and the grouped diagonal scan of rom.cpp:148-260, for regular residual coding
 * without SBT/MTS zero-out and without the range extensions. */
typedef struct {
  int w, h, lw, lh;          /* block size                                              */
  int cgw_l2, cgh_l2, cg_l2; /* coefficient-group size (g_log2SbbSize, rom.cpp:41-50)   */
  int wg, hg;                /* groups per row / column after the 32x32 zero-out        */
  int n_coded;               /* wg * hg << cg_l2                                        */
} blk_geom;

static void blk_geom_init(blk_geom *g, int lw, int lh) {
  g->lw = lw; g->lh = lh; g->w = 1 << lw; g->h = 1 << lh;
  if (lw == 0)      { g->cgw_l2 = 0; g->cgh_l2 = lh < 4 ? lh : 4; }
  else if (lh == 0) { g->cgw_l2 = lw < 4 ? lw : 4; g->cgh_l2 = 0; }
  else if (lw == 1) { g->cgw_l2 = 1; g->cgh_l2 = lh <= 2 ? 1 : 3; }
  else if (lh == 1) { g->cgh_l2 = 1; g->cgw_l2 = lw <= 2 ? 1 : 3; }
  else              { g->cgw_l2 = 2; g->cgh_l2 = 2; }
  g->cg_l2 = g->cgw_l2 + g->cgh_l2;
  g->wg = (g->w < 32 ? g->w : 32) >> g->cgw_l2;
  g->hg = (g->h < 32 ? g->h : 32) >> g->cgh_l2;
  g->n_coded = (g->wg * g->hg) << g->cg_l2;
}

/* up-right diagonal scan of a bw x bh rectangle: k-th position -> (x, y) (ScanGenerator, SCAN_DIAG, rom.cpp:72-92) */
static void diag_positions(int bw, int bh, uint8_t *xs, uint8_t *ys) {
  int k = 0;
  for (int d = 0; d <= bw + bh - 2; d++) {
    int y_hi = d < bh - 1 ? d : bh - 1, y_lo = d - (bw - 1) > 0 ? d - (bw - 1) : 0;
    for (int y = y_hi; y >= y_lo; y--, k++) { xs[k] = (uint8_t)(d - y); ys[k] = (uint8_t)y; }
  }
}

/* scan position -> (x, y); positions beyond the coded region of a zeroed-out block all map to the
 * bottom-right sample (rom.cpp:218-226).  Returns the number of entries (w*h). */
long orc_scan_order(int lw, int lh, uint32_t *out) {
  blk_geom g;
  if (lw < 0 || lw > 6 || lh < 0 || lh > 6) return -1;
  blk_geom_init(&g, lw, lh);
  uint8_t gx[64], gy[64], ix[16], iy[16];
  diag_positions(g.wg, g.hg, gx, gy);
  diag_positions(1 << g.cgw_l2, 1 << g.cgh_l2, ix, iy);
  for (int p = 0; p < g.w * g.h; p++) {
    if (p < g.n_coded) {
      int cg = p >> g.cg_l2, i = p & ((1 << g.cg_l2) - 1);
      out[p] = (uint32_t)((gx[cg] << g.cgw_l2) + ix[i]) | (uint32_t)((gy[cg] << g.cgh_l2) + iy[i]) << 16;
    } else {
      out[p] = (uint32_t)(g.w - 1) | (uint32_t)(g.h - 1) << 16;
    }
  }
  return (long)g.w * g.h;
}

typedef struct {
  int sum_abs;     /* plain sum of |neighbour| (templateAbsSum)                          */
  int sum_clip;    /* sum of min(4 + (a & 1), a)          (sigCtxIdAbs)                  */
  int n_nonzero;
} tmpl_t;

static tmpl_t tmpl_at(const int32_t *c, const blk_geom *g, int x, int y) {
  static const int8_t dx[5] = {1, 2, 1, 0, 0}, dy[5] = {0, 0, 1, 1, 2};
  tmpl_t t = {0, 0, 0};
  for (int k = 0; k < 5; k++) {
    int xx = x + dx[k], yy = y + dy[k];
    if (xx >= g->w || yy >= g->h) continue;
    int a = c[yy * g->w + xx];
    if (a < 0) a = -a;
    int lim = 4 + (a & 1);
    t.sum_abs += a;
    t.sum_clip += a < lim ? a : lim;
    t.n_nonzero += a != 0;
  }
  return t;
}

static const uint8_t k_rice_of_sum[32] = {0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2,
                                          2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3}; /* g_goRiceParsCoeff */
static unsigned rice_from(int sum_abs, int base_level) { /* deriveRice, context_modelling.hpp:262-266 */
  int v = sum_abs - 5 * base_level;
  return k_rice_of_sum[v < 0 ? 0 : v > 31 ? 31 : v];
}
static unsigned last_group_idx(unsigned p) { /* g_groupIdx, rom.cpp:21-25 */
  return p < 4 ? p : 2 * floor_log2(p) + ((p >> (floor_log2(p) - 1)) & 1);
}
static unsigned last_min_in_group(unsigned gidx) { /* g_minInGroup, rom.cpp:18-19 */
  return gidx < 4 ? gidx : (2u + (gidx & 1)) << ((gidx >> 1) - 1);
}

/* Transform-skip residual coding: CABACWriter::residual_codingTS / residual_coding_subblockTS (cabac_writer.cpp:2874-3046)
 * with the TS context selection and level mapping of CoeffCodingContext (context_modelling.hpp:268-371).  Groups and
 * positions in forward scan order; neighbours are the left and the upper sample.  TSRC Rice extension off (rice = 1). */
static int ts_mod_level(int left, int above, int a, int bdpcm) { /* deriveModCoeff, context_modelling.hpp:344-364 */
  if (a == 0 || bdpcm) return a;
  if (left < 0) left = -left;
  if (above < 0) above = -above;
  const int pred = left > above ? left : above;
  return a == pred ? 1 : (a < pred ? a + 1 : a);
}

static long residual_ts(rec_sink *r, const blk_geom *g, const uint32_t *scan, const int32_t *coeff, int bdpcm,
                        int max_log2_range) {
#define SX(p) ((int)(scan[p] & 0xffff))
#define SY(p) ((int)(scan[p] >> 16))
#define AT(x, y) (coeff[(y) * g->w + (x)])
  const int n_cg = (g->w * g->h) >> g->cg_l2, cg_size = 1 << g->cg_l2;
  uint8_t cg_sig[64];
  int n_sig = 0;
  memset(cg_sig, 0, sizeof cg_sig);
  int budget = (g->w * g->h * 7) >> 2;
  for (int cg = 0; cg < n_cg; cg++) {
    const int lo = cg << g->cg_l2, hi = lo + cg_size - 1;
    const int cgx = SX(lo) >> g->cgw_l2, cgy = SY(lo) >> g->cgh_l2;
    int sig = 0;
    for (int p = lo; p <= hi; p++) sig |= AT(SX(p), SY(p)) != 0;
    if (sig) { cg_sig[cgy * g->wg + cgx] = 1; n_sig++; }
    if (cg != n_cg - 1 || n_sig - sig != 0) { /* :2933-2942: not (last group and no group before it significant) */
      const int left = cgx > 0 ? cg_sig[cgy * g->wg + cgx - 1] : 0, above = cgy > 0 ? cg_sig[(cgy - 1) * g->wg + cgx] : 0;
      rs_put(r, CABAC_CTX_TS_SIG_COEFF_GROUP + (unsigned)(left + above), (unsigned)sig);
      if (!sig) continue;
    }
    int n_nz = 0, last1 = -1, last2 = -1, p;
    for (p = lo; p <= hi && budget >= 4; p++) { /* pass 1 */
      const int x = SX(p), y = SY(p), v = AT(x, y);
      const int left = x > 0 ? AT(x - 1, y) : 0, above = y > 0 ? AT(x, y - 1) : 0;
      const int n_nb = (left != 0) + (above != 0);
      if (n_nz || p != hi) {
        rs_put(r, CABAC_CTX_TS_SIG_FLAG + (unsigned)n_nb, v != 0);
        budget--;
      }
      if (v) {
        const int sl = (left > 0) - (left < 0), sa = (above > 0) - (above < 0);
        unsigned sctx = ((sl == 0 && sa == 0) || sl * sa < 0) ? 0u : (sl >= 0 && sa >= 0) ? 1u : 2u;
        if (bdpcm) sctx += 3;
        rs_put(r, CABAC_CTX_TS_RESIDUAL_SIGN + sctx, v < 0);
        const int m = ts_mod_level(left, above, v < 0 ? -v : v, bdpcm);
        rs_put(r, CABAC_CTX_TS_LRG1_FLAG + (unsigned)(bdpcm ? 3 : n_nb), m > 1);
        budget -= 2;
        n_nz++;
        if (m > 1) {
          rs_put(r, CABAC_CTX_TS_PAR_FLAG, (unsigned)((m - 2) & 1));
          budget--;
        }
      }
      last1 = p;
    }
    for (p = lo; p <= hi && budget >= 4; p++) { /* pass 2: greater-than-3/5/7/9 flags */
      const int x = SX(p), y = SY(p), v = AT(x, y);
      const int m = ts_mod_level(x > 0 ? AT(x - 1, y) : 0, y > 0 ? AT(x, y - 1) : 0, v < 0 ? -v : v, bdpcm);
      for (int cut = 2; cut <= 8; cut += 2)
        if (m >= cut) {
          rs_put(r, CABAC_CTX_TS_GTX_FLAG + (unsigned)(cut >> 1), m >= cut + 2);
          budget--;
        }
      last2 = p;
    }
    for (p = lo; p <= hi; p++) { /* pass 3: remainders and the signs of the bypass-coded positions */
      const int x = SX(p), y = SY(p), v = AT(x, y);
      const int cut = p <= last2 ? 10 : p <= last1 ? 2 : 0;
      const int m = ts_mod_level(x > 0 ? AT(x - 1, y) : 0, y > 0 ? AT(x, y - 1) : 0, v < 0 ? -v : v, bdpcm || !cut);
      if (m >= cut) {
        rs_rem_abs(r, p <= last1 ? (unsigned)(m - cut) >> 1 : (unsigned)m, 1, 5, max_log2_range);
        if (m && p > last1) rs_put(r, CABAC_REC_EP, v < 0);
      }
    }
  }
#undef SX
#undef SY
#undef AT
  return r->n;
}

/* Returns the number of records (written up to cap), -1 for an all-zero block (the reference throws),
 * -2 for a bad size.  info (may be NULL): scanPosLast | CABAC_TU_INFO_MTS_VIOLATION. */
long orc_residual_records(int lw, int lh, int chroma, unsigned flags, int max_log2_range, const int32_t *coeff,
                          uint16_t *out, long cap, uint32_t *info) {
  if (lw < 0 || lw > 6 || lh < 0 || lh > 6 || chroma < 0 || chroma > 1) return -2;
  if (max_log2_range == 0) max_log2_range = 15;
  blk_geom g;
  blk_geom_init(&g, lw, lh);
  rec_sink r = {out, cap, 0};
  uint32_t *scan = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(g.w * g.h));
  orc_scan_order(lw, lh, scan);
#define SX(p) ((int)(scan[p] & 0xffff))
#define SY(p) ((int)(scan[p] >> 16))
#define COEF(p) (coeff[SY(p) * g.w + SX(p)])
  const int cg_size = 1 << g.cg_l2;
  const int n_cg = g.wg * g.hg;
  uint8_t cg_sig[64]; /* by raster position in the group grid */
  memset(cg_sig, 0, sizeof cg_sig);
  int last = -1;
  for (int p = 0; p < g.n_coded; p++)
    if (COEF(p)) {
      last = p;
      int gx = SX(p) >> g.cgw_l2, gy = SY(p) >> g.cgh_l2;
      cg_sig[gy * g.wg + gx] = 1;
    }
  if (last < 0) { free(scan); return -1; }
  uint32_t info_bits = (uint32_t)last;

  if (flags & CABAC_TU_TS_FLAG) rs_put(&r, CABAC_CTX_TRANSFORM_SKIP_FLAG(chroma), (flags & CABAC_TU_TRANSFORM_SKIP) != 0);
  if (flags & CABAC_TU_TRANSFORM_SKIP) {
    const long n_ts = residual_ts(&r, &g, scan, coeff, (flags & CABAC_TU_BDPCM) != 0, max_log2_range);
    free(scan);
    if (info) *info = info_bits;
    return n_ts;
  }

  /* SBT / MTS zero-out (cabac_writer.cpp:2660-2667, :2507-2516, unit.cpp:465-479): a 32-wide (tall) luma block coded as 16 */
  const int zo = (flags & CABAC_TU_SBT_ZERO_OUT) && !chroma && g.w <= 32 && g.h <= 32;
  const int zo_w = zo && g.w == 32 ? 16 : (g.w < 32 ? g.w : 32), zo_h = zo && g.h == 32 ? 16 : (g.h < 32 ? g.h : 32);
  { /* last significant position, cabac_writer.cpp:2639-2720 */
    const unsigned px = (unsigned)SX(last), py = (unsigned)SY(last);
    static const uint8_t luma_off[7] = {0, 0, 0, 3, 6, 10, 15};
    const unsigned off_x = chroma ? 0 : luma_off[lw], off_y = chroma ? 0 : luma_off[lh];
    unsigned sh_x, sh_y;
    if (chroma) {
      sh_x = (unsigned)(g.w >> 3); if (sh_x > 2) sh_x = 2;
      sh_y = (unsigned)(g.h >> 3); if (sh_y > 2) sh_y = 2;
    } else {
      sh_x = (unsigned)(lw + 1) >> 2; sh_y = (unsigned)(lh + 1) >> 2;
    }
    const unsigned gx = last_group_idx(px), gy = last_group_idx(py);
    const unsigned max_x = last_group_idx((unsigned)zo_w - 1), max_y = last_group_idx((unsigned)zo_h - 1);
    for (unsigned k = 0; k < gx; k++) rs_put(&r, CABAC_CTX_LAST_X(chroma) + off_x + (k >> sh_x), 1);
    if (gx < max_x) rs_put(&r, CABAC_CTX_LAST_X(chroma) + off_x + (gx >> sh_x), 0);
    for (unsigned k = 0; k < gy; k++) rs_put(&r, CABAC_CTX_LAST_Y(chroma) + off_y + (k >> sh_y), 1);
    if (gy < max_y) rs_put(&r, CABAC_CTX_LAST_Y(chroma) + off_y + (gy >> sh_y), 0);
    if (gx > 3) rs_bins_ep(&r, px - last_min_in_group(gx), (gx - 2) >> 1);
    if (gy > 3) rs_bins_ep(&r, py - last_min_in_group(gy), (gy - 2) >> 1);
  }

  const unsigned trans = (flags & CABAC_TU_DEP_QUANT) ? 32040u : 0u; /* cabac_writer.cpp:2482 */
  int state = 0;
  int budget = ((zo_w * zo_h) * 28) >> 4; /* :2485-2489 */
  const int last_cg = last >> g.cg_l2;
  (void)n_cg;

  for (int cg = last_cg; cg >= 0; cg--) {
    const int lo = cg << g.cg_l2, hi = lo + cg_size - 1;
    const int cgx = SX(lo) >> g.cgw_l2, cgy = SY(lo) >> g.cgh_l2;
    const int coded_group = cg_sig[cgy * g.wg + cgx];
    if ((cgx << g.cgw_l2) >= zo_w || (cgy << g.cgh_l2) >= zo_h) continue; /* zeroed out: no flag, no coefficients (:2507-2516) */
    if (cg != last_cg && cg != 0) {
      const int right = cgx + 1 < g.wg ? cg_sig[cgy * g.wg + cgx + 1] : 0;
      const int below = cgy + 1 < g.hg ? cg_sig[(cgy + 1) * g.wg + cgx] : 0;
      rs_put(&r, CABAC_CTX_SIG_COEFF_GROUP(chroma) + (unsigned)(right | below), (unsigned)coded_group);
      if (!coded_group) continue;
    }
    if (!chroma && coded_group && (cgx > 3 || cgy > 3)) info_bits |= CABAC_TU_INFO_MTS_VIOLATION;

    const int first = cg == last_cg ? last : hi;
    const int infer = cg == last_cg ? last : (cg != 0 ? lo : -1);
    int n_nz = 0, first_nz = first, last_nz = -1;
    unsigned signs = 0;
    int p = first;
    /* pass 1: context-coded flags while the budget lasts */
    for (; p >= lo && budget >= 4; p--) {
      const int v = COEF(p);
      const int x = SX(p), y = SY(p), diag = x + y;
      const tmpl_t t = tmpl_at(coeff, &g, x, y);
      if (n_nz || p != infer) {
        int ofs = (t.sum_clip + 1) >> 1;
        if (ofs > 3) ofs = 3;
        if (diag < 2) ofs += 4;
        if (!chroma && diag < 5) ofs += 4;
        const int set = (state > 1 ? state - 1 : 0) * 2 + chroma;
        rs_put(&r, CABAC_CTX_SIG_FLAG(set) + (unsigned)ofs, v != 0);
        budget--;
      }
      if (v) {
        int ofs = 0;
        if (p != last) { /* ctxOffsetAbs: the template of this position (context_modelling.hpp:131-143) */
          int s1 = t.sum_clip - t.n_nonzero;
          ofs = (s1 < 4 ? s1 : 4) + 1;
          if (diag == 0) ofs += chroma ? 5 : 15;
          else if (!chroma) ofs += diag < 3 ? 10 : diag < 10 ? 5 : 0;
        }
        int rem = (v < 0 ? -v : v) - 1;
        if (p != last) signs <<= 1;
        if (v < 0) signs++;
        n_nz++;
        first_nz = p;
        if (p > last_nz) last_nz = p;
        rs_put(&r, CABAC_CTX_GTX_FLAG(2 + chroma) + (unsigned)ofs, rem != 0);
        budget--;
        if (rem) {
          rem--;
          rs_put(&r, CABAC_CTX_PAR_FLAG(chroma) + (unsigned)ofs, (unsigned)(rem & 1));
          rs_put(&r, CABAC_CTX_GTX_FLAG(chroma) + (unsigned)ofs, (rem >> 1) != 0);
          budget -= 2;
        }
      }
      state = (int)((trans >> ((state << 2) + ((v & 1) << 1))) & 3);
    }
    const int bypass_from = p; /* positions bypass_from .. lo are coded without contexts */
    /* pass 2: remainders of the context-coded positions */
    for (int q = first; q > bypass_from; q--) {
      const int v = COEF(q), a = v < 0 ? -v : v;
      if (a >= 4) {
        const tmpl_t t = tmpl_at(coeff, &g, SX(q), SY(q));
        rs_rem_abs(&r, (unsigned)(a - 4) >> 1, rice_from(t.sum_abs, 4), 5, max_log2_range);
      }
    }
    /* pass 3: whole levels in bypass mode */
    for (int q = bypass_from; q >= lo; q--) {
      const int v = COEF(q), a = v < 0 ? -v : v;
      const tmpl_t t = tmpl_at(coeff, &g, SX(q), SY(q));
      const unsigned rice = rice_from(t.sum_abs, 0);
      const int pos0 = (state < 2 ? 1 : 2) << rice;
      const unsigned rem = a == 0 ? (unsigned)pos0 : a <= pos0 ? (unsigned)(a - 1) : (unsigned)a;
      rs_rem_abs(&r, rem, rice, 5, max_log2_range);
      state = (int)((trans >> ((state << 2) + ((a & 1) << 1))) & 3);
      if (a) {
        n_nz++;
        first_nz = q;
        if (q > last_nz) last_nz = q;
        signs = (signs << 1) + (v < 0);
      }
    }
    unsigned n_signs = (unsigned)n_nz;
    if ((flags & CABAC_TU_SIGN_HIDING) && last_nz - first_nz >= 4) { /* hideSign, SBH_THRESHOLD = 4 */
      n_signs--;
      signs >>= 1;
    }
    rs_bins_ep(&r, signs, n_signs);
  }
#undef SX
#undef SY
#undef COEF
  free(scan);
  if (info) *info = info_bits;
  return r.n;
}

/* ---------------------------------------------------------------- residual parser (decoder side)
 * Restates CABACReader::residual_coding (cabac_reader.cpp:2647-2735), ts_flag (:2737-2752), last_sig_coeff (:2865-2938),
 * residual_coding_subblock (:2946-3128), residual_codingTS (:3130-3152) and residual_coding_subblockTS (:3154-3339): the
 * context of every bin follows from the coefficients decoded so far, so no bin/context sequence is supplied.  One
 * substream = n_tu blocks in order, then (finish != 0) encodeBinTrm(1) / finish() are checked.  coeff_out receives each
 * block at tus[t].coeff_offset (raster, stride = width; only the coded region of a 64-wide/tall block is written).
 * A block with CABAC_TU_TS_FLAG has its transform_skip_flag in the stream and is parsed as that bin says (the
 * descriptor's CABAC_TU_TRANSFORM_SKIP bit is then ignored); without it the descriptor's bit decides (BDPCM blocks: the
 * reference infers the flag).  info[t] (may be NULL): scanPosLast | CABAC_TU_INFO_MTS_VIOLATION for a regular block,
 * CABAC_TU_INFO_TS for a block parsed as transform skip.
 * Returns 0; -2 unsupported block (bad size, transform skip wider/taller than 32); -4 read past the end; -5 missing
 * terminate bin / stop pattern. */
static int parse_block_ts(bin_dec *d, const blk_geom *g, const uint32_t *scan, int bdpcm, int max_log2, int32_t *coeff) {
#define SX(p) ((int)(scan[p] & 0xffff))
#define SY(p) ((int)(scan[p] >> 16))
#define AT(x, y) (coeff[(y) * g->w + (x)])
  const int n_cg = (g->w * g->h) >> g->cg_l2, cg_size = 1 << g->cg_l2;
  uint8_t cg_sig[64];
  int n_sig = 0;
  memset(cg_sig, 0, sizeof cg_sig);
  int budget = (g->w * g->h * 7) >> 2; /* cabac_reader.cpp:3136-3137 */
  for (int cg = 0; cg < n_cg; cg++) {
    const int lo = cg << g->cg_l2, hi = lo + cg_size - 1;
    const int cgx = SX(lo) >> g->cgw_l2, cgy = SY(lo) >> g->cgh_l2;
    int sig = cg == n_cg - 1 && n_sig == 0; /* :3203: the last group of a block without any significant group */
    if (!sig) {
      const int left = cgx > 0 ? cg_sig[cgy * g->wg + cgx - 1] : 0, above = cgy > 0 ? cg_sig[(cgy - 1) * g->wg + cgx] : 0;
      sig = (int)dec_bin(d, CABAC_CTX_TS_SIG_COEFF_GROUP + (unsigned)(left + above));
    }
    if (!sig) continue;
    cg_sig[cgy * g->wg + cgx] = 1;
    n_sig++;
    int n_nz = 0, last1 = -1, last2 = -1, p;
    int nz_pos[16];
    unsigned sign_pattern = 0;
    for (p = lo; p <= hi && budget >= 4; p++) { /* pass 1: sig, sign, greater-1, parity (:3222-3271) */
      const int x = SX(p), y = SY(p);
      const int left = x > 0 ? AT(x - 1, y) : 0, above = y > 0 ? AT(x, y - 1) : 0;
      const int n_nb = (left != 0) + (above != 0);
      unsigned sf = (!n_nz && p == hi);
      if (!sf) {
        sf = dec_bin(d, CABAC_CTX_TS_SIG_FLAG + (unsigned)n_nb);
        budget--;
      }
      if (sf) {
        const int sl = (left > 0) - (left < 0), sa = (above > 0) - (above < 0);
        unsigned sctx = ((sl == 0 && sa == 0) || sl * sa < 0) ? 0u : (sl >= 0 && sa >= 0) ? 1u : 2u;
        if (bdpcm) sctx += 3;
        const unsigned sign = dec_bin(d, CABAC_CTX_TS_RESIDUAL_SIGN + sctx);
        sign_pattern += sign << n_nz;
        nz_pos[n_nz++] = p;
        const unsigned g1 = dec_bin(d, CABAC_CTX_TS_LRG1_FLAG + (unsigned)(bdpcm ? 3 : n_nb));
        budget -= 2;
        unsigned par = 0;
        if (g1) {
          par = dec_bin(d, CABAC_CTX_TS_PAR_FLAG);
          budget--;
        }
        AT(x, y) = (sign ? -1 : 1) * (int32_t)(1 + par + g1);
      }
      last1 = p;
    }
    for (p = lo; p <= hi && budget >= 4; p++) { /* pass 2: greater-than-3/5/7/9 flags on the magnitudes (:3276-3297) */
      int32_t *c = &AT(SX(p), SY(p));
      if (*c < 0) *c = -*c;
      for (int cut = 2; cut <= 8; cut += 2)
        if (*c >= cut) {
          *c += (int32_t)(dec_bin(d, CABAC_CTX_TS_GTX_FLAG + (unsigned)(cut >> 1)) << 1);
          budget--;
        }
      last2 = p;
    }
    for (p = lo; p <= hi; p++) { /* pass 3: remainders, bypass-coded levels with their signs, level un-mapping (:3299-3329) */
      const int x = SX(p), y = SY(p);
      int32_t *c = &AT(x, y);
      const int cut = p <= last2 ? 10 : p <= last1 ? 2 : 0;
      if (*c < 0) *c = -*c;
      if (*c >= cut) {
        const int32_t rem = (int32_t)dec_rem_abs(d, 1, 5, max_log2);
        *c += p <= last1 ? rem << 1 : rem;
        if (*c && p > last1) {
          sign_pattern += dec_ep(d) << n_nz;
          nz_pos[n_nz++] = p;
        }
      }
      if (!bdpcm && cut && *c > 0) { /* decDeriveModCoeff, context_modelling.hpp:367-384 */
        int left = x > 0 ? AT(x - 1, y) : 0, above = y > 0 ? AT(x, y - 1) : 0;
        if (left < 0) left = -left;
        if (above < 0) above = -above;
        const int pred = left > above ? left : above;
        *c = (*c == 1 && pred > 0) ? pred : *c - (*c <= pred);
      }
    }
    for (int k = 0; k < n_nz; k++) { /* signs (:3332-3338) */
      int32_t *c = &AT(SX(nz_pos[k]), SY(nz_pos[k]));
      if (sign_pattern & 1) *c = -*c;
      sign_pattern >>= 1;
    }
  }
#undef SX
#undef SY
#undef AT
  return 0;
}

static int parse_block(bin_dec *d, const cabac_tu_desc *tu, int32_t *coeff, uint32_t *info) {
  const int lw = tu->log2_width, lh = tu->log2_height, chroma = tu->channel;
  const unsigned flags = tu->flags;
  const int max_log2 = tu->max_log2_tr_range ? tu->max_log2_tr_range : 15;
  if (lw > 6 || lh > 6 || chroma > 1) return -2;
  blk_geom g;
  blk_geom_init(&g, lw, lh);
  /* ts_flag, cabac_reader.cpp:2737-2752 */
  unsigned ts = (flags & CABAC_TU_TRANSFORM_SKIP) != 0;
  if (flags & CABAC_TU_TS_FLAG) ts = dec_bin(d, CABAC_CTX_TRANSFORM_SKIP_FLAG(chroma));
  if (ts && (lw > 5 || lh > 5)) return -2;
  uint32_t *scan = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(g.w * g.h));
  orc_scan_order(lw, lh, scan);
#define SX(p) ((int)(scan[p] & 0xffff))
#define SY(p) ((int)(scan[p] >> 16))
  const int we = g.w < 32 ? g.w : 32, he = g.h < 32 ? g.h : 32;
  memset(coeff, 0, sizeof(int32_t) * (size_t)(g.w * g.h)); /* the reader requires a zeroed block (cabac_reader.cpp:2950) */
  if (ts) {
    const int rc_ts = parse_block_ts(d, &g, scan, (flags & CABAC_TU_BDPCM) != 0, max_log2, coeff);
    free(scan);
    if (info) *info = CABAC_TU_INFO_TS;
    return rc_ts ? rc_ts : d->underrun ? -4 : 0;
  }
  uint32_t info_bits = 0;

  /* last significant position */
  static const uint8_t luma_off[7] = {0, 0, 0, 3, 6, 10, 15};
  const unsigned off_x = chroma ? 0 : luma_off[lw], off_y = chroma ? 0 : luma_off[lh];
  unsigned sh_x, sh_y;
  if (chroma) {
    sh_x = (unsigned)(g.w >> 3); if (sh_x > 2) sh_x = 2;
    sh_y = (unsigned)(g.h >> 3); if (sh_y > 2) sh_y = 2;
  } else {
    sh_x = (unsigned)(lw + 1) >> 2; sh_y = (unsigned)(lh + 1) >> 2;
  }
  /* SBT / MTS zero-out (cabac_reader.cpp:2880-2891, :2718-2727, unit.cpp:465-479) */
  const int zo = (flags & CABAC_TU_SBT_ZERO_OUT) && !chroma && g.w <= 32 && g.h <= 32;
  const int zo_w = zo && g.w == 32 ? 16 : we, zo_h = zo && g.h == 32 ? 16 : he;
  const unsigned max_x = last_group_idx((unsigned)zo_w - 1), max_y = last_group_idx((unsigned)zo_h - 1);
  unsigned px = 0, py = 0;
  while (px < max_x && dec_bin(d, CABAC_CTX_LAST_X(chroma) + off_x + (px >> sh_x))) px++;
  while (py < max_y && dec_bin(d, CABAC_CTX_LAST_Y(chroma) + off_y + (py >> sh_y))) py++;
  if (px > 3) px = last_min_in_group(px) + dec_bins_ep(d, (px - 2) >> 1);
  if (py > 3) py = last_min_in_group(py) + dec_bins_ep(d, (py - 2) >> 1);
  int last = 0;
  for (; last < g.w * g.h - 1; last++)
    if (SX(last) == (int)px && SY(last) == (int)py) break;
  info_bits = (uint32_t)last;

  const unsigned trans = (flags & CABAC_TU_DEP_QUANT) ? 32040u : 0u;
  int state = 0, budget = (zo_w * zo_h * 28) >> 4;
  uint8_t cg_sig[64];
  memset(cg_sig, 0, sizeof cg_sig);
  const int cg_size = 1 << g.cg_l2, last_cg = last >> g.cg_l2;
  for (int cg = last_cg; cg >= 0; cg--) {
    const int lo = cg << g.cg_l2, hi = lo + cg_size - 1;
    const int cgx = SX(lo) >> g.cgw_l2, cgy = SY(lo) >> g.cgh_l2;
    if ((cgx << g.cgw_l2) >= zo_w || (cgy << g.cgh_l2) >= zo_h) continue; /* zeroed out: nothing is coded for it */
    int sig = cg == last_cg || cg == 0;
    if (!sig) {
      const int right = cgx + 1 < g.wg ? cg_sig[cgy * g.wg + cgx + 1] : 0;
      const int below = cgy + 1 < g.hg ? cg_sig[(cgy + 1) * g.wg + cgx] : 0;
      sig = (int)dec_bin(d, CABAC_CTX_SIG_COEFF_GROUP(chroma) + (unsigned)(right | below));
    }
    if (!sig) continue;
    cg_sig[cgy * g.wg + cgx] = 1;
    if (!chroma && (cgx > 3 || cgy > 3)) info_bits |= CABAC_TU_INFO_MTS_VIOLATION; /* cabac_reader.cpp:2729-2732 */
    const int first = cg == last_cg ? last : hi;
    const int infer = cg == last_cg ? last : (cg != 0 ? lo : -1);
    int n_nz = 0, first_nz = first, last_nz = -1, p;
    int nz_pos[16];
    for (p = first; p >= lo && budget >= 4; p--) { /* pass 1 */
      const int x = SX(p), y = SY(p), diag = x + y;
      const tmpl_t t = tmpl_at(coeff, &g, x, y);
      unsigned sf = (!n_nz && p == infer);
      if (!sf) {
        int ofs = (t.sum_clip + 1) >> 1;
        if (ofs > 3) ofs = 3;
        if (diag < 2) ofs += 4;
        if (!chroma && diag < 5) ofs += 4;
        sf = dec_bin(d, CABAC_CTX_SIG_FLAG((state > 1 ? state - 1 : 0) * 2 + chroma) + (unsigned)ofs);
        budget--;
      }
      if (sf) {
        int ofs = 0;
        if (p != last) {
          int s1 = t.sum_clip - t.n_nonzero;
          ofs = (s1 < 4 ? s1 : 4) + 1;
          if (diag == 0) ofs += chroma ? 5 : 15;
          else if (!chroma) ofs += diag < 3 ? 10 : diag < 10 ? 5 : 0;
        }
        nz_pos[n_nz++] = p;
        first_nz = p;
        if (p > last_nz) last_nz = p;
        const unsigned g1 = dec_bin(d, CABAC_CTX_GTX_FLAG(2 + chroma) + (unsigned)ofs);
        unsigned par = 0, g2 = 0;
        budget--;
        if (g1) {
          par = dec_bin(d, CABAC_CTX_PAR_FLAG(chroma) + (unsigned)ofs);
          g2 = dec_bin(d, CABAC_CTX_GTX_FLAG(chroma) + (unsigned)ofs);
          budget -= 2;
        }
        coeff[y * g.w + x] = (int32_t)(1 + par + g1 + (g2 << 1));
      }
      state = (int)((trans >> ((state << 2) + ((coeff[y * g.w + x] & 1) << 1))) & 3);
    }
    const int bypass_from = p;
    for (int q = first; q > bypass_from; q--) { /* pass 2 */
      int32_t *c = &coeff[SY(q) * g.w + SX(q)];
      if (*c >= 4) {
        const tmpl_t t = tmpl_at(coeff, &g, SX(q), SY(q));
        *c += (int32_t)(dec_rem_abs(d, rice_from(t.sum_abs, 4), 5, max_log2) << 1);
      }
    }
    for (int q = bypass_from; q >= lo; q--) { /* pass 3 */
      const tmpl_t t = tmpl_at(coeff, &g, SX(q), SY(q));
      const unsigned rice = rice_from(t.sum_abs, 0);
      const int pos0 = (state < 2 ? 1 : 2) << rice;
      const int rem = (int)dec_rem_abs(d, rice, 5, max_log2);
      const int v = rem == pos0 ? 0 : rem < pos0 ? rem + 1 : rem;
      state = (int)((trans >> ((state << 2) + ((v & 1) << 1))) & 3);
      if (v) {
        nz_pos[n_nz++] = q;
        first_nz = q;
        if (q > last_nz) last_nz = q;
        coeff[SY(q) * g.w + SX(q)] = v;
      }
    }
    const int hide = (flags & CABAC_TU_SIGN_HIDING) && last_nz - first_nz >= 4;
    const int n_signs = hide ? n_nz - 1 : n_nz;
    const unsigned pattern = dec_bins_ep(d, (unsigned)n_signs);
    int sum = 0;
    for (int k = 0; k < n_signs; k++) {
      int32_t *c = &coeff[SY(nz_pos[k]) * g.w + SX(nz_pos[k])];
      sum += *c;
      if ((pattern >> (n_signs - 1 - k)) & 1) *c = -*c;
    }
    if (n_nz > n_signs) { /* the hidden sign is the parity of the sum of levels, cabac_reader.cpp:3120-3126 */
      int32_t *c = &coeff[SY(nz_pos[n_signs]) * g.w + SX(nz_pos[n_signs])];
      sum += *c;
      if (sum & 1) *c = -*c;
    }
  }
#undef SX
#undef SY
  free(scan);
  if (info) *info = info_bits;
  return d->underrun ? -4 : 0;
}

int orc_residual_decode(const uint8_t *in, long n_in, int qp, int init_id, const void *tus_, long n_tu, int finish,
                        int32_t *coeff_out, uint32_t *n_bits_read, uint32_t *info) {
  const cabac_tu_desc *tus = (const cabac_tu_desc *)tus_;
  bin_dec d;
  memset(&d, 0, sizeof d);
  d.in = in;
  d.n_in = n_in;
  ctx_store_init(&d.ctx, qp, init_id);
  dec_start(&d);
  for (long t = 0; t < n_tu; t++) {
    const int rc = parse_block(&d, &tus[t], coeff_out + tus[t].coeff_offset, info ? &info[t] : NULL);
    if (rc) return rc;
  }
  int rc = 0;
  if (finish) {
    if (dec_trm(&d) != 1) rc = -5;
    if (!rc && d.underrun) rc = -4;
    if (!rc) rc = dec_finish(&d);
  }
  if (n_bits_read) *n_bits_read = (uint32_t)(8 * d.idx + d.bits_needed);
  if (!rc && d.underrun) rc = -4;
  return rc;
}
