/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's CABAC bin codec, used ONLY as the parity
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
 * library (libcabac_hip.so) never includes, links or loads anything from oracle/.
 *
 * Parity status: PINNED — checked bit-exact against the reference's own compiled sources
 * (oracle/_ref/libcabac_ref.so, built by oracle/Makefile) by tests/test_oracle_vs_reference.py
 * in the build container, and against the golden vectors under tests/golden/ (generated from
 * that compiled reference by oracle/gen_golden.py) everywhere else.
 */
#ifndef CABAC_ORACLE_H
#define CABAC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Operation stream — identical codes to oracle/ref_harness.cpp.  One op = 4 x uint32. */
enum {
  ORC_OP_ENC_BIN = 0,     /* a = bin, b = ctxId                                  */
  ORC_OP_ENC_EP = 1,      /* a = bin                                             */
  ORC_OP_ENC_BINS_EP = 2, /* a = bins, b = numBins                               */
  ORC_OP_ENC_REM_ABS = 3, /* a = value, b = rice, c = cutoff | maxLog2TrDR << 8  */
  ORC_OP_ENC_TRM = 4,     /* a = bin                                             */
  ORC_OP_ALIGN = 5,
  ORC_OP_UNARY_MAX = 6,  /* a = symbol, b = ctxId0 | ctxIdN << 16, c = maxSymbol */
  ORC_OP_UNARY_EP = 7,   /* a = symbol, b = maxSymbol                            */
  ORC_OP_EXP_GOLOMB = 8, /* a = symbol, b = count                                */
  ORC_OP_TRUNC_BIN = 9   /* a = symbol, b = maxSymbol                            */
};

#define ORC_NUM_CTX 379

/* Ctx::init(qp, initId): reference contexts.cpp:893-901, :915-920, :996-1015 */
void orc_ctx_init(int qp, int init_id, uint16_t *s0, uint16_t *s1, uint8_t *rate);

/* One context traced through n updates (contexts.cpp:903-913, :939-950). */
void orc_ctx_trace(int qp, int init_id, int ctx_id, const uint8_t *bins, int n, unsigned range,
                   uint8_t *state8, uint8_t *lps, uint16_t *s0_after, uint16_t *s1_after);

/* flags: bit0 finish(), bit1 writeByteAlignment(), bit2 probe: nothing is flushed, *n_bits = getNumWrittenBits()
 * (arith_codec.cpp:482-485) and 0 is returned.  Returns bytes written to out (whole bytes
 * plus one MSB-aligned partial byte if *n_bits % 8), or <0: -2 bad op/record, -3 capacity.
 * n_bins_out (may be NULL): {ctx, EP, TRM} BinCounter totals (arith_codec.cpp:281-316). */
long orc_encode_ops(const uint32_t *ops, long n_ops, int qp, int init_id, int flags, uint8_t *out,
                    long cap, uint32_t *n_bits, uint32_t *n_bins_out);
long orc_encode_records(const uint16_t *rec, long n, int qp, int init_id, int flags, uint8_t *out,
                        long cap, uint32_t *n_bits);

/* Returns 0; -2 bad record; -4 read past end of input (reference: "FIFO exceeded");
 * -5 stop pattern check of finish() failed.  flags bit0: run finish(). */
int orc_decode_records(const uint16_t *rec, long n, int qp, int init_id, int flags,
                       const uint8_t *in, long n_in, uint8_t *bins, uint32_t *n_bits_read);
int orc_decode_ops(const uint32_t *ops, long n_ops, int qp, int init_id, int flags,
                   const uint8_t *in, long n_in, uint32_t *values);

/* Binarisation: expand an op stream into bin records (include/cabac_hip.h format).
 * Returns the number of records (call with rec == NULL to size), or -2 on a bad op. */
long orc_ops_to_records(const uint32_t *ops, long n_ops, uint16_t *rec, long cap);

/* Batch helpers over substream descriptors laid out like cabac_substream_desc
 * (include/cabac_hip.h) — used by the parity tests and the CPU baseline. `results`
 * is an array of {n_bits, flags} pairs.  Substreams [first, first+count) are coded. */
void orc_encode_batch(const void *desc, uint32_t first, uint32_t count, const uint16_t *records,
                      uint8_t *bytes, uint32_t *results);
void orc_decode_batch(const void *desc, uint32_t first, uint32_t count, const uint16_t *records,
                      const uint8_t *bytes, uint8_t *bins, uint32_t *results);

/* bench.py's whole-batch hash (see cabac_oracle.c) */
uint64_t orc_digest_mt(const void *desc, uint32_t first, uint32_t count, const uint16_t *records, int n_threads, uint64_t *digests);
void orc_digest_slots(const void *desc, const uint32_t *results, uint32_t count, const uint8_t *bytes, uint64_t *digests);

/* OutputBitstream::countStartCodeEmulations, common/bit_stream.cpp:157-181 */
int orc_count_emulations(const uint8_t *bytes, long n);

/* Bit estimator (BitEstimator_Std, arith_codec.cpp:603-711): cost of a bin string in 1/32768 bit after
 * reset(qp, initId).  Returns 0, or -2 on a bad record / op.  A bad record leaves *frac_bits at the cost of
 * the records before it. */
int orc_estimate_records(const uint16_t *rec, long n, int qp, int init_id, uint64_t *frac_bits);
int orc_estimate_ops(const uint32_t *ops, long n_ops, int qp, int init_id, uint64_t *frac_bits);
/* the same from given context states (format of orc_ctx_init) instead of reset(qp, initId) */
int orc_estimate_records_from(const uint16_t *rec, long n, const uint16_t *s0, const uint16_t *s1,
                              const uint8_t *rate, uint64_t *frac_bits);
void orc_estimate_batch(const void *desc, uint32_t first, uint32_t count, const uint16_t *records,
                        uint64_t *frac_bits, uint32_t *flags);

/* Residual coding (CABACWriter::residual_coding, cabac_writer.cpp:2424-2872): a coefficient block -> bin records.
 * flags = CABAC_TU_* of include/cabac_hip.h.  Returns the number of records (out filled up to cap), -1 for an
 * all-zero block (the reference throws), -2 for a bad size.  orc_scan_order: scan position -> x | y << 16. */
long orc_residual_records(int log2_width, int log2_height, int chroma, unsigned flags, int max_log2_range,
                          const int32_t *coeff, uint16_t *out, long cap, uint32_t *info);
long orc_scan_order(int log2_width, int log2_height, uint32_t *out);

/* Residual parser (CABACReader::residual_coding, cabac_reader.cpp:2647-3128, regular residual coding): bytes -> coefficient
 * blocks; `tus` = n_tu cabac_tu_desc (include/cabac_hip.h), block t written at coeff_out + tus[t].coeff_offset.
 * finish: expect TRM(1) and the stop pattern after the last block.  Returns 0, -2 unsupported block, -4 read past the
 * end, -5 missing terminate bin / stop pattern. */
int orc_residual_decode(const uint8_t *in, long n_in, int qp, int init_id, const void *tus, long n_tu, int finish,
                        int32_t *coeff_out, uint32_t *n_bits_read, uint32_t *info);

#ifdef __cplusplus
}
#endif
#endif
