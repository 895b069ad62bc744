/*
 * cabac_hip.h — C ABI of the MI355X-native CABAC bin codec (libcabac_hip.so).
 *
 * This is the drop-in boundary for the hot path of p-sawicki/entropy_coding:
 * the arithmetic bin encoder/decoder + context model + binarisation helpers
 * (reference: src/entropy_codec/arith_codec.{hpp,cpp}, src/common/contexts.{hpp,cpp},
 * src/common/bit_stream.{hpp,cpp}, cabac_writer.cpp:854-882,3072-3118).
 *
 * The reference drives the codec through ~10^5 tiny virtual calls per frame
 * (BinEncIf, arith_codec.hpp:31-70).  Those cannot cross a device boundary one
 * at a time, so the boundary is *batched*: host code records the calls of one
 * independent CABAC substream (slice / tile / frame) as a flat array of 16-bit
 * bin records, and a batch of substreams is coded on the GPU — one wavefront
 * per substream.  Plain pointers and sizes only; no C++ / torch types.
 *
 * All functions return 0 (CABAC_HIP_OK) or a negative cabac_hip_status.
 * No exception crosses this boundary.  A cabac_hip_ctx is bound to one device
 * and one HIP stream; calls on the same ctx must not overlap, different ctxs
 * are independent (thread-safe per ctx).
 */
#ifndef CABAC_HIP_H
#define CABAC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Constants of the codec (reference: contexts.cpp:773-774, type_def.hpp:21-26) */
#define CABAC_NUM_CONTEXTS 379 /* Ctx::NumberOfContexts                       */
#define CABAC_INIT_B 0         /* B_SLICE init table                          */
#define CABAC_INIT_P 1         /* P_SLICE                                     */
#define CABAC_INIT_I 2         /* I_SLICE                                     */

/* ------------------------------------------------------------------ */
/* Bin record (uint16_t) — one per coded bin.
 *   bit 15     bin value (encode input; ignored on decode input)
 *   bits 14..9 zero
 *   bits 8..0  id: 0..378 = context-coded bin with that ctxId
 *                  (BinEncIf::encodeBin(bin, ctxId), arith_codec.cpp:553-582)
 *              0x1FE = bypass bin   (encodeBinEP,  arith_codec.cpp:389-399)
 *              0x1FF = terminate bin(encodeBinTrm, arith_codec.cpp:460-478)
 *              0x1FD = align(): range := 256 (arith_codec.cpp:480); codes no bin
 *              0x1FC, 0x1FB = bit-estimator pseudo-records (cabac_hip_estimate_* only; a bad
 *                  record for the codec): resetBits() / start() — the running cost := 0 — and
 *                  restart() — the running cost rounded DOWN to a whole bit
 *                  (arith_codec.cpp:615-628); contexts carry on
 * encodeBinsEP / encodeRemAbsEP (arith_codec.cpp:401-458) are recorded as their
 * individual bypass bins, MSB first — byte-identical by construction (the
 * reference's 8-at-a-time loop is arithmetically n single bypass bins).
 */
#define CABAC_REC_BIN 0x8000u
#define CABAC_REC_ID_MASK 0x01FFu
#define CABAC_REC_ALIGN 0x01FDu
#define CABAC_REC_EST_RESETBITS 0x01FCu
#define CABAC_REC_EST_RESTART 0x01FBu
#define CABAC_REC_EP 0x01FEu
#define CABAC_REC_TRM 0x01FFu

/* ------------------------------------------------------------------ */
/* Syntax-element record (2 x uint32_t) — input of the device binariser
 * (cabac_hip_binarize_device).  word0 = kind | params, word1 = value.
 *   kind (bits 3..0 of word0):
 *     0 CTX_BIN     p: ctxId[12:4]                       value = bin
 *     1 EP_BINS     p: numBins[9:4] (0..32)              value = bins, MSB first
 *                   (BinEncIf::encodeBinsEP, arith_codec.cpp:401-424)
 *     2 REM_ABS     p: rice[8:4], cutoff[13:9], maxLog2TrDR[19:14]
 *                   (BinEncIf::encodeRemAbsEP, arith_codec.cpp:426-458)
 *     3 TRM         value = bin   (encodeBinTrm)
 *     4 UNARY_MAX   p: ctxId0[12:4], ctxIdN[21:13], maxSymbol[29:22]
 *                   (CABACWriter::unary_max_symbol, cabac_writer.cpp:3072-3081)
 *     5 UNARY_EP    p: maxSymbol[9:4] (<=32)
 *                   (CABACWriter::unary_max_eqprob, cabac_writer.cpp:3083-3101)
 *     6 EXP_GOLOMB  p: count[8:4]
 *                   (CABACWriter::exp_golomb_eqprob, cabac_writer.cpp:3103-3118)
 *     7 TRUNC_BIN   p: maxSymbol[31:4]
 *                   (CABACWriter::xWriteTruncBinCode, cabac_writer.cpp:854-882)
 *     8 ALIGN
 */
#define CABAC_SE_CTX_BIN 0u
#define CABAC_SE_EP_BINS 1u
#define CABAC_SE_REM_ABS 2u
#define CABAC_SE_TRM 3u
#define CABAC_SE_UNARY_MAX 4u
#define CABAC_SE_UNARY_EP 5u
#define CABAC_SE_EXP_GOLOMB 6u
#define CABAC_SE_TRUNC_BIN 7u
#define CABAC_SE_ALIGN 8u

/* ------------------------------------------------------------------ */
/* One independent CABAC substream (private context store, private low/range,
 * private byte stream; reference: cabac_writer.cpp:16-39, :104-107).        */
typedef struct cabac_substream_desc {
  uint64_t rec_offset;    /* first record of this substream in records[] (in records) */
  uint64_t byte_offset;   /* first byte of this substream in bytes[]                 */
  uint32_t n_records;     /* number of bin records                                    */
  uint32_t byte_capacity; /* encode: room at byte_offset; decode: valid input bytes   */
                          /* byte_offset must be a multiple of 16; decode reads whole  */
                          /* aligned dwords, so bytes[] must be readable up to the next */
                          /* multiple of 4 past byte_offset + byte_capacity             */
  int32_t qp;             /* slice QP for Ctx::init (clipped to 0..63, contexts.cpp:1010) */
  uint32_t init_id;       /* bits 1..0: CABAC_INIT_B/P/I;  bits 31..8: CABAC_SUB_* flags */
} cabac_substream_desc;

/* desc.init_id flag bits */
#define CABAC_SUB_FINISH 0x100u /* encode: run BinEncoderBase::finish() (arith_codec.cpp:339-357); REQUIRED — a   \
                                   substream is coded start() .. finish() in one call, output without it is     \
                                   unspecified.  decode: run BinDecoderBase::finish() stop-pattern check (:68-73) */
#define CABAC_SUB_ALIGN_RBSP 0x200u /* encode, with FINISH: also OutputBitstream::writeByteAlignment() \
                                       (bit_stream.cpp:152-155): stop bit '1' + zero pad            */

#define CABAC_SUB_PROBE 0x400u /* encode, instead of FINISH: nothing is flushed; results[].n_bits receives what   \
                                 BinEncoderBase::getNumWrittenBits() (arith_codec.cpp:482-485) would answer after \
                                 the substream's records — bits in the bitstream + buffered bytes + the bits of   \
                                 low already shifted out, i.e. the sum of all renormalisation shifts — and the    \
                                 bytes written are unspecified                                                    */

typedef struct cabac_substream_result {
  uint32_t n_bits; /* encode: bits written (8*whole bytes + held bits, i.e.
                      OutputBitstream::getNumberOfWrittenBits(), bit_stream.cpp:60-62);
                      a trailing partial byte is MSB-aligned, zero padded.
                      decode: 8*bytes consumed + bitsNeeded (getNumBitsRead analogue) */
  uint32_t flags;  /* CABAC_RES_* */
} cabac_substream_result;

#define CABAC_RES_OVERFLOW 0x1u    /* encode: byte_capacity too small (output truncated)     */
#define CABAC_RES_BAD_RECORD 0x2u  /* record id is neither a ctxId < 379 nor a special id    */
#define CABAC_RES_UNDERRUN 0x4u    /* decode: read past byte_capacity ("FIFO exceeded",      \
                                      bit_stream.cpp:269: the reference throws there); the   \
                                      bins from that read on and n_bits are unspecified      */
#define CABAC_RES_BAD_STOP 0x8u    /* decode: finish() stop/alignment pattern check failed   */
#define CABAC_RES_RANGE 0x10u      /* residual_parse16: a coefficient does not fit int16_t (stored truncated) */

typedef enum cabac_hip_status {
  CABAC_HIP_OK = 0,
  CABAC_HIP_ERR_NO_DEVICE = -1,
  CABAC_HIP_ERR_INVALID = -2,
  CABAC_HIP_ERR_HIP = -3,      /* a HIP runtime call failed; see cabac_hip_last_error */
  CABAC_HIP_ERR_NOMEM = -4,
  CABAC_HIP_ERR_SUBSTREAM = -5 /* at least one substream result has a flag set       */
} cabac_hip_status;

typedef struct cabac_hip_ctx cabac_hip_ctx;

/* Worst-case encoded size in bytes of a substream with the given bin counts
 * (a context bin shifts out <= 6 bits, contexts.cpp:787-789; bypass 1;
 * terminate <= 7; finish() <= 3 bytes + alignment).                          */
size_t cabac_hip_encode_bound(uint64_t n_ctx_bins, uint64_t n_ep_bins, uint64_t n_trm_bins);

/* ---- lifetime ----------------------------------------------------- */
/* device: HIP device ordinal.  Fails (CABAC_HIP_ERR_NO_DEVICE) when no GPU is
 * present: there is no CPU fallback in this library.                         */
int cabac_hip_init(int device, cabac_hip_ctx **out);
/* Waits for everything the ctx has in flight (its stream and the copy / kernel streams of the host-pointer entry points),
 * then releases its streams, events, device staging and pinned bounce blocks.  Call it before the process ends: like any
 * HIP resource a ctx must not be released from a static destructor or an atexit handler (the runtime may be gone).      */
void cabac_hip_destroy(cabac_hip_ctx *ctx);
const char *cabac_hip_strerror(int status);
const char *cabac_hip_last_error(const cabac_hip_ctx *ctx);
/* STREAM ORDERING CONTRACT of the device-pointer entry points (*_device).  They only enqueue work on the ctx's stream and
 * return.  A ctx starts with a stream of its own, created hipStreamNonBlocking: work on it is NOT ordered after work the
 * caller has queued elsewhere — not after the null stream either (a hipMemset / hipMemcpyAsync or another library's fill of
 * the very buffers handed in may still be running) — and the caller's later work is not ordered after it.  Either
 *   (a) hand the ctx the stream the buffers are produced and consumed on (cabac_hip_set_stream), or
 *   (b) keep the own stream and order explicitly: record an event behind the producer and cabac_hip_wait_event() it
 *       before the call; cabac_hip_record_event() an event behind the call and make the consumer's stream wait for it
 *       (or cabac_hip_synchronize() from the host).
 * The host-pointer entry points (*_batch) are synchronous: they first wait for the ctx's stream, run on streams of their own
 * and return when the results are in the caller's memory.
 *
 * Adopt an existing HIP stream (hipStream_t passed as void*; NULL = back to a stream of the ctx's own).  The stream stays the
 * caller's: it must outlive the ctx's use of it and is not destroyed by cabac_hip_destroy.
 * The device's DEFAULT (null) stream — whose handle is the null pointer, which here means "the ctx's own" — is adopted by
 * passing CABAC_HIP_STREAM_DEFAULT: a caller whose work runs on the default stream (torch's current stream, unless it was
 * changed, has the handle 0) must pass this and not the handle, or its work and the library's are not ordered at all.   */
#define CABAC_HIP_STREAM_DEFAULT ((void *)1)
int cabac_hip_set_stream(cabac_hip_ctx *ctx, void *hip_stream);
/* hipStreamWaitEvent(ctx stream, event) / hipEventRecord(event, ctx stream); hipEvent_t passed as void*                 */
int cabac_hip_wait_event(cabac_hip_ctx *ctx, void *hip_event);
int cabac_hip_record_event(cabac_hip_ctx *ctx, void *hip_event);
int cabac_hip_synchronize(cabac_hip_ctx *ctx);
/* kernel variant: 0 = the dispatch (by batch size; the fastest verified).  Forcing one geometry — encode: 4, 6, 7; decode: 4 (four
 * substreams per wave), 8 (sixteen per wave), 1 (one per wave) — is for measurements and the parity matrix;
 * any other number makes the next launch fail                                                                            */
int cabac_hip_set_variant(cabac_hip_ctx *ctx, int encode_variant, int decode_variant);

/* ---- device-pointer entry points (asynchronous on the ctx stream) ---
 * Replace: BinEncoderBase::reset/start + TBinEncoder::encodeBin + encodeBinEP +
 * encodeBinTrm + writeOut + finish (arith_codec.cpp:329-357, :367-370, :389-399,
 * :460-478, :524-582) and Ctx::init (contexts.cpp:996-1015, :1133-1145) for a
 * batch of n_sub substreams.  All pointers are device memory.
 * Order: any.  Consecutive descriptors share a wave (decode: one, four or sixteen substreams per wave by batch size; encode: four
 * or sixteen) and a wave runs as long as its longest substream, so a batch of very unequal substreams is coded fastest with
 * its descriptors longest first — which is also what lets the decode dispatch give a few long substreams in front of many
 * short ones a wave each (the shards of sharding.py and of the C++ HipBatch are ordered that way).                        */
int cabac_hip_encode_device(cabac_hip_ctx *ctx, uint32_t n_sub,
                            const cabac_substream_desc *d_desc, const uint16_t *d_records,
                            uint8_t *d_bytes, cabac_substream_result *d_results);

/* Replace: BinDecoderBase::reset/start + TBinDecoder::decodeBin + decodeBinEP +
 * decodeBinTrm + finish (arith_codec.cpp:60-78, :100-114, :181-197, :242-277).
 * d_records supplies the ctxId / EP / TRM sequence (bin bit ignored);
 * d_bins[rec_offset + i] receives the decoded bin (0/1) of record i.         */
int cabac_hip_decode_device(cabac_hip_ctx *ctx, uint32_t n_sub,
                            const cabac_substream_desc *d_desc, const uint16_t *d_records,
                            const uint8_t *d_bytes, uint8_t *d_bins,
                            cabac_substream_result *d_results);

/* Bit estimator.  Replace: TBitEstimator::encodeBin / BitEstimatorBase::encodeBinEP /
 * encodeBinTrm / align / reset / getEstFracBits (arith_codec.cpp:603-711) with
 * BinProbModel_Std::estFracBitsUpdate / estFracBitsTrm and the m_binFracBits table
 * (contexts.cpp:791-878, :922-937) for a batch of n_sub bin strings: d_frac_bits[s]
 * receives the cost of substream s in 1/32768 bit (SCALE_BITS = 15) after
 * reset(qp, init_id & 3); only rec_offset, n_records, qp and init_id of the
 * descriptor are used.  d_flags (may be NULL) receives CABAC_RES_BAD_RECORD or 0;
 * the cost of a substream with a bad record is unspecified.  encodeBinsEP /
 * encodeRemAbsEP cost one bit per bypass bin they expand to (:640-677).        */
int cabac_hip_estimate_device(cabac_hip_ctx *ctx, uint32_t n_sub,
                              const cabac_substream_desc *d_desc, const uint16_t *d_records,
                              uint64_t *d_frac_bits, uint32_t *d_flags);

/* The same from given contexts instead of reset(qp, init_id): replaces the assignment of another coder's
 * contexts to the estimator (Ctx::operator=, contexts.hpp:254, as RDO does before costing candidates) followed
 * by resetBits().  d_state / d_rate hold context sets in the format of cabac_hip_ctx_init_device
 * (379 entries each: m_state[0] | m_state[1] << 16, m_rate); substream s starts from set d_set[s], so many
 * candidate strings can share one start state.  qp / init_id of the descriptors are ignored.            */
int cabac_hip_estimate_from_device(cabac_hip_ctx *ctx, uint32_t n_sub,
                                   const cabac_substream_desc *d_desc, const uint16_t *d_records,
                                   const uint32_t *d_state, const uint8_t *d_rate, const uint32_t *d_set,
                                   uint64_t *d_frac_bits, uint32_t *d_flags);

/* Context-store initialisation only (Ctx::init, contexts.cpp:893-901, :915-920,
 * :996-1015): d_state[(s*379 + k)] = s0 | s1 << 16, d_rate[...] = m_rate for
 * substream s = (qp[s], init_id[s]).  Used by parity tests.                  */
int cabac_hip_ctx_init_device(cabac_hip_ctx *ctx, uint32_t n_sub, const int32_t *d_qp,
                              const uint32_t *d_init_id, uint32_t *d_state, uint8_t *d_rate);

/* Device binariser: syntax-element records -> bin records (the binarisation
 * helpers listed at the SE record format above).  d_se_offset has n_sub+1
 * entries delimiting each substream's SE records.  Pass 1 (d_records == NULL)
 * only writes d_n_records[s]; pass 2 writes the records at d_rec_offset[s].   */
int cabac_hip_binarize_device(cabac_hip_ctx *ctx, uint32_t n_sub, const uint64_t *d_se_offset,
                              const uint32_t *d_se, const uint64_t *d_rec_offset,
                              uint32_t *d_n_records, uint16_t *d_records);

/* ---- residual binariser on the device (encoder side) ----
 * Transform-block coefficients -> the bin records CABACWriter::residual_coding (cabac_writer.cpp:2424-2525)
 * asks its bin encoder for: ts_flag (:2527-2534), last_sig_coeff (:2639-2720) and, per coefficient group in
 * reverse scan order, residual_coding_subblock (:2722-2872) with the context selection of CoeffCodingContext
 * (context_modelling.hpp:71-244, context_modelling.cpp:7-106) and the scans of rom.cpp:148-260.
 * Regular residual coding and transform-skip residual coding (residual_codingTS, cabac_writer.cpp:2874-3046, with
 * BDPCM) and the SBT/MTS zero-out (CABAC_TU_SBT_ZERO_OUT); the range extensions (extended Rice derivation, persistent Rice
 * adaptation, TSRC Rice) are not covered.
 * One block = one cabac_tu_desc; coefficients are int32 (the reference's TCoeff), raster, stride = width.
 * For blocks wider/taller than 32 only the top-left 32x32 region is coded (rom.cpp:218-226). */
typedef struct cabac_tu_desc {
  uint64_t coeff_offset;      /* first coefficient of the block in d_coeff (in coefficients)       */
  uint8_t log2_width;         /* 0..6                                                               */
  uint8_t log2_height;        /* 0..6                                                               */
  uint8_t channel;            /* 0 luma, 1 chroma (toChannelType(compID))                           */
  uint8_t flags;              /* CABAC_TU_*                                                         */
  uint8_t max_log2_tr_range;  /* SPS::getMaxLog2TrDynamicRange, 15 unless extended precision; 0 = 15 */
  uint8_t reserved[3];
} cabac_tu_desc;

#define CABAC_TU_DEP_QUANT 0x1u   /* Slice::getDepQuantEnabledFlag                                   */
#define CABAC_TU_SIGN_HIDING 0x2u /* Slice::getSignDataHidingEnabledFlag                             */
#define CABAC_TU_TS_FLAG 0x4u     /* TU::isTSAllowed(tu, compID): transform_skip_flag is coded (0, or 1 with TRANSFORM_SKIP) */
#define CABAC_TU_TRANSFORM_SKIP 0x8u /* mtsIdx == MTS_SKIP and TS residual coding enabled: residual_codingTS
                                      * (cabac_writer.cpp:2874-3046) instead of the regular walk; blocks up to 32 x 32  */
#define CABAC_TU_BDPCM 0x10u      /* with TRANSFORM_SKIP: cu.bdpcmMode / bdpcmModeChroma != 0                  */
#define CABAC_TU_SBT_ZERO_OUT 0x20u /* SPS::getUseMTS() && cu.sbtInfo != 0, a luma block of at most 32 x 32 that is not transform-skip
                                     * coded: a 32-wide (32-tall) block is coded as if only its left (upper) 16 columns (rows) existed —
                                     * the prefix of last_sig_coeff stops at g_groupIdx[15] (cabac_writer.cpp:2660-2667,
                                     * cabac_reader.cpp:2880-2891), coefficient groups beyond are passed over without a flag (:2507-2516,
                                     * cabac_reader.cpp:2718-2727), and the budget of context-coded bins follows the reduced area
                                     * (TransformUnit::getTbAreaAfterCoefZeroOut, unit.cpp:465-479).  No effect on other block sizes.   */

/* d_info[t] (may be NULL): scanPosLast in bits 15..0, what residual_coding records in its CUCtx argument */
#define CABAC_TU_INFO_LAST_MASK 0xFFFFu
#define CABAC_TU_INFO_TS 0x20000u            /* residual parser: the block was parsed as transform skip (mtsIdx = MTS_SKIP) */
#define CABAC_TU_INFO_MTS_VIOLATION 0x10000u /* a coded luma group with cgPosX > 3 or cgPosY > 3 (cabac_writer.cpp:2519-2522) */
#define CABAC_TU_INFO_EMPTY 0x80000000u      /* all coefficients zero: the reference throws; no records     */
#define CABAC_TU_INFO_BAD_DESC 0x40000000u   /* log2 size > 6 or channel > 1: no records                    */

/* Context-set offsets used by residual coding (Ctx::*, contexts.cpp:77-770) */
#define CABAC_CTX_SIG_COEFF_GROUP(ch) (86u + 2u * (ch))
#define CABAC_CTX_SIG_FLAG(set) ((set) == 0 ? 90u : (set) == 1 ? 102u : (set) == 2 ? 110u : (set) == 3 ? 122u : (set) == 4 ? 130u : 142u)
#define CABAC_CTX_PAR_FLAG(ch) ((ch) ? 171u : 150u)
#define CABAC_CTX_GTX_FLAG(set) ((set) == 0 ? 182u : (set) == 1 ? 203u : (set) == 2 ? 214u : 235u)
#define CABAC_CTX_LAST_X(ch) ((ch) ? 266u : 246u)
#define CABAC_CTX_LAST_Y(ch) ((ch) ? 289u : 269u)
#define CABAC_CTX_TRANSFORM_SKIP_FLAG(ch) (310u + (ch))
#define CABAC_CTX_TS_SIG_COEFF_GROUP 357u
#define CABAC_CTX_TS_SIG_FLAG 360u
#define CABAC_CTX_TS_PAR_FLAG 363u
#define CABAC_CTX_TS_GTX_FLAG 364u
#define CABAC_CTX_TS_LRG1_FLAG 369u
#define CABAC_CTX_TS_RESIDUAL_SIGN 373u

/* Upper bound on the records of one block of n = min(32,w)*min(32,h) coded coefficients:
 * 1 + 2*12 + 8 (ts flag, last position) + per coefficient 4 context bins, a 32-bin escape and a sign,
 * + one group flag per 16.                                                                              */
#define CABAC_TU_MAX_RECORDS(n) (33u + 38u * (n))

/* Pass 1 (d_records == NULL): d_n_records[t] and d_info[t] only.  Pass 2: the records of block t are written
 * at d_records + d_rec_offset[t] (the caller splices them into the substream's record buffer).          */
int cabac_hip_residual_device(cabac_hip_ctx *ctx, uint32_t n_tu, const cabac_tu_desc *d_tu, const int32_t *d_coeff,
                              const uint64_t *d_rec_offset, uint32_t *d_n_records, uint32_t *d_info,
                              uint16_t *d_records);

/* ---- coefficients -> bytes: residual records spliced into the substreams on the device (row f2, the writer's side) ----
 * In the reference the bins of CABACWriter::residual_coding go straight into the bin encoder, in between the bins of the
 * syntax elements around them (cabac_writer.cpp:2424-2525; flags :2766-2803, escapes :2822 / :2843, signs :2871).  Here the
 * caller records the syntax elements it walks itself as bin records (desc[s].rec_offset / n_records, as for
 * cabac_hip_encode_batch) and, for each transform block, a SPLICE: the block's records — exactly those
 * cabac_hip_residual_device produces for tus[tu] — are inserted in front of host record `at` of the substream (at ==
 * n_records: behind the last one; several blocks at the same `at` keep their list order).  Substream s owns
 * splices[splice_first[s] .. splice_first[s + 1]), sorted by `at`; every block of tus[] is spliced exactly once.  The
 * block records exist on the device only: sizes pass -> prefix sums -> records pass straight into the expanded substream ->
 * encode kernel; bytes (and counts) are all that comes back.  Code the transform_skip_flag of a block (ts_flag,
 * cabac_writer.cpp:2527-2534) as an ordinary host record in front of its splice and leave CABAC_TU_TS_FLAG clear.
 * byte_offset / byte_capacity of the descriptors are ignored: the library sizes the byte slots itself (the expanded
 * lengths are only known on the device).                                                                           */
typedef struct cabac_splice {
  uint32_t at; /* 0 .. desc[s].n_records */
  uint32_t tu; /* index into tus[]        */
} cabac_splice;

/* bin_counts (optional): per substream CABAC_BIN_COUNT_WORDS words — the BinCounter totals of everything the substream
 * codes, host records and spliced blocks alike (arith_codec.cpp:281-316): [ctxId] context-coded bins per context,
 * [CABAC_NUM_CONTEXTS] bypass bins, [CABAC_NUM_CONTEXTS + 1] terminate bins.                                        */
#define CABAC_BIN_COUNT_WORDS (CABAC_NUM_CONTEXTS + 2)

/* Device-pointer form.  All inputs and outputs are device memory; intermediate buffers (block sizes, expanded records, byte
 * slots) belong to the ctx and grow on demand.  The coded substreams arrive compacted, in descriptor order, in
 * d_payload[d_payload_offsets[s] .. d_payload_offsets[s + 1]) (n_sub + 1 offsets) — byte-aligned when coded with
 * CABAC_SUB_ALIGN_RBSP, i.e. what OutputBitstream::addSubstream makes of them (bit_stream.cpp:139-150).  d_tu_info (may be
 * NULL): one word per block as cabac_hip_residual_device reports it; d_bin_counts (may be NULL) as above.
 * Unlike the other *_device calls this one waits for the ctx's stream once in the middle (the expanded sizes decide the
 * buffers and launch geometry of the second half); the rest is queued on the stream when it returns.
 * Returns CABAC_HIP_ERR_INVALID for a splice list that is not sorted, points outside its substream or does not name
 * every block exactly once (nothing is coded then).                                                                 */
int cabac_hip_encode_residual_device(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *d_desc,
                                     const uint16_t *d_records, const uint32_t *d_splice_first, const cabac_splice *d_splices,
                                     uint32_t n_splice, uint32_t n_tu, const cabac_tu_desc *d_tu, const int32_t *d_coeff,
                                     uint8_t *d_payload, uint64_t payload_capacity, uint64_t *d_payload_offsets,
                                     cabac_substream_result *d_results, uint32_t *d_tu_info, uint32_t *d_bin_counts);
/* The same with the coefficients as int16_t (tus[].coeff_offset counts int16_t then): for blocks whose dynamic range is 15 bits
 * (max_log2_tr_range 15, every coefficient in [-32768, 32767] — all of the reference's cfgs), half the bytes to move.  What a
 * caller that copies its TCoeff blocks into a staging buffer anyway (the C++ shim does) narrows on the way.               */
int cabac_hip_encode_residual16_device(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *d_desc,
                                       const uint16_t *d_records, const uint32_t *d_splice_first, const cabac_splice *d_splices,
                                       uint32_t n_splice, uint32_t n_tu, const cabac_tu_desc *d_tu, const int16_t *d_coeff,
                                       uint8_t *d_payload, uint64_t payload_capacity, uint64_t *d_payload_offsets,
                                       cabac_substream_result *d_results, uint32_t *d_tu_info, uint32_t *d_bin_counts);

/* Host-pointer form (synchronous): pinned caller memory is DMA'd where it lies, pageable memory goes through the bounce ring.
 * payload / payload_offsets / results / tu_info / bin_counts as above, in host memory.  Returns CABAC_HIP_ERR_SUBSTREAM if a
 * result flag is set or a block is empty / has a bad descriptor (tu_info says which; its splice adds no records).     */
int cabac_hip_encode_batch_residual(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *desc,
                                    const uint16_t *records, uint64_t n_records_total, const uint32_t *splice_first,
                                    const cabac_splice *splices, uint32_t n_tu, const cabac_tu_desc *tus, const int32_t *coeff,
                                    uint64_t n_coeff_total, uint8_t *payload, uint64_t payload_capacity,
                                    uint64_t *payload_offsets, cabac_substream_result *results, uint32_t *tu_info,
                                    uint32_t *bin_counts);
/* ... with int16_t coefficients (see cabac_hip_encode_residual16_device): 2 instead of 4 bytes per coefficient over PCIe      */
int cabac_hip_encode_batch_residual16(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *desc,
                                      const uint16_t *records, uint64_t n_records_total, const uint32_t *splice_first,
                                      const cabac_splice *splices, uint32_t n_tu, const cabac_tu_desc *tus, const int16_t *coeff,
                                      uint64_t n_coeff_total, uint8_t *payload, uint64_t payload_capacity,
                                      uint64_t *payload_offsets, cabac_substream_result *results, uint32_t *tu_info,
                                      uint32_t *bin_counts);

/* ---- substream assembly on the device --------------
 * assemble: concatenate the coded substreams in descriptor order into d_payload — the effect of
 * OutputBitstream::addSubstream (bit_stream.cpp:139-150) on byte-aligned substreams (encode them with
 * CABAC_SUB_ALIGN_RBSP).  d_offsets receives n_sub + 1 byte offsets (the entry points; the last is the
 * total).  Bytes beyond payload_capacity are dropped.
 * split: the inverse (InputBitstream::extractSubstream, bit_stream.cpp:382-415, byte-aligned case):
 * payload[d_offsets[s] .. d_offsets[s+1]) -> d_bytes + desc[s].byte_offset.
 * count_emulations: OutputBitstream::countStartCodeEmulations (bit_stream.cpp:157-181) per substream. */
int cabac_hip_assemble_device(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *d_desc,
                              const cabac_substream_result *d_results, const uint8_t *d_bytes,
                              uint8_t *d_payload, uint64_t payload_capacity, uint64_t *d_offsets);
int cabac_hip_split_device(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *d_desc,
                           const uint64_t *d_offsets, const uint8_t *d_payload, uint8_t *d_bytes);
int cabac_hip_count_emulations_device(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *d_desc,
                                      const cabac_substream_result *d_results, const uint8_t *d_bytes,
                                      uint32_t *d_counts);
/* gather: n_seg runs of bin records, run k = d_src[d_src_off[k] .. + d_len[k]) -> d_dst[d_dst_off[k] ..) (offsets in
 * records; runs must not overlap in d_dst): one launch re-packs the substreams of a batch into the shard of one GPU
 * (the records side of the ordered-concatenation bookkeeping of bit_stream.cpp:139-150, before coding instead of after). */
int cabac_hip_gather_records_device(cabac_hip_ctx *ctx, uint32_t n_seg, const uint64_t *d_src_off, const uint64_t *d_dst_off,
                                    const uint32_t *d_len, const uint16_t *d_src, uint16_t *d_dst);

/* ---- pinned host memory: the device mirrors of the reference's host buffers ------------------------------
 * The reference keeps its byte strings in std::vector FIFOs (OutputBitstream::m_fifo, bit_stream.hpp:16-97;
 * InputBitstream::m_fifo, :103-168; filled by write(), bit_stream.cpp:70-117) and its callers' bin records would live in
 * ordinary heap memory too.  Memory obtained here is page-locked and mapped for the GPU's DMA engines: buffers in it
 * cross PCIe without a staging copy (cabac_hip_encode_batch / _decode_batch detect it, as they detect memory pinned by
 * cabac_hip_host_register or by anybody else, e.g. torch's pin_memory()).  The C++ shim's OutputBitstream /
 * InputBitstream mirrors and the recording encoders allocate from here when EntropyCodingAMD::usePinnedMirrors(true)
 * is set (host/cabac_hip_host.hpp).  No cabac_hip_ctx is needed; fails without a GPU.                              */
int cabac_hip_host_alloc(size_t bytes, void **out);
int cabac_hip_host_free(void *p);                    /* memory from cabac_hip_host_alloc only                      */
int cabac_hip_host_register(void *p, size_t bytes);  /* pin an existing allocation in place (hipHostRegister)      */
int cabac_hip_host_unregister(void *p);
int cabac_hip_host_is_pinned(const void *p, size_t bytes); /* 1 if [p, p + bytes) is DMA-able where it lies         */

/* ---- host-pointer entry points (synchronous) ---------------------------------------------------------------
 * The whole trip host -> device -> host.  A batch whose substreams lie in ascending order in records[] / bytes[] is cut
 * into chunks (two from 2 048 substreams and 16 M records up; the kernels' time is the length of the serial chains and does
 * not shrink with the chunk, so more chunks cost more than their overlap gains); the H2D copy of chunk k+1, the kernel of
 * chunk k and the D2H copy of chunk k-1 run on three kinds of HIP streams of the ctx.  Pinned caller memory (see above) is DMA'd where it
 * lies; pageable memory goes through a ring of pinned 4 MiB blocks inside the ctx, block n+1 being copied by the host
 * while block n is on the wire.  encode: the coded substreams are compacted on the device
 * (cabac_hip_assemble_device's kernels) and leave in one copy per chunk, then are placed at bytes + byte_offset.    */
int cabac_hip_encode_batch(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *desc,
                           const uint16_t *records, uint64_t n_records_total, uint8_t *bytes,
                           uint64_t bytes_total, cabac_substream_result *results);
int cabac_hip_decode_batch(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *desc,
                           const uint16_t *records, uint64_t n_records_total, const uint8_t *bytes,
                           uint64_t bytes_total, uint8_t *bins, cabac_substream_result *results);
/* The same decode with the bins packed eight to a byte: packed_bins holds (n_records_total + 7) / 8 bytes, bit (r & 7) of byte
 * r >> 3 is the bin of record r (r counted in records[], as desc[s].rec_offset does).  An eighth of the bytes come back over PCIe
 * (decodeBin on the host then is a bit test instead of a byte load).                                                      */
int cabac_hip_decode_batch_packed(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *desc,
                                  const uint16_t *records, uint64_t n_records_total, const uint8_t *bytes,
                                  uint64_t bytes_total, uint8_t *packed_bins, cabac_substream_result *results);
/* The same encode with the output as a multiplexer wants it: the coded substreams back to back in descriptor order —
 * what OutputBitstream::addSubstream (bit_stream.cpp:139-150) makes of byte-aligned substreams (code them with
 * CABAC_SUB_ALIGN_RBSP) — in payload[0 .. payload_offsets[n_sub]), substream s at payload_offsets[s] ..
 * payload_offsets[s + 1].  byte_offset / byte_capacity of the descriptors size the device-side slots only; no host pass
 * over the output bytes (cabac_hip_encode_batch places every substream at its byte_offset: one memcpy per substream). */
int cabac_hip_encode_batch_payload(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *desc,
                                   const uint16_t *records, uint64_t n_records_total, uint8_t *payload,
                                   uint64_t payload_capacity, uint64_t *payload_offsets, cabac_substream_result *results);

/* Host-pointer form of cabac_hip_estimate_device (synchronous).  flags may be NULL;
 * returns CABAC_HIP_ERR_SUBSTREAM if any substream had a bad record.            */
int cabac_hip_estimate_batch(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *desc,
                             const uint16_t *records, uint64_t n_records_total,
                             uint64_t *frac_bits, uint32_t *flags);

/* ---- residual parser on the device (decoder side) ------------------------------------
 * CABACReader::residual_coding (cabac_reader.cpp:2647-2735) with ts_flag (:2737-2752), last_sig_coeff (:2865-2938),
 * residual_coding_subblock (:2946-3128), residual_codingTS and residual_coding_subblockTS (:3130-3339) on top of the bin
 * decoder: bytes -> transform-block coefficients.  Unlike cabac_hip_decode_device no bin / context sequence is supplied:
 * every context follows from the coefficients decoded so far; only the geometry of the blocks is given.  Substream s
 * (desc[s]: byte_offset, byte_capacity, qp, init_id | CABAC_SUB_FINISH) holds the blocks d_tile_first[s] ..
 * d_tile_first[s+1]-1 of d_tu in order, then — with CABAC_SUB_FINISH — the terminate bin 1 and the stop pattern, which
 * are checked.  Block t is written to d_coeff + d_tu[t].coeff_offset (int32, raster, stride = width; of a 64-wide/tall
 * block only the coded top-left 32 x 32).
 * Block flags (cabac_tu_desc.flags): DEP_QUANT, SIGN_HIDING as for the binariser; CABAC_TU_TS_FLAG: transform_skip_flag
 * is in the stream (TU::isTSAllowed) and the block is parsed as that bin says, whatever CABAC_TU_TRANSFORM_SKIP says;
 * without it CABAC_TU_TRANSFORM_SKIP decides (the reference infers the flag for BDPCM blocks); CABAC_TU_BDPCM:
 * cu.bdpcmMode / bdpcmModeChroma; CABAC_TU_SBT_ZERO_OUT as for the binariser.  Transform-skip blocks up to 32 x 32.  The range
 * extensions are not covered.
 * d_tu_info[t] (may be NULL): for a regular block scanPosLast | CABAC_TU_INFO_MTS_VIOLATION — what residual_coding
 * leaves in its CUCtx argument (:2675-2693, :2729-2732) follows from it —, for a block parsed as transform skip
 * CABAC_TU_INFO_TS (the reader sets mtsIdx = MTS_SKIP).
 * results[s] = {bits read, CABAC_RES_UNDERRUN | CABAC_RES_BAD_STOP | CABAC_RES_BAD_RECORD (a block the parser does not
 * cover: bad size, transform skip beyond 32 x 32; parsing of the substream stops there)}.                          */
int cabac_hip_residual_parse_device(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *d_desc,
                                    const uint8_t *d_bytes, const uint32_t *d_tile_first, const cabac_tu_desc *d_tu,
                                    int32_t *d_coeff, uint32_t *d_tu_info, cabac_substream_result *d_results);
/* The same with the blocks stored as int16_t (d_tu[t].coeff_offset counts int16_t then): for streams of 15-bit dynamic range.  A
 * level that does not fit is stored truncated and sets CABAC_RES_RANGE in the substream's result.                          */
int cabac_hip_residual_parse16_device(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *d_desc,
                                      const uint8_t *d_bytes, const uint32_t *d_tile_first, const cabac_tu_desc *d_tu,
                                      int16_t *d_coeff, uint32_t *d_tu_info, cabac_substream_result *d_results);

/* Host-pointer form of cabac_hip_residual_parse_device (synchronous).  bytes_total / n_coeff_total bound the two
 * buffers; coeff receives the blocks at tus[t].coeff_offset, tu_info (may be NULL) one word per block.  Returns
 * CABAC_HIP_ERR_SUBSTREAM if any result flag is set. */
int cabac_hip_residual_parse_batch(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *desc, const uint8_t *bytes,
                                   uint64_t bytes_total, const uint32_t *tile_first, const cabac_tu_desc *tus,
                                   int32_t *coeff, uint64_t n_coeff_total, uint32_t *tu_info,
                                   cabac_substream_result *results);
/* ... with the blocks as int16_t (see cabac_hip_residual_parse16_device): half the bytes back over PCIe, and nothing up — coeff is
 * output only here: what the parser does not write (outside the coded top-left 32 x 32 of 64-wide / tall blocks, blocks behind
 * the one at which a substream stops) is zero.                                                                             */
int cabac_hip_residual_parse_batch16(cabac_hip_ctx *ctx, uint32_t n_sub, const cabac_substream_desc *desc, const uint8_t *bytes,
                                     uint64_t bytes_total, const uint32_t *tile_first, const cabac_tu_desc *tus,
                                     int16_t *coeff, uint64_t n_coeff_total, uint32_t *tu_info,
                                     cabac_substream_result *results);

/* Host-pointer form of cabac_hip_residual_device (synchronous, both passes).  `offsets` receives n_tu + 1 record
 * offsets (block t's records are records[offsets[t] .. offsets[t+1])); n_records/info as on the device, info may
 * be NULL.  If `records` is NULL or records_capacity is less than offsets[n_tu], only the sizes are produced
 * (status OK in the first case, CABAC_HIP_ERR_INVALID in the second).  Blocks flagged CABAC_TU_INFO_EMPTY /
 * CABAC_TU_INFO_BAD_DESC produce no records and make the call return CABAC_HIP_ERR_SUBSTREAM.              */
int cabac_hip_residual_batch(cabac_hip_ctx *ctx, uint32_t n_tu, const cabac_tu_desc *tus, const int32_t *coeff,
                             uint64_t n_coeff_total, uint64_t *offsets, uint32_t *info, uint16_t *records,
                             uint64_t records_capacity);

/* ---- per-launch timing (HIP events on the ctx stream) ----------------
 * cabac_hip_profile_enable(ctx, capacity): from now on every device call (encode, decode, binarize, ...) is
 * bracketed by its own pair of HIP events on the stream it is launched on (up to `capacity` calls;
 * 0 disables and frees).  cabac_hip_profile_read synchronises the stream, writes kind[i]
 * (0 encode, 1 decode, 2 binarize, 3 ctx_init, 4 estimate, 5 residual, 6 assemble, 7 split, 8 count_emulations, 9 residual_parse,
 * 10 splice plan / scan / expand, 11 bin counts) and ms[i] for the recorded calls in launch order,
 * returns their number and resets the ring.                                 */
int cabac_hip_profile_enable(cabac_hip_ctx *ctx, uint32_t capacity);
int cabac_hip_profile_read(cabac_hip_ctx *ctx, int32_t *kind, float *ms, uint32_t max_entries);

/* ---- timing of the last device call (HIP events on the ctx stream) --- */
/* milliseconds between the events that bracket the kernel(s) of the last
 * encode/decode/binarize device call; <0 if unavailable.  Synchronises.      */
float cabac_hip_last_kernel_ms(cabac_hip_ctx *ctx);

/* ---- synthetic workload generator (host, deterministic) ---------------
 * The residual-heavy bin mix of SURVEY.md §8(d): SplitMix64 seeded with
 * seed ^ substream_index; ctx_permille of the bins context coded (60 % of them
 * from ctxIds 90..245, 15 % from 246..291, 25 % uniform over the rest, each
 * context with a fixed P(1) out of {0.03,0.1,0.25,0.5,0.75,0.9}); the rest fair
 * bypass bins; the last record is TRM(1).  Writes n_bins records.            */
void cabac_synth_records(uint64_t seed, uint64_t substream_index, uint32_t n_bins,
                         uint32_t ctx_permille, uint16_t *out_records);

#ifdef __cplusplus
}
#endif
#endif /* CABAC_HIP_H */
