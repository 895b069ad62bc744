// TEST INFRASTRUCTURE.  Runs the reference's OWN CABACWriter (cabac_writer.hpp) on top of
// BinEncoderHipRef (the GPU recording encoder) and, side by side, on top of the reference's
// BinEncoder_Std, from the same op stream; returns both byte strings so the test can compare them.
// Built by oracle/Makefile into oracle/_ref/libadapter_test.so (needs the reference headers and
// objects, so it is built in the build container only; the .so travels to the GPU box).
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#define private public
#define protected public
#include "cabac_reader.hpp"
#define class struct  // Logger's stream is an implicitly private member (log.hpp:131-132); adapter_log_mark() flushes it
#include "log.hpp"
#undef class
#include "cabac_writer.hpp"
#undef private
#undef protected

// BinDecoderBase's only constructor is a member template defined in arith_codec.cpp:50-52; the reference instantiates it
// implicitly (inlined into TBinDecoder's constructor, no symbol is exported), so a class deriving from BinDecoderBase
// outside that file cannot link.  A reference tree that adopts BinDecoderHipRef adds one explicit instantiation line to
// arith_codec.cpp; this test library, which links the unmodified reference, supplies the same
// constructor as an explicit specialisation instead.
namespace EntropyCoding {
template <>
BinDecoderBase::BinDecoderBase(const Common::BinProbModel_Std *dummy)
    : Common::Ctx(dummy), m_Bitstream(nullptr), m_Range(0), m_Value(0), m_bitsNeeded(0) {}
}  // namespace EntropyCoding

#include "reference_adapter.hpp"
#include "ref_rig.hpp"  // oracle/: test rig around TransformUnit

using namespace EntropyCoding;
using namespace Common;

enum { OP_BIN = 0, OP_EP, OP_BINS_EP, OP_REM_ABS, OP_TRM, OP_ALIGN, OP_UNARY_MAX, OP_UNARY_EP, OP_EXP_GOLOMB, OP_TRUNC_BIN };

static thread_local char g_err[512];

static void drive(CABACWriter &w, BinEncIf &e, const uint32_t *ops, long n) {
  for (long i = 0; i < n; i++) {
    const uint32_t *o = ops + 4 * i;
    switch (o[0]) {
    case OP_BIN: e.encodeBin(o[1], o[2]); break;
    case OP_EP: e.encodeBinEP(o[1]); break;
    case OP_BINS_EP: e.encodeBinsEP(o[1], o[2]); break;
    case OP_REM_ABS: e.encodeRemAbsEP(o[1], o[2], o[3] & 0xff, (int)(o[3] >> 8)); break;
    case OP_TRM: e.encodeBinTrm(o[1]); break;
    case OP_ALIGN: e.align(); break;
    case OP_UNARY_MAX: w.unary_max_symbol(o[1], o[2] & 0xffff, o[2] >> 16, o[3]); break;  // reference's binarisers
    case OP_UNARY_EP: w.unary_max_eqprob(o[1], o[2]); break;
    case OP_EXP_GOLOMB: w.exp_golomb_eqprob(o[1], o[2]); break;
    case OP_TRUNC_BIN: w.xWriteTruncBinCode(o[1], o[2]); break;
    }
  }
}

static long dump(OutputBitstream &bs, uint8_t *out, long cap, uint32_t *nbits) {
  *nbits = bs.getNumberOfWrittenBits();
  long n = (long)bs.getFIFO().size(), total = n + ((*nbits & 7) ? 1 : 0);
  if (total > cap) return -3;
  if (n) memcpy(out, bs.getFIFO().data(), n);
  if (*nbits & 7) out[n] = bs.getHeldBits();
  return total;
}

extern "C" {
const char *adapter_last_error() { return g_err; }

// which = 0: reference BinEncoder_Std; 1: BinEncoderHipRef (GPU).  Both under the reference CABACWriter,
// ending with the reference's end_of_slice() (TRM(1) + finish()) and VTM's writeByteAlignment().
long adapter_encode(int which, const uint32_t *ops, long n_ops, int qp, int initId, uint8_t *out, long cap,
                    uint32_t *nbits, uint32_t *numBins) {
  try {
    OutputBitstream bs;
    if (which == 0) {
      BinEncoder_Std enc;
      CABACWriter w(enc);
      w.initBitstream(&bs);
      enc.reset(qp, initId);
      drive(w, enc, ops, n_ops);
      w.end_of_slice();
      *numBins = w.getNumBins();
    } else {
      EntropyCodingAMD::HipBatch batch(0);
      EntropyCodingAMD::BinEncoderHipRef enc(batch);
      CABACWriter w(enc);
      w.initBitstream(&bs);
      enc.reset(qp, initId);
      drive(w, enc, ops, n_ops);
      w.end_of_slice();
      *numBins = w.getNumBins();
      batch.flush();
    }
    bs.writeByteAlignment();
    return dump(bs, out, cap, nbits);
  } catch (std::exception &e) {
    strncpy(g_err, e.what(), sizeof g_err - 1);
    return -1;
  }
}

// getNumWrittenBits() (arith_codec.cpp:482-485) asked after every `every`-th op, on the reference's BinEncoder_Std (which 0) and
// on BinEncoderHipRef in Immediate mode (which 1: one probing launch per question); the bitstream already holds `lead_bits`
// bits, as a substream's does after earlier ones.  Returns the number of answers.
long adapter_num_written_bits(int which, const uint32_t *ops, long n_ops, int qp, int initId, int every, int lead_bits,
                              uint32_t *answers, long cap) {
  try {
    OutputBitstream bs;
    if (lead_bits) bs.write((1u << lead_bits) - 1u, lead_bits);
    EntropyCodingAMD::HipBatch batch(0);
    BinEncoder_Std std_enc;
    EntropyCodingAMD::BinEncoderHipRef hip_enc(batch, EntropyCodingAMD::BinEncoderHipRef::Immediate);
    BinEncIf &e = which == 0 ? static_cast<BinEncIf &>(std_enc) : static_cast<BinEncIf &>(hip_enc);
    CABACWriter w(e);
    w.initBitstream(&bs);
    e.reset(qp, initId);
    long n = 0;
    for (long i = 0; i < n_ops; i++) {
      drive(w, e, ops + 4 * i, 1);
      if ((i + 1) % every == 0 || i + 1 == n_ops) {
        if (n >= cap) return -3;
        answers[n++] = e.getNumWrittenBits();
      }
    }
    return n;
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof g_err - 1);
    return -1;
  }
}

// CPU only: the bin store of the window-size training path (arith_codec.cpp:585-601) under the reference's CABACWriter: with
// setBinStorage(true) BinEncoder_Std (which 0) and BinEncoderHipRef (which 1) must hold the same bins per context, and
// getTestBinEncoder() a fresh encoder.  out[ctx] = a digest of that context's bin vector (length * 2654435761 + bits folded).
long adapter_bin_store(int which, const uint32_t *ops, long n_ops, uint32_t *out, int *has_test_encoder) {
  try {
    EntropyCodingAMD::HipBatch batch(0);
    BinEncoder_Std std_enc;
    EntropyCodingAMD::BinEncoderHipRef hip_enc(batch);
    BinEncIf &e = which == 0 ? static_cast<BinEncIf &>(std_enc) : static_cast<BinEncIf &>(hip_enc);
    OutputBitstream bs;
    CABACWriter w(e);
    w.initBitstream(&bs);
    e.setBinStorage(true);
    e.reset(30, 2);
    drive(w, e, ops, n_ops);
    const BinStore *store = e.getBinStore();
    if (!store || !store->inUse()) return -4;
    long total = 0;
    for (unsigned c = 0; c < Ctx::NumberOfContexts; c++) {
      const std::vector<bool> &v = store->getBinVector(c);
      uint32_t h = uint32_t(v.size()) * 2654435761u;
      for (size_t i = 0; i < v.size(); i++) h = (h << 1 | h >> 31) ^ (v[i] ? 0x9E3779B9u : 0x7F4A7C15u);
      out[c] = h;
      total += long(v.size());
    }
    BinEncIf *t = e.getTestBinEncoder();
    *has_test_encoder = t != nullptr;
    delete t;
    e.setBinStorage(false);
    BinEncIf *none = e.getTestBinEncoder();
    if (none) { delete none; return -5; }
    return total;
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof g_err - 1);
    return -1;
  }
}

// CPU only: what the adapter recorded under the reference's CABACWriter
long adapter_record(const uint32_t *ops, long n_ops, uint16_t *rec, long cap, uint32_t *numBins) {
  try {
    EntropyCodingAMD::HipBatch batch(0);
    EntropyCodingAMD::BinEncoderHipRef enc(batch);
    OutputBitstream bs;
    CABACWriter w(enc);
    w.initBitstream(&bs);
    enc.reset(32, 2);
    drive(w, enc, ops, n_ops);
    *numBins = w.getNumBins();
    long n = (long)enc.records().size();
    if (n > cap) return -3;
    if (n) memcpy(rec, enc.records().data(), n * 2);
    return n;
  } catch (std::exception &e) {
    strncpy(g_err, e.what(), sizeof g_err - 1);
    return -1;
  }
}
// Bit estimator: the reference's CABACWriter on the reference's BitEstimator_Std (which = 0) or on
// BitEstimatorHipRef (which = 1, GPU); the op stream is applied in segments with resetBits() (kind 0), start()
// (1) or restart() (2) before every segment but the first; costs[i] = getEstFracBits() after segment i.
long adapter_estimate(int which, const uint32_t *ops, const long *seg_end, const int *seg_kind, int n_seg, int qp,
                      int initId, uint64_t *costs) {
  try {
    EntropyCodingAMD::HipBatch batch(0);
    BitEstimator_Std ref_est;
    EntropyCodingAMD::BitEstimatorHipRef hip_est(batch);
    BinEncIf &e = which == 0 ? static_cast<BinEncIf &>(ref_est) : static_cast<BinEncIf &>(hip_est);
    CABACWriter w(e);
    e.reset(qp, initId);
    long begin = 0;
    for (int i = 0; i < n_seg; i++) {
      if (i > 0) {
        if (seg_kind[i] == 0) e.resetBits();
        else if (seg_kind[i] == 1) e.start();
        else e.restart();
      }
      drive(w, e, ops + 4 * begin, seg_end[i] - begin);
      begin = seg_end[i];
      costs[i] = e.getEstFracBits();
    }
    return 0;
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// Residual coding: n blocks (sizes wh[2i], wh[2i+1], component comp[i], coefficients back to back in coeff) coded into
// ONE substream, so that the contexts adapt from block to block, (which = 0) by the reference's
// CABACWriter::residual_coding on BinEncoder_Std, (which = 1) by ResidualCoderHipRef::residual_coding block by block,
// (which = 2) by ResidualCoderHipRef::queue for all blocks and one flush(); then TRM(1), finish.  out = the bytes;
// cu[0..4] = what the calls left in one shared CUCtx.  rig_flags as ref_residual_records (oracle/ref_harness.cpp).
long adapter_residual(int which, int n, const int *wh, const int *comp, int rig_flags, const int32_t *coeff, int qp,
                      uint8_t *out, long cap, int32_t *cu) {
  try {
    static ResidualRig rig;
    EntropyCodingAMD::HipBatch batch(0);
    BinEncoder_Std enc;
    OutputBitstream bs;
    CABACWriter w(enc);
    w.initBitstream(&bs);
    enc.reset(qp, 2);
    EntropyCodingAMD::ResidualCoderHipRef hip(batch, enc);
    CUCtx cuCtx(0);
    std::vector<std::unique_ptr<TransformUnit>> tus;
    std::vector<std::vector<TCoeff>> bufs(n);
    const int32_t *c = coeff;
    for (int i = 0; i < n; i++) {
      tus.emplace_back(new TransformUnit);
      rig.make_tu(*tus[i], bufs[i], wh[2 * i], wh[2 * i + 1], comp[i], rig_flags, c);
      c += wh[2 * i] * wh[2 * i + 1];
      if (which == 0) w.residual_coding(*tus[i], ComponentID(comp[i]), &cuCtx);
      else if (which == 1) hip.residual_coding(*tus[i], ComponentID(comp[i]), &cuCtx);
      else hip.queue(*tus[i], ComponentID(comp[i]), &cuCtx);
    }
    if (which == 2) hip.flush();
    static_cast<BinEncIf &>(enc).encodeBinTrm(1);
    static_cast<BinEncIf &>(enc).finish();
    bs.writeByteAlignment();
    cu[0] = int(cuCtx.violatesLfnstConstrained[CHANNEL_TYPE_LUMA]) | int(cuCtx.violatesLfnstConstrained[CHANNEL_TYPE_CHROMA]) << 1;
    cu[1] = cuCtx.lfnstLastScanPos;
    cu[2] = cuCtx.violatesMtsCoeffConstraint;
    cu[3] = cuCtx.mtsLastScanPos;
    const std::vector<uint8_t> &f = bs.getFIFO();
    if ((long)f.size() > cap) { strcpy(g_err, "capacity"); return -3; }
    if (!f.empty()) memcpy(out, f.data(), f.size());
    return (long)f.size();
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// The same blocks, each preceded by a few other syntax elements (ops[op_off[i] .. op_off[i + 1]) before block i, the rest after
// the last one), coded (which = 0) by the reference's CABACWriter on BinEncoder_Std and (which = 1) by the reference's
// CABACWriter on BinEncoderHipRef with ResidualCoderHipRef in its splice form: the coefficients go to the device, the block
// bins never exist on the host, one HipBatch::flush() codes the substream.  cu[0..3] as adapter_residual, read BEFORE the
// flush (a writer decides lfnst_idx / mts_idx from them while it walks); counts = {getNumBins(), getNumBins(ctx) for the
// contexts listed in probe_ctx[0..7]} read after it.
long adapter_residual_spliced(int which, int n, const int *wh, const int *comp, int rig_flags, const int32_t *coeff,
                              const uint32_t *ops, const long *op_off, int qp, uint8_t *out, long cap, int32_t *cu,
                              const int *probe_ctx, uint32_t *counts) {
  try {
    static ResidualRig rig;
    EntropyCodingAMD::HipBatch batch(0);
    BinEncoder_Std std_enc;
    EntropyCodingAMD::BinEncoderHipRef hip_enc(batch);
    BinEncIf &e = which == 0 ? static_cast<BinEncIf &>(std_enc) : static_cast<BinEncIf &>(hip_enc);
    OutputBitstream bs;
    CABACWriter w(e);
    w.initBitstream(&bs);
    e.reset(qp, 2);
    EntropyCodingAMD::ResidualCoderHipRef hip(batch, hip_enc);
    CUCtx cuCtx(0);
    std::vector<std::unique_ptr<TransformUnit>> tus;
    std::vector<std::vector<TCoeff>> bufs(n);
    const int32_t *c = coeff;
    for (int i = 0; i < n; i++) {
      drive(w, e, ops + 4 * op_off[i], op_off[i + 1] - op_off[i]);
      tus.emplace_back(new TransformUnit);
      rig.make_tu(*tus[i], bufs[i], wh[2 * i], wh[2 * i + 1], comp[i], rig_flags, c);
      c += wh[2 * i] * wh[2 * i + 1];
      if (which == 0) w.residual_coding(*tus[i], ComponentID(comp[i]), &cuCtx);
      else hip.queue(*tus[i], ComponentID(comp[i]), &cuCtx);
    }
    drive(w, e, ops + 4 * op_off[n], op_off[n + 1] - op_off[n]);
    cu[0] = int(cuCtx.violatesLfnstConstrained[CHANNEL_TYPE_LUMA]) | int(cuCtx.violatesLfnstConstrained[CHANNEL_TYPE_CHROMA]) << 1;
    cu[1] = cuCtx.lfnstLastScanPos;
    cu[2] = cuCtx.violatesMtsCoeffConstraint;
    cu[3] = cuCtx.mtsLastScanPos;
    e.encodeBinTrm(1);
    e.finish();
    batch.flush();
    bs.writeByteAlignment();
    counts[0] = e.getNumBins();
    for (int k = 0; k < 8; k++) counts[1 + k] = static_cast<const BinEncIf &>(e).getNumBins(unsigned(probe_ctx[k]));
    const std::vector<uint8_t> &f = bs.getFIFO();
    if ((long)f.size() > cap) { strcpy(g_err, "capacity"); return -3; }
    if (!f.empty()) memcpy(out, f.data(), f.size());
    return (long)f.size();
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// ---- decoder side -----------------------------------------------------------------------------------------------------
// n residual blocks of one substream, parsed (which = 0) by the reference's own CABACReader::residual_coding on
// BinDecoder_Std, (which = 1) by ResidualParserHipRef on the device.  block_flags: rig flags per block
// (oracle/ref_rig.hpp).  coeff_out: the TransformUnits' coefficient buffers afterwards; tu_out[i] = mtsIdx of block i;
// cu[0..3] = one CUCtx shared by all blocks, as a reader's walk over a coding unit shares it.
long adapter_residual_parse(int which, int n, const int *wh, const int *comp, const int *block_flags, const uint8_t *in,
                            long n_in, int qp, int32_t *coeff_out, int32_t *tu_out, int32_t *cu) {
  try {
    static ResidualRig rig;
    InputBitstream ib;
    ib.getFifo().assign(in, in + n_in);
    CUCtx cuCtx(0);
    std::vector<std::unique_ptr<TransformUnit>> tus;
    std::vector<std::vector<TCoeff>> bufs(n);
    EntropyCodingAMD::HipBatch batch(0);
    BinDecoder_Std dec;
    CABACReader r(dec);
    EntropyCodingAMD::ResidualParserHipRef hip(batch);
    // The rig's slice / SPS / CU are shared objects that make_tu sets per block, and both parsers read them while they
    // parse (the reader at once, the GPU parser's queue call at once too): every block is made right before its call.
    if (which == 0) {
      r.initBitstream(&ib);
      dec.reset(qp, 2);
    } else {
      hip.beginSubstream(&ib, qp, 2);
    }
    for (int i = 0; i < n; i++) {
      const int w = wh[2 * i], h = wh[2 * i + 1];
      std::vector<int32_t> zeros((size_t)w * h, 0);
      tus.emplace_back(new TransformUnit);
      rig.make_tu(*tus[i], bufs[i], w, h, comp[i], block_flags[i], zeros.data());
      if (which == 0) r.residual_coding(*tus[i], ComponentID(comp[i]), cuCtx);
      else hip.residual_coding(*tus[i], ComponentID(comp[i]), cuCtx);
    }
    if (which == 0) {
      if (dec.decodeBinTrm() != 1) { strcpy(g_err, "terminate bin is not 1"); return -5; }
      dec.finish();
    } else {
      hip.endSubstream();
    }
    int32_t *out = coeff_out;
    for (int i = 0; i < n; i++) {
      for (size_t k = 0; k < bufs[i].size(); k++) out[k] = (int32_t)bufs[i][k];
      out += bufs[i].size();
      tu_out[i] = tus[i]->mtsIdx[comp[i]];
    }
    cu[0] = int(cuCtx.violatesLfnstConstrained[CHANNEL_TYPE_LUMA]) | int(cuCtx.violatesLfnstConstrained[CHANNEL_TYPE_CHROMA]) << 1;
    cu[1] = cuCtx.lfnstLastScanPos;
    cu[2] = cuCtx.violatesMtsCoeffConstraint;
    cu[3] = cuCtx.mtsLastScanPos;
    return (long)ib.getByteLocation();
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// The supplied-ctxId decoder behind the reference's BinDecoderBase: a record stream is planned, run on the device and
// served — the context bins through a `BinDecoderBase &` (decodeBin is the interface's one virtual), bypass / terminate
// bins through the class itself.  which = 0 decodes the same with BinDecoder_Std for comparison.
long adapter_decode_replay(int which, const uint16_t *rec, long n, int qp, int initId, const uint8_t *in, long n_in,
                           uint8_t *bins, uint32_t *fifo_idx_after) {
  try {
    InputBitstream ib;
    ib.getFifo().assign(in, in + n_in);
    if (which == 0) {
      BinDecoder_Std dec;
      dec.init(&ib);
      dec.reset(qp, initId);
      for (long i = 0; i < n; i++) {
        const unsigned id = rec[i] & CABAC_REC_ID_MASK;
        bins[i] = uint8_t(id < CABAC_NUM_CONTEXTS ? dec.decodeBin(id) : id == CABAC_REC_EP ? dec.decodeBinEP() : dec.decodeBinTrm());
      }
      dec.finish();
    } else {
      EntropyCodingAMD::HipBatch batch(0);
      EntropyCodingAMD::BinDecoderHipRef dec(batch);
      BinDecoderBase &base = dec;
      base.init(&ib);
      dec.reset(qp, initId);
      dec.planRecords(rec, size_t(n));
      dec.run(true);
      for (long i = 0; i < n; i++) {
        const unsigned id = rec[i] & CABAC_REC_ID_MASK;
        bins[i] = uint8_t(id < CABAC_NUM_CONTEXTS ? base.decodeBin(id) : id == CABAC_REC_EP ? dec.decodeBinEP() : dec.decodeBinTrm());
      }
    }
    *fifo_idx_after = ib.getByteLocation();
    return 0;
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// ---- a syntax walk with its bin_log.txt ---------------------------------------------------------
// The reference's own CABACWriter walks n_sub substreams; an item of a substream is one coding unit's worth of syntax
// that needs nothing but the rig: mvd_coding (cabac_writer.cpp:2152-2210), cu_qp_delta (:2356-2379), cu_chroma_qp_offset
// (:2381-2400) and residual_coding of one transform block (:2424-2525), each of which writes its bin_log.txt lines through
// binLogger (log.hpp:131-168) before its bins; a substream ends with end_of_slice() (:104-107) and VTM's
// writeByteAlignment().  which = 0: on the reference's BinEncoder_Std; 1: on BinEncoderHipRef, all substreams coded by one
// HipBatch::flush() on the device; 2: on BinEncoderHipRef, recording only (no device) — rec / rec_off receive the bin
// records of every substream.  items: 8 ints each {width, height, comp, rig flags, mvd hor, mvd ver, pred QP, CU QP}.
// When this file is built against the reference compiled with ENABLE_LOGGING, adapter_log_mark() flushes bin_log.txt
// (created in the working directory the library was loaded in) and returns its length so far.
long adapter_log_mark() {
  binLogger.fs.flush();
  return (long)binLogger.fs.tellp();
}

long adapter_walk(int which, int n_sub, const int *sub_first, const int *qp, const int32_t *items, const int32_t *coeff,
                  uint8_t *out, long cap, long *out_off, uint32_t *nbits, uint16_t *rec, long rec_cap, long *rec_off) {
  try {
    static ResidualRig rig;
    EntropyCodingAMD::HipBatch batch(0);
    std::vector<OutputBitstream> bs(n_sub);
    std::vector<std::unique_ptr<BinEncIf>> encs;
    std::vector<long> coeff_at(size_t(sub_first[n_sub]) + 1, 0);
    for (int i = 0; i < sub_first[n_sub]; i++) coeff_at[i + 1] = coeff_at[i] + long(items[8 * i]) * items[8 * i + 1];
    for (int s = 0; s < n_sub; s++) {
      if (which == 0) encs.emplace_back(new BinEncoder_Std);
      else encs.emplace_back(new EntropyCodingAMD::BinEncoderHipRef(batch));
      BinEncIf &enc = *encs.back();
      CABACWriter w(enc);
      w.initBitstream(&bs[s]);
      enc.reset(qp[s], 2);
      CUCtx cuCtx(0);
      for (int i = sub_first[s]; i < sub_first[s + 1]; i++) {
        const int32_t *it = items + 8 * i;
        w.mvd_coding(Mv(it[4], it[5]), 0);
        w.cu_qp_delta(*rig.cu, it[6], int8_t(it[7]));
        w.cu_chroma_qp_offset(*rig.cu);
        std::vector<TCoeff> buf;
        TransformUnit tu;
        rig.make_tu(tu, buf, it[0], it[1], it[2], it[3], coeff + coeff_at[i]);
        w.residual_coding(tu, ComponentID(it[2]), &cuCtx);
      }
      w.end_of_slice();
    }
    if (which == 2) {
      const auto &pend = batch.pendingSubstreams();
      long at = 0;
      for (int s = 0; s < n_sub; s++) {
        rec_off[s] = at;
        const auto &r = pend[s].records;
        if (at + (long)r.size() > rec_cap) return -3;
        if (!r.empty()) memcpy(rec + at, r.data(), r.size() * 2);
        at += (long)r.size();
      }
      rec_off[n_sub] = at;
      return 0;
    }
    if (which == 1) batch.flush();
    long at = 0;
    for (int s = 0; s < n_sub; s++) {
      bs[s].writeByteAlignment();
      out_off[s] = at;
      const long n = dump(bs[s], out + at, cap - at, &nbits[s]);
      if (n < 0) return -3;
      at += n;
    }
    out_off[n_sub] = at;
    return 0;
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

} // extern "C"
