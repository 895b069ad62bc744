// DROP-IN ADAPTER — compiled only where the reference's headers are available
// (-I<reference>/src/common -I<reference>/src/entropy_codec -I<reference>/src/log).
//
// BinEncoderHipRef IS-A EntropyCoding::BinEncIf (reference arith_codec.hpp:31-70), so the reference's
// own CABACWriter (cabac_writer.hpp:14-184) runs on top of it unchanged:
//
//     EntropyCodingAMD::HipBatch batch(device);
//     EntropyCodingAMD::BinEncoderHipRef enc(batch);          // instead of BinEncoder_Std
//     EntropyCoding::CABACWriter writer(enc);                 // reference class, unmodified
//     writer.initBitstream(&substream); writer.initCtxModels(slice); ... writer.end_of_slice();
//     batch.flush();                                          // once per picture / batch of substreams
//     substream.writeByteAlignment();                         // as VTM does after end_of_slice
//
// Every encode* call is recorded as 16-bit bin records; finish() queues the substream; flush()
// codes all queued substreams in one GPU launch and appends the bytes to the reference's
// Common::OutputBitstream exactly as BinEncoderBase::finish() would have (FIFO + held bits).
// No reference source is copied: this file only derives from its interface.
#ifndef CABAC_HIP_REFERENCE_ADAPTER_HPP
#define CABAC_HIP_REFERENCE_ADAPTER_HPP

#include "arith_codec.hpp"  // reference
#include "coding_structure.hpp"  // reference (TransformUnit, CodingStructure)
#include "context_modelling.hpp"  // reference (CUCtx)
#include "unit_tools.hpp"  // reference (TU::isTSAllowed)
#include "cabac_hip_host.hpp"
#include "cabac_rem_abs.hpp"

// the reference's CHECK/THROW name `Exception` unqualified (type_def.hpp:319-326); inside this
// namespace that would pick EntropyCodingAMD::Exception, so throw the reference's type explicitly
#define HIPREF_THROW(msg) throw Common::Exception(std::string("\nERROR: ") + (msg))
#define HIPREF_CHECK(c, msg) \
  if (c) HIPREF_THROW(msg)

namespace EntropyCodingAMD {

class BinEncoderHipRef : public EntropyCoding::BinEncIf, public EntropyCoding::BinCounter {
public:
  enum Mode { Deferred, Immediate };
  explicit BinEncoderHipRef(HipBatch &batch, Mode mode = Deferred)
      : EntropyCoding::BinEncIf(static_cast<const Common::BinProbModel_Std *>(nullptr)), m_batch(batch), m_mode(mode) {}

  void init(Common::OutputBitstream *bitstream) override { m_Bitstream = bitstream; }
  void uninit() override { m_Bitstream = nullptr; }
  void start() override {
    m_records.clear();
    m_splices.clear();
    m_blocks.clear();
    EntropyCoding::BinCounter::reset();
    std::fill(m_devCtx.begin(), m_devCtx.end(), 0u);
    m_BinStore.reset();  // arith_codec.cpp:336
  }
  void finish() override {
    HIPREF_CHECK(!m_Bitstream, "finish() without a bitstream");
    HipBatch::Pending p;
    p.records.swap(m_records);
    p.qp = m_qp;
    p.initId = m_initId;
    p.nEp = EntropyCoding::BinCounter::getEP();
    p.nTrm = EntropyCoding::BinCounter::getTrm();
    p.nCtx = EntropyCoding::BinCounter::getAll() - p.nEp - p.nTrm;
    Common::OutputBitstream *bs = m_Bitstream;
    p.deliver = [bs](const uint8_t *bytes, uint32_t whole, uint32_t tail_bits) {
      for (uint32_t i = 0; i < whole; i++) bs->write(bytes[i], 8);  // same calls writeOut()/finish() make
      if (tail_bits) bs->write(uint32_t(bytes[whole]) >> (8 - tail_bits), tail_bits);
    };
    if (!m_blocks.empty()) {  // residual blocks to be spliced in on the device (spliceResidual)
      p.splices.swap(m_splices);
      p.blocks.swap(m_blocks);
      p.hostCounts.assign(CABAC_BIN_COUNT_WORDS, 0u);
      for (unsigned k = 0; k < CABAC_NUM_CONTEXTS; k++) p.hostCounts[k] = EntropyCoding::BinCounter::getCtx(k);
      p.hostCounts[CABAC_NUM_CONTEXTS] = uint32_t(p.nEp);
      p.hostCounts[CABAC_NUM_CONTEXTS + 1] = uint32_t(p.nTrm);
      BinEncoderHipRef *self = this;  // must outlive the flush (as the bitstream must)
      p.counted = [self](const uint32_t *counts, const uint32_t *host) {
        // the blocks' bins, known now: the reference's BinCounter keeps its per-context array private, so the context bins are
        // kept here and added in getNumBins() / getNumBins(ctxId); the bypass bins go into the base class
        for (unsigned k = 0; k < CABAC_NUM_CONTEXTS; k++) self->m_devCtx[k] += counts[k] - host[k];
        self->EntropyCoding::BinCounter::addEP(counts[CABAC_NUM_CONTEXTS] - host[CABAC_NUM_CONTEXTS]);
      };
      std::vector<std::function<void(uint32_t)>> cbs;
      cbs.swap(m_blockInfo);
      p.blockInfo = [cbs](size_t k, uint32_t info) {
        if (k < cbs.size() && cbs[k]) cbs[k](info);
      };
    }
    m_batch.submit(std::move(p));
    if (m_mode == Immediate) m_batch.flush();
  }
  // The bins CABACWriter::residual_coding would put into this encoder for block `b` (cabac_writer.cpp:2424-2525), without
  // binarising on the host: the coefficients are staged for the device and a splice marks the place of the block's bins
  // among the recorded ones (ResidualCoderHipRef does this for a TransformUnit).  onInfo (optional) receives the block's
  // scanPosLast | CABAC_TU_INFO_* at flush().
  void spliceResidual(const HipBatch::ResidualBlock &b, std::function<void(uint32_t)> onInfo = nullptr) {
    HIPREF_CHECK(!b.coeff || b.tsFlag, "spliceResidual: code transform_skip_flag with encodeBin first; coefficients required");
    HIPREF_CHECK(m_BinStore.inUse(), "spliceResidual: the bin store needs the block's bins on the host (ResidualCoderHipRef replays them then)");
    cabac_tu_desc t;
    try {
      t = makeTuDesc(b, 0);
    } catch (const EntropyCodingAMD::Exception &e) {
      HIPREF_THROW(e.what());
    }
    t.coeff_offset = m_batch.stageCoefficients(b.coeff, size_t(b.width) * b.height, b.maxLog2TrDynamicRange);
    m_splices.push_back(cabac_splice{uint32_t(m_records.size()), uint32_t(m_blocks.size())});
    m_blocks.push_back(t);
    m_blockInfo.push_back(std::move(onInfo));
  }
  void restart() override { HIPREF_CHECK(!m_records.empty(), "restart() on a non-empty recording"); }
  void reset(int qp, int initId) override {
    Common::Ctx::init(qp, initId);  // keeps the host-visible context store / GRAdaptStats as the reference has them
    m_qp = qp;
    m_initId = initId;
    start();
  }
  void resetBits() override {
    HIPREF_CHECK(!m_records.empty(), "resetBits() on a non-empty recording");
    EntropyCoding::BinCounter::reset();
  }
  uint64_t getEstFracBits() const override { HIPREF_THROW("not supported"); }
  // (the bins of spliced residual blocks are in these numbers from the flush() that codes them on)
  unsigned getNumBins(unsigned ctxId) const override { return EntropyCoding::BinCounter::getCtx(ctxId) + m_devCtx[ctxId]; }

  void encodeBin(unsigned bin, unsigned ctxId) override {
    HIPREF_CHECK(ctxId >= CABAC_NUM_CONTEXTS, "ctxId out of range");
    EntropyCoding::BinCounter::addCtx(ctxId);
    put(ctxId, bin);
    m_BinStore.addBin(bin, ctxId);  // arith_codec.cpp:581 (a no-op unless setBinStorage(true))
  }
  void encodeBinEP(unsigned bin) override {
    EntropyCoding::BinCounter::addEP();
    put(CABAC_REC_EP, bin);
  }
  void encodeBinsEP(unsigned bins, unsigned numBins) override {
    HIPREF_CHECK(numBins > 32 || (numBins < 32 && (bins >> numBins) != 0), "encodeBinsEP: value does not fit numBins");
    EntropyCoding::BinCounter::addEP(numBins);
    for (int i = int(numBins) - 1; i >= 0; i--) put(CABAC_REC_EP, (bins >> i) & 1u);
  }
  void encodeRemAbsEP(unsigned bins, unsigned goRicePar, unsigned cutoff, int maxLog2TrDynamicRange) override {
    // the code word is this repo's statement of it (host/cabac_rem_abs.hpp); its bypass bins as records, in order
    const cabac_code::RemAbsCode c = cabac_code::rem_abs_code(bins, goRicePar, cutoff, unsigned(maxLog2TrDynamicRange));
    EntropyCoding::BinCounter::addEP(c.length());
    for (uint32_t i = 0; i < c.ones; i++) put(CABAC_REC_EP, 1);
    if (c.stop) put(CABAC_REC_EP, 0);
    for (uint32_t i = c.tail_bits; i-- > 0;) put(CABAC_REC_EP, (c.tail >> i) & 1u);
  }
  void encodeBinTrm(unsigned bin) override {
    EntropyCoding::BinCounter::addTrm();
    put(CABAC_REC_TRM, bin);
  }
  void align() override { put(CABAC_REC_ALIGN, 0); }
  uint32_t getNumBins() override {
    uint32_t n = EntropyCoding::BinCounter::getAll();
    for (uint32_t c : m_devCtx) n += c;
    return n;
  }
  bool isEncoding() override { return true; }
  unsigned getNumWrittenBits() override {  // arith_codec.cpp:482-485; Immediate mode: one probing launch (HipBatch::numWrittenBits)
    HIPREF_CHECK(m_mode != Immediate, "getNumWrittenBits: nothing is coded before HipBatch::flush() in Deferred mode");
    HIPREF_CHECK(!m_blocks.empty(), "getNumWrittenBits: the bins of spliced residual blocks are not known before flush()");
    HIPREF_CHECK(!m_Bitstream, "getNumWrittenBits() without a bitstream");
    try {
      return m_Bitstream->getNumberOfWrittenBits() + m_batch.numWrittenBits(m_records.data(), m_records.size(), m_qp, m_initId);
    } catch (const EntropyCodingAMD::Exception &e) {
      HIPREF_THROW(e.what());
    }
  }
  // The bin store of the window-size training path (arith_codec.cpp:585-601, the reference's own BinStore class): the bins are
  // all here on the host as they are recorded, so the store is kept exactly as TBinEncoder keeps it.  While it is in use
  // residual blocks must be binarised where their bins can be stored: spliceResidual refuses, ResidualCoderHipRef replays.
  void setBinStorage(bool b) override { m_BinStore.setUse(b); }
  const EntropyCoding::BinStore *getBinStore() const override { return &m_BinStore; }
  EntropyCoding::BinEncIf *getTestBinEncoder() const override {
    return m_BinStore.inUse() ? new BinEncoderHipRef(m_batch, Immediate) : nullptr;  // a fresh encoder, as :594-601; Immediate: it is asked getNumWrittenBits()
  }
  bool binStoreInUse() const { return m_BinStore.inUse(); }

  const EntropyCodingAMD::RecordVector &records() const { return m_records; }

private:
  void put(unsigned id, unsigned bin) { m_records.push_back(uint16_t(id | (bin ? CABAC_REC_BIN : 0u))); }
  HipBatch &m_batch;
  Mode m_mode;
  Common::OutputBitstream *m_Bitstream = nullptr;
  EntropyCodingAMD::RecordVector m_records;  // page-locked when EntropyCodingAMD::usePinnedMirrors(true) is set
  std::vector<cabac_splice> m_splices;
  std::vector<cabac_tu_desc> m_blocks;
  std::vector<std::function<void(uint32_t)>> m_blockInfo;
  std::vector<uint32_t> m_devCtx = std::vector<uint32_t>(CABAC_NUM_CONTEXTS, 0u);
  EntropyCoding::BinStore m_BinStore;
  int m_qp = 0, m_initId = 0;
};

// BitEstimatorHipRef IS-A EntropyCoding::BinEncIf with the behaviour of the reference's BitEstimator_Std
// (arith_codec.hpp:159-213): the reference's CABACWriter runs on it unchanged, the calls are recorded
// (resetBits() / start() / restart() as pseudo-records, so contexts carry on as in the reference) and
// getEstFracBits() has the recording costed on the device.  Bulk: HipBatch::estimate over records().
class BitEstimatorHipRef : public EntropyCoding::BinEncIf {
public:
  explicit BitEstimatorHipRef(HipBatch &batch)
      : EntropyCoding::BinEncIf(static_cast<const Common::BinProbModel_Std *>(nullptr)), m_est(batch) {}
  void init(Common::OutputBitstream *) override {}
  void uninit() override {}
  void start() override { m_est.start(); }
  void finish() override {}
  void restart() override { m_est.restart(); }
  void reset(int qp, int initId) override {
    Common::Ctx::init(qp, initId);  // host-visible context store / GRAdaptStats as the reference has them
    m_est.reset(qp, initId);
  }
  void resetBits() override { m_est.resetBits(); }
  uint64_t getEstFracBits() const override { return m_est.getEstFracBits(); }
  unsigned getNumBins(unsigned) const override { HIPREF_THROW("not supported for BitEstimator"); }
  void encodeBin(unsigned bin, unsigned ctxId) override {
    HIPREF_CHECK(ctxId >= CABAC_NUM_CONTEXTS, "ctxId out of range");
    m_est.encodeBin(bin, ctxId);
  }
  void encodeBinEP(unsigned bin) override { m_est.encodeBinEP(bin); }
  void encodeBinsEP(unsigned bins, unsigned numBins) override { m_est.encodeBinsEP(bins, numBins); }
  void encodeRemAbsEP(unsigned bins, unsigned goRicePar, unsigned cutoff, int maxLog2TrDynamicRange) override {
    m_est.encodeRemAbsEP(bins, goRicePar, cutoff, maxLog2TrDynamicRange);
  }
  void encodeBinTrm(unsigned bin) override { m_est.encodeBinTrm(bin); }
  void align() override { m_est.align(); }
  uint32_t getNumBins() override { HIPREF_THROW("Not supported"); }
  bool isEncoding() override { return false; }
  unsigned getNumWrittenBits() override { return 0; }
  void setBinStorage(bool) override {}
  const EntropyCoding::BinStore *getBinStore() const override { return nullptr; }
  EntropyCoding::BinEncIf *getTestBinEncoder() const override { return nullptr; }

  const std::vector<uint16_t> &records() const { return m_est.records(); }

private:
  BitEstimatorHip m_est;
};

// CABACWriter::residual_coding (cabac_writer.cpp:2424-2525) with the binarisation done on the GPU: same arguments,
// same bins into the bin encoder, same CUCtx side effects.  `residual_coding` is the drop-in for one block (one launch
// per call: for checking, not for speed); `queue` + `flush` is the shape a writer uses — blocks queued while the
// syntax walk runs, one launch for all of them, each block's bins handed to the encoder in order.
// Covered: regular and transform-skip residual coding (with BDPCM), the SBT/MTS zero-out.  Not covered (throws): the range
// extensions (Rice extension, persistent Rice adaptation, TSRC Rice).
//
// Two forms.  On a BinEncoderHipRef (second constructor) a block costs the host one copy of its coefficients and one
// pass over them for the CUCtx side effects: ts_flag is recorded as an ordinary bin, the coefficients are staged, a splice
// marks the place of the block's bins, and HipBatch::flush() has them binarised, spliced into the substream and coded on
// the device — the bins go "straight into the encoder" as in the reference (:2766-2803, :2822, :2843, :2871) and never
// exist on the host.  On any other BinEncIf (first constructor; e.g. the reference's BinEncoder_Std, for checking) the
// block records come back from the device and are replayed into the encoder call by call.
class ResidualCoderHipRef {
public:
  ResidualCoderHipRef(HipBatch &batch, EntropyCoding::BinEncIf &enc) : m_batch(batch), m_enc(enc) {}
  ResidualCoderHipRef(HipBatch &batch, BinEncoderHipRef &enc) : m_batch(batch), m_enc(enc), m_hip(&enc) {}

  void residual_coding(const Common::TransformUnit &tu, Common::ComponentID compID, Common::CUCtx *cuCtx) {
    queue(tu, compID, cuCtx);
    flush();
  }

  void queue(const Common::TransformUnit &tu, Common::ComponentID compID, Common::CUCtx *cuCtx) {
    using namespace Common;
    if (compID == COMPONENT_Cr && tu.jointCbCr == 3) return;  // cabac_writer.cpp:2428-2430
    Item it;
    it.tsAllowed = TU::isTSAllowed(tu, compID);                // ts_flag, cabac_writer.cpp:2527-2534
    it.tsFlag = tu.mtsIdx[compID] == MTS_SKIP ? 1 : 0;
    it.chroma = !isLuma(compID);
    const SPS &sps = *tu.cs->sps;
    it.transformSkip = tu.mtsIdx[compID] == MTS_SKIP && !tu.cs->slice->getTSResidualCodingDisabledFlag();  // :2434-2438
    it.bdpcm = (isLuma(compID) ? tu.cu->bdpcmMode : tu.cu->bdpcmModeChroma) != 0;
    HIPREF_CHECK(sps.getSpsRangeExtension().getRrcRiceExtensionEnableFlag() ||
                     sps.getSpsRangeExtension().getPersistentRiceAdaptationEnabledFlag() ||
                     (it.transformSkip && sps.getSpsRangeExtension().getTSRCRicePresentFlag()),
                 "range-extension Rice derivation is not covered by the GPU binariser");
    // SBT / MTS zero-out of 32-wide / tall luma blocks (cabac_writer.cpp:2660-2667, :2507-2516, unit.cpp:465-479)
    it.sbtZeroOut = sps.getUseMTS() && tu.cu->sbtInfo != 0 && compID == COMPONENT_Y && tu.blocks[compID].width <= 32 && tu.blocks[compID].height <= 32;
    const CompArea &blk = tu.blocks[compID];
    it.width = blk.width;
    it.height = blk.height;
    it.cuCtx = cuCtx;
    it.notSkip = tu.mtsIdx[compID] != MTS_SKIP;
    const TCoeff *c = tu.getCoeffs(compID).buf;
    it.coeff.assign(c, c + size_t(blk.width) * blk.height);
    it.depQuant = tu.cs->slice->getDepQuantEnabledFlag();
    it.signHiding = tu.cs->slice->getSignDataHidingEnabledFlag();
    it.maxLog2 = sps.getMaxLog2TrDynamicRange(toChannelType(compID));
    if (m_hip && !m_hip->binStoreInUse()) return splice(tu, compID, it);
    m_items.push_back(std::move(it));   // (with the bin store in use the bins are replayed into the encoder, which stores them)
  }

  void flush() {
    using namespace Common;
    if (m_items.empty()) return;  // spliced blocks are coded by HipBatch::flush() with their substream
    std::vector<HipBatch::ResidualBlock> blocks;
    for (const Item &it : m_items) {
      HipBatch::ResidualBlock b;
      b.coeff = it.coeff.data();
      b.width = it.width;
      b.height = it.height;
      b.chroma = it.chroma;
      b.depQuant = it.depQuant;
      b.signHiding = it.signHiding;
      b.tsFlag = false;  // coded below through the encoder
      b.transformSkip = it.transformSkip;
      b.bdpcm = it.bdpcm;
      b.sbtZeroOut = it.sbtZeroOut;
      b.maxLog2TrDynamicRange = it.maxLog2;
      blocks.push_back(b);
    }
    HipBatch::ResidualResult r;
    try {
      r = m_batch.residual(blocks);
    } catch (const EntropyCodingAMD::Exception &e) {
      m_items.clear();
      HIPREF_THROW(e.what());
    }
    for (size_t t = 0; t < m_items.size(); t++) {
      const Item &it = m_items[t];
      if (it.tsAllowed) m_enc.encodeBin(it.tsFlag, Ctx::TransformSkipFlag(it.chroma ? 1 : 0));
      for (uint64_t k = r.offsets[t]; k < r.offsets[t + 1]; k++) {
        const unsigned id = r.records[k] & CABAC_REC_ID_MASK, bin = r.records[k] >> 15;
        if (id == CABAC_REC_EP) m_enc.encodeBinEP(bin);
        else m_enc.encodeBin(bin, id);
      }
      if (CUCtx *cu = it.cuCtx) {  // cabac_writer.cpp:2461-2477, :2519-2522
        const int last = int(r.info[t] & CABAC_TU_INFO_LAST_MASK);
        const ChannelType ch = it.chroma ? CHANNEL_TYPE_CHROMA : CHANNEL_TYPE_LUMA;
        if (it.notSkip && it.height >= 4 && it.width >= 4) {
          const int maxLfnstPos = ((it.height == 4 && it.width == 4) || (it.height == 8 && it.width == 8)) ? 7 : 15;
          cu->violatesLfnstConstrained[ch] |= last > maxLfnstPos;
          cu->lfnstLastScanPos |= last >= (it.chroma ? LFNST_LAST_SIG_CHROMA : LFNST_LAST_SIG_LUMA);
        }
        if (!it.chroma && it.notSkip) cu->mtsLastScanPos |= last >= 1;
        if (!it.chroma && (r.info[t] & CABAC_TU_INFO_MTS_VIOLATION)) cu->violatesMtsCoeffConstraint = true;
      }
    }
    m_items.clear();
  }

private:
  struct Item {
    std::vector<int32_t> coeff;
    unsigned width = 0, height = 0;
    bool chroma = false, depQuant = false, signHiding = false, tsAllowed = false, notSkip = true, transformSkip = false, bdpcm = false;
    bool sbtZeroOut = false;
    unsigned tsFlag = 0;
    int maxLog2 = 15;
    Common::CUCtx *cuCtx = nullptr;
  };

  // the splice form of one block: ts_flag as a bin, the coefficients to the device, the CUCtx side effects at once
  void splice(const Common::TransformUnit &tu, Common::ComponentID compID, const Item &it) {
    using namespace Common;
    if (it.tsAllowed) m_enc.encodeBin(it.tsFlag, Ctx::TransformSkipFlag(it.chroma ? 1 : 0));  // ts_flag, cabac_writer.cpp:2527-2534
    // What the writer's walk leaves in the CUCtx depends on where the last significant coefficient lies and on which
    // coefficient groups are significant (cabac_writer.cpp:2447-2477, :2519-2522).  A later syntax element may depend on it
    // (lfnst_idx, mts_idx), so it cannot wait for the device: one pass over the coefficients in the reference's own scan
    // order, with the reference's own CoeffCodingContext.  The device reports the same per block; flush() compares.
    int last = -1;
    bool violation = false;
    if (!it.transformSkip) {
      CoeffCodingContext cctx(tu, compID, it.signHiding);
      const TCoeff *coeff = tu.getCoeffs(compID).buf;
      int group = -1;
      for (int scanPos = 0; scanPos < int(cctx.maxNumCoeff()); scanPos++) {
        if (!coeff[cctx.blockPos(scanPos)]) continue;
        last = scanPos;
        const int sub = scanPos >> cctx.log2CGSize();
        if (sub == group) continue;
        group = sub;
        cctx.initSubblock(sub, true);
        violation = violation || (isLuma(compID) && (cctx.cgPosY() > 3 || cctx.cgPosX() > 3));
      }
      HIPREF_CHECK(last < 0, "Coefficient coding called for empty TU");  // cabac_writer.cpp:2458
      if (CUCtx *cu = it.cuCtx) {
        const ChannelType ch = it.chroma ? CHANNEL_TYPE_CHROMA : CHANNEL_TYPE_LUMA;
        if (it.notSkip && it.height >= 4 && it.width >= 4) {
          const int maxLfnstPos = ((it.height == 4 && it.width == 4) || (it.height == 8 && it.width == 8)) ? 7 : 15;
          cu->violatesLfnstConstrained[ch] |= last > maxLfnstPos;
          cu->lfnstLastScanPos |= last >= (it.chroma ? LFNST_LAST_SIG_CHROMA : LFNST_LAST_SIG_LUMA);
        }
        if (!it.chroma && it.notSkip) cu->mtsLastScanPos |= last >= 1;
        if (!it.chroma && violation) cu->violatesMtsCoeffConstraint = true;
      }
    }
    HipBatch::ResidualBlock b;
    b.coeff = it.coeff.data();
    b.width = it.width;
    b.height = it.height;
    b.chroma = it.chroma;
    b.depQuant = it.depQuant;
    b.signHiding = it.signHiding;
    b.tsFlag = false;
    b.transformSkip = it.transformSkip;
    b.bdpcm = it.bdpcm;
    b.sbtZeroOut = it.sbtZeroOut;
    b.maxLog2TrDynamicRange = it.maxLog2;
    const bool regular = !it.transformSkip;
    const bool luma = !it.chroma;
    m_hip->spliceResidual(b, [last, violation, regular, luma](uint32_t info) {
      if (!regular) return;
      if (int(info & CABAC_TU_INFO_LAST_MASK) != last || (luma && bool(info & CABAC_TU_INFO_MTS_VIOLATION) != violation))
        HIPREF_THROW("device and host disagree about a block's last significant position");
    });
  }

  HipBatch &m_batch;
  EntropyCoding::BinEncIf &m_enc;
  BinEncoderHipRef *m_hip = nullptr;
  std::vector<Item> m_items;
};

// ---------------------------------------------------------------------------------------------------------------
// Decoder side.
//
// BinDecoderHipRef IS-A EntropyCoding::BinDecoderBase (reference arith_codec.hpp:215-260) for the supplied-ctxId mode of
// the device decoder: the ctxId / bypass / terminate sequence of a substream is planned (plan*), decoded by one launch
// (run), and served back through the decoder's calls, which check that the caller follows the plan.  Of the reference's
// decoder interface only decodeBin is virtual (arith_codec.hpp:241); decodeBinEP / decodeBinsEP / decodeRemAbsEP /
// decodeBinTrm / finish are not, so a caller that holds a `BinDecoderBase &` — the reference's CABACReader does,
// cabac_reader.hpp:142 — reaches this class's versions of them only when those five members are declared virtual in
// arith_codec.hpp (a five-keyword change); through a `BinDecoderHipRef &` every call is served.
class BinDecoderHipRef : public EntropyCoding::BinDecoderBase {
public:
  explicit BinDecoderHipRef(HipBatch &batch)
      : EntropyCoding::BinDecoderBase(static_cast<const Common::BinProbModel_Std *>(nullptr)), m_batch(batch) {}

  void reset(int qp, int initId) {  // arith_codec.cpp:75-78
    Common::Ctx::init(qp, initId);
    m_qp = qp;
    m_initId = initId;
    m_plan.clear();
    m_bins.clear();
    m_pos = 0;
  }
  void planBin(unsigned ctxId) { m_plan.push_back(uint16_t(ctxId)); }
  void planBinEP(unsigned n = 1) { m_plan.insert(m_plan.end(), n, uint16_t(CABAC_REC_EP)); }
  void planBinTrm() { m_plan.push_back(uint16_t(CABAC_REC_TRM)); }
  void planRecords(const uint16_t *rec, size_t n) { m_plan.insert(m_plan.end(), rec, rec + n); }

  // decode the plan from the bitstream handed to init() (from its current byte position); checkFinish: the stop
  // pattern check of finish() (arith_codec.cpp:68-73).  Afterwards the bitstream stands where the reference's decoder
  // would leave it.
  void run(bool checkFinish) {
    HIPREF_CHECK(!m_Bitstream, "run(): no bitstream");
    const std::vector<uint8_t> &fifo = m_Bitstream->getFifo();
    const uint32_t at = m_Bitstream->getByteLocation();
    HipBatch::DecodeJob job;
    job.records = m_plan.data();
    job.n_records = uint32_t(m_plan.size());
    job.bytes = fifo.data() + at;
    job.n_bytes = uint32_t(fifo.size()) - at;
    job.qp = m_qp;
    job.initId = m_initId;
    job.finish = checkFinish;
    std::vector<std::vector<uint8_t>> bins;
    std::vector<uint32_t> bitsRead;
    try {
      m_batch.decode({job}, bins, &bitsRead);
    } catch (const EntropyCodingAMD::Exception &e) {
      HIPREF_THROW(e.what());
    }
    m_bins.swap(bins[0]);
    m_bitsRead = bitsRead[0];
    m_Bitstream->m_fifo_idx += (m_bitsRead + 8) / 8;  // bytes the reference decoder would have consumed
    m_pos = 0;
  }

  unsigned decodeBin(unsigned ctxId) override { return next(ctxId); }  // arith_codec.cpp:242-277
  unsigned decodeBinEP() { return next(CABAC_REC_EP); }                // :100-114
  unsigned decodeBinTrm() { return next(CABAC_REC_TRM); }              // :181-197
  unsigned decodeBinsEP(unsigned numBins) {                            // :116-151
    unsigned bins = 0;
    for (unsigned i = 0; i < numBins; i++) bins = (bins << 1) | next(CABAC_REC_EP);
    return bins;
  }
  unsigned decodeRemAbsEP(unsigned goRicePar, unsigned cutoff, int maxLog2TrDynamicRange) {  // :153-179
    const unsigned maxPrefix = 32u - unsigned(maxLog2TrDynamicRange);
    unsigned prefix = 0;
    while (prefix < maxPrefix && decodeBinEP()) prefix++;
    if (prefix < cutoff) return (prefix << goRicePar) + decodeBinsEP(goRicePar);
    const unsigned offset = ((1u << (prefix - cutoff)) + cutoff - 1u) << goRicePar;
    return offset + decodeBinsEP(prefix == maxPrefix ? unsigned(maxLog2TrDynamicRange) : goRicePar + prefix - cutoff);
  }
  void finish() {}  // the stop pattern was checked on the device (run(true))
  unsigned getNumBitsRead() const { return m_bitsRead; }

private:
  unsigned next(unsigned id) {
    HIPREF_CHECK(m_pos >= m_plan.size(), "decode call beyond the planned sequence");
    HIPREF_CHECK((m_plan[m_pos] & CABAC_REC_ID_MASK) != id, "decode call does not match the planned ctxId sequence");
    return m_bins[m_pos++];
  }
  HipBatch &m_batch;
  std::vector<uint16_t> m_plan;
  std::vector<uint8_t> m_bins;
  size_t m_pos = 0;
  int m_qp = 0, m_initId = 0;
  uint32_t m_bitsRead = 0;
};

// CABACReader::residual_coding (cabac_reader.cpp:2647-2735, with ts_flag :2737-2752 and residual_codingTS :3130-3339)
// with the parsing done on the GPU: `residual_coding(tu, compID, cuCtx)` has the reader's signature and queues the block —
// its geometry and where its coefficients go; endSubstream() has every queued block of the substream parsed by one launch
// (the contexts are derived on the device from the coefficients decoded so far), writes the coefficients into the
// TransformUnits' buffers, sets mtsIdx as ts_flag does, updates the CUCtx arguments exactly as the reader does, checks the
// terminate bin and the stop pattern (end_of_slice / finish) and moves the bitstream on.  The device decoder cannot stop
// in the middle of a substream and hand over to the host, so this binds where a substream (or the residual partition of
// one) holds nothing but residual blocks whose sizes are known before their coefficients; INTEGRATION.md section 6.
// Not covered (throws): the range extensions, TS residual coding disabled by the slice.
class ResidualParserHipRef {
public:
  explicit ResidualParserHipRef(HipBatch &batch) : m_batch(batch) {}

  // CABACReader::initBitstream + the decoder's reset(qp, initId): the substream starts at the bitstream's position
  void beginSubstream(Common::InputBitstream *bitstream, int qp, int initId) {
    m_Bitstream = bitstream;
    m_qp = qp;
    m_initId = initId;
    m_items.clear();
  }

  void residual_coding(Common::TransformUnit &tu, Common::ComponentID compID, Common::CUCtx &cuCtx) {
    using namespace Common;
    if (compID == COMPONENT_Cr && tu.jointCbCr == 3) return;  // cabac_reader.cpp:2651-2653
    const SPS &sps = *tu.cs->sps;
    HIPREF_CHECK(sps.getSpsRangeExtension().getRrcRiceExtensionEnableFlag() ||
                     sps.getSpsRangeExtension().getPersistentRiceAdaptationEnabledFlag() ||
                     sps.getSpsRangeExtension().getTSRCRicePresentFlag(),
                 "range-extension Rice derivation is not covered by the GPU parser");
    HIPREF_CHECK(tu.cs->slice->getTSResidualCodingDisabledFlag(), "slice_ts_residual_coding_disabled_flag is not covered by the GPU parser");
    Item it;
    it.tu = &tu;
    it.compID = compID;
    it.cuCtx = &cuCtx;
    it.b.coeff = nullptr;
    it.b.width = tu.blocks[compID].width;
    it.b.height = tu.blocks[compID].height;
    it.b.chroma = !isLuma(compID);
    it.b.depQuant = tu.cs->slice->getDepQuantEnabledFlag();
    it.b.signHiding = tu.cs->slice->getSignDataHidingEnabledFlag();
    // ts_flag (cabac_reader.cpp:2737-2752): in the stream where transform skip is allowed, otherwise inferred
    it.b.bdpcm = (isLuma(compID) ? tu.cu->bdpcmMode : tu.cu->bdpcmModeChroma) != 0;
    it.b.tsFlag = TU::isTSAllowed(tu, compID);
    it.b.transformSkip = it.b.bdpcm || tu.mtsIdx[compID] == MTS_SKIP;
    it.b.sbtZeroOut = sps.getUseMTS() && tu.cu->sbtInfo != 0 && compID == COMPONENT_Y && tu.blocks[compID].width <= 32 &&
                      tu.blocks[compID].height <= 32;  // cabac_reader.cpp:2880-2891, :2718-2727
    it.b.maxLog2TrDynamicRange = sps.getMaxLog2TrDynamicRange(toChannelType(compID));
    m_items.push_back(it);
  }

  // terminate: the substream ends with end_of_slice()'s terminate bin and the stop pattern, which are checked
  void endSubstream() {
    using namespace Common;
    HIPREF_CHECK(!m_Bitstream, "endSubstream(): no bitstream");
    const std::vector<uint8_t> &fifo = m_Bitstream->getFifo();
    const uint32_t at = m_Bitstream->getByteLocation();
    HipBatch::ParseJob job;
    job.bytes = fifo.data() + at;
    job.n_bytes = uint32_t(fifo.size()) - at;
    job.qp = m_qp;
    job.initId = m_initId;
    for (const Item &it : m_items) job.blocks.push_back(it.b);
    std::vector<std::vector<int32_t>> coeff;
    std::vector<uint32_t> info;
    try {
      coeff = m_batch.residualParse({job}, &info);
    } catch (const EntropyCodingAMD::Exception &e) {
      m_items.clear();
      HIPREF_THROW(e.what());
    }
    m_Bitstream->m_fifo_idx = uint32_t(fifo.size());  // a substream is consumed whole (its length is what delimited it)
    const int32_t *c = coeff[0].data();
    for (size_t t = 0; t < m_items.size(); t++) {
      const Item &it = m_items[t];
      const unsigned w = it.b.width, h = it.b.height;
      TCoeff *dst = it.tu->getCoeffs(it.compID).buf;
      for (size_t k = 0; k < size_t(w) * h; k++) dst[k] = TCoeff(c[k]);
      c += size_t(w) * h;
      const bool ts = (info[t] & CABAC_TU_INFO_TS) != 0;
      it.tu->mtsIdx[it.compID] = ts ? MTS_SKIP : MTS_DCT2_DCT2;  // ts_flag, :2751
      if (ts) continue;                                           // residual_codingTS touches no CUCtx
      CUCtx &cu = *it.cuCtx;                                      // :2668-2693, :2729-2732
      const int last = int(info[t] & CABAC_TU_INFO_LAST_MASK);
      const ChannelType ch = it.b.chroma ? CHANNEL_TYPE_CHROMA : CHANNEL_TYPE_LUMA;
      if (h >= 4 && w >= 4) {
        const int maxLfnstPos = ((h == 4 && w == 4) || (h == 8 && w == 8)) ? 7 : 15;
        cu.violatesLfnstConstrained[ch] |= last > maxLfnstPos;
        cu.lfnstLastScanPos |= last >= (it.b.chroma ? LFNST_LAST_SIG_CHROMA : LFNST_LAST_SIG_LUMA);
      }
      if (!it.b.chroma) cu.mtsLastScanPos |= last >= 1;
      if (!it.b.chroma && (info[t] & CABAC_TU_INFO_MTS_VIOLATION)) cu.violatesMtsCoeffConstraint = true;
    }
    m_items.clear();
  }

private:
  struct Item {
    Common::TransformUnit *tu = nullptr;
    Common::ComponentID compID = Common::COMPONENT_Y;
    Common::CUCtx *cuCtx = nullptr;
    HipBatch::ResidualBlock b;
  };
  HipBatch &m_batch;
  Common::InputBitstream *m_Bitstream = nullptr;
  int m_qp = 0, m_initId = 0;
  std::vector<Item> m_items;
};

}  // namespace EntropyCodingAMD
#endif
