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
    EntropyCoding::BinCounter::reset();
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
    m_batch.submit(std::move(p));
    if (m_mode == Immediate) m_batch.flush();
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
  unsigned getNumBins(unsigned ctxId) const override { return EntropyCoding::BinCounter::getCtx(ctxId); }

  void encodeBin(unsigned bin, unsigned ctxId) override {
    HIPREF_CHECK(ctxId >= CABAC_NUM_CONTEXTS, "ctxId out of range");
    EntropyCoding::BinCounter::addCtx(ctxId);
    put(ctxId, bin);
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
    // the binarisation is this repo's (host/cabac_hip_host.cpp), re-used through a tiny recorder
    struct Fwd : EntropyCodingAMD::BinEncoderHip {
      using BinEncoderHip::BinEncoderHip;
    };
    Fwd tmp(m_batch);
    tmp.encodeRemAbsEP(bins, goRicePar, cutoff, maxLog2TrDynamicRange);
    EntropyCoding::BinCounter::addEP(unsigned(tmp.records().size()));
    m_records.insert(m_records.end(), tmp.records().begin(), tmp.records().end());
  }
  void encodeBinTrm(unsigned bin) override {
    EntropyCoding::BinCounter::addTrm();
    put(CABAC_REC_TRM, bin);
  }
  void align() override { put(CABAC_REC_ALIGN, 0); }
  uint32_t getNumBins() override { return EntropyCoding::BinCounter::getAll(); }
  bool isEncoding() override { return true; }
  unsigned getNumWrittenBits() override { HIPREF_THROW("getNumWrittenBits: not available from a recording encoder"); }
  void setBinStorage(bool) override {}
  const EntropyCoding::BinStore *getBinStore() const override { return nullptr; }
  EntropyCoding::BinEncIf *getTestBinEncoder() const override { return nullptr; }

  const EntropyCodingAMD::RecordVector &records() const { return m_records; }

private:
  void put(unsigned id, unsigned bin) { m_records.push_back(uint16_t(id | (bin ? CABAC_REC_BIN : 0u))); }
  HipBatch &m_batch;
  Mode m_mode;
  Common::OutputBitstream *m_Bitstream = nullptr;
  EntropyCodingAMD::RecordVector m_records;  // page-locked when EntropyCodingAMD::usePinnedMirrors(true) is set
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
// Covered: regular and transform-skip residual coding (with BDPCM).  Not covered (throws): the SBT/MTS zero-out of
// last_sig_coeff and the range extensions (Rice extension, persistent Rice adaptation, TSRC Rice).
class ResidualCoderHipRef {
public:
  ResidualCoderHipRef(HipBatch &batch, EntropyCoding::BinEncIf &enc) : m_batch(batch), m_enc(enc) {}

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
    HIPREF_CHECK(sps.getUseMTS() && tu.cu->sbtInfo != 0, "SBT zero-out is not covered by the GPU binariser");
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
    m_items.push_back(std::move(it));
  }

  void flush() {
    using namespace Common;
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
    unsigned tsFlag = 0;
    int maxLog2 = 15;
    Common::CUCtx *cuCtx = nullptr;
  };
  HipBatch &m_batch;
  EntropyCoding::BinEncIf &m_enc;
  std::vector<Item> m_items;
};

}  // namespace EntropyCodingAMD
#endif
