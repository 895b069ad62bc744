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

  const std::vector<uint16_t> &records() const { return m_records; }

private:
  void put(unsigned id, unsigned bin) { m_records.push_back(uint16_t(id | (bin ? CABAC_REC_BIN : 0u))); }
  HipBatch &m_batch;
  Mode m_mode;
  Common::OutputBitstream *m_Bitstream = nullptr;
  std::vector<uint16_t> m_records;
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

}  // namespace EntropyCodingAMD
#endif
