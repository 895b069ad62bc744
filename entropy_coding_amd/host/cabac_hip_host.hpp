// Host-side C++ mirror of the reference's bin-codec interface, on top of the C ABI (cabac_hip.h).
//
// The reference's callers (CABACWriter / CABACReader) talk to the codec through
//   BinEncIf          entropy_codec/arith_codec.hpp:31-70   (encodeBin, encodeBinEP, encodeBinsEP,
//                                                             encodeRemAbsEP, encodeBinTrm, align, ...)
//   BinDecoderBase    entropy_codec/arith_codec.hpp:215-260 (decodeBin, decodeBinEP, decodeBinsEP, ...)
//   OutputBitstream / InputBitstream   common/bit_stream.hpp:16-168
// and make 10^4..10^6 tiny virtual calls per frame.  Those cannot cross to the GPU one by one, so
// BinEncoderHip *records* them as 16-bit bin records (cabac_hip.h) and a HipBatch codes whole
// substreams on the device; afterwards each substream's OutputBitstream is in exactly the state the
// reference leaves it in after finish() (same FIFO bytes, same held bits).
//
// Names, argument meaning and error behaviour follow the reference (errors are C++ exceptions,
// type_def.hpp:295-329).  integration/reference_adapter.hpp shows the same recorder deriving from
// the reference's own BinEncIf for a true drop-in.
#ifndef CABAC_HIP_HOST_HPP
#define CABAC_HIP_HOST_HPP

#include <cstdint>
#include <exception>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "cabac_hip.h"

namespace EntropyCodingAMD {

class Exception : public std::exception {
public:
  explicit Exception(std::string s) : m_str(std::move(s)) {}
  const char *what() const noexcept override { return m_str.c_str(); }

private:
  std::string m_str;
};

// ---------------------------------------------------------------------------------------------
// Pinned host / device mirrors of the reference's host buffers.  The FIFOs of the byte-stream containers below and the
// record buffers of the recording encoders are std::vectors over HostAllocator: ordinary heap memory by default, and
// page-locked memory mapped for the GPU's DMA engines (cabac_hip_host_alloc, cabac_hip.h) for every allocation made
// while usePinnedMirrors(true) is in force — such buffers cross PCIe without a staging copy.  Switching it on needs a
// GPU (allocations throw std::bad_alloc otherwise); a buffer remembers how it was allocated.
void usePinnedMirrors(bool on);
bool pinnedMirrors();
void *hostAllocate(size_t bytes);
void hostDeallocate(void *p) noexcept;

template <class T>
struct HostAllocator {
  using value_type = T;
  HostAllocator() = default;
  template <class U>
  HostAllocator(const HostAllocator<U> &) {}
  T *allocate(size_t n) { return static_cast<T *>(hostAllocate(n * sizeof(T))); }
  void deallocate(T *p, size_t) noexcept { hostDeallocate(p); }
  template <class U>
  bool operator==(const HostAllocator<U> &) const { return true; }
  template <class U>
  bool operator!=(const HostAllocator<U> &) const { return false; }
};
using ByteVector = std::vector<uint8_t, HostAllocator<uint8_t>>;
using RecordVector = std::vector<uint16_t, HostAllocator<uint16_t>>;

// ---------------------------------------------------------------------------------------------
// Byte-stream containers (reference: common/bit_stream.hpp:16-97, :103-168).  Only the members the
// bin codec and its callers use are mirrored; public data members keep the reference's names.
class OutputBitstream {
public:
  ByteVector m_fifo;
  uint32_t m_num_held_bits = 0;
  uint8_t m_held_bits = 0;

  void write(uint32_t uiBits, uint32_t uiNumberOfBits);  // bit_stream.cpp:70-117
  void writeAlignZero();                                 // :125-132
  void writeAlignOne();                                  // :119-123
  void writeByteAlignment();                             // :152-155
  void addSubstream(OutputBitstream *pcSubstream);       // :139-150
  void clear();
  int getNumBitsUntilByteAligned() const { return (8 - m_num_held_bits) & 0x7; }
  uint32_t getNumberOfWrittenBits() const { return uint32_t(m_fifo.size()) * 8 + m_num_held_bits; }
  ByteVector &getFIFO() { return m_fifo; }
  const ByteVector &getFIFO() const { return m_fifo; }
  uint8_t getHeldBits() const { return m_held_bits; }
  uint8_t *getByteStream() { return m_fifo.data(); }
  uint32_t getByteStreamLength() const { return uint32_t(m_fifo.size()); }
};

class InputBitstream {
public:
  ByteVector m_fifo;
  uint32_t m_fifo_idx = 0;
  uint32_t m_num_held_bits = 0;  // unread bits of the last byte taken from the FIFO: its low m_num_held_bits bits
  uint8_t m_held_bits = 0;
  uint32_t m_numBitsRead = 0;

  ByteVector &getFifo() { return m_fifo; }
  const ByteVector &getFifo() const { return m_fifo; }
  void read(uint32_t uiNumberOfBits, uint32_t &ruiBits);  // bit_stream.cpp:204-268 (throws "Exceeded FIFO size")
  uint32_t read(uint32_t numberOfBits) {
    uint32_t v;
    read(numberOfBits, v);
    return v;
  }
  uint32_t readByte();                      // bit_stream.cpp:268-274 (throws "FIFO exceeded")
  void peekPreviousByte(uint32_t &byte);    // :276-279
  uint32_t readOutTrailingBits();           // :355-364
  uint32_t readByteAlignment();             // :417-430: the stop bit '1', then zero bits up to the byte boundary
  uint8_t getHeldBits() const { return m_held_bits; }
  uint32_t getByteLocation() const { return m_fifo_idx; }
  uint32_t getNumBitsUntilByteAligned() const { return m_num_held_bits & 7u; }
  uint32_t getNumBitsLeft() const { return 8 * (uint32_t(m_fifo.size()) - m_fifo_idx) + m_num_held_bits; }
  uint32_t getNumBitsRead() const { return m_numBitsRead; }
  InputBitstream *extractSubstream(uint32_t uiNumBits);  // :382-415, any bit position and bit count
};

// ---------------------------------------------------------------------------------------------
// BinCounter (arith_codec.hpp:72-93, arith_codec.cpp:281-316)
class BinCounter {
public:
  BinCounter() : m_NumBinsCtx(CABAC_NUM_CONTEXTS, 0) {}
  void reset();
  void addCtx(unsigned ctxId) { m_NumBinsCtx[ctxId]++; }
  void addEP(unsigned num) { m_NumBinsEP += num; }
  void addEP() { m_NumBinsEP++; }
  void addTrm() { m_NumBinsTrm++; }
  // counts that only became known on the device (the bins of spliced residual blocks, HipBatch::flush): `counts` holds
  // CABAC_BIN_COUNT_WORDS words as cabac_hip_encode_batch_residual reports them, `host` the part already counted here
  void addFromDevice(const uint32_t *counts, const uint32_t *host);
  void snapshot(uint32_t *out) const;  // CABAC_BIN_COUNT_WORDS words: per context, bypass, terminate
  uint32_t getAll() const;
  uint32_t getCtx(unsigned ctxId) const { return m_NumBinsCtx[ctxId]; }
  uint32_t getEP() const { return m_NumBinsEP; }
  uint32_t getTrm() const { return m_NumBinsTrm; }

private:
  std::vector<uint32_t> m_NumBinsCtx;
  uint32_t m_NumBinsEP = 0, m_NumBinsTrm = 0;
};

// ---------------------------------------------------------------------------------------------
// A session on one GPU: collects finished substreams and codes them in one launch.
// Objects of these classes are laid out by the caller's compiler from THIS header and filled in by code inside
// libcabac_hip.so: a caller built against another version of the header (a prebuilt test driver, a plugin) corrupts the heap
// or the stack without any diagnostic — in round 2 of this repo such a stale test library is the most probable cause of an
// intermittent SIGABRT at the end of a test run, and in round 3 one segfaulted the same way.  The constructors therefore hand
// the library the caller's view of the layout, and the library refuses a caller whose view differs from its own.
size_t hostLayoutFingerprint(size_t batch, size_t pending, size_t encoder, size_t estimator, size_t decoder, size_t out, size_t in);

class HipBatch {
public:
  explicit HipBatch(int device = 0) : m_device(device) { checkCaller(); }
  // Several GPUs behind one batch (the launcher-free form: every GPU DMAs its share straight from this
  // process's host memory; no torch.distributed, no collective).  flush() deals the pending substreams to the devices
  // longest first (LPT by records + coefficients), codes every share on its own device at the same time — one host thread
  // per device for the duration of the flush — and hands the bytes to each substream's bitstream as a single device
  // does; the order of the substreams in the caller's bitstreams is the caller's (OutputBitstream::addSubstream,
  // bit_stream.cpp:139-150), whatever device coded them.  A device may be listed more than once (two contexts on it).
  // decode() is dealt out the same way; the other calls run on the first device.
  explicit HipBatch(const std::vector<int> &devices) : m_device(devices.empty() ? 0 : devices[0]) {
    checkCaller();
    addPeers(devices);
  }
  size_t deviceCount() const { return 1 + m_peers.size(); }
  ~HipBatch();
  HipBatch(const HipBatch &) = delete;
  HipBatch &operator=(const HipBatch &) = delete;

  // Encode every pending substream and append its bytes to its OutputBitstream.
  void flush();
  size_t pending() const { return m_pending.size(); }
  cabac_hip_ctx *handle();  // initialises the device on first use; throws if there is no GPU

  // Decode: ctx/EP/TRM record sequences against byte-aligned substreams (supplied-ctxId replay).
  struct DecodeJob {
    const uint16_t *records;
    uint32_t n_records;
    const uint8_t *bytes;
    uint32_t n_bytes;
    int qp;
    int initId;
    bool finish;
  };
  // bins[i] receives the decoded bins of job i; throws Exception on UNDERRUN / BAD_STOP like the
  // reference's CHECKs ("FIFO exceeded", "No proper stop/alignment pattern ...").
  void decode(const std::vector<DecodeJob> &jobs, std::vector<std::vector<uint8_t>> &bins,
              std::vector<uint32_t> *bitsRead = nullptr);

  // Bit estimator: the cost (1/32768 bit) of each bin string after reset(qp, initId), one launch for all
  // (BitEstimator_Std, arith_codec.cpp:603-711).  Strings may contain the estimator pseudo-records
  // CABAC_REC_EST_RESETBITS / CABAC_REC_EST_RESTART (cabac_hip.h).
  struct EstimateJob {
    const uint16_t *records;
    uint32_t n_records;
    int qp;
    int initId;
  };
  std::vector<uint64_t> estimate(const std::vector<EstimateJob> &jobs);

  // Residual binariser: the bin records CABACWriter::residual_coding (cabac_writer.cpp:2424-2525) would ask its bin
  // encoder for, for many transform blocks in one launch.  Regular and transform-skip residual coding (no SBT/MTS
  // zero-out included, no range-extension Rice derivation).  Throws Exception("Coefficient coding called for empty
  // TU") for an all-zero block, as the reference's CHECK does (cabac_writer.cpp:2458).
  struct ResidualBlock {
    const int32_t *coeff;  // width * height coefficients, raster (TransformUnit::getCoeffs(compID).buf)
    unsigned width, height;
    bool chroma;           // toChannelType(compID) == CHANNEL_TYPE_CHROMA
    bool depQuant;         // Slice::getDepQuantEnabledFlag
    bool signHiding;       // Slice::getSignDataHidingEnabledFlag
    bool tsFlag;           // TU::isTSAllowed: code transform_skip_flag first (1 for a transform-skip block)
    bool transformSkip = false;  // mtsIdx == MTS_SKIP with TS residual coding enabled: residual_codingTS
    bool bdpcm = false;          // with transformSkip: cu.bdpcmMode / bdpcmModeChroma
    bool sbtZeroOut = false;     // SPS::getUseMTS() && cu.sbtInfo != 0 (luma, at most 32 x 32): cabac_writer.cpp:2660-2667, :2507-2516
    int maxLog2TrDynamicRange = 15;
  };
  struct ResidualResult {
    std::vector<uint16_t> records;   // all blocks back to back
    std::vector<uint64_t> offsets;   // n + 1 entries
    std::vector<uint32_t> info;      // scanPosLast | CABAC_TU_INFO_MTS_VIOLATION
  };
  ResidualResult residual(const std::vector<ResidualBlock> &blocks);

  // Residual parser (CABACReader::residual_coding, cabac_reader.cpp:2647-3339): the blocks of each substream decoded from
  // its bytes, every context derived on the device.  blocks[i] gives the geometry (coeff is ignored); tsFlag: the
  // transform_skip_flag is in the stream and decides between regular and transform-skip residual coding, otherwise
  // transformSkip does (with bdpcm as given).  The result holds the blocks of all substreams back to back (width * height
  // each); info (optional) one word per block: scanPosLast | CABAC_TU_INFO_MTS_VIOLATION, or CABAC_TU_INFO_TS for a block
  // parsed as transform skip.  Throws Exception like the reference's CHECKs on a substream that runs out of bytes or
  // misses its terminate bin / stop pattern.
  struct ParseJob {
    const uint8_t *bytes;
    uint32_t n_bytes;
    int qp;
    int initId;
    std::vector<ResidualBlock> blocks;
  };
  std::vector<std::vector<int32_t>> residualParse(const std::vector<ParseJob> &jobs, std::vector<uint32_t> *info = nullptr);

  // Coefficients of blocks whose bins are spliced into a substream on the device (BinEncoderHip::encodeResidual): one
  // staging area per batch (page-locked under usePinnedMirrors), so that a flush hands the C ABI one contiguous region.
  // Returns the offset (in coefficients) of the copy.  The copy is narrowed to int16 on the way — half the bytes over PCIe
  // (cabac_hip_encode_batch_residual16) — as long as every staged block has a 15-bit dynamic range (maxLog2TrDynamicRange
  // 15, the value of all of the reference's cfgs) and its coefficients keep to it; the first block that does not turns the
  // staging area back into the reference's 32-bit TCoeff.
  uint64_t stageCoefficients(const int32_t *coeff, size_t n, int maxLog2TrDynamicRange = 15);

  // One finished, not yet coded substream.  Either `sink` (this namespace's OutputBitstream) or
  // `deliver` (any other container, e.g. the reference's Common::OutputBitstream through
  // integration/reference_adapter.hpp) receives the result: `whole` bytes + `tail_bits` (MSB-aligned
  // in bytes[whole]) exactly as the reference's finish() leaves them.
  struct Pending {
    RecordVector records;
    int qp = 0, initId = 0;
    uint64_t nCtx = 0, nEp = 0, nTrm = 0;
    OutputBitstream *sink = nullptr;
    std::function<void(const uint8_t *bytes, uint32_t whole, uint32_t tail_bits)> deliver;
    // residual blocks whose records the device splices in (cabac_hip_encode_batch_residual): splices[k].tu indexes blocks
    std::vector<cabac_splice> splices;
    std::vector<cabac_tu_desc> blocks;          // coeff_offset: into the batch's coefficient staging
    std::vector<uint32_t> hostCounts;           // BinCounter of the host-recorded bins (CABAC_BIN_COUNT_WORDS), if counted is set
    std::function<void(const uint32_t *counts, const uint32_t *host)> counted;  // the substream's totals, once known
    std::function<void(size_t block, uint32_t info)> blockInfo;                 // per block: scanPosLast | CABAC_TU_INFO_*
  };
  // BinEncoderBase::getNumWrittenBits() (arith_codec.cpp:482-485) of an encoder that has coded `records` since
  // reset(qp, initId) and flushed nothing yet: one launch with CABAC_SUB_PROBE (cabac_hip.h).  For the recording encoders'
  // Immediate mode — the reference asks this mid-substream only in its window-size training helper (cabac_writer.cpp:83-96).
  uint32_t numWrittenBits(const uint16_t *records, size_t n_records, int qp, int initId);

  void submit(Pending &&p) { m_pending.push_back(std::move(p)); }
  const std::vector<Pending> &pendingSubstreams() const { return m_pending; }  // finished, not yet coded (flush() codes them)

private:
  friend class BinEncoderHip;
  inline void checkCaller();  // defined at the end of this header, where every class is complete
  static void checkLayout(size_t callers);  // throws Exception("... built against another version of cabac_hip_host.hpp")
  void addPeers(const std::vector<int> &devices);
  int m_device;
  cabac_hip_ctx *m_ctx = nullptr;
  std::vector<Pending> m_pending;
  // the batch as the C ABI wants it (all records back to back, one byte slot per substream); kept between flushes, so
  // that with pinned mirrors the page-locking is paid once
  RecordVector m_stageRecords;
  ByteVector m_stageBytes;
  std::vector<int32_t, HostAllocator<int32_t>> m_stageCoeff;    // used when !m_narrow
  std::vector<int16_t, HostAllocator<int16_t>> m_stageCoeff16;  // used while m_narrow
  bool m_narrow = true;
  size_t m_stagedBlocksOpen = 0;  // blocks staged by encoders that have not been flushed yet
  size_t stagedCoefficients() const { return m_narrow ? m_stageCoeff16.size() : m_stageCoeff.size(); }
  void clearStage();
  uint64_t restage(HipBatch &from, uint64_t at, size_t n);  // a block of `from`'s staging area into this one (multi-device flush)
  std::vector<std::unique_ptr<HipBatch>> m_peers;  // the other devices of a multi-device batch (each a plain one-device batch)
#ifdef CABAC_HOST_TEST_OTHER_LAYOUT  // tests/test_host_shim.py: a caller compiled from "another version" of this header
  void *m_memberOfAnotherVersion = nullptr;
#endif
  void flushLocal(std::vector<Pending> &done);
  void flushSpliced(std::vector<Pending> &done);
  void deliverBytes(Pending &p, const uint8_t *src, uint32_t nbits);
};

// ---------------------------------------------------------------------------------------------
// Encoder interface — same virtuals as the reference's BinEncIf (arith_codec.hpp:39-69).
class BinEncIf {
public:
  virtual ~BinEncIf() = default;
  virtual void init(OutputBitstream *bitstream) = 0;
  virtual void uninit() = 0;
  virtual void start() = 0;
  virtual void finish() = 0;
  virtual void restart() = 0;
  virtual void reset(int qp, int initId) = 0;
  virtual void resetBits() = 0;
  virtual uint64_t getEstFracBits() const = 0;
  virtual unsigned getNumBins(unsigned ctxId) const = 0;
  virtual void encodeBin(unsigned bin, unsigned ctxId) = 0;
  virtual void encodeBinEP(unsigned bin) = 0;
  virtual void encodeBinsEP(unsigned bins, unsigned numBins) = 0;
  virtual void encodeRemAbsEP(unsigned bins, unsigned goRicePar, unsigned cutoff, int maxLog2TrDynamicRange) = 0;
  virtual void encodeBinTrm(unsigned bin) = 0;
  virtual void align() = 0;
  virtual uint32_t getNumBins() = 0;
  virtual bool isEncoding() = 0;
  virtual unsigned getNumWrittenBits() = 0;

  // Ctx side state the syntax layer reads/writes through the encoder (contexts.hpp:273-274,
  // contexts.cpp:1147-1166; RExt__GOLOMB_RICE_ADAPTATION_STATISTICS_SETS = 3)
  unsigned &getGRAdaptStats(unsigned id) { return m_GRAdaptStats[id]; }
  void riceStatReset(int bitDepth);

protected:
  unsigned m_GRAdaptStats[3] = {0, 0, 0};
};

// Recording encoder.  mode Deferred: finish() queues the substream, bytes appear after
// HipBatch::flush().  mode Immediate: finish() flushes at once (reference semantics, one launch per
// substream — for drop-in tests, not for throughput).
class BinEncoderHip : public BinEncIf, public BinCounter {
public:
  enum Mode { Deferred, Immediate };
  explicit BinEncoderHip(HipBatch &batch, Mode mode = Deferred) : m_batch(batch), m_mode(mode) {}

  void init(OutputBitstream *bitstream) override { m_Bitstream = bitstream; }  // arith_codec.cpp:323-325
  void uninit() override { m_Bitstream = nullptr; }
  void start() override;                  // arith_codec.cpp:329-337
  void finish() override;                 // arith_codec.cpp:339-357 (deferred to the device)
  void restart() override;                // :359-365 — only legal on an empty recording
  void reset(int qp, int initId) override;  // :367-370
  void resetBits() override;              // :372-378
  uint64_t getEstFracBits() const override { throw Exception("not supported"); }  // as :380-383
  unsigned getNumBins(unsigned ctxId) const override { return BinCounter::getCtx(ctxId); }
  void encodeBin(unsigned bin, unsigned ctxId) override;
  void encodeBinEP(unsigned bin) override;
  void encodeBinsEP(unsigned bins, unsigned numBins) override;
  void encodeRemAbsEP(unsigned bins, unsigned goRicePar, unsigned cutoff, int maxLog2TrDynamicRange) override;
  void encodeBinTrm(unsigned bin) override;
  void align() override;
  uint32_t getNumBins() override { return BinCounter::getAll(); }
  bool isEncoding() override { return true; }
  // arith_codec.cpp:482-485.  The arithmetic state lives on the device: in Immediate mode the bins recorded so far are
  // coded by one launch that flushes nothing and reports the count (HipBatch::numWrittenBits); in Deferred mode nothing
  // is coded before flush(), and the call throws.  (The reference asks this only in estBits, cabac_writer.cpp:83-96.)
  unsigned getNumWrittenBits() override;

  // CABACWriter::residual_coding's bins (cabac_writer.cpp:2424-2525) without binarising on the host: the block's
  // coefficients are staged for the device and a splice marks the place of its bins among the recorded ones; flush() has
  // them binarised, spliced in and coded on the device (cabac_hip_encode_batch_residual).  Code the block's
  // transform_skip_flag, if it has one, with encodeBin before this call (b.tsFlag must be false).  The block's bins enter
  // getNumBins() / getNumBins(ctxId) when they are known, at flush() — this encoder must still exist then.
  // Throws for an all-zero block at flush() ("Coefficient coding called for empty TU", cabac_writer.cpp:2458).
  void encodeResidual(const HipBatch::ResidualBlock &b);
  size_t splicedBlocks() const { return m_blocks.size(); }

  const RecordVector &records() const { return m_records; }

private:
  void put(unsigned id, unsigned bin) { m_records.push_back(uint16_t(id | (bin ? CABAC_REC_BIN : 0u))); }
  HipBatch &m_batch;
  Mode m_mode;
  OutputBitstream *m_Bitstream = nullptr;
  RecordVector m_records;
  std::vector<cabac_splice> m_splices;
  std::vector<cabac_tu_desc> m_blocks;
  int m_qp = 0, m_initId = 0;
};

// a ResidualBlock's geometry and flags as the C ABI wants them (throws for sizes that are not powers of two up to 64)
cabac_tu_desc makeTuDesc(const HipBatch::ResidualBlock &b, uint64_t coeff_offset);

// Recording bit estimator with the interface of the reference's BitEstimator_Std (arith_codec.hpp:159-213).
// The calls since reset(qp, initId) are kept as bin records (resetBits() / start() / restart() as pseudo-records,
// so that contexts carry on exactly as in the reference); getEstFracBits() has them costed on the device in one
// launch and caches the answer until the next call that changes it.  For many candidate strings use
// HipBatch::estimate on records() — one launch for all.
class BitEstimatorHip : public BinEncIf {
public:
  explicit BitEstimatorHip(HipBatch &batch) : m_batch(batch) {}
  void init(OutputBitstream *) override {}                 // arith_codec.cpp:611
  void uninit() override {}                                // :613
  void start() override { put(CABAC_REC_EST_RESETBITS); }  // :615
  void finish() override {}                                // :617
  void restart() override { put(CABAC_REC_EST_RESTART); }  // :619-621
  void reset(int qp, int initId) override;                 // :623-626
  void resetBits() override { put(CABAC_REC_EST_RESETBITS); }  // :628
  uint64_t getEstFracBits() const override;                // :630
  unsigned getNumBins(unsigned) const override { throw Exception("not supported for BitEstimator"); }  // :632-635
  void encodeBin(unsigned bin, unsigned ctxId) override;
  void encodeBinEP(unsigned bin) override { put(CABAC_REC_EP | (bin ? CABAC_REC_BIN : 0u)); }
  void encodeBinsEP(unsigned bins, unsigned numBins) override;
  void encodeRemAbsEP(unsigned bins, unsigned goRicePar, unsigned cutoff, int maxLog2TrDynamicRange) override;
  void encodeBinTrm(unsigned bin) override { put(CABAC_REC_TRM | (bin ? CABAC_REC_BIN : 0u)); }
  void align() override { put(CABAC_REC_ALIGN); }
  uint32_t getNumBins() override { throw Exception("Not supported"); }  // :644-647
  bool isEncoding() override { return false; }                           // :651
  unsigned getNumWrittenBits() override { return 0; }                    // :649

  const std::vector<uint16_t> &records() const { return m_records; }
  int qp() const { return m_qp; }
  int initId() const { return m_initId; }

private:
  void put(unsigned rec) {
    m_records.push_back(uint16_t(rec));
    m_valid = false;
  }
  HipBatch &m_batch;
  std::vector<uint16_t> m_records;
  int m_qp = 0, m_initId = 0;
  mutable bool m_valid = true;
  mutable uint64_t m_cached = 0;
};

// ---------------------------------------------------------------------------------------------
// Binarisation helpers of the syntax layer (private members of CABACWriter in the reference,
// cabac_writer.cpp:3072-3118, :854-882) as free functions over any BinEncIf.
void unary_max_symbol(BinEncIf &e, unsigned symbol, unsigned ctxId0, unsigned ctxIdN, unsigned maxSymbol);
void unary_max_eqprob(BinEncIf &e, unsigned symbol, unsigned maxSymbol);
void exp_golomb_eqprob(BinEncIf &e, unsigned symbol, unsigned count);
void xWriteTruncBinCode(BinEncIf &e, uint32_t symbol, uint32_t maxSymbol);

// ---------------------------------------------------------------------------------------------
// Replay decoder: the ctxId/EP/TRM sequence is planned first (plan*), decoded in one launch
// (run), then served through the BinDecoderBase-shaped calls, which verify that the caller asks
// for exactly the planned sequence.  (A real syntax walk chooses ctxIds from decoded values —
// that feedback is the "next" row f2 of SURVEY.md §8; this class covers the supplied-ctxId path.)
class BinDecoderHip {
public:
  explicit BinDecoderHip(HipBatch &batch) : m_batch(batch) {}
  void init(InputBitstream *bitstream) { m_Bitstream = bitstream; }  // arith_codec.cpp:54-56
  void uninit() { m_Bitstream = nullptr; }
  void reset(int qp, int initId);                                      // :75-78
  void planBin(unsigned ctxId) { m_plan.push_back(uint16_t(ctxId)); }
  void planBinEP(unsigned n = 1) { m_plan.insert(m_plan.end(), n, uint16_t(CABAC_REC_EP)); }
  void planBinTrm() { m_plan.push_back(uint16_t(CABAC_REC_TRM)); }
  void run(bool checkFinish);

  unsigned decodeBin(unsigned ctxId);          // :242-277
  unsigned decodeBinEP();                      // :100-114
  unsigned decodeBinsEP(unsigned numBins);     // :116-151
  unsigned decodeRemAbsEP(unsigned goRicePar, unsigned cutoff, int maxLog2TrDynamicRange);  // :153-179
  void planRemAbsEP(unsigned value, unsigned goRicePar, unsigned cutoff, int maxLog2TrDynamicRange);  // its bypass bins
  unsigned decodeBinTrm();                     // :181-197
  unsigned getNumBitsRead() const { return m_bitsRead; }  // :201-203

private:
  unsigned next(unsigned id);
  HipBatch &m_batch;
  InputBitstream *m_Bitstream = nullptr;
  std::vector<uint16_t> m_plan;
  std::vector<uint8_t> m_bins;
  size_t m_pos = 0;
  int m_qp = 0, m_initId = 0;
  uint32_t m_bitsRead = 0;
};

inline void HipBatch::checkCaller() {
  checkLayout(hostLayoutFingerprint(sizeof(HipBatch), sizeof(HipBatch::Pending), sizeof(BinEncoderHip), sizeof(BitEstimatorHip),
                                    sizeof(BinDecoderHip), sizeof(OutputBitstream), sizeof(InputBitstream)));
}

}  // namespace EntropyCodingAMD
#endif
