// Host-side mirror of the reference's bin-codec interface — see cabac_hip_host.hpp.
#include "cabac_hip_host.hpp"
#include "cabac_rem_abs.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>

namespace EntropyCodingAMD {

namespace {
[[noreturn]] void fail(const std::string &what) { throw Exception("\nERROR: " + what); }

void check_status(cabac_hip_ctx *ctx, int rc, const char *what) {
  if (rc == CABAC_HIP_OK) return;
  std::string msg = std::string(what) + ": " + cabac_hip_strerror(rc);
  if (ctx) msg += std::string(" (") + cabac_hip_last_error(ctx) + ")";
  fail(msg);
}
}  // namespace

// ------------------------------------------------------------------ pinned mirrors
namespace {
std::atomic<bool> g_pinned{false};
}
void usePinnedMirrors(bool on) { g_pinned.store(on); }
bool pinnedMirrors() { return g_pinned.load(); }

void *hostAllocate(size_t bytes) {
  void *p = nullptr;
  if (g_pinned.load()) {
    if (cabac_hip_host_alloc(bytes, &p) != CABAC_HIP_OK) throw std::bad_alloc();
  } else {
    p = std::malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
  }
  return p;
}

void hostDeallocate(void *p) noexcept {
  if (!p) return;
  if (cabac_hip_host_free(p) == CABAC_HIP_ERR_INVALID) std::free(p);  // not from cabac_hip_host_alloc: heap memory
}

// ------------------------------------------------------------------ OutputBitstream
void OutputBitstream::write(uint32_t uiBits, uint32_t uiNumberOfBits) {
  // same observable behaviour as bit_stream.cpp:70-117: MSB-first packing, whole bytes to the FIFO
  if (uiNumberOfBits > 32) fail("Number of bits is exceeds '32'");
  if (uiNumberOfBits != 32 && (uiBits & (~0u << uiNumberOfBits)) != 0) fail("Unsupported parameters");
  uint64_t acc = (uint64_t(m_held_bits >> (8 - m_num_held_bits)) << uiNumberOfBits) | uiBits;
  if (m_num_held_bits == 0) acc = uiBits;
  uint32_t total = m_num_held_bits + uiNumberOfBits;
  while (total >= 8) {
    m_fifo.push_back(uint8_t(acc >> (total - 8)));
    total -= 8;
  }
  m_num_held_bits = total;
  m_held_bits = total ? uint8_t((acc & ((1u << total) - 1)) << (8 - total)) : 0;
}

void OutputBitstream::writeAlignZero() {
  if (m_num_held_bits == 0) return;
  m_fifo.push_back(m_held_bits);
  m_held_bits = 0;
  m_num_held_bits = 0;
}

void OutputBitstream::writeAlignOne() {
  uint32_t n = uint32_t(getNumBitsUntilByteAligned());
  write((1u << n) - 1, n);
}

void OutputBitstream::writeByteAlignment() {
  write(1, 1);
  writeAlignZero();
}

void OutputBitstream::addSubstream(OutputBitstream *sub) {
  uint32_t nbits = sub->getNumberOfWrittenBits();
  for (uint8_t b : sub->getFIFO()) write(b, 8);
  if (nbits & 7) write(sub->getHeldBits() >> (8 - (nbits & 7)), nbits & 7);
}

void OutputBitstream::clear() {
  m_fifo.clear();
  m_held_bits = 0;
  m_num_held_bits = 0;
}

// ------------------------------------------------------------------ InputBitstream
uint32_t InputBitstream::readByte() {
  if (m_fifo_idx >= m_fifo.size()) fail("FIFO exceeded");
  return m_fifo[m_fifo_idx++];
}

void InputBitstream::peekPreviousByte(uint32_t &byte) {
  if (m_fifo_idx == 0) fail("FIFO empty");
  byte = m_fifo[m_fifo_idx - 1];
}

void InputBitstream::read(uint32_t uiNumberOfBits, uint32_t &ruiBits) {
  // MSB-first extraction; the unread part of the last byte stays in m_held_bits (its low m_num_held_bits bits).
  // Whether the FIFO holds the bytes this read needs is settled BEFORE anything is taken from it, as the reference does
  // (bit_stream.cpp:240-242): a read that fails leaves the position and the held bits where they were (only the bit
  // counter has moved, as there).
  if (uiNumberOfBits > 32) fail("Too many bits read");
  m_numBitsRead += uiNumberOfBits;
  uint64_t acc = m_held_bits & ((1u << m_num_held_bits) - 1u);
  uint32_t have = m_num_held_bits;
  if (uiNumberOfBits > have) {
    const uint32_t bytes = (uiNumberOfBits - have + 7u) >> 3;  // whole bytes to take: at least one
    if (uint64_t(m_fifo_idx) + bytes > m_fifo.size()) fail("Exceeded FIFO size");
    for (uint32_t i = 0; i < bytes; i++) {
      m_held_bits = m_fifo[m_fifo_idx++];
      acc = (acc << 8) | m_held_bits;
    }
    have += 8 * bytes;
  }
  m_num_held_bits = have - uiNumberOfBits;
  const uint64_t v = acc >> m_num_held_bits;
  ruiBits = uiNumberOfBits == 32 ? uint32_t(v) : uint32_t(v & ((uint64_t(1) << uiNumberOfBits) - 1u));
}

uint32_t InputBitstream::readOutTrailingBits() {
  uint32_t count = 0;
  while (getNumBitsLeft() > 0 && getNumBitsUntilByteAligned() != 0) {
    count++;
    (void)read(1);
  }
  return count;
}

uint32_t InputBitstream::readByteAlignment() {
  if (read(1) != 1) fail("Code is not '1'");
  const uint32_t numBits = getNumBitsUntilByteAligned();
  if (numBits) {
    if (numBits > getNumBitsLeft()) fail("More bits available than left");
    if (read(numBits) != 0) fail("Code not '0'");
  }
  return numBits + 1;
}

InputBitstream *InputBitstream::extractSubstream(uint32_t uiNumBits) {
  const uint32_t nbytes = uiNumBits / 8;
  std::unique_ptr<InputBitstream> r(new InputBitstream);  // a read below may throw
  r->m_fifo.reserve((uiNumBits + 7) >> 3);
  if (m_num_held_bits == 0) {  // byte-aligned source: a plain copy, zero padded past the end of the FIFO
    const uint32_t avail = std::min<uint32_t>(nbytes, uint32_t(m_fifo.size()) - m_fifo_idx);
    r->m_fifo.assign(m_fifo.begin() + m_fifo_idx, m_fifo.begin() + m_fifo_idx + avail);
    r->m_fifo.resize(nbytes, 0);
    m_fifo_idx += avail;
  } else {
    for (uint32_t i = 0; i < nbytes; i++) r->m_fifo.push_back(uint8_t(read(8)));
  }
  if (const uint32_t tail = uiNumBits & 7u) r->m_fifo.push_back(uint8_t(read(tail) << (8 - tail)));  // MSB-aligned
  return r.release();
}

// ------------------------------------------------------------------ BinCounter
void BinCounter::reset() {
  std::fill(m_NumBinsCtx.begin(), m_NumBinsCtx.end(), 0u);
  m_NumBinsEP = 0;
  m_NumBinsTrm = 0;
}

void BinCounter::snapshot(uint32_t *out) const {
  std::copy(m_NumBinsCtx.begin(), m_NumBinsCtx.end(), out);
  out[CABAC_NUM_CONTEXTS] = m_NumBinsEP;
  out[CABAC_NUM_CONTEXTS + 1] = m_NumBinsTrm;
}

void BinCounter::addFromDevice(const uint32_t *counts, const uint32_t *host) {
  for (unsigned k = 0; k < CABAC_NUM_CONTEXTS; k++) m_NumBinsCtx[k] += counts[k] - host[k];
  m_NumBinsEP += counts[CABAC_NUM_CONTEXTS] - host[CABAC_NUM_CONTEXTS];
  m_NumBinsTrm += counts[CABAC_NUM_CONTEXTS + 1] - host[CABAC_NUM_CONTEXTS + 1];
}

uint32_t BinCounter::getAll() const {
  uint32_t count = m_NumBinsEP + m_NumBinsTrm;
  for (uint32_t c : m_NumBinsCtx) count += c;
  return count;
}

// ------------------------------------------------------------------ BinEncIf
void BinEncIf::riceStatReset(int bitDepth) {
  // Ctx::riceStatReset, contexts.cpp:1147-1166 (JVET_W0178 off): 2*floorLog2(bitDepth-10) above 10 bit
  unsigned v = 0;
  if (bitDepth > 10) {
    unsigned x = unsigned(bitDepth - 10), l = 0;
    while (x >>= 1) l++;
    v = 2 * l;
  }
  for (unsigned &s : m_GRAdaptStats) s = v;
}

// ------------------------------------------------------------------ HipBatch
size_t hostLayoutFingerprint(size_t batch, size_t pending, size_t encoder, size_t estimator, size_t decoder, size_t out, size_t in) {
  size_t h = 1469598103934665603ull;
  for (size_t v : {batch, pending, encoder, estimator, decoder, out, in}) h = (h ^ v) * 1099511628211ull;
  return h;
}

void HipBatch::checkLayout(size_t callers) {
  const size_t mine = hostLayoutFingerprint(sizeof(HipBatch), sizeof(HipBatch::Pending), sizeof(BinEncoderHip), sizeof(BitEstimatorHip),
                                            sizeof(BinDecoderHip), sizeof(OutputBitstream), sizeof(InputBitstream));
  if (callers != mine)
    fail("this caller was built against another version of cabac_hip_host.hpp than libcabac_hip.so (class layouts differ): rebuild it");
}

void HipBatch::addPeers(const std::vector<int> &devices) {
  for (size_t k = 1; k < devices.size(); k++) m_peers.emplace_back(new HipBatch(devices[k]));
}

HipBatch::~HipBatch() { cabac_hip_destroy(m_ctx); }

cabac_hip_ctx *HipBatch::handle() {
  // Recording needs no device; coding does.  No GPU -> exception (there is no CPU path).
  if (!m_ctx) check_status(nullptr, cabac_hip_init(m_device, &m_ctx), "cabac_hip_init");
  return m_ctx;
}

cabac_tu_desc makeTuDesc(const HipBatch::ResidualBlock &b, uint64_t coeff_offset) {
  unsigned lw = 0, lh = 0;
  while ((1u << lw) < b.width) lw++;
  while ((1u << lh) < b.height) lh++;
  if ((1u << lw) != b.width || (1u << lh) != b.height || lw > 6 || lh > 6)
    throw Exception("residual: block sizes must be powers of two up to 64");
  cabac_tu_desc t{};
  t.coeff_offset = coeff_offset;
  t.log2_width = uint8_t(lw);
  t.log2_height = uint8_t(lh);
  t.channel = b.chroma ? 1 : 0;
  t.flags = uint8_t((b.depQuant ? CABAC_TU_DEP_QUANT : 0u) | (b.signHiding ? CABAC_TU_SIGN_HIDING : 0u) |
                    (b.tsFlag ? CABAC_TU_TS_FLAG : 0u) | (b.transformSkip ? CABAC_TU_TRANSFORM_SKIP : 0u) |
                    (b.transformSkip && b.bdpcm ? CABAC_TU_BDPCM : 0u) | (b.sbtZeroOut && !b.transformSkip ? CABAC_TU_SBT_ZERO_OUT : 0u));
  t.max_log2_tr_range = uint8_t(b.maxLog2TrDynamicRange);
  return t;
}

uint64_t HipBatch::stageCoefficients(const int32_t *coeff, size_t n, int maxLog2TrDynamicRange) {
  const uint64_t at = stagedCoefficients();
  if (m_narrow && (maxLog2TrDynamicRange == 0 || maxLog2TrDynamicRange <= 15)) {
    m_stageCoeff16.resize(at + n);
    int16_t *dst = m_stageCoeff16.data() + at;
    uint32_t outside = 0;
    for (size_t i = 0; i < n; i++) {
      dst[i] = int16_t(coeff[i]);
      outside |= (uint32_t(coeff[i]) + 32768u) >> 16;  // non-zero iff the value does not fit
    }
    if (!outside) {
      m_stagedBlocksOpen++;
      return at;
    }
    m_stageCoeff16.resize(at);  // a coefficient beyond the range the block declares: keep it as it is, in 32 bits
  }
  if (m_narrow) {  // from here on the staging area is the reference's TCoeff
    m_stageCoeff.assign(m_stageCoeff16.begin(), m_stageCoeff16.end());
    m_stageCoeff16.clear();
    m_narrow = false;
  }
  m_stageCoeff.insert(m_stageCoeff.end(), coeff, coeff + n);
  m_stagedBlocksOpen++;
  return at;
}

void HipBatch::clearStage() {
  m_stageCoeff.clear();
  m_stageCoeff16.clear();
  m_narrow = true;
}

uint64_t HipBatch::restage(HipBatch &from, uint64_t at, size_t n) {
  if (!from.m_narrow) return stageCoefficients(from.m_stageCoeff.data() + at, n, 16);  // (wide stays wide)
  const uint64_t to = stagedCoefficients();
  if (m_narrow) {
    m_stageCoeff16.insert(m_stageCoeff16.end(), from.m_stageCoeff16.begin() + at, from.m_stageCoeff16.begin() + at + n);
  } else {
    m_stageCoeff.insert(m_stageCoeff.end(), from.m_stageCoeff16.begin() + at, from.m_stageCoeff16.begin() + at + n);
  }
  m_stagedBlocksOpen++;
  return to;
}

// the coded substream into its container, leaving it as the reference's finish() leaves its bitstream
void HipBatch::deliverBytes(Pending &p, const uint8_t *src, uint32_t nbits) {
  const uint32_t whole = nbits / 8, tail = nbits & 7;
  if (p.deliver) p.deliver(src, whole, tail);
  OutputBitstream *sink = p.sink;
  if (!sink) return;
  if (sink->m_num_held_bits == 0) {
    sink->m_fifo.insert(sink->m_fifo.end(), src, src + whole);
  } else {
    for (uint32_t i = 0; i < whole; i++) sink->write(src[i], 8);
  }
  if (tail) sink->write(uint32_t(src[whole]) >> (8 - tail), tail);
}

namespace {
// longest-processing-time-first: item k (weight w[k]) to the least loaded of n_bins bins; a bin holds its items longest first
// (a wave decodes consecutive substreams and runs as long as its longest; and a share whose few long substreams come first gets
// a wave per substream from the decode dispatch)
std::vector<std::vector<uint32_t>> lptAssign(const std::vector<uint64_t> &w, size_t n_bins) {
  std::vector<uint32_t> order(w.size());
  for (uint32_t k = 0; k < order.size(); k++) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return w[a] > w[b]; });
  std::vector<uint64_t> load(n_bins, 0);
  std::vector<std::vector<uint32_t>> bins(n_bins);
  for (uint32_t k : order) {
    const size_t b = size_t(std::min_element(load.begin(), load.end()) - load.begin());
    bins[b].push_back(k);
    load[b] += w[k] + 1;
  }
  return bins;
}

// run job(k) for k = 1 .. n - 1 on threads of their own and job(0) here; the first exception is rethrown after all have ended
template <class F>
void onEveryDevice(size_t n, F job) {
  std::vector<std::exception_ptr> err(n);
  std::vector<std::thread> th;
  for (size_t k = 1; k < n; k++)
    th.emplace_back([&, k] {
      try {
        job(k);
      } catch (...) {
        err[k] = std::current_exception();
      }
    });
  try {
    job(0);
  } catch (...) {
    err[0] = std::current_exception();
  }
  for (std::thread &t : th) t.join();
  for (const std::exception_ptr &e : err)
    if (e) std::rethrow_exception(e);
}
}  // namespace

void HipBatch::flush() {
  if (m_pending.empty()) return;
  std::vector<Pending> done;
  done.swap(m_pending);
  std::vector<uint64_t> weight(done.size());
  for (size_t k = 0; k < done.size(); k++) {
    weight[k] = done[k].records.size();
    for (const cabac_tu_desc &t : done[k].blocks) weight[k] += uint64_t(1) << (t.log2_width + t.log2_height);
  }
  if (m_peers.empty()) {
    // one device: the substreams longest first — consecutive substreams share a wave / a workgroup, which runs as long as its
    // longest (every substream's bytes go to its own bitstream, so the order is nobody else's business)
    std::vector<uint32_t> order(done.size());
    for (uint32_t k = 0; k < order.size(); k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return weight[a] > weight[b]; });
    std::vector<Pending> sorted;
    sorted.reserve(done.size());
    for (uint32_t k : order) sorted.push_back(std::move(done[k]));
    return flushLocal(sorted);
  }
  // several devices: deal the substreams out, longest first, and code every share at the same time
  const std::vector<std::vector<uint32_t>> share = lptAssign(weight, deviceCount());
  std::vector<std::vector<Pending>> part(deviceCount());
  size_t moved_blocks = 0;
  for (size_t dv = 0; dv < share.size(); dv++)
    for (uint32_t k : share[dv]) {
      Pending &p = done[k];
      if (dv > 0 && !p.blocks.empty()) {  // its coefficients move to the staging of the device that will read them
        HipBatch &peer = *m_peers[dv - 1];
        for (cabac_tu_desc &t : p.blocks) {
          const size_t n = size_t(1) << (t.log2_width + t.log2_height);
          t.coeff_offset = peer.restage(*this, t.coeff_offset, n);
        }
        moved_blocks += p.blocks.size();
      }
      part[dv].push_back(std::move(p));
    }
  m_stagedBlocksOpen -= std::min(m_stagedBlocksOpen, moved_blocks);
  onEveryDevice(deviceCount(), [&](size_t dv) {
    if (part[dv].empty()) return;
    if (dv == 0) flushLocal(part[0]);
    else m_peers[dv - 1]->flushLocal(part[dv]);
  });
  if (m_stagedBlocksOpen == 0) clearStage();
}

void HipBatch::flushLocal(std::vector<Pending> &done) {
  for (const Pending &p : done)
    if (!p.blocks.empty()) return flushSpliced(done);
  const uint32_t n = uint32_t(done.size());
  std::vector<cabac_substream_desc> desc(n);
  uint64_t rec_total = 0, byte_total = 0;
  for (uint32_t s = 0; s < n; s++) {
    Pending &p = done[s];
    desc[s].rec_offset = rec_total;
    desc[s].byte_offset = byte_total;
    desc[s].n_records = uint32_t(p.records.size());
    desc[s].byte_capacity = uint32_t(cabac_hip_encode_bound(p.nCtx, p.nEp, p.nTrm));
    desc[s].qp = p.qp;
    desc[s].init_id = uint32_t(p.initId) | CABAC_SUB_FINISH;
    rec_total += p.records.size();
    byte_total += desc[s].byte_capacity;  // encode_bound is a multiple of 16
  }
  RecordVector &records = m_stageRecords;
  ByteVector &bytes = m_stageBytes;
  if (records.size() < rec_total + 1) records.resize(rec_total + 1);
  if (bytes.size() < byte_total + 1) bytes.resize(byte_total + 1);
  for (uint32_t s = 0; s < n; s++)
    if (!done[s].records.empty())
      std::memcpy(records.data() + desc[s].rec_offset, done[s].records.data(), done[s].records.size() * 2);
  std::vector<cabac_substream_result> res(n);
  int rc = cabac_hip_encode_batch(handle(), n, desc.data(), records.data(), rec_total, bytes.data(), byte_total,
                                  res.data());
  check_status(m_ctx, rc, "cabac_hip_encode_batch");
  for (uint32_t s = 0; s < n; s++) deliverBytes(done[s], bytes.data() + desc[s].byte_offset, res[s].n_bits);
}

// at least one substream has residual blocks to be spliced in on the device: coefficients -> bytes
// (cabac_hip_encode_batch_residual; the bins of residual_coding go "straight into the encoder", cabac_writer.cpp:2424-2525)
void HipBatch::flushSpliced(std::vector<Pending> &done) {
  const uint32_t n = uint32_t(done.size());
  std::vector<cabac_substream_desc> desc(n);
  std::vector<uint32_t> first(size_t(n) + 1, 0);
  std::vector<cabac_splice> splices;
  std::vector<cabac_tu_desc> tus;
  uint64_t rec_total = 0, n_blocks = 0;
  bool want_counts = false;
  for (uint32_t s = 0; s < n; s++) {
    Pending &p = done[s];
    desc[s] = cabac_substream_desc{};
    desc[s].rec_offset = rec_total;
    desc[s].n_records = uint32_t(p.records.size());
    desc[s].qp = p.qp;
    desc[s].init_id = uint32_t(p.initId) | CABAC_SUB_FINISH;
    rec_total += p.records.size();
    for (const cabac_splice &sp : p.splices) splices.push_back(cabac_splice{sp.at, uint32_t(tus.size()) + sp.tu});
    tus.insert(tus.end(), p.blocks.begin(), p.blocks.end());
    first[s + 1] = uint32_t(splices.size());
    n_blocks += p.blocks.size();
    want_counts = want_counts || bool(p.counted);
  }
  RecordVector &records = m_stageRecords;
  if (records.size() < rec_total + 1) records.resize(rec_total + 1);
  for (uint32_t s = 0; s < n; s++)
    if (!done[s].records.empty())
      std::memcpy(records.data() + desc[s].rec_offset, done[s].records.data(), done[s].records.size() * 2);
  std::vector<cabac_substream_result> res(n);
  std::vector<uint64_t> offsets(size_t(n) + 1, 0);
  std::vector<uint32_t> info(tus.size() ? tus.size() : 1, 0), counts(want_counts ? size_t(n) * CABAC_BIN_COUNT_WORDS : 0);
  // The coded size is only known on the device.  A first guess from the input sizes; the call says when it does not fit.
  size_t capacity = std::max<size_t>(size_t(1) << 16, rec_total + stagedCoefficients() / 2);
  int rc = CABAC_HIP_OK;
  for (int attempt = 0; attempt < 6; attempt++) {
    if (m_stageBytes.size() < capacity) m_stageBytes.resize(capacity);
    rc = m_narrow ? cabac_hip_encode_batch_residual16(handle(), n, desc.data(), records.data(), rec_total, first.data(), splices.data(),
                                                      uint32_t(tus.size()), tus.data(), m_stageCoeff16.data(), m_stageCoeff16.size(),
                                                      m_stageBytes.data(), m_stageBytes.size(), offsets.data(), res.data(), info.data(),
                                                      want_counts ? counts.data() : nullptr)
                  : cabac_hip_encode_batch_residual(handle(), n, desc.data(), records.data(), rec_total, first.data(), splices.data(),
                                                    uint32_t(tus.size()), tus.data(), m_stageCoeff.data(), m_stageCoeff.size(),
                                                    m_stageBytes.data(), m_stageBytes.size(), offsets.data(), res.data(), info.data(),
                                                    want_counts ? counts.data() : nullptr);
    if (rc != CABAC_HIP_ERR_INVALID || std::string(cabac_hip_last_error(m_ctx)) != "payload_capacity too small") break;
    capacity *= 4;
  }
  m_stagedBlocksOpen -= std::min<size_t>(m_stagedBlocksOpen, size_t(n_blocks));
  if (m_stagedBlocksOpen == 0) clearStage();  // (blocks of encoders still recording keep their place otherwise)
  if (rc == CABAC_HIP_ERR_SUBSTREAM)
    for (uint32_t t = 0; t < tus.size(); t++)
      if (info[t] & CABAC_TU_INFO_EMPTY) throw Exception("Coefficient coding called for empty TU");
  check_status(m_ctx, rc, "cabac_hip_encode_batch_residual");
  size_t t0 = 0;
  for (uint32_t s = 0; s < n; s++) {
    Pending &p = done[s];
    if (p.blockInfo)
      for (size_t k = 0; k < p.blocks.size(); k++) p.blockInfo(k, info[t0 + k]);
    t0 += p.blocks.size();
    if (p.counted) p.counted(counts.data() + size_t(s) * CABAC_BIN_COUNT_WORDS, p.hostCounts.data());
    deliverBytes(p, m_stageBytes.data() + offsets[s], res[s].n_bits);
  }
}

uint32_t HipBatch::numWrittenBits(const uint16_t *records, size_t n_records, int qp, int initId) {
  cabac_substream_desc d{};
  d.n_records = uint32_t(n_records);
  d.qp = qp;
  d.init_id = uint32_t(initId) | CABAC_SUB_PROBE;
  d.byte_capacity = uint32_t(cabac_hip_encode_bound(n_records, n_records, n_records));  // any record kind, worst case
  std::vector<uint8_t> scratch(d.byte_capacity + 1u);
  const uint16_t none = 0;
  cabac_substream_result res{};
  const int rc = cabac_hip_encode_batch(handle(), 1, &d, n_records ? records : &none, n_records, scratch.data(), d.byte_capacity, &res);
  check_status(m_ctx, rc, "cabac_hip_encode_batch (probe)");
  return res.n_bits;
}

std::vector<uint64_t> HipBatch::estimate(const std::vector<EstimateJob> &jobs) {
  const uint32_t n = uint32_t(jobs.size());
  std::vector<uint64_t> cost(n, 0);
  if (n == 0) return cost;
  std::vector<cabac_substream_desc> desc(n);
  uint64_t rec_total = 0;
  for (uint32_t s = 0; s < n; s++) {
    desc[s] = cabac_substream_desc{};
    desc[s].rec_offset = rec_total;
    desc[s].n_records = jobs[s].n_records;
    desc[s].qp = jobs[s].qp;
    desc[s].init_id = uint32_t(jobs[s].initId);
    rec_total += jobs[s].n_records;
  }
  std::vector<uint16_t> records(rec_total ? rec_total : 1);
  for (uint32_t s = 0; s < n; s++)
    if (jobs[s].n_records) std::memcpy(records.data() + desc[s].rec_offset, jobs[s].records, size_t(jobs[s].n_records) * 2);
  const int rc = cabac_hip_estimate_batch(handle(), n, desc.data(), records.data(), rec_total, cost.data(), nullptr);
  check_status(m_ctx, rc, "cabac_hip_estimate_batch");
  return cost;
}

HipBatch::ResidualResult HipBatch::residual(const std::vector<ResidualBlock> &blocks) {
  ResidualResult r;
  const uint32_t n = uint32_t(blocks.size());
  r.offsets.assign(size_t(n) + 1, 0);
  r.info.assign(n, 0);
  if (n == 0) return r;
  std::vector<cabac_tu_desc> tus(n);
  uint64_t total = 0;
  for (uint32_t t = 0; t < n; t++) {
    const ResidualBlock &b = blocks[t];
    if (!b.coeff) throw Exception("residual: block without coefficients");
    tus[t] = makeTuDesc(b, total);
    total += uint64_t(b.width) * b.height;
  }
  std::vector<int32_t> coeff(total);
  for (uint32_t t = 0; t < n; t++)
    std::memcpy(coeff.data() + tus[t].coeff_offset, blocks[t].coeff, size_t(blocks[t].width) * blocks[t].height * sizeof(int32_t));
  int rc = cabac_hip_residual_batch(handle(), n, tus.data(), coeff.data(), total, r.offsets.data(), r.info.data(), nullptr, 0);
  if (rc == CABAC_HIP_ERR_SUBSTREAM) {
    for (uint32_t t = 0; t < n; t++)
      if (r.info[t] & CABAC_TU_INFO_EMPTY) throw Exception("Coefficient coding called for empty TU");
  }
  check_status(m_ctx, rc, "cabac_hip_residual_batch");
  r.records.assign(size_t(r.offsets[n]) ? size_t(r.offsets[n]) : 1, 0);
  rc = cabac_hip_residual_batch(m_ctx, n, tus.data(), coeff.data(), total, r.offsets.data(), r.info.data(), r.records.data(),
                                r.records.size());
  check_status(m_ctx, rc, "cabac_hip_residual_batch");
  r.records.resize(size_t(r.offsets[n]));
  return r;
}

std::vector<std::vector<int32_t>> HipBatch::residualParse(const std::vector<ParseJob> &jobs, std::vector<uint32_t> *info_out) {
  const uint32_t n = uint32_t(jobs.size());
  std::vector<std::vector<int32_t>> out(n);
  if (n == 0) return out;
  std::vector<cabac_substream_desc> desc(n);
  std::vector<uint32_t> first(size_t(n) + 1, 0);
  std::vector<cabac_tu_desc> tus;
  uint64_t byte_total = 0, coeff_total = 0;
  for (uint32_t s = 0; s < n; s++) {
    desc[s] = cabac_substream_desc{};
    desc[s].byte_offset = byte_total;
    desc[s].byte_capacity = jobs[s].n_bytes;
    desc[s].qp = jobs[s].qp;
    desc[s].init_id = uint32_t(jobs[s].initId) | CABAC_SUB_FINISH;
    byte_total += (uint64_t(jobs[s].n_bytes) + 15) / 16 * 16;
    for (const ResidualBlock &b : jobs[s].blocks) {
      unsigned lw = 0, lh = 0;
      while ((1u << lw) < b.width) lw++;
      while ((1u << lh) < b.height) lh++;
      if ((1u << lw) != b.width || (1u << lh) != b.height || lw > 6 || lh > 6)
        throw Exception("residualParse: power-of-two blocks up to 64");
      cabac_tu_desc t{};
      t.coeff_offset = coeff_total;
      t.log2_width = uint8_t(lw);
      t.log2_height = uint8_t(lh);
      t.channel = b.chroma ? 1 : 0;
      t.flags = uint8_t((b.depQuant ? CABAC_TU_DEP_QUANT : 0u) | (b.signHiding ? CABAC_TU_SIGN_HIDING : 0u) |
                        (b.tsFlag ? CABAC_TU_TS_FLAG : 0u) | (b.transformSkip ? CABAC_TU_TRANSFORM_SKIP : 0u) |
                        (b.bdpcm ? CABAC_TU_BDPCM : 0u) | (b.sbtZeroOut ? CABAC_TU_SBT_ZERO_OUT : 0u));
      t.max_log2_tr_range = uint8_t(b.maxLog2TrDynamicRange);
      tus.push_back(t);
      coeff_total += uint64_t(b.width) * b.height;
    }
    first[s + 1] = uint32_t(tus.size());
  }
  std::vector<uint8_t> bytes(byte_total ? byte_total : 16, 0);
  for (uint32_t s = 0; s < n; s++)
    if (jobs[s].n_bytes) std::memcpy(bytes.data() + desc[s].byte_offset, jobs[s].bytes, jobs[s].n_bytes);
  std::vector<cabac_substream_result> res(n);
  bool narrow = !tus.empty();  // every block of 15-bit dynamic range: the blocks come back as int16 (half the bytes over PCIe)
  for (const cabac_tu_desc &t : tus) narrow = narrow && (t.max_log2_tr_range == 0 || t.max_log2_tr_range <= 15);
  if (tus.empty()) tus.push_back(cabac_tu_desc{});
  if (info_out) info_out->assign(tus.size(), 0u);
  std::vector<int16_t> coeff16;
  std::vector<int32_t> coeff;
  int rc = CABAC_HIP_OK;
  if (narrow) {
    coeff16.assign(coeff_total ? coeff_total : 1, 0);
    rc = cabac_hip_residual_parse_batch16(handle(), n, desc.data(), bytes.data(), bytes.size(), first.data(), tus.data(), coeff16.data(),
                                          coeff_total, info_out ? info_out->data() : nullptr, res.data());
    if (rc == CABAC_HIP_ERR_SUBSTREAM)
      for (uint32_t s = 0; s < n; s++)
        if (res[s].flags & CABAC_RES_RANGE) narrow = false;  // a level beyond what the stream declares: once more, in 32 bits
  }
  if (!narrow) {
    coeff.assign(coeff_total ? coeff_total : 1, 0);
    rc = cabac_hip_residual_parse_batch(handle(), n, desc.data(), bytes.data(), bytes.size(), first.data(), tus.data(), coeff.data(),
                                        coeff_total, info_out ? info_out->data() : nullptr, res.data());
  }
  if (rc == CABAC_HIP_ERR_SUBSTREAM) {
    for (uint32_t s = 0; s < n; s++) {
      if (res[s].flags & CABAC_RES_UNDERRUN) throw Exception("FIFO exceeded");
      if (res[s].flags & CABAC_RES_BAD_STOP) throw Exception("No proper stop/alignment pattern at end of CABAC stream.");
      if (res[s].flags) throw Exception("residualParse: block not covered by the parser");
    }
  }
  check_status(m_ctx, rc, "cabac_hip_residual_parse_batch");
  uint64_t at = 0;
  for (uint32_t s = 0; s < n; s++) {
    uint64_t len = 0;
    for (const ResidualBlock &b : jobs[s].blocks) len += uint64_t(b.width) * b.height;
    if (narrow) out[s].assign(coeff16.begin() + at, coeff16.begin() + at + len);
    else out[s].assign(coeff.begin() + at, coeff.begin() + at + len);
    at += len;
  }
  return out;
}

void HipBatch::decode(const std::vector<DecodeJob> &jobs, std::vector<std::vector<uint8_t>> &bins,
                      std::vector<uint32_t> *bitsRead) {
  const uint32_t n = uint32_t(jobs.size());
  if (!m_peers.empty() && n > 1) {  // several devices: the jobs dealt out longest first, every share decoded at the same time
    std::vector<uint64_t> weight(n);
    for (uint32_t k = 0; k < n; k++) weight[k] = jobs[k].n_records;
    const std::vector<std::vector<uint32_t>> share = lptAssign(weight, deviceCount());
    std::vector<std::vector<DecodeJob>> part(deviceCount());
    std::vector<std::vector<std::vector<uint8_t>>> part_bins(deviceCount());
    std::vector<std::vector<uint32_t>> part_bits(deviceCount());
    for (size_t dv = 0; dv < share.size(); dv++)
      for (uint32_t k : share[dv]) part[dv].push_back(jobs[k]);
    onEveryDevice(deviceCount(), [&](size_t dv) {
      if (part[dv].empty()) return;
      HipBatch single(dv == 0 ? m_device : m_peers[dv - 1]->m_device);  // (a one-device view of that device's context)
      HipBatch &owner = dv == 0 ? *this : *m_peers[dv - 1];
      single.m_ctx = owner.handle();
      try {
        single.decode(part[dv], part_bins[dv], &part_bits[dv]);
      } catch (...) {
        single.m_ctx = nullptr;
        throw;
      }
      single.m_ctx = nullptr;
    });
    bins.assign(n, {});
    if (bitsRead) bitsRead->assign(n, 0);
    for (size_t dv = 0; dv < share.size(); dv++)
      for (size_t i = 0; i < share[dv].size(); i++) {
        bins[share[dv][i]].swap(part_bins[dv][i]);
        if (bitsRead) (*bitsRead)[share[dv][i]] = part_bits[dv][i];
      }
    return;
  }
  bins.assign(n, {});
  if (bitsRead) bitsRead->assign(n, 0);
  if (n == 0) return;
  std::vector<cabac_substream_desc> desc(n);
  uint64_t rec_total = 0, byte_total = 0;
  for (uint32_t s = 0; s < n; s++) {
    desc[s].rec_offset = rec_total;
    desc[s].byte_offset = byte_total;
    desc[s].n_records = jobs[s].n_records;
    desc[s].byte_capacity = jobs[s].n_bytes;
    desc[s].qp = jobs[s].qp;
    desc[s].init_id = uint32_t(jobs[s].initId) | (jobs[s].finish ? CABAC_SUB_FINISH : 0u);
    rec_total += jobs[s].n_records;
    byte_total += (uint64_t(jobs[s].n_bytes) + 15) / 16 * 16;
  }
  std::vector<uint16_t> records(rec_total ? rec_total : 1);
  std::vector<uint8_t> bytes(byte_total ? byte_total : 16, 0);
  for (uint32_t s = 0; s < n; s++) {
    if (jobs[s].n_records) std::memcpy(records.data() + desc[s].rec_offset, jobs[s].records, jobs[s].n_records * 2);
    if (jobs[s].n_bytes) std::memcpy(bytes.data() + desc[s].byte_offset, jobs[s].bytes, jobs[s].n_bytes);
  }
  std::vector<uint8_t> out(rec_total ? rec_total : 1);
  std::vector<cabac_substream_result> res(n);
  int rc = cabac_hip_decode_batch(handle(), n, desc.data(), records.data(), rec_total, bytes.data(), byte_total,
                                  out.data(), res.data());
  if (rc == CABAC_HIP_ERR_SUBSTREAM) {
    for (uint32_t s = 0; s < n; s++) {
      if (res[s].flags & CABAC_RES_UNDERRUN) fail("FIFO exceeded");
      if (res[s].flags & CABAC_RES_BAD_STOP) fail("No proper stop/alignment pattern at end of CABAC stream.");
      if (res[s].flags & CABAC_RES_BAD_RECORD) fail("invalid bin record");
    }
  }
  check_status(m_ctx, rc, "cabac_hip_decode_batch");
  for (uint32_t s = 0; s < n; s++) {
    bins[s].assign(out.begin() + desc[s].rec_offset, out.begin() + desc[s].rec_offset + jobs[s].n_records);
    if (bitsRead) (*bitsRead)[s] = res[s].n_bits;
  }
}

// ------------------------------------------------------------------ BinEncoderHip
void BinEncoderHip::start() {
  m_records.clear();
  m_splices.clear();
  m_blocks.clear();
  BinCounter::reset();
}

void BinEncoderHip::encodeResidual(const HipBatch::ResidualBlock &b) {
  if (!b.coeff) fail("encodeResidual: block without coefficients");
  if (b.tsFlag) fail("encodeResidual: code transform_skip_flag with encodeBin before the block (tsFlag must be false)");
  cabac_tu_desc t = makeTuDesc(b, 0);
  t.coeff_offset = m_batch.stageCoefficients(b.coeff, size_t(b.width) * b.height, b.maxLog2TrDynamicRange);
  m_splices.push_back(cabac_splice{uint32_t(m_records.size()), uint32_t(m_blocks.size())});
  m_blocks.push_back(t);
}

void BinEncoderHip::restart() {
  if (!m_records.empty()) fail("restart(): bins already recorded for this substream");
}

void BinEncoderHip::reset(int qp, int initId) {
  if (initId < 0 || initId > 2) fail("Invalid initId");
  m_qp = qp;
  m_initId = initId;
  for (unsigned &s : m_GRAdaptStats) s = 0;  // Ctx::init, contexts.cpp:1141-1144
  start();
}

void BinEncoderHip::resetBits() {
  if (!m_records.empty()) fail("resetBits(): bins already recorded for this substream");
  BinCounter::reset();
}

// ---- BitEstimatorHip ------------------------------------------------------------------------------------
void BitEstimatorHip::reset(int qp, int initId) {  // Ctx::init + cost := 0, arith_codec.cpp:623-626
  m_records.clear();
  m_qp = qp;
  m_initId = initId;
  m_valid = true;
  m_cached = 0;
}

uint64_t BitEstimatorHip::getEstFracBits() const {
  if (!m_valid) {
    HipBatch::EstimateJob job{m_records.data(), uint32_t(m_records.size()), m_qp, m_initId};
    m_cached = m_batch.estimate({job})[0];
    m_valid = true;
  }
  return m_cached;
}

void BitEstimatorHip::encodeBin(unsigned bin, unsigned ctxId) {
  if (ctxId >= CABAC_NUM_CONTEXTS) fail("ctxId out of range");
  put(ctxId | (bin ? CABAC_REC_BIN : 0u));
}

void BitEstimatorHip::encodeBinsEP(unsigned, unsigned numBins) {  // arith_codec.cpp:640-642: numBins bits
  for (unsigned i = 0; i < numBins; i++) put(CABAC_REC_EP);
}

void BitEstimatorHip::encodeRemAbsEP(unsigned bins, unsigned goRicePar, unsigned cutoff, int maxLog2TrDynamicRange) {
  // one bit per bypass bin of the code word (arith_codec.cpp:653-677)
  const unsigned n = cabac_code::rem_abs_code(bins, goRicePar, cutoff, unsigned(maxLog2TrDynamicRange)).length();
  for (unsigned i = 0; i < n; i++) put(CABAC_REC_EP);
}

void BinEncoderHip::encodeBin(unsigned bin, unsigned ctxId) {
  if (ctxId >= CABAC_NUM_CONTEXTS) fail("ctxId out of range");
  BinCounter::addCtx(ctxId);
  put(ctxId, bin);
}

void BinEncoderHip::encodeBinEP(unsigned bin) {
  BinCounter::addEP();
  put(CABAC_REC_EP, bin);
}

void BinEncoderHip::encodeBinsEP(unsigned bins, unsigned numBins) {
  // arith_codec.cpp:401-424: arithmetically numBins single bypass bins, MSB first
  if (numBins > 32) fail("encodeBinsEP: more than 32 bins");
  if (numBins < 32 && (bins >> numBins) != 0) fail("encodeBinsEP: value has bits above numBins");
  BinCounter::addEP(numBins);
  for (int i = int(numBins) - 1; i >= 0; i--) put(CABAC_REC_EP, (bins >> i) & 1u);
}

void BinEncoderHip::encodeRemAbsEP(unsigned bins, unsigned goRicePar, unsigned cutoff, int maxLog2TrDynamicRange) {
  // arith_codec.cpp:426-458: the code word's bypass bins as records, in order
  const cabac_code::RemAbsCode c = cabac_code::rem_abs_code(bins, goRicePar, cutoff, unsigned(maxLog2TrDynamicRange));
  BinCounter::addEP(c.length());
  for (uint32_t i = 0; i < c.ones; i++) put(CABAC_REC_EP, 1);
  if (c.stop) put(CABAC_REC_EP, 0);
  for (uint32_t i = c.tail_bits; i-- > 0;) put(CABAC_REC_EP, (c.tail >> i) & 1u);
}

unsigned BinEncoderHip::getNumWrittenBits() {
  if (m_mode != Immediate) fail("getNumWrittenBits: nothing is coded before HipBatch::flush() in Deferred mode");
  if (!m_Bitstream) fail("getNumWrittenBits: no bitstream (init not called)");
  if (!m_blocks.empty()) fail("getNumWrittenBits: the bins of spliced residual blocks are not known before flush()");
  return m_Bitstream->getNumberOfWrittenBits() + m_batch.numWrittenBits(m_records.data(), m_records.size(), m_qp, m_initId);
}

void BinEncoderHip::encodeBinTrm(unsigned bin) {
  BinCounter::addTrm();
  put(CABAC_REC_TRM, bin);
}

void BinEncoderHip::align() { put(CABAC_REC_ALIGN, 0); }

void BinEncoderHip::finish() {
  if (!m_Bitstream) fail("finish(): no bitstream (init not called)");
  HipBatch::Pending p;
  p.records.swap(m_records);
  p.qp = m_qp;
  p.initId = m_initId;
  p.nEp = BinCounter::getEP();
  p.nTrm = BinCounter::getTrm();
  p.nCtx = BinCounter::getAll() - p.nEp - p.nTrm;
  p.sink = m_Bitstream;
  if (!m_blocks.empty()) {
    p.splices.swap(m_splices);
    p.blocks.swap(m_blocks);
    p.hostCounts.resize(CABAC_BIN_COUNT_WORDS);
    BinCounter::snapshot(p.hostCounts.data());
    BinCounter *counter = this;  // the spliced blocks' bins are counted when the device has told how many there were
    p.counted = [counter](const uint32_t *counts, const uint32_t *host) { counter->addFromDevice(counts, host); };
  }
  m_batch.submit(std::move(p));
  if (m_mode == Immediate) m_batch.flush();
}

// ------------------------------------------------------------------ binarisation helpers
void unary_max_symbol(BinEncIf &e, unsigned symbol, unsigned ctxId0, unsigned ctxIdN, unsigned maxSymbol) {
  if (symbol > maxSymbol) fail("symbol > maxSymbol");
  const unsigned total = std::min(symbol + 1, maxSymbol);
  for (unsigned k = 0; k < total; ++k) e.encodeBin(symbol > k, k == 0 ? ctxId0 : ctxIdN);
}

void unary_max_eqprob(BinEncIf &e, unsigned symbol, unsigned maxSymbol) {
  if (maxSymbol == 0) return;
  const bool codeLast = maxSymbol > symbol;
  unsigned bins = 0, numBins = 0;
  while (symbol--) {
    bins = (bins << 1) + 1;
    numBins++;
  }
  if (codeLast) {
    bins <<= 1;
    numBins++;
  }
  if (numBins > 32) fail("Unspecified error");
  e.encodeBinsEP(bins, numBins);
}

void exp_golomb_eqprob(BinEncIf &e, unsigned symbol, unsigned count) {
  unsigned bins = 0, numBins = 0;
  while (symbol >= (1u << count)) {
    bins = (bins << 1) + 1;
    numBins++;
    symbol -= 1u << count;
    count++;
  }
  bins <<= 1;
  numBins++;
  e.encodeBinsEP(bins, numBins);
  e.encodeBinsEP(symbol, count);
}

void xWriteTruncBinCode(BinEncIf &e, uint32_t symbol, uint32_t maxSymbol) {
  if (maxSymbol == 0 || symbol >= maxSymbol) fail("xWriteTruncBinCode: symbol >= maxSymbol");
  unsigned thresh = 0;  // g_tbMax[maxSymbol] == floor(log2(maxSymbol)), rom.hpp:43-54
  for (uint32_t x = maxSymbol; x >>= 1;) thresh++;
  const uint32_t val = 1u << thresh, b = maxSymbol - val;
  if (symbol < val - b) {
    e.encodeBinsEP(symbol, thresh);
  } else {
    e.encodeBinsEP(symbol + val - b, thresh + 1);
  }
}

// ------------------------------------------------------------------ BinDecoderHip
void BinDecoderHip::reset(int qp, int initId) {
  m_qp = qp;
  m_initId = initId;
  m_plan.clear();
  m_bins.clear();
  m_pos = 0;
}

void BinDecoderHip::run(bool checkFinish) {
  if (!m_Bitstream) fail("run(): no bitstream");
  HipBatch::DecodeJob job;
  job.records = m_plan.data();
  job.n_records = uint32_t(m_plan.size());
  job.bytes = m_Bitstream->m_fifo.data() + m_Bitstream->m_fifo_idx;
  job.n_bytes = uint32_t(m_Bitstream->m_fifo.size()) - m_Bitstream->m_fifo_idx;
  job.qp = m_qp;
  job.initId = m_initId;
  job.finish = checkFinish;
  std::vector<std::vector<uint8_t>> bins;
  std::vector<uint32_t> bitsRead;
  m_batch.decode({job}, bins, &bitsRead);
  m_bins.swap(bins[0]);
  m_bitsRead = bitsRead[0];
  m_Bitstream->m_fifo_idx += (m_bitsRead + 8) / 8;  // bytes the reference decoder would have consumed
  m_pos = 0;
}

unsigned BinDecoderHip::next(unsigned id) {
  if (m_pos >= m_plan.size()) fail("decode call beyond the planned sequence");
  if ((m_plan[m_pos] & CABAC_REC_ID_MASK) != id) fail("decode call does not match the planned ctxId sequence");
  return m_bins[m_pos++];
}

unsigned BinDecoderHip::decodeBin(unsigned ctxId) { return next(ctxId); }
unsigned BinDecoderHip::decodeBinEP() { return next(CABAC_REC_EP); }
unsigned BinDecoderHip::decodeBinTrm() { return next(CABAC_REC_TRM); }
unsigned BinDecoderHip::decodeBinsEP(unsigned numBins) {
  unsigned bins = 0;
  for (unsigned i = 0; i < numBins; i++) bins = (bins << 1) | next(CABAC_REC_EP);
  return bins;
}

unsigned BinDecoderHip::decodeRemAbsEP(unsigned goRicePar, unsigned cutoff, int maxLog2TrDynamicRange) {
  // unary prefix of at most 32 - maxLog2TrDynamicRange ones, then a suffix whose length follows from the prefix
  const unsigned maxPrefix = 32u - unsigned(maxLog2TrDynamicRange);
  unsigned prefix = 0;
  while (prefix < maxPrefix && decodeBinEP()) prefix++;
  if (prefix < cutoff) return (prefix << goRicePar) + decodeBinsEP(goRicePar);
  const unsigned offset = ((1u << (prefix - cutoff)) + cutoff - 1u) << goRicePar;
  const unsigned length = prefix == maxPrefix ? unsigned(maxLog2TrDynamicRange) : goRicePar + prefix - cutoff;
  return offset + decodeBinsEP(length);
}

void BinDecoderHip::planRemAbsEP(unsigned value, unsigned goRicePar, unsigned cutoff, int maxLog2TrDynamicRange) {
  // a planner that knows the value (a replay of recorded syntax) reserves exactly the bypass bins encodeRemAbsEP codes
  planBinEP(cabac_code::rem_abs_code(value, goRicePar, cutoff, unsigned(maxLog2TrDynamicRange)).length());
}

}  // namespace EntropyCodingAMD
