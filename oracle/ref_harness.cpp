// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// C-callable driver around the *reference's own* classes, compiled together with
// the reference sources where they lie under /root/reference (see oracle/Makefile,
// target _ref/libcabac_ref.so).  Nothing from the reference is copied into this
// repository: this file only #includes its headers at build time.
//
// It is used (a) to validate oracle/cabac_oracle.c, (b) to generate the golden
// vectors under tests/golden/ (oracle/gen_golden.py) and (c) optionally as the
// "reference" CPU baseline of bench.py.  Product code never links or loads it.
//
// Private binarisation helpers of CABACWriter/CABACReader (cabac_writer.hpp:165-174,
// cabac_reader.hpp:125-132) are reached with the usual access-macro trick, in this
// translation unit only; the reference objects are compiled unmodified.
// every standard header the reference pulls in is included first, so that the access
// macros below only ever apply to the reference's own class definitions
#include <algorithm>
#include <array>
#include <bitset>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <fstream>
#include <functional>
#include <initializer_list>
#include <iomanip>
#include <iostream>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>
#include <thread>
#include <numeric>
#include <set>
#include <sstream>
#include <stack>
#include <string>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

#define private public
#define protected public
#include "cabac_reader.hpp"
#include "cabac_writer.hpp"
#undef private
#undef protected
#include "ref_rig.hpp"
#include "arith_codec.hpp"
#include "bit_stream.hpp"
#include "contexts.hpp"

using namespace EntropyCoding;
using namespace Common;

// Operation stream shared by ref_harness.cpp, cabac_oracle.c and the host shim tests.
// One op = 4 x uint32 {code, a, b, c}.
enum {
  OP_ENC_BIN = 0,      // a = bin, b = ctxId
  OP_ENC_EP = 1,       // a = bin
  OP_ENC_BINS_EP = 2,  // a = bins, b = numBins
  OP_ENC_REM_ABS = 3,  // a = value, b = rice, c = cutoff | maxLog2TrDR << 8
  OP_ENC_TRM = 4,      // a = bin
  OP_ALIGN = 5,
  OP_UNARY_MAX = 6,   // a = symbol, b = ctxId0 | ctxIdN << 16, c = maxSymbol
  OP_UNARY_EP = 7,    // a = symbol, b = maxSymbol
  OP_EXP_GOLOMB = 8,  // a = symbol, b = count
  OP_TRUNC_BIN = 9,   // a = symbol, b = maxSymbol
};

static thread_local char g_err[512];

extern "C" {

const char *ref_last_error() { return g_err; }

int ref_num_contexts() { return (int)Ctx::NumberOfContexts; }

int ref_sizeof_prob_model() { return (int)sizeof(BinProbModel_Std); }

// row 0..2 = B,P,I slice init values, row 3 = window sizes (contexts.cpp:73-74)
int ref_init_table(int row, uint8_t *out) {
  try {
    const std::vector<uint8_t> &t = ContextSetCfg::getInitTable((unsigned)row);
    memcpy(out, t.data(), t.size());
    return (int)t.size();
  } catch (std::exception &e) {
    strncpy(g_err, e.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// Ctx::init(qp, initId) -> per-context (state0, state1, rate)
int ref_ctx_init(int qp, int initId, uint16_t *s0, uint16_t *s1, uint8_t *rate) {
  try {
    BinEncoder_Std enc;
    static_cast<BinEncIf &>(enc).getCtx().init(qp, initId);
    CtxStore<BinProbModel_Std> &st =
        static_cast<CtxStore<BinProbModel_Std> &>(static_cast<BinEncIf &>(enc).getCtx());
    for (unsigned k = 0; k < Ctx::NumberOfContexts; k++) {
      s0[k] = st[k].getState0();
      s1[k] = st[k].getState1();
      rate[k] = st[k].getRate();
    }
    return 0;
  } catch (std::exception &e) {
    strncpy(g_err, e.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// Scripted update of one context: init (qp, initId), then update(bin) for each bin;
// records state()/mps()/getLPS(range) *before* each update and the states after.
int ref_ctx_trace(int qp, int initId, int ctxId, const uint8_t *bins, int n, unsigned range,
                  uint8_t *state8, uint8_t *lps, uint16_t *s0_after, uint16_t *s1_after) {
  try {
    BinEncoder_Std enc;
    static_cast<BinEncIf &>(enc).getCtx().init(qp, initId);
    CtxStore<BinProbModel_Std> &st =
        static_cast<CtxStore<BinProbModel_Std> &>(static_cast<BinEncIf &>(enc).getCtx());
    BinProbModel_Std &m = st[ctxId];
    for (int i = 0; i < n; i++) {
      state8[i] = m.state();
      lps[i] = m.getLPS(range);
      m.update(bins[i]);
      s0_after[i] = m.getState0();
      s1_after[i] = m.getState1();
    }
    return 0;
  } catch (std::exception &e) {
    strncpy(g_err, e.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// Run an op stream through BinEncoder_Std (+ CABACWriter helpers).
//   flags bit0: call finish() at the end; bit1: then writeByteAlignment()
// out receives m_fifo followed (if any held bits) by the held byte; *n_bits =
// getNumberOfWrittenBits(); n_bins_out[0..2] = ctx/EP/TRM BinCounter totals.
long ref_encode_ops(const uint32_t *ops, long n_ops, int qp, int initId, int flags, uint8_t *out,
                    long cap, uint32_t *n_bits, uint32_t *n_bins_out) {
  try {
    BinEncoder_Std enc;
    OutputBitstream bs;
    CABACWriter w(enc);
    w.initBitstream(&bs);
    enc.reset(qp, initId);
    BinEncIf &e = enc;
    for (long i = 0; i < n_ops; i++) {
      const uint32_t *o = ops + 4 * i;
      switch (o[0]) {
      case OP_ENC_BIN: e.encodeBin(o[1], o[2]); break;
      case OP_ENC_EP: e.encodeBinEP(o[1]); break;
      case OP_ENC_BINS_EP: e.encodeBinsEP(o[1], o[2]); break;
      case OP_ENC_REM_ABS: e.encodeRemAbsEP(o[1], o[2], o[3] & 0xff, (int)(o[3] >> 8)); break;
      case OP_ENC_TRM: e.encodeBinTrm(o[1]); break;
      case OP_ALIGN: e.align(); break;
      case OP_UNARY_MAX: w.unary_max_symbol(o[1], o[2] & 0xffff, o[2] >> 16, o[3]); break;
      case OP_UNARY_EP: w.unary_max_eqprob(o[1], o[2]); break;
      case OP_EXP_GOLOMB: w.exp_golomb_eqprob(o[1], o[2]); break;
      case OP_TRUNC_BIN: w.xWriteTruncBinCode(o[1], o[2]); break;
      default: strcpy(g_err, "bad op"); return -2;
      }
    }
    if (n_bins_out) {
      uint32_t nctx = 0;
      for (unsigned k = 0; k < Ctx::NumberOfContexts; k++) nctx += e.getNumBins(k);
      n_bins_out[0] = nctx;
      n_bins_out[1] = enc.getEP();
      n_bins_out[2] = enc.getTrm();
    }
    if (flags & 1) e.finish();
    if (flags & 2) bs.writeByteAlignment();
    *n_bits = bs.getNumberOfWrittenBits();
    const std::vector<uint8_t> &f = bs.getFIFO();
    long n = (long)f.size();
    long total = n + ((*n_bits & 7) ? 1 : 0);
    if (total > cap) { strcpy(g_err, "capacity"); return -3; }
    if (n) memcpy(out, f.data(), n);
    if (*n_bits & 7) out[n] = bs.getHeldBits();
    return total;
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// Bin-record stream (include/cabac_hip.h) through BinEncoder_Std.
long ref_encode_records(const uint16_t *rec, long n, int qp, int initId, int flags, uint8_t *out,
                        long cap, uint32_t *n_bits) {
  try {
    BinEncoder_Std enc;
    OutputBitstream bs;
    enc.init(&bs);
    enc.reset(qp, initId);
    BinEncIf &e = enc;
    for (long i = 0; i < n; i++) {
      unsigned id = rec[i] & 0x1ff, bin = rec[i] >> 15;
      if (id < Ctx::NumberOfContexts) e.encodeBin(bin, id);
      else if (id == 0x1fe) e.encodeBinEP(bin);
      else if (id == 0x1ff) e.encodeBinTrm(bin);
      else if (id == 0x1fd) e.align();
      else { strcpy(g_err, "bad record"); return -2; }
    }
    if (flags & 4) {  // probe: the reference's own getNumWrittenBits(), nothing flushed
      *n_bits = e.getNumWrittenBits();
      return 0;
    }
    if (flags & 1) e.finish();
    if (flags & 2) bs.writeByteAlignment();
    *n_bits = bs.getNumberOfWrittenBits();
    const std::vector<uint8_t> &f = bs.getFIFO();
    long nb = (long)f.size();
    long total = nb + ((*n_bits & 7) ? 1 : 0);
    if (total > cap) { strcpy(g_err, "capacity"); return -3; }
    if (nb) memcpy(out, f.data(), nb);
    if (*n_bits & 7) out[nb] = bs.getHeldBits();
    return total;
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// Bin-record stream through BinDecoder_Std.  flags bit0: call finish() at the end.
// Returns 0, or -1 with ref_last_error() on a reference exception (e.g. FIFO exceeded,
// missing stop pattern).  *n_bits_read = 8*m_fifo_idx + bitsNeeded.
int ref_decode_records(const uint16_t *rec, long n, int qp, int initId, int flags,
                       const uint8_t *in, long n_in, uint8_t *bins, uint32_t *n_bits_read) {
  try {
    BinDecoder_Std dec;
    InputBitstream ib;
    ib.getFifo().assign(in, in + n_in);
    dec.init(&ib);
    dec.reset(qp, initId);
    for (long i = 0; i < n; i++) {
      unsigned id = rec[i] & 0x1ff;
      if (id < Ctx::NumberOfContexts) bins[i] = (uint8_t)dec.decodeBin(id);
      else if (id == 0x1fe) bins[i] = (uint8_t)dec.decodeBinEP();
      else if (id == 0x1ff) bins[i] = (uint8_t)dec.decodeBinTrm();
      else if (id == 0x1fd) { dec.align(); bins[i] = 0; }
      else { strcpy(g_err, "bad record"); return -2; }
    }
    if (n_bits_read) *n_bits_read = 8u * ib.getByteLocation() + (uint32_t)dec.m_bitsNeeded;
    if (flags & 1) dec.finish();
    return 0;
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// Decode-side op stream: values[i] receives the decoded symbol of op i.
//   OP_ENC_BIN -> decodeBin(b)         OP_ENC_EP -> decodeBinEP()
//   OP_ENC_BINS_EP -> decodeBinsEP(b)  OP_ENC_REM_ABS -> decodeRemAbsEP(b, c&0xff, c>>8)
//   OP_ENC_TRM -> decodeBinTrm()       OP_UNARY_MAX/UNARY_EP/EXP_GOLOMB/TRUNC_BIN -> reader twins
int ref_decode_ops(const uint32_t *ops, long n_ops, int qp, int initId, int flags,
                   const uint8_t *in, long n_in, uint32_t *values) {
  try {
    BinDecoder_Std dec;
    InputBitstream ib;
    ib.getFifo().assign(in, in + n_in);
    CABACReader r(dec);
    r.initBitstream(&ib);
    dec.reset(qp, initId);
    for (long i = 0; i < n_ops; i++) {
      const uint32_t *o = ops + 4 * i;
      uint32_t v = 0;
      switch (o[0]) {
      case OP_ENC_BIN: v = dec.decodeBin(o[2]); break;
      case OP_ENC_EP: v = dec.decodeBinEP(); break;
      case OP_ENC_BINS_EP: v = dec.decodeBinsEP(o[2]); break;
      case OP_ENC_REM_ABS: v = dec.decodeRemAbsEP(o[2], o[3] & 0xff, (int)(o[3] >> 8)); break;
      case OP_ENC_TRM: v = dec.decodeBinTrm(); break;
      case OP_ALIGN: dec.align(); break;
      case OP_UNARY_MAX: v = r.unary_max_symbol(o[2] & 0xffff, o[2] >> 16, o[3]); break;
      case OP_UNARY_EP: v = r.unary_max_eqprob(o[2]); break;
      case OP_EXP_GOLOMB: v = r.exp_golomb_eqprob(o[2]); break;
      case OP_TRUNC_BIN: r.xReadTruncBinCode(v, o[2]); break;
      default: strcpy(g_err, "bad op"); return -2;
      }
      values[i] = v;
    }
    if (flags & 1) dec.finish();
    return 0;
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// OutputBitstream::countStartCodeEmulations on a byte string
int ref_count_emulations(const uint8_t *bytes, long n) {
  try {
    OutputBitstream bs;
    bs.getFIFO().assign(bytes, bytes + n);
    return bs.countStartCodeEmulations();
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// CPU baseline of bench.py on all host cores: substreams [first, first + count) of a batch (descriptor layout of
// include/cabac_hip.h: rec_offset u64, byte_offset u64, n_records u32, byte_capacity u32, qp i32, init_id u32) through
// BinEncoder_Std and back through BinDecoder_Std, one substream per task on a std::thread pool.  out = {bins coded,
// failed substreams, sum over threads of encode ns, of decode ns}; returns the wall time in ns.
struct RefDesc { uint64_t rec_offset, byte_offset; uint32_t n_records, byte_capacity; int32_t qp; uint32_t init_id; };
uint64_t ref_roundtrip_mt(const void *desc_, uint32_t first, uint32_t count, const uint16_t *records, int n_threads,
                          uint64_t *out) {
  const RefDesc *desc = static_cast<const RefDesc *>(desc_);
  std::atomic<uint32_t> next{0};
  std::atomic<uint64_t> bins{0}, bad{0}, enc_ns{0}, dec_ns{0};
  auto worker = [&]() {
    std::vector<uint8_t> buf, got;
    for (;;) {
      const uint32_t k = next.fetch_add(1);
      if (k >= count) break;
      const RefDesc &d = desc[first + k];
      const uint16_t *rec = records + d.rec_offset;
      buf.resize(size_t(d.n_records) + 64);
      got.resize(size_t(d.n_records) + 1);
      uint32_t nbits = 0, nread = 0;
      const auto t0 = std::chrono::steady_clock::now();
      const long nb = ref_encode_records(rec, d.n_records, d.qp, int(d.init_id & 3), 3, buf.data(), long(buf.size()), &nbits);
      const auto t1 = std::chrono::steady_clock::now();
      int rc = nb < 0 ? -1 : ref_decode_records(rec, d.n_records, d.qp, int(d.init_id & 3), 1, buf.data(), nb, got.data(), &nread);
      const auto t2 = std::chrono::steady_clock::now();
      for (uint32_t i = 0; rc == 0 && i < d.n_records; i++)
        if (got[i] != (rec[i] >> 15)) rc = -1;
      bins += d.n_records;
      enc_ns += uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count());
      dec_ns += uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count());
      if (rc) bad += 1;
    }
  };
  if (n_threads < 1) n_threads = 1;
  const auto w0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; t++) pool.emplace_back(worker);
  worker();
  for (auto &t : pool) t.join();
  const auto w1 = std::chrono::steady_clock::now();
  out[0] = bins;
  out[1] = bad;
  out[2] = enc_ns;
  out[3] = dec_ns;
  return uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(w1 - w0).count());
}

// A digest per substream of the bytes the reference's BinEncoder_Std produces for it (finish() + writeByteAlignment()):
// FNV-1a 64 over the bytes, then over the four bytes of the bit count.  bench.py compares ALL substreams of its batch with
// the device's bytes through this (orc_digest_slots computes the same over the device's output).
uint64_t ref_digest_mt(const void *desc_, uint32_t first, uint32_t count, const uint16_t *records, int n_threads, uint64_t *digests) {
  const RefDesc *desc = static_cast<const RefDesc *>(desc_);
  std::atomic<uint32_t> next{0};
  std::atomic<uint64_t> bad{0};
  auto worker = [&]() {
    std::vector<uint8_t> buf;
    for (;;) {
      const uint32_t k = next.fetch_add(1);
      if (k >= count) break;
      const RefDesc &d = desc[first + k];
      buf.resize(size_t(d.n_records) + 64);
      uint32_t nbits = 0;
      const long nb = ref_encode_records(records + d.rec_offset, d.n_records, d.qp, int(d.init_id & 3), 3, buf.data(), long(buf.size()), &nbits);
      uint64_t h = 0xcbf29ce484222325ull;
      for (long i = 0; i < nb; i++) h = (h ^ buf[size_t(i)]) * 0x100000001b3ull;
      for (int i = 0; i < 4; i++) h = (h ^ ((nbits >> (8 * i)) & 0xffu)) * 0x100000001b3ull;
      digests[k] = h;
      if (nb < 0) bad += 1;
    }
  };
  if (n_threads < 1) n_threads = 1;
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; t++) pool.emplace_back(worker);
  worker();
  for (auto &t : pool) t.join();
  return bad.load();
}

// InputBitstream (bit_stream.cpp:183-430) driven by the script format of tests/csrc/host_shim_driver.cpp's
// shim_input_bitstream_script, so that the host mirror can be compared with it step by step
long ref_input_bitstream_script(const uint8_t *bytes, long n_bytes, const uint32_t *script, long n_steps, uint32_t *out,
                                uint8_t *sub, long sub_cap) {
  InputBitstream bs;
  bs.getFifo().assign(bytes, bytes + n_bytes);
  long n_sub = 0;
  bool go_on = false;  // op 10: from here on a throwing step reports 0xFFFFFFFF and the script continues
  for (long i = 0; i < n_steps; i++) {
    const uint32_t op = script[2 * i], arg = script[2 * i + 1];
    try {
      switch (op) {
      case 7: out[i] = bs.getByteLocation(); break;
      case 8: out[i] = bs.getNumBitsRead(); break;
      case 9: out[i] = bs.getHeldBits(); break;
      case 10: go_on = true; out[i] = 0; break;
      case 0: out[i] = bs.read(arg); break;
      case 1: out[i] = bs.readByte(); break;
      case 2: {
        std::unique_ptr<InputBitstream> r(bs.extractSubstream(arg));
        out[i] = uint32_t(r->getFifo().size());
        if (n_sub + (long)r->getFifo().size() > sub_cap) return -3;
        if (!r->getFifo().empty()) memcpy(sub + n_sub, r->getFifo().data(), r->getFifo().size());
        n_sub += (long)r->getFifo().size();
        break;
      }
      case 3: out[i] = bs.readOutTrailingBits(); break;
      case 4: out[i] = bs.getNumBitsLeft(); break;
      case 5: out[i] = bs.getNumBitsUntilByteAligned(); break;
      case 6: out[i] = bs.readByteAlignment(); break;
      default: return -2;
      }
    } catch (std::exception &ex) {
      strncpy(g_err, ex.what(), sizeof(g_err) - 1);
      out[i] = 0xFFFFFFFFu;
      if (!go_on) return n_sub;
    }
  }
  return n_sub;
}

// ---- BitEstimator_Std (arith_codec.cpp:603-711): fractional-bit cost of a bin string -----------------
// contexts.hpp:143 keeps the table protected; "contexts.hpp" is first included through cabac_writer.hpp
// above, inside the access macros, so it is reachable here.
int ref_frac_bits_table(uint32_t *out) {  // 256 x {bits of bin 0, bits of bin 1}, SCALE_BITS = 15
  for (int q = 0; q < 256; q++) {
    out[2 * q] = ProbModelTables::m_binFracBits[q].intBits[0];
    out[2 * q + 1] = ProbModelTables::m_binFracBits[q].intBits[1];
  }
  return 256;
}

int ref_estimate_ops(const uint32_t *ops, long n_ops, int qp, int initId, uint64_t *frac_bits) {
  try {
    BitEstimator_Std est;
    CABACWriter w(est);
    est.reset(qp, initId);
    BinEncIf &e = est;
    for (long i = 0; i < n_ops; i++) {
      const uint32_t *o = ops + 4 * i;
      switch (o[0]) {
      case OP_ENC_BIN: e.encodeBin(o[1], o[2]); break;
      case OP_ENC_EP: e.encodeBinEP(o[1]); break;
      case OP_ENC_BINS_EP: e.encodeBinsEP(o[1], o[2]); break;
      case OP_ENC_REM_ABS: e.encodeRemAbsEP(o[1], o[2], o[3] & 0xff, (int)(o[3] >> 8)); break;
      case OP_ENC_TRM: e.encodeBinTrm(o[1]); break;
      case OP_ALIGN: e.align(); break;
      case OP_UNARY_MAX: w.unary_max_symbol(o[1], o[2] & 0xffff, o[2] >> 16, o[3]); break;
      case OP_UNARY_EP: w.unary_max_eqprob(o[1], o[2]); break;
      case OP_EXP_GOLOMB: w.exp_golomb_eqprob(o[1], o[2]); break;
      case OP_TRUNC_BIN: w.xWriteTruncBinCode(o[1], o[2]); break;
      default: strcpy(g_err, "bad op"); return -2;
      }
    }
    *frac_bits = est.getEstFracBits();
    return 0;
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

int ref_estimate_records(const uint16_t *rec, long n, int qp, int initId, uint64_t *frac_bits) {
  try {
    BitEstimator_Std est;
    est.reset(qp, initId);
    BinEncIf &e = est;
    for (long i = 0; i < n; i++) {
      unsigned id = rec[i] & 0x1ff, bin = rec[i] >> 15;
      if (id < Ctx::NumberOfContexts) e.encodeBin(bin, id);
      else if (id == 0x1fe) e.encodeBinEP(bin);
      else if (id == 0x1ff) e.encodeBinTrm(bin);
      else if (id == 0x1fd) e.align();
      else if (id == 0x1fc) { if (i & 1) e.resetBits(); else e.start(); }  // both zero the cost, nothing else
      else if (id == 0x1fb) e.restart();
      else { strcpy(g_err, "bad record"); return -2; }
    }
    *frac_bits = est.getEstFracBits();
    return 0;
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// Bit estimator started from the contexts another coder has reached (RDO: estimator.getCtx() = encoder.getCtx(),
// contexts.hpp:254, then resetBits()).  `hist` is coded first (after reset(qp, initId)) to produce that state,
// which is also dumped (s0, s1, rate) so that the oracle / the GPU can start from the same arrays.
int ref_estimate_from_history(const uint16_t *hist, long n_hist, const uint16_t *rec, long n, int qp, int initId,
                              uint64_t *frac_bits, uint16_t *s0, uint16_t *s1, uint8_t *rate) {
  try {
    auto run = [](BinEncIf &e, const uint16_t *r, long cnt) {
      for (long i = 0; i < cnt; i++) {
        unsigned id = r[i] & 0x1ff, bin = r[i] >> 15;
        if (id < Ctx::NumberOfContexts) e.encodeBin(bin, id);
        else if (id == 0x1fe) e.encodeBinEP(bin);
        else if (id == 0x1ff) e.encodeBinTrm(bin);
        else if (id == 0x1fd) e.align();
        else if (id == 0x1fc) e.resetBits();
        else if (id == 0x1fb) e.restart();
      }
    };
    BitEstimator_Std a, b;
    a.reset(qp, initId);
    run(a, hist, n_hist);
    b.reset(0, 0);
    static_cast<BinEncIf &>(b).getCtx() = static_cast<BinEncIf &>(a).getCtx();
    b.resetBits();
    CtxStore<BinProbModel_Std> &st = static_cast<CtxStore<BinProbModel_Std> &>(static_cast<BinEncIf &>(b).getCtx());
    for (unsigned k = 0; k < Ctx::NumberOfContexts; k++) {
      s0[k] = st[k].getState0();
      s1[k] = st[k].getState1();
      rate[k] = st[k].getRate();
    }
    run(b, rec, n);
    *frac_bits = b.getEstFracBits();
    return 0;
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}


// ---- residual coding ---------------------------------------------------
// The reference's own CABACWriter::residual_coding (cabac_writer.cpp:2424-2525, with last_sig_coeff
// :2639-2720 and residual_coding_subblock :2722-2872) run on a TransformUnit built here around a
// caller-supplied coefficient block; the bins it asks the encoder for are captured as bin records.
} // extern "C"
namespace {
struct RecordingEncoder : public BinEncoder_Std {  // encodeRemAbsEP stays the reference's (it calls encodeBinsEP)
  std::vector<uint16_t> rec;
  void encodeBin(unsigned bin, unsigned ctxId) override { rec.push_back(uint16_t((bin & 1u) << 15 | ctxId)); }
  void encodeBinEP(unsigned bin) override { rec.push_back(uint16_t((bin & 1u) << 15 | 0x1fe)); }
  void encodeBinsEP(unsigned bins, unsigned numBins) override {
    for (int i = int(numBins) - 1; i >= 0; i--) rec.push_back(uint16_t(((bins >> i) & 1u) << 15 | 0x1fe));
  }
  void encodeBinTrm(unsigned bin) override { rec.push_back(uint16_t((bin & 1u) << 15 | 0x1ff)); }
};

}  // namespace
extern "C" {

// flags: bit0 dep_quant, bit1 sign_data_hiding, bit2 transform-skip enabled in the SPS (max TS size 32),
//        bit3 the cuCtx pointer is passed (info[1..4] = violatesLfnstConstrained[luma|chroma<<1], lfnstLastScanPos,
//        violatesMtsCoeffConstraint, mtsLastScanPos); bits 8..15: if non-zero, extended_precision_processing with
//        this bit depth (SPS::getMaxLog2TrDynamicRange = min(20, depth + 6), slice.hpp:180-192).  comp: 0 Y, 1 Cb, 2 Cr.
long ref_residual_records(int width, int height, int comp, int flags, const int32_t *coeff, uint16_t *out, long cap,
                          int32_t *info) {
  try {
    static ResidualRig rig;
    std::vector<TCoeff> buf;
    TransformUnit tu;
    rig.make_tu(tu, buf, width, height, comp, flags, coeff);
    const ComponentID cid = ComponentID(comp);
    RecordingEncoder enc;
    CABACWriter w(enc);
    CUCtx cuCtx(0);
    w.residual_coding(tu, cid, (flags & 8) ? &cuCtx : nullptr);
    if (info) {
      info[0] = (int32_t)enc.rec.size();
      info[1] = int(cuCtx.violatesLfnstConstrained[CHANNEL_TYPE_LUMA]) | int(cuCtx.violatesLfnstConstrained[CHANNEL_TYPE_CHROMA]) << 1;
      info[2] = cuCtx.lfnstLastScanPos;
      info[3] = cuCtx.violatesMtsCoeffConstraint;
      info[4] = cuCtx.mtsLastScanPos;
    }
    if ((long)enc.rec.size() > cap) { strcpy(g_err, "capacity"); return -3; }
    if (!enc.rec.empty()) memcpy(out, enc.rec.data(), enc.rec.size() * sizeof(uint16_t));
    return (long)enc.rec.size();
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

// The scan the reference's ROM holds for a block (rom.cpp:148-260): out[scanPos] = x | y << 16.
long ref_scan_order(int width, int height, uint32_t *out) {
  static ResidualRig rig;
  const ScanElement *scan = g_scanOrder[SCAN_GROUPED_4x4][SCAN_DIAG][gp_sizeIdxInfo->idxFrom(width)][gp_sizeIdxInfo->idxFrom(height)];
  if (!scan) return -1;
  for (long i = 0; i < (long)width * height; i++) out[i] = scan[i].x | uint32_t(scan[i].y) << 16;
  return (long)width * height;
}

// The reference's own CABACReader::residual_coding (cabac_reader.cpp:2647-2735) on n blocks in one substream: sizes
// wh[2i], wh[2i+1], component comp[i], rig flags (as ref_residual_records) rig_flags for every block or, if block_flags
// is given, block_flags[i]; coefficients written back to back into coeff_out.  With transform skip enabled in the rig
// (bit2) the reader takes transform_skip_flag from the stream (ts_flag, :2737-2752) unless BDPCM (bit5) infers it.
// info (may be NULL), 5 ints per block from a fresh CUCtx: {mtsIdx == MTS_SKIP after the call, violatesLfnstConstrained
// luma | chroma << 1, lfnstLastScanPos, violatesMtsCoeffConstraint, mtsLastScanPos}.
// finish: then decodeBinTrm() must give 1 and finish() is called.
long ref_residual_decode(int n, const int *wh, const int *comp, int rig_flags, const uint8_t *in, long n_in, int qp,
                         int finish, int32_t *coeff_out, uint32_t *n_bits_read, const int *block_flags, int32_t *info) {
  try {
    static ResidualRig rig;
    BinDecoder_Std dec;
    InputBitstream ib;
    ib.getFifo().assign(in, in + n_in);
    CABACReader r(dec);
    r.initBitstream(&ib);
    dec.reset(qp, 2);
    int32_t *out = coeff_out;
    for (int i = 0; i < n; i++) {
      const int w = wh[2 * i], h = wh[2 * i + 1];
      std::vector<int32_t> zeros((size_t)w * h, 0);
      std::vector<TCoeff> buf;
      TransformUnit tu;
      rig.make_tu(tu, buf, w, h, comp[i], block_flags ? block_flags[i] : rig_flags, zeros.data());
      CUCtx cuCtx(0);
      r.residual_coding(tu, ComponentID(comp[i]), cuCtx);
      for (size_t k = 0; k < buf.size(); k++) out[k] = (int32_t)buf[k];
      out += (size_t)w * h;
      if (info) {
        info[5 * i + 0] = tu.mtsIdx[comp[i]] == MTS_SKIP;
        info[5 * i + 1] = int(cuCtx.violatesLfnstConstrained[CHANNEL_TYPE_LUMA]) | int(cuCtx.violatesLfnstConstrained[CHANNEL_TYPE_CHROMA]) << 1;
        info[5 * i + 2] = cuCtx.lfnstLastScanPos;
        info[5 * i + 3] = cuCtx.violatesMtsCoeffConstraint;
        info[5 * i + 4] = cuCtx.mtsLastScanPos;
      }
    }
    if (finish) {
      if (dec.decodeBinTrm() != 1) { strcpy(g_err, "terminate bin is not 1"); return -5; }
      dec.finish();
    }
    if (n_bits_read) *n_bits_read = 8u * ib.getByteLocation() + (uint32_t)dec.m_bitsNeeded;
    return 0;
  } catch (std::exception &ex) {
    strncpy(g_err, ex.what(), sizeof(g_err) - 1);
    return -1;
  }
}

} // extern "C"
