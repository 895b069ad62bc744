// TEST INFRASTRUCTURE — NOT PRODUCT CODE, never shipped, never loaded by the package.
//
// A CPU stand-in for the part of the C ABI (include/cabac_hip.h) that the C++ host shim (entropy_coding_amd/host) and the
// reference adapters (integration/reference_adapter.hpp) call, answered by the oracle (oracle/cabac_oracle.c).  Its only
// purpose is the host sanitizer job (tests/test_sanitizers.py): the shim, the adapters and their test drivers
// are compiled together with this file under -fsanitize=address,undefined and driven by the ordinary tests in the build
// container, which has no GPU (GPU AddressSanitizer is not available on the pool either).  What is checked there is the
// HOST code's memory behaviour, not parity: results come from the oracle on both sides.
// The product library has no CPU path; this file is linked into *_san.so test objects only.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "cabac_hip.h"
#include "cabac_oracle.h"

struct cabac_hip_ctx {
  std::string last_error;
  int device = 0;
};

namespace {
std::mutex g_mu;
std::unordered_map<void *, size_t> g_owned;  // memory handed out by cabac_hip_host_alloc
int g_live_ctx = 0;

int fail(cabac_hip_ctx *c, int rc, const char *what) {
  if (c) c->last_error = what;
  return rc;
}
}  // namespace

extern "C" {

size_t cabac_hip_encode_bound(uint64_t n_ctx_bins, uint64_t n_ep_bins, uint64_t n_trm_bins) {
  const uint64_t bits = 6 * n_ctx_bins + n_ep_bins + 7 * n_trm_bins;
  return (size_t)(((bits + 7) / 8 + 8 + 15) / 16 * 16);
}

int cabac_hip_init(int device, cabac_hip_ctx **out) {
  if (!out) return CABAC_HIP_ERR_INVALID;
  *out = nullptr;
  if (device != 0) return CABAC_HIP_ERR_INVALID;
  cabac_hip_ctx *c = new (std::nothrow) cabac_hip_ctx;
  if (!c) return CABAC_HIP_ERR_NOMEM;
  c->device = device;
  std::lock_guard<std::mutex> lk(g_mu);
  g_live_ctx++;
  *out = c;
  return CABAC_HIP_OK;
}

void cabac_hip_destroy(cabac_hip_ctx *c) {
  if (!c) return;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_live_ctx--;
  }
  delete c;
}

int cabac_hip_stub_live_contexts() {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_live_ctx;
}

const char *cabac_hip_strerror(int status) {
  switch (status) {
  case CABAC_HIP_OK: return "ok";
  case CABAC_HIP_ERR_NO_DEVICE: return "no HIP device (this library has no CPU path)";
  case CABAC_HIP_ERR_INVALID: return "invalid argument";
  case CABAC_HIP_ERR_HIP: return "HIP runtime error";
  case CABAC_HIP_ERR_NOMEM: return "out of memory";
  case CABAC_HIP_ERR_SUBSTREAM: return "a substream reported an error flag";
  default: return "unknown status";
  }
}

const char *cabac_hip_last_error(const cabac_hip_ctx *c) { return c ? c->last_error.c_str() : ""; }

int cabac_hip_host_alloc(size_t bytes, void **out) {
  if (!out) return CABAC_HIP_ERR_INVALID;
  void *p = std::malloc(bytes ? bytes : 16);
  if (!p) return CABAC_HIP_ERR_NOMEM;
  std::lock_guard<std::mutex> lk(g_mu);
  g_owned[p] = bytes ? bytes : 16;
  *out = p;
  return CABAC_HIP_OK;
}

int cabac_hip_host_free(void *p) {
  if (!p) return CABAC_HIP_OK;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_owned.erase(p)) return CABAC_HIP_ERR_INVALID;
  }
  std::free(p);
  return CABAC_HIP_OK;
}

int cabac_hip_host_is_pinned(const void *p, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (const auto &kv : g_owned) {
    const uint8_t *b = static_cast<const uint8_t *>(kv.first), *q = static_cast<const uint8_t *>(p);
    if (q >= b && q + bytes <= b + kv.second) return 1;
  }
  return 0;
}

static int check_desc(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, uint64_t n_records_total,
                      uint64_t bytes_total) {
  for (uint32_t s = 0; s < n_sub; s++) {
    const cabac_substream_desc &d = desc[s];
    if (d.rec_offset > n_records_total || d.n_records > n_records_total - d.rec_offset)
      return fail(c, CABAC_HIP_ERR_INVALID, "records out of range");
    if (d.byte_offset > bytes_total || d.byte_capacity > bytes_total - d.byte_offset)
      return fail(c, CABAC_HIP_ERR_INVALID, "bytes out of range");
    if (d.byte_offset & 15u) return fail(c, CABAC_HIP_ERR_INVALID, "byte_offset must be 16-byte aligned");
    if ((d.init_id & 3u) > 2u) return fail(c, CABAC_HIP_ERR_INVALID, "init_id must be 0..2");
  }
  return CABAC_HIP_OK;
}

// The device library reads the caller's arrays only inside [0, n_records_total) / [0, bytes_total): copies of exactly that
// size make any shim-side overrun of its own staging buffers visible to the sanitizer.
int cabac_hip_encode_batch(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                           uint64_t n_records_total, uint8_t *bytes, uint64_t bytes_total, cabac_substream_result *results) {
  if (!c || (n_sub && (!desc || !results))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  if (n_sub == 0) return CABAC_HIP_OK;
  if (int rc = check_desc(c, n_sub, desc, n_records_total, bytes_total)) return rc;
  std::vector<uint16_t> rec(records, records + n_records_total);
  std::vector<uint8_t> out(bytes_total, 0);
  rec.push_back(0);
  out.push_back(0);
  orc_encode_batch(desc, 0, n_sub, rec.data(), out.data(), reinterpret_cast<uint32_t *>(results));
  int status = CABAC_HIP_OK;
  for (uint32_t s = 0; s < n_sub; s++) {
    const uint64_t n = (uint64_t(results[s].n_bits) + 7) / 8;
    if (n > desc[s].byte_capacity) return fail(c, CABAC_HIP_ERR_INVALID, "stub: oracle wrote past the slot");
    std::memcpy(bytes + desc[s].byte_offset, out.data() + desc[s].byte_offset, n);
    if (results[s].flags) status = CABAC_HIP_ERR_SUBSTREAM;
  }
  if (status) c->last_error = "substream flag set (see results[].flags)";
  return status;
}

int cabac_hip_decode_batch(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                           uint64_t n_records_total, const uint8_t *bytes, uint64_t bytes_total, uint8_t *bins,
                           cabac_substream_result *results) {
  if (!c || (n_sub && (!desc || !results))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  if (n_sub == 0) return CABAC_HIP_OK;
  if (int rc = check_desc(c, n_sub, desc, n_records_total, bytes_total)) return rc;
  std::vector<uint16_t> rec(records, records + n_records_total);
  std::vector<uint8_t> in(bytes, bytes + bytes_total), out(n_records_total + 1, 0);
  rec.push_back(0);
  in.push_back(0);
  orc_decode_batch(desc, 0, n_sub, rec.data(), in.data(), out.data(), reinterpret_cast<uint32_t *>(results));
  if (bins && n_records_total) std::memcpy(bins, out.data(), n_records_total);
  int status = CABAC_HIP_OK;
  for (uint32_t s = 0; s < n_sub; s++)
    if (results[s].flags) status = CABAC_HIP_ERR_SUBSTREAM;
  if (status) c->last_error = "substream flag set (see results[].flags)";
  return status;
}

int cabac_hip_decode_batch_packed(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                                  uint64_t n_records_total, const uint8_t *bytes, uint64_t bytes_total, uint8_t *packed_bins,
                                  cabac_substream_result *results) {
  std::vector<uint8_t> bins(n_records_total ? n_records_total : 1, 0);
  const int rc = cabac_hip_decode_batch(c, n_sub, desc, records, n_records_total, bytes, bytes_total, packed_bins ? bins.data() : nullptr, results);
  if (packed_bins) {
    std::fill(packed_bins, packed_bins + (n_records_total + 7) / 8, uint8_t(0));
    for (uint64_t r = 0; r < n_records_total; r++) packed_bins[r >> 3] |= uint8_t((bins[r] & 1u) << (r & 7));
  }
  return rc;
}

int cabac_hip_estimate_batch(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                             uint64_t n_records_total, uint64_t *frac_bits, uint32_t *flags) {
  if (!c || (n_sub && (!desc || !frac_bits))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  if (n_sub == 0) return CABAC_HIP_OK;
  for (uint32_t s = 0; s < n_sub; s++)
    if (desc[s].rec_offset > n_records_total || desc[s].n_records > n_records_total - desc[s].rec_offset)
      return fail(c, CABAC_HIP_ERR_INVALID, "records out of range");
  std::vector<uint16_t> rec(records, records + n_records_total);
  rec.push_back(0);
  std::vector<uint32_t> fl(n_sub, 0);
  orc_estimate_batch(desc, 0, n_sub, rec.data(), frac_bits, fl.data());
  int status = CABAC_HIP_OK;
  for (uint32_t s = 0; s < n_sub; s++) {
    if (flags) flags[s] = fl[s];
    if (fl[s]) status = CABAC_HIP_ERR_SUBSTREAM;
  }
  return status;
}

int cabac_hip_residual_batch(cabac_hip_ctx *c, uint32_t n_tu, const cabac_tu_desc *tus, const int32_t *coeff,
                             uint64_t n_coeff_total, uint64_t *offsets, uint32_t *info, uint16_t *records,
                             uint64_t records_capacity) {
  if (!c || !offsets || (n_tu && (!tus || !coeff))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  offsets[0] = 0;
  std::vector<int32_t> co(coeff, coeff + n_coeff_total);  // exactly the declared size
  std::vector<std::vector<uint16_t>> rec(n_tu);
  int status = CABAC_HIP_OK;
  for (uint32_t t = 0; t < n_tu; t++) {
    uint32_t inf = 0;
    if (tus[t].log2_width > 6 || tus[t].log2_height > 6 || tus[t].channel > 1) {
      inf = CABAC_TU_INFO_BAD_DESC;
    } else {
      const uint64_t n = uint64_t(1) << (tus[t].log2_width + tus[t].log2_height);
      if (tus[t].coeff_offset > n_coeff_total || n > n_coeff_total - tus[t].coeff_offset)
        return fail(c, CABAC_HIP_ERR_INVALID, "coefficients out of range");
      const unsigned lw = tus[t].log2_width < 5 ? tus[t].log2_width : 5, lh = tus[t].log2_height < 5 ? tus[t].log2_height : 5;
      rec[t].resize(CABAC_TU_MAX_RECORDS(1u << (lw + lh)));
      const long k = orc_residual_records(tus[t].log2_width, tus[t].log2_height, tus[t].channel, tus[t].flags,
                                          tus[t].max_log2_tr_range, co.data() + tus[t].coeff_offset, rec[t].data(),
                                          (long)rec[t].size(), &inf);
      if (k == -1) inf = CABAC_TU_INFO_EMPTY;
      else if (k < 0) inf = CABAC_TU_INFO_BAD_DESC;
      rec[t].resize(k > 0 ? size_t(k) : 0);
    }
    if (inf & (CABAC_TU_INFO_EMPTY | CABAC_TU_INFO_BAD_DESC)) status = CABAC_HIP_ERR_SUBSTREAM;
    if (info) info[t] = inf;
    offsets[t + 1] = offsets[t] + rec[t].size();
  }
  if (status) c->last_error = "empty block or bad descriptor (see info[])";
  if (!records) return status;
  if (records_capacity < offsets[n_tu]) return fail(c, CABAC_HIP_ERR_INVALID, "records_capacity too small");
  for (uint32_t t = 0; t < n_tu; t++)
    if (!rec[t].empty()) std::memcpy(records + offsets[t], rec[t].data(), rec[t].size() * 2);
  return status;
}

// coefficients -> bytes: the splices resolved on the host with the oracle's block records
int cabac_hip_encode_batch_residual(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                                    uint64_t n_records_total, const uint32_t *splice_first, const cabac_splice *splices, uint32_t n_tu,
                                    const cabac_tu_desc *tus, const int32_t *coeff, uint64_t n_coeff_total, uint8_t *payload,
                                    uint64_t payload_capacity, uint64_t *payload_offsets, cabac_substream_result *results,
                                    uint32_t *tu_info, uint32_t *bin_counts) {
  if (!c || !payload_offsets || (n_sub && (!desc || !splice_first || !results || !payload)) || (n_tu && (!tus || !coeff || !splices)))
    return fail(c, CABAC_HIP_ERR_INVALID, "null");
  payload_offsets[0] = 0;
  if (n_sub == 0) return CABAC_HIP_OK;
  if (splice_first[0] != 0 || splice_first[n_sub] != n_tu) return fail(c, CABAC_HIP_ERR_INVALID, "every block must be spliced exactly once");
  std::vector<uint16_t> rec(records, records + n_records_total);  // exactly the declared sizes
  std::vector<int32_t> co(coeff, coeff + n_coeff_total);
  std::vector<cabac_splice> sp(splices, splices + n_tu);
  std::vector<uint8_t> seen(n_tu ? n_tu : 1, 0);
  std::vector<uint32_t> info(n_tu ? n_tu : 1, 0);
  int status = CABAC_HIP_OK;
  uint64_t at_payload = 0;
  for (uint32_t s = 0; s < n_sub; s++) {
    const cabac_substream_desc &d = desc[s];
    if (d.rec_offset > n_records_total || d.n_records > n_records_total - d.rec_offset) return fail(c, CABAC_HIP_ERR_INVALID, "records out of range");
    if (splice_first[s] > splice_first[s + 1]) return fail(c, CABAC_HIP_ERR_INVALID, "splice_first must not decrease");
    std::vector<uint16_t> full;
    uint32_t prev = 0;
    for (uint32_t j = splice_first[s]; j < splice_first[s + 1]; j++) {
      if (sp[j].tu >= n_tu || sp[j].at > d.n_records || sp[j].at < prev || seen[sp[j].tu]++)
        return fail(c, CABAC_HIP_ERR_INVALID, "splice list: not sorted, outside its substream, or a block not spliced exactly once");
      full.insert(full.end(), rec.begin() + d.rec_offset + prev, rec.begin() + d.rec_offset + sp[j].at);
      prev = sp[j].at;
      const cabac_tu_desc &t = tus[sp[j].tu];
      if (t.log2_width > 6 || t.log2_height > 6 || t.channel > 1) {
        info[sp[j].tu] = CABAC_TU_INFO_BAD_DESC;
        status = CABAC_HIP_ERR_SUBSTREAM;
        continue;
      }
      const uint64_t n = uint64_t(1) << (t.log2_width + t.log2_height);
      if (t.coeff_offset > n_coeff_total || n > n_coeff_total - t.coeff_offset) return fail(c, CABAC_HIP_ERR_INVALID, "coefficients out of range");
      const unsigned lw = t.log2_width < 5 ? t.log2_width : 5, lh = t.log2_height < 5 ? t.log2_height : 5;
      std::vector<uint16_t> blk(CABAC_TU_MAX_RECORDS(1u << (lw + lh)));
      const long k = orc_residual_records(t.log2_width, t.log2_height, t.channel, t.flags, t.max_log2_tr_range, co.data() + t.coeff_offset,
                                          blk.data(), (long)blk.size(), &info[sp[j].tu]);
      if (k < 0) {
        info[sp[j].tu] = k == -1 ? CABAC_TU_INFO_EMPTY : CABAC_TU_INFO_BAD_DESC;
        status = CABAC_HIP_ERR_SUBSTREAM;
        continue;
      }
      full.insert(full.end(), blk.begin(), blk.begin() + k);
    }
    full.insert(full.end(), rec.begin() + d.rec_offset + prev, rec.begin() + d.rec_offset + d.n_records);
    std::vector<uint8_t> out(cabac_hip_encode_bound(full.size(), full.size(), full.size()) + 1);
    uint32_t nbits = 0;
    const int fl = ((d.init_id & CABAC_SUB_FINISH) ? 1 : 0) | ((d.init_id & CABAC_SUB_ALIGN_RBSP) ? 2 : 0);
    const long nb = orc_encode_records(full.data(), (long)full.size(), d.qp, int(d.init_id & 3u), fl, out.data(), (long)out.size() - 1, &nbits);
    results[s].n_bits = nbits;
    results[s].flags = nb == -2 ? CABAC_RES_BAD_RECORD : nb == -3 ? CABAC_RES_OVERFLOW : 0;
    if (results[s].flags) status = CABAC_HIP_ERR_SUBSTREAM;
    const uint64_t nbytes = nb > 0 ? uint64_t(nb) : 0;
    if (at_payload + nbytes > payload_capacity) return fail(c, CABAC_HIP_ERR_INVALID, "payload_capacity too small");
    if (nbytes) std::memcpy(payload + at_payload, out.data(), nbytes);
    at_payload += nbytes;
    payload_offsets[s + 1] = at_payload;
    if (bin_counts) {
      uint32_t *cnt = bin_counts + size_t(s) * CABAC_BIN_COUNT_WORDS;
      std::memset(cnt, 0, CABAC_BIN_COUNT_WORDS * sizeof(uint32_t));
      for (uint16_t r : full) {
        const unsigned id = r & CABAC_REC_ID_MASK;
        if (id < CABAC_NUM_CONTEXTS) cnt[id]++;
        else if (id == CABAC_REC_EP) cnt[CABAC_NUM_CONTEXTS]++;
        else if (id == CABAC_REC_TRM) cnt[CABAC_NUM_CONTEXTS + 1]++;
      }
    }
  }
  for (uint32_t t = 0; t < n_tu; t++)
    if (!seen[t]) return fail(c, CABAC_HIP_ERR_INVALID, "splice list: not sorted, outside its substream, or a block not spliced exactly once");
  if (tu_info && n_tu) std::memcpy(tu_info, info.data(), n_tu * sizeof(uint32_t));
  if (status) c->last_error = "substream flag set, or an empty / badly described block (see results[].flags, tu_info[])";
  return status;
}

int cabac_hip_encode_batch_residual16(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                                      uint64_t n_records_total, const uint32_t *splice_first, const cabac_splice *splices, uint32_t n_tu,
                                      const cabac_tu_desc *tus, const int16_t *coeff, uint64_t n_coeff_total, uint8_t *payload,
                                      uint64_t payload_capacity, uint64_t *payload_offsets, cabac_substream_result *results,
                                      uint32_t *tu_info, uint32_t *bin_counts) {
  if (n_tu && !coeff) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  std::vector<int32_t> wide(coeff, coeff + (coeff ? n_coeff_total : 0));
  return cabac_hip_encode_batch_residual(c, n_sub, desc, records, n_records_total, splice_first, splices, n_tu, tus, wide.data(), n_coeff_total,
                                         payload, payload_capacity, payload_offsets, results, tu_info, bin_counts);
}

int cabac_hip_residual_parse_batch(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint8_t *bytes,
                                   uint64_t bytes_total, const uint32_t *tile_first, const cabac_tu_desc *tus, int32_t *coeff,
                                   uint64_t n_coeff_total, uint32_t *tu_info, cabac_substream_result *results) {
  if (!c || (n_sub && (!desc || !bytes || !tile_first || !tus || !coeff || !results))) return fail(c, CABAC_HIP_ERR_INVALID, "null");
  if (n_sub == 0) return CABAC_HIP_OK;
  const uint32_t n_tu = tile_first[n_sub];
  for (uint32_t t = 0; t < n_tu; t++) {
    if (tus[t].log2_width > 6 || tus[t].log2_height > 6) continue;
    const uint64_t n = uint64_t(1) << (tus[t].log2_width + tus[t].log2_height);
    if (tus[t].coeff_offset > n_coeff_total || n > n_coeff_total - tus[t].coeff_offset)
      return fail(c, CABAC_HIP_ERR_INVALID, "coefficients out of range");
  }
  std::vector<int32_t> co(coeff, coeff + n_coeff_total);
  std::vector<uint32_t> inf(n_tu ? n_tu : 1, 0);
  int status = CABAC_HIP_OK;
  for (uint32_t s = 0; s < n_sub; s++) {
    if (tile_first[s] > tile_first[s + 1]) return fail(c, CABAC_HIP_ERR_INVALID, "tile_first must not decrease");
    if (desc[s].byte_offset > bytes_total || desc[s].byte_capacity > bytes_total - desc[s].byte_offset)
      return fail(c, CABAC_HIP_ERR_INVALID, "bytes out of range");
    std::vector<uint8_t> in(bytes + desc[s].byte_offset, bytes + desc[s].byte_offset + desc[s].byte_capacity);
    uint32_t nbits = 0;
    const int rc = orc_residual_decode(in.data(), (long)in.size(), desc[s].qp, int(desc[s].init_id & 3u), tus + tile_first[s],
                                       long(tile_first[s + 1] - tile_first[s]), (desc[s].init_id & CABAC_SUB_FINISH) ? 1 : 0,
                                       co.data(), &nbits, inf.data() + tile_first[s]);
    results[s].n_bits = nbits;
    results[s].flags = rc == -4 ? CABAC_RES_UNDERRUN : rc == -5 ? CABAC_RES_BAD_STOP : rc == -2 ? CABAC_RES_BAD_RECORD : 0;
    if (results[s].flags) status = CABAC_HIP_ERR_SUBSTREAM;
  }
  if (n_coeff_total) std::memcpy(coeff, co.data(), n_coeff_total * sizeof(int32_t));
  if (tu_info && n_tu) std::memcpy(tu_info, inf.data(), n_tu * sizeof(uint32_t));
  if (status) c->last_error = "substream flag set (see results[].flags)";
  return status;
}

int cabac_hip_residual_parse_batch16(cabac_hip_ctx *c, uint32_t n_sub, const cabac_substream_desc *desc, const uint8_t *bytes,
                                     uint64_t bytes_total, const uint32_t *tile_first, const cabac_tu_desc *tus, int16_t *coeff,
                                     uint64_t n_coeff_total, uint32_t *tu_info, cabac_substream_result *results) {
  std::vector<int32_t> wide(n_coeff_total ? n_coeff_total : 1, 0);
  const int rc = cabac_hip_residual_parse_batch(c, n_sub, desc, bytes, bytes_total, tile_first, tus, wide.data(), n_coeff_total, tu_info, results);
  if (coeff)
    for (uint64_t i = 0; i < n_coeff_total; i++) coeff[i] = int16_t(wide[i]);
  return rc;
}

}  // extern "C"
