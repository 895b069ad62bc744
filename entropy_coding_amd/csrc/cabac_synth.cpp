/* Synthetic bin-record generator (host, deterministic) — SURVEY.md §8(d).
 *
 * The reference ships no bin traces (its only inputs are three y4m clips that are absent,
 * .MISSING_LARGE_BLOBS:1-3), so throughput is measured on synthetic bin buffers of the shape
 * BASELINE.json states.  The mix imitates intra residual coding, where ~all bins come from
 * (reference cabac_writer.cpp:2724-2872): 60 % of the context-coded bins use the
 * SigFlag/ParFlag/GtxFlag contexts (ctxId 90..245), 15 % LastX/LastY (246..291), 25 % the
 * rest; every context has a fixed P(1); bypass bins are fair; the substream ends with the
 * end_of_slice terminate bin (cabac_writer.cpp:104-107).
 */
#include "cabac_hip.h"

static inline uint64_t splitmix64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* P(bin = 1) in 1/65536 units, picked per context by a multiplicative hash of its id */
static inline uint32_t ctx_p_one(uint32_t ctx_id) {
  static const uint32_t p[6] = {1966, 6554, 16384, 32768, 49152, 58982}; /* .03 .1 .25 .5 .75 .9 */
  return p[((ctx_id * 2654435761u) >> 16) % 6u];
}

void cabac_synth_records(uint64_t seed, uint64_t substream_index, uint32_t n_bins,
                         uint32_t ctx_permille, uint16_t *out) {
  uint64_t s = seed ^ substream_index;
  if (n_bins == 0) return;
  for (uint32_t i = 0; i + 1 < n_bins; i++) {
    uint64_t r = splitmix64(&s);
    if ((uint32_t)((r >> 32) % 1000u) < ctx_permille) {
      uint64_t r2 = splitmix64(&s);
      uint32_t sel = (uint32_t)(r2 % 100u);
      uint32_t pick = (uint32_t)(r2 >> 8) & 0xffffffu;
      uint32_t id;
      if (sel < 60) {
        id = 90 + pick % 156u;
      } else if (sel < 75) {
        id = 246 + pick % 46u;
      } else {
        uint32_t k = pick % 177u; /* 0..89 and 292..378 */
        id = k < 90 ? k : k + 202;
      }
      uint32_t bin = (uint32_t)((r2 >> 40) & 0xffffu) < ctx_p_one(id);
      out[i] = (uint16_t)(id | (bin ? CABAC_REC_BIN : 0));
    } else {
      out[i] = (uint16_t)(CABAC_REC_EP | (((r >> 8) & 1u) ? CABAC_REC_BIN : 0));
    }
  }
  out[n_bins - 1] = (uint16_t)(CABAC_REC_TRM | CABAC_REC_BIN);
}
