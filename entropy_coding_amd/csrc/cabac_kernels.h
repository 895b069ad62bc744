// Internal launch interface between the kernels (cabac_kernels.hip) and the C ABI (cabac_capi.cpp).
#ifndef CABAC_KERNELS_H
#define CABAC_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cabac_hip.h"

namespace cabac {

hipError_t launch_ctx_init(hipStream_t st, uint32_t n_sub, const int32_t *qp, const uint32_t *init_id, uint32_t *state,
                           uint8_t *rate);
// in_flight: the number of substreams on the device at the same time when this launch is one chunk of a batch whose
// chunks run concurrently on several streams (0 = this launch is alone); the workgroup geometry follows the whole batch
hipError_t launch_encode(hipStream_t st, int variant, uint32_t n_sub, const cabac_substream_desc *desc,
                         const uint16_t *records, uint8_t *bytes, cabac_substream_result *results, uint32_t in_flight = 0);
hipError_t launch_decode(hipStream_t st, int variant, uint32_t n_sub, const cabac_substream_desc *desc,
                         const uint16_t *records, const uint8_t *bytes, uint8_t *bins,
                         cabac_substream_result *results, uint32_t in_flight = 0, uint32_t *select = nullptr);

// v4 "quad" kernels (cabac_kernels_v4.hip): four substreams per wave
hipError_t launch_encode_v4(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                            uint8_t *bytes, cabac_substream_result *results);
hipError_t launch_encode_v6(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                            uint8_t *bytes, cabac_substream_result *results, uint32_t in_flight = 0);
hipError_t launch_encode_v7(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                            uint8_t *bytes, cabac_substream_result *results, uint32_t in_flight = 0);
// lanes_per_sub: 16 = the quad decoder (four substreams per wave), 4 = sixteen substreams per wave, 0 = by the batch: from
// 9 216 substreams in flight the choice is made on the device (select: one device word of the caller's, not shared with
// another launch in flight; null = the quad decoder)
hipError_t launch_decode_v4(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                            const uint8_t *bytes, uint8_t *bins, cabac_substream_result *results, uint32_t in_flight = 0,
                            int lanes_per_sub = 16, uint32_t *select = nullptr);

// bit estimator (cabac_kernels_v4.hip)
hipError_t launch_estimate(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                           uint64_t *frac_bits, uint32_t *flags, const uint32_t *start_state = nullptr,
                           const uint8_t *start_rate = nullptr, const uint32_t *start_set = nullptr);

// device binariser (cabac_binarize.hip)
hipError_t launch_binarize(hipStream_t st, uint32_t n_sub, const uint64_t *se_offset, const uint32_t *se,
                           const uint64_t *rec_offset, uint32_t *n_records, uint16_t *records);

// residual binariser (cabac_residual.hip)
// scratch: residual_scratch_bytes(n_tu) bytes of device memory the launch may overwrite (block ordering)
size_t residual_scratch_bytes(uint32_t n_tu);
// order_ready: `scratch` still holds the block order of an earlier launch over the SAME tus[] (the sizes pass of this call)
// coeff_bytes: 4 (int32_t, the reference's TCoeff) or 2 (int16_t: blocks of 15-bit dynamic range)
hipError_t launch_residual(hipStream_t st, uint32_t n_tu, const cabac_tu_desc *tus, const void *coeff, int coeff_bytes,
                           const uint64_t *rec_offset, uint32_t *n_records, uint32_t *info, uint16_t *records,
                           void *scratch, bool order_ready = false);

// residual parser (cabac_residual.hip): bytes -> coefficient blocks, one substream = blocks [tile_first[s], tile_first[s+1])
// (cabac_residual_parse.hip); tu_info (may be null): per block scanPosLast | CABAC_TU_INFO_*
hipError_t launch_residual_parse(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint8_t *bytes,
                                 const uint32_t *tile_first, const cabac_tu_desc *tus, void *coeff, int coeff_bytes /* 4 or 2 */,
                                 uint32_t *tu_info, cabac_substream_result *results);

// residual records spliced into host-recorded substreams (cabac_splice.hip); array sizes: pre n_splice + n_sub + 1,
// sub_n / sub_cap / rec_base / byte_base n_sub, seen n_tu, err 1, totals 3 ({records, bytes, error})
hipError_t launch_splice_plan(hipStream_t st, uint32_t n_sub, uint32_t n_tu, const cabac_substream_desc *desc,
                              const uint32_t *splice_first, const cabac_splice *splices, uint32_t n_splice,
                              const uint32_t *tu_n_records, uint32_t *pre, uint32_t *sub_n, uint32_t *sub_cap, uint32_t *seen, uint32_t *err,
                              uint64_t *rec_base, uint64_t *byte_base, uint64_t *totals);
hipError_t launch_splice_expand(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *host_records,
                                const uint32_t *splice_first, const cabac_splice *splices, const uint32_t *pre,
                                const uint32_t *sub_n, const uint32_t *sub_cap, const uint64_t *rec_base, const uint64_t *byte_base,
                                cabac_substream_desc *desc_out, uint64_t *tu_offset, uint16_t *records);
hipError_t launch_tu_range_check(hipStream_t st, uint32_t n_tu, const cabac_tu_desc *tus, uint64_t n_coeff_total, uint32_t *err);
hipError_t launch_tu_info_any(hipStream_t st, uint32_t n_tu, const uint32_t *info, uint32_t *flag);
hipError_t launch_bin_count(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                            uint32_t *counts);

// substream assembly (cabac_assemble.hip)
hipError_t launch_assemble(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc,
                           const cabac_substream_result *results, const uint8_t *bytes, uint8_t *payload,
                           uint64_t payload_capacity, uint64_t *offsets);
hipError_t launch_pack_bins(hipStream_t st, uint64_t n, const uint8_t *bins, uint8_t *packed);
hipError_t launch_gather_records(hipStream_t st, uint32_t n_seg, const uint64_t *src_off, const uint64_t *dst_off, const uint32_t *len,
                                 const uint16_t *src, uint16_t *dst);
hipError_t launch_split(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint64_t *offsets,
                        const uint8_t *payload, uint8_t *bytes);
hipError_t launch_count_emulations(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc,
                                   const cabac_substream_result *results, const uint8_t *bytes, uint32_t *counts);

}  // namespace cabac
#endif
