// MI355X (gfx950) CABAC bin codec: context-store initialisation kernel and the dispatch of the codec kernels
// (cabac_kernels_v4.hip).  What the kernels restate, with the reference lines each piece must match bit for bit
// (paths relative to /root/reference/src):
//   context init        BinProbModel_Std::init / setLog2WindowSize / CtxStore::init
//                       common/contexts.cpp:893-901, :915-920, :996-1015
//   probability model   state / mps / getLPS / update / getRenormBitsLPS
//                       common/contexts.cpp:903-913, :939-954, :787-789
//   bin encoder         start / encodeBin / encodeBinEP / encodeBinTrm / writeOut / finish
//                       entropy_codec/arith_codec.cpp:329-337, :553-582, :389-399, :460-478, :524-546, :339-357
//   bin decoder         start / decodeBin / decodeBinEP / decodeBinTrm / finish
//                       entropy_codec/arith_codec.cpp:60-73, :242-277, :100-114, :181-197
//   byte I/O            OutputBitstream::write / writeByteAlignment, InputBitstream::readByte
//                       common/bit_stream.cpp:70-117, :152-155, :268-274
// The round-1 generations (v1 wave-serial, v2 lane-per-substream, v3 phased wave, v5 three-wave encoder) were never
// dispatched after round 1 and were retired in round 3; DESIGN.md section 3 keeps their measurements, the history their code.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cabac_device.h"
#include "cabac_hip.h"
#include "cabac_kernels.h"

namespace cabac {

// ------------------------------------------------------------------------------------------
// ctx-init only kernel (parity tests of a2/a3 in SURVEY.md section 8a)
__global__ __launch_bounds__(64) void ctx_init_kernel(uint32_t n_sub, const int32_t *qp, const uint32_t *init_id,
                                                      uint32_t *state, uint8_t *rate) {
  uint32_t s = blockIdx.x;
  if (s >= n_sub) return;
  int q = qp[s];
  q = q < 0 ? 0 : (q > 63 ? 63 : q);
  uint32_t id = init_id[s] & 3u;
  for (int k = threadIdx.x; k < kNumCtx; k += 64) {
    state[(size_t)s * kNumCtx + k] = ctx_init_state(q, c_init_tables[id * kNumCtx + k]);
    uint32_t r = ctx_init_rates(c_init_tables[3 * kNumCtx + k]);
    rate[(size_t)s * kNumCtx + k] = (uint8_t)(16 * (r & 0xff) + (r >> 8));  // m_rate layout
  }
}

// ------------------------------------------------------------------------------------------
// launchers (called from cabac_capi.cpp)
hipError_t launch_ctx_init(hipStream_t st, uint32_t n_sub, const int32_t *qp, const uint32_t *init_id, uint32_t *state,
                           uint8_t *rate) {
  if (n_sub == 0) return hipSuccess;
  hipLaunchKernelGGL(ctx_init_kernel, dim3(n_sub), dim3(64), 0, st, n_sub, qp, init_id, state, rate);
  return hipGetLastError();
}

// variant: 0 = auto, 4 = one-wave quad encoder (the round-1 baseline of this family), 6, 7; anything else is refused
hipError_t launch_encode(hipStream_t st, int variant, uint32_t n_sub, const cabac_substream_desc *desc,
                         const uint16_t *records, uint8_t *bytes, cabac_substream_result *results, uint32_t in_flight) {
  if (n_sub == 0) return hipSuccess;
  const int kind = variant & 0xff;
  // auto (measured, DESIGN.md section 3, tools/enc_scaling.py): from 3 072 substreams the lane-serial encoder (v7: C4 0.64, C5 10.3 ms,
  // 16 384 substreams 1.90 ms against v6's 0.82 / 12.8 / 2.81); below, one unit per workgroup, the four-wave quad encoder (v6:
  // C2 0.61, C3 12.8 ms, 1 024 substreams 0.43, 2 048: 0.57 ms against v7's 0.78 / 16.5 / 0.65 / 0.64)
  if (kind == 7 || (kind == 0 && max(n_sub, in_flight) >= 3072u)) return launch_encode_v7(st, n_sub, desc, records, bytes, results, in_flight);
  if (kind == 6 || kind == 0) return launch_encode_v6(st, n_sub, desc, records, bytes, results, in_flight);
  if (kind == 4) return launch_encode_v4(st, n_sub, desc, records, bytes, results);
  return hipErrorInvalidValue;
}

// variant: 0 = auto, 4 = quad decoder (four substreams per wave), 8 = sixteen substreams per wave (big batches), 1 = one substream per
// wave (few substreams); anything else is refused
hipError_t launch_decode(hipStream_t st, int variant, uint32_t n_sub, const cabac_substream_desc *desc,
                         const uint16_t *records, const uint8_t *bytes, uint8_t *bins,
                         cabac_substream_result *results, uint32_t in_flight, uint32_t *select) {
  if (n_sub == 0) return hipSuccess;
  const int kind = variant & 0xff;
  // auto: up to 1 024 substreams in flight one substream per wave (C2: 10, C3: 256); then the quad decoder (four substreams per
  // wave), the shortest chain per bin up to two quad waves per SIMD (C4: 4 096); from 9 216 about equally long substreams in flight sixteen per wave
  // decode 12 288 substreams in 2.23 ms against 3.14, 16 384 in 2.24 against 4.05
  if (kind == 0) return launch_decode_v4(st, n_sub, desc, records, bytes, bins, results, in_flight, 0, select);
  if (kind == 4) return launch_decode_v4(st, n_sub, desc, records, bytes, bins, results, in_flight, 16);
  if (kind == 8) return launch_decode_v4(st, n_sub, desc, records, bytes, bins, results, in_flight, 4);
  if (kind == 1) return launch_decode_v4(st, n_sub, desc, records, bytes, bins, results, in_flight, 64);
  return hipErrorInvalidValue;
}

}  // namespace cabac
