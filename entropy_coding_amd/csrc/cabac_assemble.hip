// Substream assembly on the device: what the reference does on the host after
// the bin codec, restated so that coded substreams can leave (or enter) the GPU as one contiguous payload.
//   assemble  == OutputBitstream::addSubstream for byte-aligned substreams      common/bit_stream.cpp:139-150
//                (each substream already carries its rbsp stop bit: CABAC_SUB_ALIGN_RBSP)
//   split     == InputBitstream::extractSubstream, byte-aligned case             common/bit_stream.cpp:382-415
//   count     == OutputBitstream::countStartCodeEmulations                       common/bit_stream.cpp:157-181
// All three are plain byte movers / scanners: HBM-bound, no arithmetic to speak of.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cabac_hip.h"
#include "cabac_kernels.h"

namespace cabac {

// exclusive scan of the substream sizes: one workgroup, tiles of 1024 with a running carry
__global__ __launch_bounds__(1024) void sizes_scan_kernel(uint32_t n_sub, const cabac_substream_result *__restrict__ results,
                                                          uint64_t *__restrict__ offsets) {
  __shared__ uint64_t wave_sum[16];
  __shared__ uint64_t carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t tile = 0; tile < n_sub; tile += 1024) {
    const uint32_t s = tile + tid;
    const uint64_t sz = s < n_sub ? (uint64_t)((results[s].n_bits + 7u) >> 3) : 0;
    uint64_t incl = sz;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t up = __shfl_up(incl, d);
      if ((int)lane >= d) incl += up;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint64_t base = carry;
    for (uint32_t k = 0; k < wave; k++) base += wave_sum[k];
    if (s < n_sub) offsets[s] = base + incl - sz;
    __syncthreads();
    if (tid == 1023) carry = base + incl;
    __syncthreads();
  }
  if (tid == 0) offsets[n_sub] = carry;
}

// TO_PAYLOAD: slot (16-B aligned) -> payload[offsets[s]..]; else the inverse
template <bool TO_PAYLOAD>
__global__ __launch_bounds__(256) void copy_substreams_kernel(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                              const uint64_t *__restrict__ offsets,
                                                              uint8_t *__restrict__ slots, uint8_t *__restrict__ payload,
                                                              uint64_t payload_capacity) {
  const uint32_t s = blockIdx.x;
  if (s >= n_sub) return;
  const uint64_t off = offsets[s];
  uint64_t n = offsets[s + 1] - off;
  if (TO_PAYLOAD && off + n > payload_capacity) n = off < payload_capacity ? payload_capacity - off : 0;
  if (!TO_PAYLOAD && n > desc[s].byte_capacity) n = desc[s].byte_capacity;
  uint8_t *slot = slots + desc[s].byte_offset;
  uint8_t *pay = payload + off;
  // slots are 16-byte aligned: move aligned dwords on that side, bytes on the payload side
  const uint64_t n4 = n >> 2;
  for (uint64_t i = threadIdx.x; i < n4; i += blockDim.x) {
    if (TO_PAYLOAD) {
      const uint32_t w = reinterpret_cast<const uint32_t *>(slot)[i];
      pay[4 * i + 0] = (uint8_t)w;
      pay[4 * i + 1] = (uint8_t)(w >> 8);
      pay[4 * i + 2] = (uint8_t)(w >> 16);
      pay[4 * i + 3] = (uint8_t)(w >> 24);
    } else {
      const uint32_t w = (uint32_t)pay[4 * i] | ((uint32_t)pay[4 * i + 1] << 8) | ((uint32_t)pay[4 * i + 2] << 16) |
                         ((uint32_t)pay[4 * i + 3] << 24);
      reinterpret_cast<uint32_t *>(slot)[i] = w;
    }
  }
  for (uint64_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
    if (TO_PAYLOAD) pay[i] = slot[i];
    else slot[i] = pay[i];
  }
}

// countStartCodeEmulations, bit_stream.cpp:157-181: greedy, non-overlapping 00 00 {00..03} matches; the byte
// after a match may start the next one.  One lane per substream (streams are a few KB).
__global__ __launch_bounds__(64) void count_emulations_kernel(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                              const cabac_substream_result *__restrict__ results,
                                                              const uint8_t *__restrict__ bytes, uint32_t *__restrict__ counts) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_sub) return;
  const uint8_t *p = bytes + desc[s].byte_offset;
  uint32_t n = (results[s].n_bits + 7u) >> 3;  // FIFO bytes (a trailing partial byte counts as held bits in the
  if (results[s].n_bits & 7u) n -= 1;           // reference and is not part of the FIFO)
  uint32_t cnt = 0, zeros = 0;
  // search_n(found, end - 1, 2, 0): the pair of zeros must lie in [0, n-1), the third byte is p[i]
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t b = p[i];
    if (zeros >= 2 && b <= 3) {
      cnt++;
      zeros = (b == 0) ? 1 : 0;  // the scan resumes AT this byte
    } else {
      zeros = (b == 0) ? zeros + 1 : 0;
    }
  }
  counts[s] = cnt;
}

hipError_t launch_assemble(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc,
                           const cabac_substream_result *results, const uint8_t *bytes, uint8_t *payload,
                           uint64_t payload_capacity, uint64_t *offsets) {
  hipLaunchKernelGGL(sizes_scan_kernel, dim3(1), dim3(1024), 0, st, n_sub, results, offsets);
  if (n_sub)
    hipLaunchKernelGGL(copy_substreams_kernel<true>, dim3(n_sub), dim3(256), 0, st, n_sub, desc, offsets,
                       const_cast<uint8_t *>(bytes), payload, payload_capacity);
  return hipGetLastError();
}

hipError_t launch_split(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint64_t *offsets,
                        const uint8_t *payload, uint8_t *bytes) {
  if (n_sub)
    hipLaunchKernelGGL(copy_substreams_kernel<false>, dim3(n_sub), dim3(256), 0, st, n_sub, desc, offsets, bytes,
                       const_cast<uint8_t *>(payload), ~0ull);
  return hipGetLastError();
}

hipError_t launch_count_emulations(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc,
                                   const cabac_substream_result *results, const uint8_t *bytes, uint32_t *counts) {
  if (n_sub)
    hipLaunchKernelGGL(count_emulations_kernel, dim3((n_sub + 63) / 64), dim3(64), 0, st, n_sub, desc, results, bytes,
                       counts);
  return hipGetLastError();
}

}  // namespace cabac
