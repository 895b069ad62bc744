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
// (a substream whose encode overflowed its slot — CABAC_RES_OVERFLOW — counts on past byte_capacity; only what the slot
// holds is assembled)
__global__ __launch_bounds__(1024) void sizes_scan_kernel(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                          const cabac_substream_result *__restrict__ results,
                                                          uint64_t *__restrict__ offsets) {
  __shared__ uint64_t wave_sum[16];
  __shared__ uint64_t carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t tile = 0; tile < n_sub; tile += 1024) {
    const uint32_t s = tile + tid;
    const uint64_t sz = s < n_sub ? (uint64_t)min((results[s].n_bits + 7u) >> 3, desc[s].byte_capacity) : 0;
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
// after a match may start the next one, i.e. after a match the zero count restarts at this byte.  In a run of L
// zeros that makes the 3rd, 5th, 7th ... zero a match, and leaves "two zeros seen" after the run iff L is even; a
// following byte 1..3 then matches once more.  So a byte's fate depends only on the length k of the zero run that
// ends at it (or just before it): one wave per substream, 64 bytes per step, k from the ballot of the zero bytes
// (count of ones ending at the lane's bit) plus the run length carried in from the previous step.
__global__ __launch_bounds__(256) void count_emulations_kernel(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                               const cabac_substream_result *__restrict__ results,
                                                               const uint8_t *__restrict__ bytes, uint32_t *__restrict__ counts) {
  const uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (s >= n_sub) return;
  const uint8_t *p = bytes + desc[s].byte_offset;
  uint32_t n = (results[s].n_bits + 7u) >> 3;  // FIFO bytes (a trailing partial byte counts as held bits in the
  if (results[s].n_bits & 7u) n -= 1;           // reference and is not part of the FIFO)
  n = min(n, desc[s].byte_capacity);            // an overflowed encode counts on past its slot
  uint32_t cnt = 0, carry = 0;                  // carry: length of the zero run that ends at the last byte seen
  for (uint32_t base = 0; base < n; base += 64u) {
    const uint32_t i = base + lane;
    const bool in = i < n;
    const uint32_t b = in ? p[i] : 0xffu;
    const uint64_t zmask = __ballot(in && b == 0u);
    // zero run ending at this lane (0 if the byte is not zero), and the one ending just before it; a run that
    // reaches the first byte of this step continues the one carried in
    const uint64_t not_here = ~(zmask << (63u - lane));                    // bit 63 = this lane, 0 where the byte is zero
    const uint32_t ones = not_here ? (uint32_t)__builtin_clzll(not_here) : 64u;
    const uint32_t k_here = ones > lane ? lane + 1u + carry : ones;
    const uint64_t not_before = lane ? ~(zmask << (64u - lane)) : ~0ull;   // bit 63 = lane - 1
    const uint32_t ones_b = not_before ? (uint32_t)__builtin_clzll(not_before) : 64u;
    const uint32_t k_prev = ones_b >= lane ? lane + carry : ones_b;
    const bool match = in && (b == 0u ? (k_here >= 3u && (k_here & 1u)) : (b <= 3u && k_prev >= 2u && !(k_prev & 1u)));
    cnt += (uint32_t)__builtin_popcountll(__ballot(match));
    // the run that ends at the last byte of this step
    const uint32_t k_last = (uint32_t)__shfl((int)k_here, 63);
    carry = k_last;
  }
  if (lane == 0u) counts[s] = cnt;
}

// segments of bin records gathered into one buffer: segment k = src[src_off[k] .. + len[k]) -> dst[dst_off[k] ..) (offsets
// in records).  What re-packing a batch's substreams into per-GPU shards needs (entropy_coding_amd/sharding.py).
__global__ __launch_bounds__(256) void gather_records_kernel(uint32_t n_seg, const uint64_t *__restrict__ src_off,
                                                             const uint64_t *__restrict__ dst_off, const uint32_t *__restrict__ len,
                                                             const uint16_t *__restrict__ src, uint16_t *__restrict__ dst) {
  const uint32_t k = blockIdx.x;
  if (k >= n_seg) return;
  const uint16_t *s = src + src_off[k];
  uint16_t *d = dst + dst_off[k];
  const uint32_t n = len[k];
  // dwords where source and destination are two-byte-misaligned alike, single records at the ends and otherwise
  uint32_t head = (uint32_t)(((uintptr_t)d >> 1) & 1u);
  if (((((uintptr_t)s) ^ ((uintptr_t)d)) & 2u) != 0 || n < 8u) head = n;
  head = head < n ? head : n;
  for (uint32_t i = threadIdx.x; i < head; i += 256u) d[i] = s[i];
  const uint32_t pairs = (n - head) >> 1;
  const uint32_t *s32 = reinterpret_cast<const uint32_t *>(s + head);
  uint32_t *d32 = reinterpret_cast<uint32_t *>(d + head);
  for (uint32_t i = threadIdx.x; i < pairs; i += 256u) d32[i] = s32[i];
  if (threadIdx.x == 0 && ((n - head) & 1u)) d[n - 1] = s[n - 1];
}

hipError_t launch_gather_records(hipStream_t st, uint32_t n_seg, const uint64_t *src_off, const uint64_t *dst_off, const uint32_t *len,
                                 const uint16_t *src, uint16_t *dst) {
  if (n_seg) hipLaunchKernelGGL(gather_records_kernel, dim3(n_seg), dim3(256), 0, st, n_seg, src_off, dst_off, len, src, dst);
  return hipGetLastError();
}

// decoded bins, one byte each, packed eight to a byte: bit (r & 7) of packed[r >> 3] = bins[r] & 1.  What the host-pointer
// decode hands back when asked to (cabac_hip_decode_batch_packed): an eighth of the bytes over PCIe.
__global__ __launch_bounds__(256) void pack_bins_kernel(uint64_t n, const uint8_t *__restrict__ bins, uint8_t *__restrict__ packed) {
  const uint64_t b = (uint64_t)blockIdx.x * 256u + threadIdx.x, i = b * 8u;
  if (i >= n) return;
  uint32_t v = 0;
  if (i + 8u <= n) {
    const uint64_t w = *reinterpret_cast<const uint64_t *>(bins + i) & 0x0101010101010101ull;   // (hipMalloc'd, i a multiple of 8)
    v = (uint32_t)((w * 0x0102040810204080ull) >> 56);
  } else {
    for (uint64_t k = i; k < n; k++) v |= (uint32_t)(bins[k] & 1u) << (k - i);
  }
  packed[b] = (uint8_t)v;
}

hipError_t launch_pack_bins(hipStream_t st, uint64_t n, const uint8_t *bins, uint8_t *packed) {
  const uint64_t bytes = (n + 7u) / 8u;
  if (bytes) hipLaunchKernelGGL(pack_bins_kernel, dim3((uint32_t)((bytes + 255u) / 256u)), dim3(256), 0, st, n, bins, packed);
  return hipGetLastError();
}

hipError_t launch_assemble(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc,
                           const cabac_substream_result *results, const uint8_t *bytes, uint8_t *payload,
                           uint64_t payload_capacity, uint64_t *offsets) {
  hipLaunchKernelGGL(sizes_scan_kernel, dim3(1), dim3(1024), 0, st, n_sub, desc, results, offsets);
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
    hipLaunchKernelGGL(count_emulations_kernel, dim3((n_sub + 3) / 4), dim3(256), 0, st, n_sub, desc, results, bytes,
                       counts);
  return hipGetLastError();
}

}  // namespace cabac
