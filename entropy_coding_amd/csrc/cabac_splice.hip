// Splicing residual records into host-recorded substreams on the device (the writer's side).
//
// In the reference every bin of CABACWriter::residual_coding goes straight into the bin encoder, between the bins of the
// syntax elements around it (cabac_writer.cpp:2424-2525; the flags :2766-2803, escapes :2822 / :2843, signs :2871).  Here the
// host records the syntax elements it walks itself as bin records and marks where a transform block's bins belong — a
// *splice*: (position in the substream's host records, block index) — and hands the coefficients over instead of bins.  The
// kernels below put the two together without the block records ever existing on the host:
//   residual sizes pass (cabac_residual.hip)  ->  n_records per block
//   splice_plan_kernel    per substream: running sum of its blocks' sizes at every splice, expanded length, byte-slot size
//   splice_scan_kernel    across substreams: where each expanded substream starts (records, bytes); totals; list check
//   splice_expand_kernel  per substream: its final descriptor, every block's destination, the host records moved apart
//   residual records pass (cabac_residual.hip) writes each block's records at its destination
//   encode kernel (cabac_kernels_v4.hip) on the expanded substreams; bin_count_kernel for the BinCounter totals
// No atomics on the data path (one atomicAdd per splice checks that every block is spliced exactly once).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cabac_hip.h"
#include "cabac_kernels.h"

namespace cabac {

namespace {

// block-wide exclusive scan of one 64-bit value per thread (256 threads); returns the exclusive prefix, total in *total
__device__ __forceinline__ uint64_t block_excl_scan256(uint64_t v, uint64_t *wave_sum /* [4] LDS */, uint64_t *total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint64_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t up = __shfl_up(incl, d);
    if ((int)lane >= d) incl += up;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  uint64_t base = 0, all = 0;
  for (uint32_t k = 0; k < 4; k++) {
    if (k < wave) base += wave_sum[k];
    all += wave_sum[k];
  }
  __syncthreads();
  *total = all;
  return base + incl - v;
}

// worst case bytes of n coded records: a record shifts out at most 7 bits (terminate bin; a context bin at most 6,
// contexts.cpp:787-789), finish() up to 4 bytes more, the alignment one; rounded up to the slots' 16-byte alignment
__device__ __forceinline__ uint64_t slot_bytes(uint64_t n_records) { return ((7u * n_records + 7u) / 8u + 8u + 15u) / 16u * 16u; }

}  // namespace

// pre[j + s] = records of the substream's blocks spliced in before splice j (pre[j1 + s] = of all of them)
__global__ __launch_bounds__(256) void splice_plan_kernel(uint32_t n_sub, uint32_t n_tu, uint32_t n_splice, const cabac_substream_desc *__restrict__ desc,
                                                          const uint32_t *__restrict__ splice_first,
                                                          const cabac_splice *__restrict__ splices,
                                                          const uint32_t *__restrict__ tu_n_records, uint32_t *__restrict__ pre,
                                                          uint32_t *__restrict__ sub_n, uint32_t *__restrict__ sub_cap,
                                                          uint32_t *__restrict__ seen, uint32_t *__restrict__ err) {
  __shared__ uint64_t wave_sum[4];
  const uint32_t s = blockIdx.x;
  if (s >= n_sub) return;
  uint32_t j0 = splice_first[s], j1 = splice_first[s + 1];
  const uint32_t n_host = desc[s].n_records;
  uint64_t carry = 0;
  // a list that is not one: nothing of it is read (the arrays hold n_splice splices and n_splice + n_sub + 1 sums)
  uint32_t bad = (j1 < j0 || j1 > n_splice || (s == 0u && j0 != 0u) || (s + 1u == n_sub && j1 != n_splice)) ? 1u : 0u;
  if (bad) j0 = j1 = 0u;
  for (uint32_t base = j0; base < j1; base += 256u) {
    const uint32_t j = base + threadIdx.x;
    const bool valid = j < j1;
    uint64_t sz = 0;
    if (valid) {
      const cabac_splice sp = splices[j];
      const bool ok = sp.tu < n_tu && sp.at <= n_host && (j == j0 || splices[j - 1].at <= sp.at);
      if (ok) {
        sz = tu_n_records[sp.tu];
        atomicAdd(&seen[sp.tu], 1u);
      } else {
        bad = 1;
      }
    }
    uint64_t total;
    const uint64_t excl = block_excl_scan256(sz, wave_sum, &total);
    if (valid) pre[j + s] = (uint32_t)(carry + excl);
    carry += total;
  }
  const uint64_t expanded = (uint64_t)n_host + carry;
  if (expanded > 0xfffffff0ull || slot_bytes(expanded) > 0xfffffff0ull) bad = 1;
  if (threadIdx.x == 0) {
    pre[j1 + s] = (uint32_t)carry;
    sub_n[s] = (uint32_t)expanded;
    sub_cap[s] = (uint32_t)slot_bytes(expanded);
  }
  if (bad) atomicOr(err, 1u);
}

// every block spliced exactly once (the whole grid looks: one workgroup doing it alone took 0.26 ms for 1.6 M blocks)
__global__ __launch_bounds__(256) void splice_seen_kernel(uint32_t n_tu, const uint32_t *__restrict__ seen, uint32_t *__restrict__ err) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const bool bad = t < n_tu && seen[t] != 1u;
  if (__ballot(bad) != 0ull && (threadIdx.x & 63u) == 0u) atomicOr(err, 1u);
}

// one workgroup: exclusive scans over the substreams and the totals {records, bytes, error}
__global__ __launch_bounds__(1024) void splice_scan_kernel(uint32_t n_sub, const uint32_t *__restrict__ sub_n,
                                                           const uint32_t *__restrict__ sub_cap,
                                                           uint64_t *__restrict__ rec_base, uint64_t *__restrict__ byte_base,
                                                           const uint32_t *__restrict__ err, uint64_t *__restrict__ totals) {
  __shared__ uint64_t wave_a[16], wave_b[16];
  __shared__ uint64_t carry_a, carry_b;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) carry_a = carry_b = 0;
  __syncthreads();
  for (uint32_t tile = 0; tile < n_sub; tile += 1024u) {
    const uint32_t s = tile + tid;
    const uint64_t a = s < n_sub ? sub_n[s] : 0, b = s < n_sub ? sub_cap[s] : 0;
    uint64_t ia = a, ib = b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t ua = __shfl_up(ia, d), ub = __shfl_up(ib, d);
      if ((int)lane >= d) {
        ia += ua;
        ib += ub;
      }
    }
    if (lane == 63) {
      wave_a[wave] = ia;
      wave_b[wave] = ib;
    }
    __syncthreads();
    uint64_t ba = carry_a, bb = carry_b;
    for (uint32_t k = 0; k < wave; k++) {
      ba += wave_a[k];
      bb += wave_b[k];
    }
    if (s < n_sub) {
      rec_base[s] = ba + ia - a;
      byte_base[s] = bb + ib - b;
    }
    __syncthreads();
    if (tid == 1023) {
      carry_a = ba + ia;
      carry_b = bb + ib;
    }
    __syncthreads();
  }
  if (tid == 0) {
    totals[0] = carry_a;
    totals[1] = carry_b;
    totals[2] = *err ? 1u : 0u;
  }
}

// the expanded substream: descriptor, block destinations, host records moved to their places
__global__ __launch_bounds__(256) void splice_expand_kernel(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                            const uint16_t *__restrict__ host_records,
                                                            const uint32_t *__restrict__ splice_first,
                                                            const cabac_splice *__restrict__ splices, const uint32_t *__restrict__ pre,
                                                            const uint32_t *__restrict__ sub_n, const uint32_t *__restrict__ sub_cap,
                                                            const uint64_t *__restrict__ rec_base, const uint64_t *__restrict__ byte_base,
                                                            cabac_substream_desc *__restrict__ desc_out, uint64_t *__restrict__ tu_offset,
                                                            uint16_t *__restrict__ records) {
  const uint32_t s = blockIdx.x;
  if (s >= n_sub) return;
  const cabac_substream_desc d = desc[s];
  const uint32_t j0 = splice_first[s], ns = splice_first[s + 1] - j0;
  const uint64_t base = rec_base[s];
  if (threadIdx.x == 0) {
    cabac_substream_desc o = d;
    o.rec_offset = base;
    o.byte_offset = byte_base[s];
    o.n_records = sub_n[s];
    o.byte_capacity = sub_cap[s];
    desc_out[s] = o;
  }
  const cabac_splice *sp = splices + j0;
  const uint32_t *p = pre + j0 + s;
  for (uint32_t k = threadIdx.x; k < ns; k += 256u) tu_offset[sp[k].tu] = base + sp[k].at + p[k];
  const uint16_t *src = host_records + d.rec_offset;
  for (uint32_t i = threadIdx.x; i < d.n_records; i += 256u) {
    // the blocks spliced in at or before host record i: the first k splices, k = #(at <= i); `at` is sorted
    uint32_t lo = 0, hi = ns;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (sp[mid].at <= i) lo = mid + 1;
      else hi = mid;
    }
    records[base + i + p[lo]] = src[i];
  }
}

// BinCounter totals of the expanded substreams (arith_codec.cpp:281-316): per substream 379 context-bin counts, then the
// bypass and the terminate bins
__global__ __launch_bounds__(256) void bin_count_kernel(uint32_t n_sub, const cabac_substream_desc *__restrict__ desc,
                                                        const uint16_t *__restrict__ records, uint32_t *__restrict__ counts) {
  __shared__ uint32_t hist[CABAC_BIN_COUNT_WORDS];
  const uint32_t s = blockIdx.x;
  if (s >= n_sub) return;
  for (uint32_t k = threadIdx.x; k < CABAC_BIN_COUNT_WORDS; k += 256u) hist[k] = 0;
  __syncthreads();
  const cabac_substream_desc d = desc[s];
  const uint16_t *rec = records + d.rec_offset;
  for (uint32_t i = threadIdx.x; i < d.n_records; i += 256u) {
    const uint32_t id = rec[i] & CABAC_REC_ID_MASK;
    if (id < CABAC_NUM_CONTEXTS) atomicAdd(&hist[id], 1u);
    else if (id == CABAC_REC_EP) atomicAdd(&hist[CABAC_NUM_CONTEXTS], 1u);
    else if (id == CABAC_REC_TRM) atomicAdd(&hist[CABAC_NUM_CONTEXTS + 1], 1u);
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < CABAC_BIN_COUNT_WORDS; k += 256u) counts[(size_t)s * CABAC_BIN_COUNT_WORDS + k] = hist[k];
}

// descriptor checks the host-pointer entry point leaves to the device (one word per million blocks is cheaper here than a
// host loop): a block whose coefficients would lie outside [0, n_coeff_total) sets *err
__global__ __launch_bounds__(256) void tu_range_check_kernel(uint32_t n_tu, const cabac_tu_desc *__restrict__ tus, uint64_t n_coeff_total,
                                                             uint32_t *__restrict__ err) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= n_tu) return;
  const cabac_tu_desc d = tus[t];
  if (d.log2_width > 6 || d.log2_height > 6) return;  // flagged by the binariser (CABAC_TU_INFO_BAD_DESC), reads nothing
  const uint64_t n = 1ull << (d.log2_width + d.log2_height);
  if (d.coeff_offset > n_coeff_total || n > n_coeff_total - d.coeff_offset) atomicOr(err, 1u);
}

// any block empty or with a bad descriptor (its splice adds no records; the caller is told)
__global__ __launch_bounds__(256) void tu_info_any_kernel(uint32_t n_tu, const uint32_t *__restrict__ info, uint32_t *__restrict__ flag) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t < n_tu && (info[t] & (CABAC_TU_INFO_EMPTY | CABAC_TU_INFO_BAD_DESC))) atomicOr(flag, 1u);
}

hipError_t launch_tu_range_check(hipStream_t st, uint32_t n_tu, const cabac_tu_desc *tus, uint64_t n_coeff_total, uint32_t *err) {
  hipError_t e = hipMemsetAsync(err, 0, sizeof(uint32_t), st);
  if (e != hipSuccess) return e;
  if (n_tu) hipLaunchKernelGGL(tu_range_check_kernel, dim3((n_tu + 255u) / 256u), dim3(256), 0, st, n_tu, tus, n_coeff_total, err);
  return hipGetLastError();
}

hipError_t launch_tu_info_any(hipStream_t st, uint32_t n_tu, const uint32_t *info, uint32_t *flag) {
  hipError_t e = hipMemsetAsync(flag, 0, sizeof(uint32_t), st);
  if (e != hipSuccess) return e;
  if (n_tu) hipLaunchKernelGGL(tu_info_any_kernel, dim3((n_tu + 255u) / 256u), dim3(256), 0, st, n_tu, info, flag);
  return hipGetLastError();
}

hipError_t launch_splice_plan(hipStream_t st, uint32_t n_sub, uint32_t n_tu, const cabac_substream_desc *desc,
                              const uint32_t *splice_first, const cabac_splice *splices, uint32_t n_splice,
                              const uint32_t *tu_n_records, uint32_t *pre, uint32_t *sub_n, uint32_t *sub_cap, uint32_t *seen, uint32_t *err,
                              uint64_t *rec_base, uint64_t *byte_base, uint64_t *totals) {
  hipError_t e = hipMemsetAsync(seen, 0, sizeof(uint32_t) * (n_tu ? n_tu : 1u), st);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(err, 0, sizeof(uint32_t), st);
  if (e != hipSuccess) return e;
  if (n_sub)
    hipLaunchKernelGGL(splice_plan_kernel, dim3(n_sub), dim3(256), 0, st, n_sub, n_tu, n_splice, desc, splice_first, splices, tu_n_records,
                       pre, sub_n, sub_cap, seen, err);
  if (n_tu) hipLaunchKernelGGL(splice_seen_kernel, dim3((n_tu + 255u) / 256u), dim3(256), 0, st, n_tu, seen, err);
  hipLaunchKernelGGL(splice_scan_kernel, dim3(1), dim3(1024), 0, st, n_sub, sub_n, sub_cap, rec_base, byte_base, err, totals);
  return hipGetLastError();
}

hipError_t launch_splice_expand(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *host_records,
                                const uint32_t *splice_first, const cabac_splice *splices, const uint32_t *pre,
                                const uint32_t *sub_n, const uint32_t *sub_cap, const uint64_t *rec_base, const uint64_t *byte_base,
                                cabac_substream_desc *desc_out, uint64_t *tu_offset, uint16_t *records) {
  if (n_sub)
    hipLaunchKernelGGL(splice_expand_kernel, dim3(n_sub), dim3(256), 0, st, n_sub, desc, host_records, splice_first, splices, pre,
                       sub_n, sub_cap, rec_base, byte_base, desc_out, tu_offset, records);
  return hipGetLastError();
}

hipError_t launch_bin_count(hipStream_t st, uint32_t n_sub, const cabac_substream_desc *desc, const uint16_t *records,
                            uint32_t *counts) {
  if (n_sub) hipLaunchKernelGGL(bin_count_kernel, dim3(n_sub), dim3(256), 0, st, n_sub, desc, records, counts);
  return hipGetLastError();
}

}  // namespace cabac
