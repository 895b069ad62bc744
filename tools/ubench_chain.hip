// Micro-benchmark of the v4/v5 encode chain alone (no phase (a), no barriers): cycles per 16-step group.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ientropy_coding_amd/csrc tools/ubench_chain.hip -o tools/ubench_chain
#include "../entropy_coding_amd/csrc/cabac_kernels_v4.hip"
#include <cstdio>
#include <vector>
using namespace cabac;

__global__ void chain_bench(unsigned long long *cyc, unsigned *sink, int iters, unsigned seed, uint8_t *scratch) {
  const uint32_t lane = threadIdx.x & 63u;
  QuadEnc e;
  e.low = 0; e.range = 510; e.pend = 0; e.buf = 0; e.nbuf = 0; e.pos = 0; e.dst = scratch + (blockIdx.x * 4 + (lane >> 4)) * 65536; e.cap = 65536;
  uint32_t x = seed * 2654435761u + lane * 40503u + blockIdx.x * 977u;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    x = x * 1664525u + 1013904223u;
    const uint32_t kind = (x >> 28);                 // 75 % context bins, 25 % bypass
    const bool is_ep = kind >= 12;
    uint32_t info = 0;
    if (!is_ep) info = ((x >> 8) & 31u) | (8u << 5) | (((x >> 20) & 7u) == 0 ? (1u << 9) : 0u);
    else info = (1u << 10) | (((x >> 13) & 1u) << 11);
    const QuadEncInfo f = quad_unpack(info);
    quad_enc_steps<false, false>(f, e, false, QuadPost());
    if (e.pos > 60000) e.pos = 0;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[blockIdx.x * 64 + lane] = (unsigned)e.low ^ e.range ^ e.pos;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  unsigned long long *cyc; unsigned *sink; uint8_t *scratch;
  hipMalloc(&cyc, 8 * 8192); hipMalloc(&sink, 4 * 64 * 8192); hipMalloc(&scratch, (size_t)8192 * 4 * 65536);
  const int iters = 2000;
  for (int grid : {1, 256, 1024, 2048, 4096}) {
    chain_bench<<<grid, 64>>>(cyc, sink, iters, 3, scratch); hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); chain_bench<<<grid, 64>>>(cyc, sink, iters, 5, scratch); hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(grid); hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    printf("grid %5d single-wave WGs: %.0f cycles per 16-step group (memtime), %.1f per step; wall %.3f ms -> %.1f ns per group\n", grid,
           avg / iters, avg / iters / 16.0, ms, ms * 1e6 / iters);
  }
  return 0;
}
