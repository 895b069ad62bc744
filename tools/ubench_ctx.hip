// What the context wave's pieces cost, one wave per SIMD: the 9-round match-any, the whole quad_phase_a (records from
// arithmetic), and a plain chain of as many vector instructions for comparison.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ientropy_coding_amd/csrc tools/ubench_ctx.hip -o tools/ubench_ctx
#include "../entropy_coding_amd/csrc/cabac_kernels_v4.hip"
#include <cstdio>
using namespace cabac;

__global__ __launch_bounds__(64) void k_match(unsigned *out, int iters, unsigned seed) {
  unsigned key = (threadIdx.x * 7u + seed) & 511u, acc = 0;
  for (int i = 0; i < iters; i++) {
    const uint64_t m = match_any_bits<9>(key, ~0ull);
    acc += (unsigned)m ^ (unsigned)(m >> 32);
    key = (key * 5u + acc + i) & 511u;
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(64) void k_phase(unsigned *out, int iters, unsigned seed) {
  __shared__ uint32_t ctx[kQuadSubs * kQuadCtxStride];
  const uint32_t lane = threadIdx.x, row = lane >> 4, j = lane & 15u;
  quad_ctx_init(ctx + row * kQuadCtxStride, 32, 2, j);
  __syncthreads();
  unsigned acc = 0, bad = 0;
  for (int i = 0; i < iters; i++) {
    const unsigned r = ((i * 37u + lane * 11u + seed) % 379u) | (((i + lane) & 1u) << 15);
    acc += quad_phase_a(r, true, lane, row, ctx + row * kQuadCtxStride, bad);
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc + bad;
}
__global__ __launch_bounds__(64) void k_phase_lds(unsigned *out, int iters, unsigned seed) {
  __shared__ uint32_t ctx[kQuadSubs * kQuadCtxStride];
  __shared__ uint32_t match[kMatchWords];
  __shared__ uint2 rate_tab[512];
  const uint32_t lane = threadIdx.x, row = lane >> 4, j = lane & 15u;
  quad_ctx_init(ctx + row * kQuadCtxStride, 32, 2, j);
  quad_rate_tab_init(rate_tab, lane, 64u);
  for (uint32_t k = lane; k < kMatchWords; k += 64) match[k] = 0;
  __syncthreads();
  unsigned acc = 0, bad = 0;
  for (int i = 0; i < iters; i++) {
    const unsigned r = ((i * 37u + lane * 11u + seed) % 379u) | (((i + lane) & 1u) << 15);
    acc += quad_phase_a<true>(r, true, lane, row, ctx + row * kQuadCtxStride, bad, match, rate_tab);
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc + bad;
}
#define R8(x) x x x x x x x x
__global__ __launch_bounds__(64) void k_plain(unsigned *out, int iters, unsigned seed) {
  unsigned x = threadIdx.x + seed, y = seed | 1u;
  for (int i = 0; i < iters; i++) {   // 192 dependent vector instructions per iteration
    asm volatile(R8(R8("v_mad_u32_u24 %0, %0, %1, %1\n v_xor_b32 %0, %0, %1\n v_sub_u32 %0, %0, %1\n")) : "+v"(x) : "v"(y));
  }
  out[blockIdx.x * 64 + threadIdx.x] = x;
}
int main() {
  unsigned *out; (void)hipMalloc(&out, 1024 * 64 * 4);
  const int iters = 4000;
#define RUN(name, what) { hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); name<<<1024, 64>>>(out, 10, 3); (void)hipEventRecord(e0); \
    name<<<1024, 64>>>(out, iters, 3); (void)hipEventRecord(e1); (void)hipDeviceSynchronize(); float ms; (void)hipEventElapsedTime(&ms, e0, e1); \
    printf("%-44s %.1f ns per iteration\n", what, ms * 1e6 / iters); }
  RUN(k_match, "match_any_bits<9> (+ 6 instructions)")
  RUN(k_phase, "quad_phase_a, records from arithmetic")
  RUN(k_phase_lds, "quad_phase_a, same-id lanes through LDS")
  RUN(k_plain, "192 dependent vector instructions")
  return 0;
}
