// Issue rate vs. instruction-level parallelism for one wave (and for several waves per SIMD):
// K independent dependent-chains of v_add_u32 interleaved in one asm block.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_ilp.hip -o tools/ubench_ilp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define A(r) "v_add_u32 %" #r ", %" #r ", %8\n\t"
#define X(r) "v_xor_b32 %" #r ", %" #r ", %8\n\t"
#define REP8(x) x x x x x x x x

template <int K>
__global__ void k_ilp(unsigned long long *cyc, unsigned *sink, int iters) {
  unsigned v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  unsigned inc = threadIdx.x | 1;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (K == 1) asm volatile(REP8(A(0) A(0) A(0) A(0) A(0) A(0) A(0) A(0)) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(inc));
    if (K == 2) asm volatile(REP8(A(0) A(1) A(0) A(1) A(0) A(1) A(0) A(1)) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(inc));
    if (K == 3) asm volatile(REP8(A(0) A(1) A(2) A(0) A(1) A(2) A(0) A(1)) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(inc));
    if (K == 4) asm volatile(REP8(A(0) A(1) A(2) A(3) A(0) A(1) A(2) A(3)) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(inc));
    if (K == 8) asm volatile(REP8(A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7)) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(inc));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  if (threadIdx.x == blockDim.x - 1) cyc[1] = t1 - t0;
}

template <int K>
void run(int threads, unsigned long long *cyc, unsigned *sink) {
  const int iters = 100000;
  k_ilp<K><<<1, threads>>>(cyc, sink, 100);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  k_ilp<K><<<1, threads>>>(cyc, sink, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long h[2];
  hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  printf("K=%d chains, %4d threads (%d waves/SIMD): memtime ticks per instr: first wave %.2f, last wave %.2f; wall %.2f ns per instr per wave\n", K, threads,
         (threads + 255) / 256, (double)h[0] / (iters * 64.0), (double)h[1] / (iters * 64.0), ms * 1e6 / (iters * 64.0));
}

int main() {
  unsigned long long *cyc; unsigned *sink;
  (void)hipMalloc(&cyc, 8 * 64); (void)hipMalloc(&sink, 4 * 4096);
  for (int threads : {64, 256, 512, 1024}) {
    run<1>(threads, cyc, sink); run<2>(threads, cyc, sink); run<3>(threads, cyc, sink); run<4>(threads, cyc, sink); run<8>(threads, cyc, sink);
  }
  return 0;
}
