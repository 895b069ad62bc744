// Micro-benchmark: cost model of dependent instruction chains on gfx950 (used to size the CABAC kernels).
// hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void chain(unsigned *out, unsigned long long *cyc, int iters, unsigned seed) {
  unsigned x = seed + threadIdx.x, y = seed * 3 + 1;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (KIND == 0) x = x + y;                                  // v_add_u32 dependent
      else if (KIND == 1) x = (x * y) >> 1;                      // v_mul_lo_u32 + shift
      else if (KIND == 2) x = (x < 1000u) ? x + y : x - y;       // cmp + cndmask + add/sub
      else if (KIND == 3) { unsigned long long v = ((unsigned long long)x << 32) | y; v <<= (x & 7); x = (unsigned)(v >> 32) + 1; } // 64-bit shift
      else if (KIND == 4) x = __builtin_clz(x | 1) + x;          // ffbh
      else if (KIND == 5) { unsigned s = __builtin_amdgcn_readfirstlane(x); s = s * 3 + 1; x = s; } // SALU chain via readfirstlane
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void lds_chain(unsigned *out, unsigned long long *cyc, int iters) {
  __shared__ unsigned tab[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab[i] = (i * 7 + 1) & 1023;
  __syncthreads();
  unsigned x = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters * 16; i++) x = tab[x & 1023];
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  unsigned *out; unsigned long long *cyc;
  CHK(hipMalloc(&out, 1 << 24)); CHK(hipMalloc(&cyc, 1 << 20));
  const int iters = 2000;
  const char *names[] = {"v_add dep", "v_mul_lo+shift dep", "cmp+cndmask+add dep", "64b shift dep", "ffbh+add dep", "readfirstlane+salu"};
  struct Cfg { int blocks, threads; const char *what; } cfgs[] = {
      {1, 64, "1 wave on the chip"}, {256, 64, "1 wave per CU"}, {1024, 64, "1024 single-wave WGs"}, {256, 256, "4-wave WG per CU"},
      {256, 512, "8-wave WG per CU"}, {256, 1024, "16-wave WG per CU"}, {4096, 64, "4096 single-wave WGs"}};
  for (auto c : cfgs) {
    printf("--- %s (grid %d x %d)\n", c.what, c.blocks, c.threads);
    for (int kind = 0; kind < 6; kind++) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      auto launch = [&]() {
        switch (kind) {
          case 0: chain<0><<<c.blocks, c.threads>>>(out, cyc, iters, 5); break;
          case 1: chain<1><<<c.blocks, c.threads>>>(out, cyc, iters, 5); break;
          case 2: chain<2><<<c.blocks, c.threads>>>(out, cyc, iters, 5); break;
          case 3: chain<3><<<c.blocks, c.threads>>>(out, cyc, iters, 5); break;
          case 4: chain<4><<<c.blocks, c.threads>>>(out, cyc, iters, 5); break;
          case 5: chain<5><<<c.blocks, c.threads>>>(out, cyc, iters, 5); break;
        }
      };
      launch(); CHK(hipDeviceSynchronize());
      hipEventRecord(a); launch(); hipEventRecord(b); CHK(hipDeviceSynchronize());
      float ms; hipEventElapsedTime(&ms, a, b);
      std::vector<unsigned long long> h(c.blocks);
      CHK(hipMemcpy(h.data(), cyc, c.blocks * 8, hipMemcpyDeviceToHost));
      double avg = 0; for (auto v : h) avg += v; avg /= c.blocks;
      printf("  %-24s wall %.3f ms  memtime/iter-step %.2f (100MHz ticks? raw %.0f)  ns per chain step %.2f\n", names[kind], ms,
             avg / (iters * 16.0), avg, ms * 1e6 / (iters * 16.0));
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    lds_chain<<<c.blocks, c.threads>>>(out, cyc, iters); CHK(hipDeviceSynchronize());
    hipEventRecord(a); lds_chain<<<c.blocks, c.threads>>>(out, cyc, iters); hipEventRecord(b); CHK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("  %-24s wall %.3f ms  ns per dependent LDS read %.2f\n", "ds_read dep chain", ms, ms * 1e6 / (iters * 16.0));
  }
  return 0;
}
