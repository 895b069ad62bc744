// Micro-benchmark: VALU<->SALU crossing latency and uniform-branch cost on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// KIND 0: readlane -> 1 salu -> forward (v_cmp+v_cndmask) -> readlane ...      (the v1 state loop skeleton)
// KIND 1: same with 10 dependent SALU ops between
// KIND 2: pure SALU chain of 12 ops with a data-dependent (unpredictable) uniform branch per step
// KIND 3: pure SALU chain of 12 ops, branch-free selects
template <int KIND>
__global__ void k(unsigned *out, unsigned long long *cyc, int iters, unsigned seed) {
  unsigned v = threadIdx.x * 7 + seed;   // per-lane value
  unsigned key = threadIdx.x & 7;
  unsigned s = seed;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (KIND == 0 || KIND == 1) {
        unsigned x = __builtin_amdgcn_readlane(v, j * 5 + 1);
        x = x * 3 + s;
        if (KIND == 1) {
#pragma unroll
          for (int q = 0; q < 5; q++) { x = (x >> 3) ^ (x + 77); }
        }
        s = x;
        v = (key == (x & 7)) ? x : v;
      } else if (KIND == 2) {
        unsigned x = s;
#pragma unroll
        for (int q = 0; q < 3; q++) { x = (x >> 3) ^ (x + 77); }
        if (x & 16) { x = x * 5 + 1; x ^= x >> 7; } else { x = x + 3; x ^= x << 3; }
        s = x;
      } else {
        unsigned x = s;
#pragma unroll
        for (int q = 0; q < 3; q++) { x = (x >> 3) ^ (x + 77); }
        unsigned a = x * 5 + 1; a ^= a >> 7;
        unsigned b = x + 3; b ^= b << 3;
        s = (x & 16) ? a : b;
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = v + s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  unsigned *out; unsigned long long *cyc;
  CHK(hipMalloc(&out, 1 << 22)); CHK(hipMalloc(&cyc, 1 << 16));
  const int iters = 4000;
  const char *names[] = {"readlane+2salu+fwd", "readlane+17salu+fwd", "12 salu + unpredictable branch", "12+ salu selects"};
  int grids[] = {1, 1024, 4096};
  for (int g : grids) {
    printf("--- %d single-wave workgroups\n", g);
    for (int kind = 0; kind < 4; kind++) {
      auto launch = [&]() {
        switch (kind) {
          case 0: k<0><<<g, 64>>>(out, cyc, iters, 5); break;
          case 1: k<1><<<g, 64>>>(out, cyc, iters, 5); break;
          case 2: k<2><<<g, 64>>>(out, cyc, iters, 5); break;
          case 3: k<3><<<g, 64>>>(out, cyc, iters, 5); break;
        }
      };
      launch(); CHK(hipDeviceSynchronize());
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a); launch(); hipEventRecord(b); CHK(hipDeviceSynchronize());
      float ms; hipEventElapsedTime(&ms, a, b);
      unsigned long long h; CHK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
      printf("  %-32s  %.1f cycles per step (memtime), wall %.3f ms = %.2f ns/step\n", names[kind], (double)h / (iters * 8.0), ms,
             ms * 1e6 / (iters * 8.0));
    }
  }
  return 0;
}
