// Do scalar and vector instruction streams of different waves of a SIMD overlap?  16 waves per CU (4 per SIMD) run
// a 64-instruction dependent chain per iteration: all vector, all scalar, or every other wave scalar.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R16(x) x x x x x x x x x x x x x x x x
#define VBLK R16("v_mad_u32_u24 %0, %0, %2, %3\n v_xor_b32 %0, %0, %2\n v_sub_u32 %0, %0, %3\n v_lshrrev_b32 %0, 1, %0\n")
#define SBLK R16("s_mul_i32 %1, %1, %4\n s_xor_b32 %1, %1, %4\n s_sub_u32 %1, %1, %5\n s_lshr_b32 %1, %1, 1\n")
__global__ __launch_bounds__(256) void k_mix(unsigned *out, int iters, int mode, unsigned a, unsigned b) {
  unsigned v = a + threadIdx.x, s = a, y = b | 1u, z = b + 3;
  const unsigned wave = threadIdx.x >> 6;
  const bool scalar = mode == 1 || (mode == 2 && (wave & 1u)) || (mode == 3 && (blockIdx.x & 1u));
  if (scalar) {
    for (int i = 0; i < iters; i++) asm volatile(SBLK : "+v"(v), "+s"(s) : "v"(y), "v"(z), "s"(a | 1u), "s"(b) : "scc");
  } else {
    for (int i = 0; i < iters; i++) asm volatile(VBLK : "+v"(v), "+s"(s) : "v"(y), "v"(z), "s"(a | 1u), "s"(b));
  }
  out[blockIdx.x * 256 + threadIdx.x] = v + s;
}
int main() {
  unsigned *out; (void)hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 2000;
  const char *names[] = {"all waves vector", "all waves scalar", "odd waves scalar (same SIMD pairs differ)", "odd workgroups scalar"};
  for (int wgs = 256; wgs <= 1024; wgs *= 4)
    for (int mode = 0; mode < 4; mode++) {
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      k_mix<<<wgs, 256>>>(out, 10, mode, 5, 7);
      (void)hipEventRecord(e0);
      k_mix<<<wgs, 256>>>(out, iters, mode, 5, 7);
      (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%4d workgroups of 4 waves (%d waves per SIMD), %-44s %.3f ms = %.2f ns per instruction per wave\n", wgs, wgs / 256, names[mode], ms, ms * 1e6 / (iters * 64.0));
    }
  return 0;
}
