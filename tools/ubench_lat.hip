// Dependent-chain latency of single VALU opcodes on gfx950 (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(name, body)                                                              \
  __global__ void name(unsigned long long *cyc, unsigned *out, int iters, unsigned a, unsigned b) {   \
    unsigned x = a + threadIdx.x, y = b | 1u, z = b + 3;                                \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                               \
    for (int i = 0; i < iters; i++) { REP16(body) }                                     \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                               \
    out[threadIdx.x] = x;                                                               \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                    \
  }
KERNEL(k_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));)
KERNEL(k_mad24, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));)
KERNEL(k_bfe, asm volatile("v_bfe_u32 %0, %0, 1, 20" : "+v"(x));)
KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(y), "v"(z));)
KERNEL(k_lshl, asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x) : "v"(y));)
KERNEL(k_ffbh, asm volatile("v_ffbh_u32 %0, %0" : "+v"(x));)
KERNEL(k_movdpp, asm volatile("v_mov_b32_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));)
KERNEL(k_add_after_dpp, asm volatile("v_mov_b32_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32 %0, %1, %0" : "+v"(x), "+v"(z));)
KERNEL(k_xor, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y));)
KERNEL(k_sub, asm volatile("v_sub_u32 %0, %1, %0" : "+v"(x) : "v"(y));)
KERNEL(k_add3, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));)
KERNEL(k_cmp_cnd, asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(y), "v"(z) : "vcc");)
KERNEL(k_lshl64, { unsigned long long v = ((unsigned long long)z << 32) | x; asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(v)); x = (unsigned)v; })
KERNEL(k_indep4, asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(x), "+v"(z) : "v"(y));)
int main() {
  unsigned long long *cyc; unsigned *out; hipMalloc(&cyc, 8 * 2048); hipMalloc(&out, 4096);
  const int iters = 4000;
#define RUN(name, nins) { name<<<1024, 64>>>(cyc, out, iters, 5, 7); hipDeviceSynchronize(); unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("%-18s %.2f cycles per instruction (chain of %d per rep)\n", #name, (double)h / (iters * 16.0 * nins), nins); }
  RUN(k_add, 1) RUN(k_mad24, 1) RUN(k_bfe, 1) RUN(k_bfi, 1) RUN(k_lshl, 1) RUN(k_ffbh, 1) RUN(k_movdpp, 1) RUN(k_add_after_dpp, 2)
  RUN(k_xor, 1) RUN(k_sub, 1) RUN(k_add3, 1) RUN(k_cmp_cnd, 2) RUN(k_lshl64, 1) RUN(k_indep4, 2)
  return 0;
}
