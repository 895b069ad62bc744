// Dependent-chain latency of single VALU opcodes on gfx950, one wave per SIMD.  Every chain is ONE asm block of eight
// dependent instructions (hipcc puts an s_nop after each separate asm statement whose result is read next, which is what
// an earlier version of this file measured), so the figures are what the hardware does with back-to-back dependent issue.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(x) x x x x x x x x
#define KERNEL(name, nins, body)                                                          \
  __global__ void name(unsigned long long *cyc, unsigned *out, int iters, unsigned a, unsigned b) {   \
    unsigned x = a + threadIdx.x, y = b | 1u, z = b + 3, t0v = 0, t1v = 0, t2v = 0;       \
    unsigned long long v = ((unsigned long long)z << 32) | x;                             \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                 \
    for (int i = 0; i < iters; i++) { asm volatile(body : "+v"(x), "+v"(y), "+v"(z), "+v"(t0v), "+v"(t1v), "+v"(t2v), "+v"(v) : : "vcc"); }  \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                 \
    out[threadIdx.x] = x + t0v + t1v + t2v + (unsigned)v + y;                             \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                      \
  }                                                                                       \
  static const int name##_n = nins;
KERNEL(k_add, 8, R8("v_add_u32 %0, %0, %1\n"))
KERNEL(k_add_indep2, 8, "v_add_u32 %0, %0, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %3, %3, %1\n")
KERNEL(k_mad24, 8, R8("v_mad_u32_u24 %0, %0, %1, %2\n"))
KERNEL(k_mul24, 8, R8("v_mul_u32_u24 %0, %0, %1\n"))
KERNEL(k_bfe, 8, R8("v_bfe_i32 %0, %0, 1, 20\n"))
KERNEL(k_bfi, 8, R8("v_bfi_b32 %0, %1, %0, %2\n"))
KERNEL(k_bitop3, 8, R8("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x48\n"))
KERNEL(k_ffbh, 8, R8("v_ffbh_u32 %0, %0\n"))
KERNEL(k_ashr, 8, R8("v_ashrrev_i32 %0, 1, %0\n"))
KERNEL(k_add_sdwa, 8, R8("v_add_u32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"))
KERNEL(k_pk_mad, 8, R8("v_pk_mad_u16 %0, %0, %1, %2\n s_nop 0\n"))
KERNEL(k_pk_lshr, 8, R8("v_pk_lshrrev_b16 %0, %1, %0\n s_nop 0\n"))
KERNEL(k_lshl64, 8, R8("v_lshlrev_b64 %6, %1, %6\n"))
KERNEL(k_dpp_nop2, 8, R8("v_mov_b32_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"))
KERNEL(k_dpp_fill2, 8, R8("v_mov_b32_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n"))
KERNEL(k_cmp_cnd, 8, R8("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n"))
// the dependent part of one decode bin (state broadcast -> LPS width -> compare -> bin -> state update), 4 bins per block
#define DEC_BIN \
  "v_mov_b32_dpp %3, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
  "v_add_u32_sdwa %3, %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
  "v_bfe_i32 %4, %3, 15, 1\n" \
  "v_lshrrev_b32 %3, 10, %3\n" \
  "v_bitop3_b32 %3, %4, 31, %3 bitop3:0x48\n" \
  "v_mad_u32_u24 %3, %3, %1, %2\n" \
  "v_lshrrev_b32 %3, 1, %3\n" \
  "v_sub_u32 %5, %1, %3\n" \
  "v_mul_u32_u24 %5, %5, %2\n" \
  "v_sub_u32 %5, %0, %5\n" \
  "v_ashrrev_i32 %5, 31, %5\n" \
  "v_bitop3_b32 %4, %5, 1, %4 bitop3:0x84\n" \
  "v_pk_mad_u16 %4, %2, %4, %0\n" \
  "v_bfi_b32 %1, %5, %1, %3\n" \
  "v_cndmask_b32 %0, %0, %4, vcc\n" \
  "v_add_u32 %1, %1, %2\n" \
  "v_add_u32 %2, %1, %2\n"
KERNEL(k_dec_bin, 68, DEC_BIN DEC_BIN DEC_BIN DEC_BIN)
int main() {
  unsigned long long *cyc; unsigned *out; (void)hipMalloc(&cyc, 8 * 2048); (void)hipMalloc(&out, 4096);
  const int iters = 4000;
#define RUN(name) { hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0); name<<<1024, 64>>>(cyc, out, iters, 5, 7); (void)hipEventRecord(e1); (void)hipDeviceSynchronize(); \
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("%-16s %6.2f memtime ticks per instruction, %6.2f ns per instruction (block of %d)\n", #name, (double)h / (iters * (double)name##_n), ms * 1e6 / (iters * (double)name##_n), name##_n); }
  RUN(k_add) RUN(k_add) RUN(k_add_indep2) RUN(k_mad24) RUN(k_mul24) RUN(k_bfe) RUN(k_bfi) RUN(k_bitop3) RUN(k_ffbh) RUN(k_ashr) RUN(k_add_sdwa)
  RUN(k_pk_mad) RUN(k_pk_lshr) RUN(k_lshl64) RUN(k_dpp_nop2) RUN(k_dpp_fill2) RUN(k_cmp_cnd) RUN(k_dec_bin)
  return 0;
}
