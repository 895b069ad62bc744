// Issue cost of single vector opcodes for ONE wave per SIMD: 64 independent instructions per block (four rotating
// destination registers), so neither dependency latency nor the loop matters much.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R16(x) x x x x x x x x x x x x x x x x
#define KERNEL(name, body)                                                                     \
  __global__ __launch_bounds__(64) void name(unsigned *out, int iters, unsigned a, unsigned b) {  \
    unsigned x0 = a + threadIdx.x, x1 = a * 3 + threadIdx.x, x2 = a * 5 + 1, x3 = a * 7 + 2, y = b | 1u, z = b + 3;  \
    unsigned long long w0 = x0, w1 = x1;                                                       \
    for (int i = 0; i < iters; i++) asm volatile(R16(body) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(w0), "+v"(w1) : "v"(y), "v"(z) : "vcc"); \
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + (unsigned)w0 + (unsigned)w1;     \
  }
KERNEL(k_add, "v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %6\n")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %6, %7\n v_mad_u32_u24 %1, %1, %6, %7\n v_mad_u32_u24 %2, %2, %6, %7\n v_mad_u32_u24 %3, %3, %6, %7\n")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %6\n v_mul_u32_u24 %1, %1, %6\n v_mul_u32_u24 %2, %2, %6\n v_mul_u32_u24 %3, %3, %6\n")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %6, %7 bitop3:0x48\n v_bitop3_b32 %1, %1, %6, %7 bitop3:0x48\n v_bitop3_b32 %2, %2, %6, %7 bitop3:0x48\n v_bitop3_b32 %3, %3, %6, %7 bitop3:0x48\n")
KERNEL(k_bfi, "v_bfi_b32 %0, %6, %0, %7\n v_bfi_b32 %1, %6, %1, %7\n v_bfi_b32 %2, %6, %2, %7\n v_bfi_b32 %3, %6, %3, %7\n")
KERNEL(k_ffbh, "v_ffbh_u32 %0, %0\n v_ffbh_u32 %1, %1\n v_ffbh_u32 %2, %2\n v_ffbh_u32 %3, %3\n")
KERNEL(k_sdwa, "v_add_u32_sdwa %0, %0, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %7 row_newbcast:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
KERNEL(k_pk_mad, "v_pk_mad_u16 %0, %0, %6, %7\n v_pk_mad_u16 %1, %1, %6, %7\n v_pk_mad_u16 %2, %2, %6, %7\n v_pk_mad_u16 %3, %3, %6, %7\n")
KERNEL(k_pk_lshr, "v_pk_lshrrev_b16 %0, %6, %0\n v_pk_lshrrev_b16 %1, %6, %1\n v_pk_lshrrev_b16 %2, %6, %2\n v_pk_lshrrev_b16 %3, %6, %3\n")
KERNEL(k_lshl64, "v_lshlrev_b64 %4, %6, %4\n v_lshlrev_b64 %5, %6, %5\n v_lshlrev_b64 %4, %7, %4\n v_lshlrev_b64 %5, %7, %5\n")
KERNEL(k_cmp_vcc, "v_cmp_eq_u32 vcc, %0, %6\n v_cmp_eq_u32 vcc, %1, %6\n v_cmp_eq_u32 vcc, %2, %6\n v_cmp_eq_u32 vcc, %3, %6\n")
KERNEL(k_cmp_cnd, "v_cmp_eq_u32 vcc, %0, %6\n v_cndmask_b32 %1, %1, %7, vcc\n v_cmp_eq_u32 vcc, %2, %6\n v_cndmask_b32 %3, %3, %7, vcc\n")
KERNEL(k_cmp_far_cnd, "v_cmp_eq_u32 vcc, %0, %6\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %6\n v_cndmask_b32 %1, %1, %7, vcc\n")
KERNEL(k_min, "v_min_u32 %0, %0, %6\n v_min_u32 %1, %1, %6\n v_min_u32 %2, %2, %6\n v_min_u32 %3, %3, %6\n")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1\n v_ashrrev_i32 %2, 31, %2\n v_ashrrev_i32 %3, 31, %3\n")
int main() {
  unsigned *out; (void)hipMalloc(&out, 1024 * 64 * 4);
  const int iters = 3000;
#define RUN(name) { hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); name<<<1024, 64>>>(out, 10, 3, 5); (void)hipEventRecord(e0); \
    name<<<1024, 64>>>(out, iters, 3, 5); (void)hipEventRecord(e1); (void)hipDeviceSynchronize(); float ms; (void)hipEventElapsedTime(&ms, e0, e1); \
    printf("%-16s %.2f ns per instruction\n", #name, ms * 1e6 / (iters * 64.0)); }
  RUN(k_add) RUN(k_mad24) RUN(k_mul24) RUN(k_bitop3) RUN(k_bfi) RUN(k_ffbh) RUN(k_sdwa) RUN(k_dpp) RUN(k_pk_mad) RUN(k_pk_lshr) RUN(k_lshl64)
  RUN(k_cmp_vcc) RUN(k_cmp_cnd) RUN(k_cmp_far_cnd) RUN(k_min) RUN(k_ashr)
  return 0;
}
