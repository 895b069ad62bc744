// Where the three waves of the v5 encoder spend their cycles (workgroup 0, s_memtime around the parts of the loop).
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCABAC_V5_PROFILE -Iinclude -Ientropy_coding_amd/csrc tools/ubench_v5.hip entropy_coding_amd/csrc/cabac_synth.cpp -o tools/ubench_v5
#include "../entropy_coding_amd/csrc/cabac_kernels_v4.hip"
#include <cstdio>
#include <vector>
using namespace cabac;

int main() {
  const uint32_t n_sub = 4096, n_bins = 16384;
  std::vector<uint16_t> rec((size_t)n_sub * n_bins);
  for (uint32_t s = 0; s < n_sub; s++) cabac_synth_records(0xC4, s, n_bins, 750, rec.data() + (size_t)s * n_bins);
  std::vector<cabac_substream_desc> desc(n_sub);
  const uint32_t cap = 16384 * 2 / 8 + 4096;
  for (uint32_t s = 0; s < n_sub; s++) {
    desc[s].rec_offset = (uint64_t)s * n_bins; desc[s].byte_offset = (uint64_t)s * cap; desc[s].n_records = n_bins;
    desc[s].byte_capacity = cap; desc[s].qp = 32; desc[s].init_id = 2 | CABAC_SUB_FINISH | CABAC_SUB_ALIGN_RBSP;
  }
  cabac_substream_desc *d_desc; uint16_t *d_rec; uint8_t *d_bytes; cabac_substream_result *d_res;
  (void)hipMalloc(&d_desc, desc.size() * sizeof(desc[0])); (void)hipMalloc(&d_rec, rec.size() * 2);
  (void)hipMalloc(&d_bytes, (size_t)n_sub * cap); (void)hipMalloc(&d_res, n_sub * sizeof(cabac_substream_result));
  (void)hipMemcpy(d_desc, desc.data(), desc.size() * sizeof(desc[0]), hipMemcpyHostToDevice);
  (void)hipMemcpy(d_rec, rec.data(), rec.size() * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; rep++) {
    unsigned long long zero[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_v5_prof), zero, sizeof(zero));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(encode_kernel_v5<1>, dim3(n_sub / 4), dim3(192), 0, 0, n_sub, d_desc, d_rec, d_bytes, d_res);
    (void)hipEventRecord(b); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long p[16];
    (void)hipMemcpyFromSymbol(p, HIP_SYMBOL(g_v5_prof), sizeof(p));
    const double g = n_bins / 16.0;
    {
      // the decoder on the encoder's output: cycles per step of wave 0 of workgroup 0
      uint8_t *d_bins; (void)hipMalloc(&d_bins, (size_t)n_sub * n_bins);
      std::vector<cabac_substream_result> res(n_sub);
      (void)hipMemcpy(res.data(), d_res, n_sub * sizeof(res[0]), hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(decode_kernel_v4<4>, dim3(n_sub / 16), dim3(256), 0, 0, n_sub, d_desc, d_rec, d_bytes, d_bins, d_res);
      (void)hipDeviceSynchronize();
      unsigned long long zero2[16] = {};
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_v5_prof), zero2, sizeof(zero2));
      hipEvent_t c, d; (void)hipEventCreate(&c); (void)hipEventCreate(&d);
      (void)hipEventRecord(c);
      hipLaunchKernelGGL(decode_kernel_v4<4>, dim3(n_sub / 16), dim3(256), 0, 0, n_sub, d_desc, d_rec, d_bytes, d_bins, d_res);
      (void)hipEventRecord(d); (void)hipDeviceSynchronize();
      float dms; (void)hipEventElapsedTime(&dms, c, d);
      unsigned long long q[16];
      (void)hipMemcpyFromSymbol(q, HIP_SYMBOL(g_v5_prof), sizeof(q));
      const double gg = n_bins / 16.0;
      printf("decode %.3f ms; cycles per 16-bin step, wave 0: record wait %.0f prologue %.0f chain+refill %.0f epilogue %.0f\n", dms,
             q[8] / gg, q[9] / gg, q[10] / gg, q[11] / gg);
      (void)hipFree(d_bins);
    }
    printf("kernel %.3f ms; cycles per 16-bin step, workgroup 0: context wave phase(a) %.0f barrier %.0f | chain wave chain %.0f barrier %.0f | output wave emit %.0f list %.0f\n",
           ms, p[2] / g, p[3] / g, p[4] / g, p[5] / g, p[0] / g, p[1] / g);
    {  // the four-wave encoder (v6), same probes
      unsigned long long z[16] = {};
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_v5_prof), z, sizeof(z));
      hipEvent_t c, d; (void)hipEventCreate(&c); (void)hipEventCreate(&d);
      (void)hipEventRecord(c);
      hipLaunchKernelGGL((encode_kernel_v6<4, 1>), dim3(n_sub / 16), dim3(1024), 0, 0, n_sub, d_desc, d_rec, d_bytes, d_res);
      (void)hipEventRecord(d); (void)hipDeviceSynchronize();
      float vms; (void)hipEventElapsedTime(&vms, c, d);
      unsigned long long q[16];
      (void)hipMemcpyFromSymbol(q, HIP_SYMBOL(g_v5_prof), sizeof(q));
      printf("v6 kernel %.3f ms; cycles per step, workgroup 0 (sum over its 4 units): context phase(a) %.0f barrier %.0f | chain %.0f post %.0f barrier %.0f | low %.0f barrier %.0f | emit %.0f barrier %.0f\n",
             vms, q[2] / g, q[3] / g, q[4] / g, q[6] / g, q[5] / g, q[1] / g, q[7] / g, q[0] / g, q[12] / g);
    }
    for (int uu = 0; uu < 2; uu++) {  // the lane-serial encoder (v7): U = 4 (16 substreams per workgroup) and U = 1
      unsigned long long z[16] = {};
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_v5_prof), z, sizeof(z));
      hipEvent_t c, d; (void)hipEventCreate(&c); (void)hipEventCreate(&d);
      (void)hipEventRecord(c);
      if (uu == 0) hipLaunchKernelGGL(encode_kernel_v7<4>, dim3(n_sub / 16), dim3(512), 0, 0, n_sub, d_desc, d_rec, d_bytes, d_res);
      else hipLaunchKernelGGL(encode_kernel_v7<1>, dim3(n_sub / 4), dim3(256), 0, 0, n_sub, d_desc, d_rec, d_bytes, d_res);
      (void)hipEventRecord(d); (void)hipDeviceSynchronize();
      float vms; (void)hipEventElapsedTime(&vms, c, d);
      unsigned long long q[16];
      (void)hipMemcpyFromSymbol(q, HIP_SYMBOL(g_v5_prof), sizeof(q));
      printf("v7<%d> kernel %.3f ms; ticks per step, workgroup 0: context busy %.0f barrier %.0f | chain busy %.0f barrier %.0f | low busy %.0f barrier %.0f | output busy %.0f barrier %.0f\n",
             uu == 0 ? 4 : 1, vms, q[0] / g, q[1] / g, q[2] / g, q[3] / g, q[4] / g, q[5] / g, q[6] / g, q[7] / g);
    }
  }
  return 0;
}
