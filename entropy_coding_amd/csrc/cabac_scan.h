// Scan tables and small ROM functions shared by the residual binariser (cabac_residual.hip) and the residual parser
// (cabac_residual_parse.hip): the grouped up-right diagonal scan (rom.cpp:72-92, :148-260), g_groupIdx / g_minInGroup
// (rom.cpp:18-25) and g_goRiceParsCoeff (rom.cpp:27-29).
#ifndef CABAC_SCAN_H
#define CABAC_SCAN_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cabac {
namespace {

// k-th position of the up-right diagonal scan of a bw x bh rectangle, packed x | y << 4.
struct DiagLut {
  uint8_t grid[4][4][64];   // [log2 groups per row][log2 groups per column][group index]
  uint8_t in_cg[5][5][16];  // [log2 group width][log2 group height][position in group]
  // the five template neighbours (x+1,y) (x+2,y) (x+1,y+1) (x,y+1) (x,y+2) of a position as positions of the SAME group,
  // 5 bits each, 31 = outside the group: for blocks that are one group (the template never leaves the block's lanes)
  uint32_t nbr[5][5][16];
};

constexpr void fill_diag(uint8_t *out, int bw, int bh) {
  int k = 0;
  for (int d = 0; d <= bw + bh - 2; d++) {
    const int y_hi = d < bh - 1 ? d : bh - 1;
    const int y_lo = d - (bw - 1) > 0 ? d - (bw - 1) : 0;
    for (int y = y_hi; y >= y_lo; y--, k++) out[k] = (uint8_t)((d - y) | (y << 4));
  }
}

constexpr DiagLut make_lut() {
  DiagLut t{};
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) fill_diag(t.grid[a][b], 1 << a, 1 << b);
  for (int a = 0; a < 5; a++)
    for (int b = 0; b + a < 5; b++) fill_diag(t.in_cg[a][b], 1 << a, 1 << b);
  for (int a = 0; a < 5; a++)
    for (int b = 0; b + a < 5; b++) {
      const int n = 1 << (a + b);
      for (int k = 0; k < n; k++) {
        const int x = t.in_cg[a][b][k] & 15, y = t.in_cg[a][b][k] >> 4;
        const int nx[5] = {x + 1, x + 2, x + 1, x, x}, ny[5] = {y, y, y + 1, y + 1, y + 2};
        uint32_t word = 0;
        for (int q = 0; q < 5; q++) {
          uint32_t at = 31;
          if (nx[q] < (1 << a) && ny[q] < (1 << b))
            for (int m = 0; m < n; m++)
              if ((t.in_cg[a][b][m] & 15) == nx[q] && (t.in_cg[a][b][m] >> 4) == ny[q]) at = (uint32_t)m;
          word |= at << (5 * q);
        }
        t.nbr[a][b][k] = word;
      }
    }
  return t;
}

__constant__ DiagLut c_diag = make_lut();

// g_goRiceParsCoeff (rom.cpp:27-29) as thresholds: 0 below 7, 1 below 14, 2 below 28, else 3
__device__ __forceinline__ uint32_t rice_of(int sum_abs, int base_level) {
  int v = sum_abs - 5 * base_level;
  v = v < 0 ? 0 : v;
  return (uint32_t)(v >= 7) + (uint32_t)(v >= 14) + (uint32_t)(v >= 28);
}

__device__ __forceinline__ uint32_t group_idx(uint32_t p) {  // g_groupIdx, rom.cpp:21-25
  const uint32_t fl = 31u - (uint32_t)__builtin_clz(p | 1u);
  return p < 4u ? p : 2u * fl + ((p >> (fl - 1u)) & 1u);
}
__device__ __forceinline__ uint32_t min_in_group(uint32_t g) {  // g_minInGroup, rom.cpp:18-19
  return g < 4u ? g : (2u + (g & 1u)) << ((g >> 1) - 1u);
}

}  // namespace
}  // namespace cabac
#endif
