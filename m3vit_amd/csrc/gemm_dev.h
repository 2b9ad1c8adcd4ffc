// Definitions shared by the NT GEMM kernels (gemm.hip: 128 x 128 tiles; gemm_big.hip: 256 x 256 tiles for long
// contractions): the device-side argument block, the grouped tile -> expert map and the epilogue kinds.
#pragma once
#include "common.h"

namespace m3 {

constexpr int BM = 128, BN = 128, ROWB = 128;  // ROWB: bytes of K per row per step
constexpr int GEMM_THREADS = 256;

struct GemmDev {
  const char *A; int64_t lda_b;                 // byte strides
  const int32_t *a_row_idx; int32_t a_row_div; int32_t a_row_sh;   // a_row_sh: log2(a_row_div) if a power of two, else -1
  const char *B; int64_t ldb_b; int64_t b_group_b;
  char *C; int64_t ldc; int32_t c_f32;
  const int32_t *c_row_idx;
  const float *bias;
  char *pre_out; int64_t ld_pre;
  const char *gpre; int64_t ld_gpre;
  const float *residual; int64_t ld_res;
  const float *row_scale; int32_t row_scale_div;   // value *= row_scale[srow / div] in front of the residual add,
  const int32_t *row_scale_idx;                    // srow = row_scale_idx ? row_scale_idx[m] : crow
  int32_t act;
  int64_t M; int32_t N; int32_t K;
  int32_t G;
  const int32_t *group_offsets;
  const int32_t *tile_starts;
  int32_t n_tiles;
  int32_t m_band;                  // row tiles per band of the tile order (tile_of): 1 = row-tile major
  int32_t m_tiles_max;
  int32_t vec8;                                 // N and all leading dims multiples of 8: staged epilogue
};

// Grouped call: which (group, first row, end row) owns row tile mt, and how many workgroups are live.  G <= 64: ONE
// vector load of the tile prefix (lane l holds tile_starts[l + 1]) + a ballot instead of a chain of up to G dependent
// scalar loads in front of every tile (the expert GEMMs run ~800 row tiles x 3 column tiles per launch).
struct TileOwner { int g; int64_t m_begin, m_end; };
__device__ __forceinline__ int grouped_live_tiles(const int32_t *tile_starts, int G, int lane, int &ts_lane) {
  if (G <= 64) {
    ts_lane = lane < G ? tile_starts[lane + 1] : 0x7fffffff;
    return __builtin_amdgcn_readfirstlane(__shfl(ts_lane, G - 1, 64));
  }
  ts_lane = 0;
  return tile_starts[G];
}
__device__ __forceinline__ TileOwner grouped_tile_owner(const int32_t *tile_starts, const int32_t *group_offsets, int G,
                                                        int mt, int lane, int ts_lane) {
  TileOwner o;
  int g = 0, t0;
  if (G <= 64) {
    // groups whose END prefix is <= mt lie wholly before the tile (the prefix is monotone; the last group never counts)
    g = __popcll(__ballot(lane < G - 1 && ts_lane <= mt));
    t0 = g ? __shfl(ts_lane, g - 1, 64) : 0;
  } else {
    while (g + 1 < G && tile_starts[g + 1] <= mt) ++g;
    t0 = tile_starts[g];
  }
  // (everything here is wave-uniform: say so, or the compiler carries the tile bounds in vector registers)
  g = __builtin_amdgcn_readfirstlane(g);
  t0 = __builtin_amdgcn_readfirstlane(t0);
  o.g = g;
  o.m_begin = (int64_t)__builtin_amdgcn_readfirstlane(group_offsets[g]) + (int64_t)(mt - t0) * 128;
  o.m_end = __builtin_amdgcn_readfirstlane(group_offsets[g + 1]);
  return o;
}

__device__ __forceinline__ int dma_swz(int row) { return (row >> 1) & 7; }

// EPI: the epilogue's kind as a template constant (see gemm_nt_dma_kernel in gemm.hip).
//   PLAIN  C = acc (+ bias), optional scatter                      qkv, every plain input gradient, expert FC2 forward
//   GELU   pre_out = acc + bias ; C = GELU(pre_out)                fc1 / expert FC1 forward
//   GPRE   C = acc * GELU'(gpre)                                   fc2 / expert FC2 input gradient
//   RES    C(fp32) = acc (+ bias) + residual                       proj, fc2 forward
enum { DMA_EPI_ANY = 0, DMA_EPI_GPRE = 1, DMA_EPI_RES = 2, DMA_EPI_PLAIN = 3, DMA_EPI_GELU = 4 };

// gemm_big.hip: 256 x 256 tiles, eight waves, LDS-DMA ring.  Returns false when the call is not one it takes.
bool gemm_big_eligible(const GemmDev &d, int dtype_size_bytes, bool force);
int launch_gemm_big(const GemmDev &d, int dtype, int epi, hipStream_t s);
// gemm_ws.hip (EXPERIMENTAL builds): 0 launched, 1 not a call it takes, < 0 error
int launch_gemm_ws(const GemmDev &d, const m3_gemm_args *a, int64_t mt, int ws_mode, hipStream_t s);


// Logical tile id -> (row tile, column tile).  Row-tile major (band 1): the n_tiles column tiles of a row tile are neighbours
// (one XCD, one moment: the A rows come from HBM once).  When a column-tile's weight panel set does not fit the XCD's L2
// (the ViT-Base N = 2304 / 3072 launches: 18-24 panels of 196 KB), that order re-fetches every panel for every row tile; in
// bands of m_band row tiles - column tile major inside a band - the 32 workgroups an XCD runs at a time cover m_band row
// tiles x 32 / m_band panels, so a panel is fetched once per band and the band's A rows stay resident while its column tiles
// go by.  live_m: live row tiles of the launch.
__device__ __forceinline__ void tile_of(int t, int n_tiles, int m_band, int live_m, int &mt, int &nt) {
  if (m_band <= 1) { mt = t / n_tiles; nt = t - mt * n_tiles; return; }
  const int per = m_band * n_tiles;
  const int band = t / per, r = t - band * per;
  const int rows = min(m_band, live_m - band * m_band);
  nt = r / rows;
  mt = band * m_band + (r - nt * rows);
}
}  // namespace m3
