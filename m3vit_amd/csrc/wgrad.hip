// Weight-gradient GEMM ("TN": contraction over token rows) for gfx950, grouped or dense:
//   dW[g][n,k] = sum_{m in group g} dC[crow(m), n] * A[arow(m), k]
//
// Replaces fastmoe's linear_backward weight/bias part behind FMoELinear
// (models/moe/ckpt/custom_moe_layer.py:32-33) and the nn.Linear weight grads of the
// attention block / dense Mlp (models/moe/ckpt/vision_transformer_moe.py:255-261,295-313);
// the reference's torch-only twin is ParallelLinear.backward
// (models/moe/parallel_experts.py:51-82: d_weight = input^T grad, d_bias = sum grad).
//
// Both operands are row-major with the contraction index as the ROW, so fragments must
// be read transposed out of LDS:
//   f16: ds_read_b64_tr_b16 (4 rows x 16 cols per 16-lane group) on a [32][128]+pad
//        image, row stride 288 B -> the 8 rows one 32-lane half touches land on
//        8 disjoint 8-bank windows (conflict free);
//   f32: ds_read_b32 (each lane one element) on a 528 B stride image (conflict free).
// 128(n) x 128(k) output tile per 256-thread workgroup (2x2 waves of 64x64), 32 rows per
// barrier step, double-buffered register staging.  Rows are split `splits` ways;
// each split writes an fp32 slab, m3_wgrad_reduce adds the slabs in a fixed order
// (deterministic, unlike atomics).
#include "common.h"
#include <type_traits>

namespace m3 {

constexpr int WG_T = 128;        // tile edge (n and k)
constexpr int WG_ROWS = 64;      // granule of the row splits (= the largest per-dtype step below)
constexpr int WG_THREADS = 256;

struct WgradDev {
  const char *dC; int64_t lddc_b; const int32_t *c_row_idx;
  int32_t c_row_div; const float *c_row_scale;   // dC row of slot m = c_row_scale[c_row_idx[m]] * dC[c_row_idx[m] / c_row_div]
  int32_t a_row_sh, c_row_sh;                    // log2 of the divisors when they are powers of two, else -1
  const char *A; int64_t lda_b; const int32_t *a_row_idx; int32_t a_row_div;
  int64_t M; int32_t N; int32_t K; int32_t G;
  const int32_t *group_offsets;
  int32_t splits;
  float *ws;
  float *bias_ws;                  // optional [splits][G][N]: column sums of dC (bias grads), fused
  int32_t tiles_k;
  int32_t chunk_rows;              // > 0: balanced grouped mode - a work unit is `chunk_rows` rows of ONE group
  // The slab reduction of the PREVIOUS weight-gradient call of the stream, done by this launch's leading blocks
  // (m3_wgrad_args.prev): rd_blocks > 0 switches it on; layouts as m3_wgrad_reduce / m3_wgrad_reduce_grouped take them
  int32_t rd_blocks, rd_zslices;   // reduce blocks (flattened x, group) and the grid z slices they occupy
  int32_t rd_nbx, rd_nbw;          // blocks per group (weight + bias part), of those for the weight elements
  int32_t rd_cols;                 // 16-byte columns per reduce block: 256, or 64 with four threads per column (dense, many slabs)
  const float *rd_ws; int32_t rd_splits; int64_t rd_e4;
  const int32_t *rd_off; int32_t rd_G, rd_chunk;
  float *rd_dW; int32_t rd_beta;
  const float *rd_bws; int64_t rd_b4; float *rd_db; int32_t rd_beta_db;
  // direct mode (splits == 1, no balanced units: every (group, tile) belongs to exactly ONE workgroup): the result tiles are
  // added into dW [G][N][K] (the column sums into db [G][N]) by the kernel itself - no slabs, no reduction
  float *direct_dW; float *direct_db; int32_t direct_beta, direct_beta_db;
  int32_t lpt;                     // grouped, one part per group: the groups are taken longest first (wgrad_lpt_group)
};

// the result of a workgroup: one 128 x 128 fp32 tile (lane holds k = kb + 4 lg + r, n = nb + li) to its slab, or - direct
// mode - read-add-written into dW
__device__ __forceinline__ void wgrad_store_tile(const WgradDev &p, const f32x4 (&acc)[4][4], int64_t slab_id, int g, int n0, int k0,
                                                 int wr, int wc, int li, int lg) {
  float *out = p.direct_dW ? p.direct_dW + (int64_t)g * p.N * p.K : p.ws + slab_id * (int64_t)p.N * p.K;
  const bool add = p.direct_dW && p.direct_beta;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = n0 + wc * 64 + ni * 16 + li;
    if (n >= p.N) continue;
    f32x4 old[4];
    if (add) {
#pragma unroll
      for (int ki = 0; ki < 4; ++ki) {
        const int k = k0 + wr * 64 + ki * 16 + 4 * lg;
        old[ki] = k < p.K ? *(const f32x4 *)(out + (int64_t)n * p.K + k) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int ki = 0; ki < 4; ++ki) {
      const int k = k0 + wr * 64 + ki * 16 + 4 * lg;
      if (k >= p.K) continue;
      *(f32x4 *)(out + (int64_t)n * p.K + k) = add ? acc[ki][ni] + old[ki] : acc[ki][ni];
    }
  }
}
__device__ __forceinline__ void wgrad_store_bias(const WgradDev &p, float v, int64_t slab_id, int g, int n) {
  if (p.direct_db) {
    float *d = p.direct_db + (int64_t)g * p.N + n;
    *d = p.direct_beta_db ? *d + v : v;
  } else {
    p.bias_ws[slab_id * p.N + n] = v;
  }
}

// balanced grouped mode: units are dealt to the groups in order, n_g = ceil(rows_g / chunk) each, a group's rows
// divided evenly over its units (a hot expert gets proportionally more units; the slab of unit u is ws[u]).
// One lane per group (G <= 64): ONE load of the offsets per wave and a shuffle scan instead of G dependent loads.
// Returns this lane's group's (rows, n, exclusive prefix of n).
__device__ __forceinline__ void wgrad_unit_scan(const int32_t *off, int G, int chunk, int lane, int &rows, int &n, int &first) {
  rows = lane < G ? off[lane + 1] - off[lane] : 0;
  n = (rows + chunk - 1) / chunk;
  int incl = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  first = incl - n;
}

// logical id (after the XCD remap over the LIVE workgroups only: tiles x sum of n_g - the grid is sized for the upper
// bound, and remapping over the whole grid would park the surplus ids, i.e. no work at all, on the last XCDs)
// -> (tile, unit, group, first row, end row); false: this workgroup is surplus
__device__ __forceinline__ bool wgrad_unit(const int32_t *off, int G, int chunk, int lin, int tiles, int lane, int &tile,
                                           int &u, int &g, int64_t &r0, int64_t &r1) {
  int rows, n, first;
  wgrad_unit_scan(off, G, chunk, lane, rows, n, first);
  const int units = __shfl(first + n, 63, 64);
  if (lin >= units * tiles) return false;
  const int log_id = xcd_remap(lin, units * tiles);
  tile = log_id % tiles;
  u = log_id / tiles;
  const unsigned long long m = __ballot(u >= first && u < first + n);
  g = __ffsll((long long)m) - 1;
  rows = __shfl(rows, g, 64); n = __shfl(n, g, 64); first = __shfl(first, g, 64);
  const int per = ((rows + n - 1) / n + WG_ROWS - 1) / WG_ROWS * WG_ROWS;      // 32-row granules; per <= chunk
  r0 = (int64_t)off[g] + (int64_t)(u - first) * per;
  r1 = r0 + per < off[g + 1] ? r0 + per : off[g + 1];
  return true;
}

// Groups of unequal size, one part per (group, tile) workgroup (the experts' weight gradients in direct mode): a workgroup's
// life is proportional to its expert's rows, and dealt out in expert order the hot experts' workgroups can all start in the
// last round (configs[3] / [4] with the learned router: +30..40 % over the same launch with uniform routing).  Longest first:
// unit u of the launch (in dispatch order: the XCD remap hands each XCD one contiguous eighth of the units) takes the
// group of size rank 8 * (u mod G/8) + u / (G/8) - every XCD gets every eighth-largest group, largest first.  One lane per
// group (G <= 64, G a multiple of 8; else the identity), ranks by 64 shuffles; which workgroup computes a tile never
// changes the tile's value.
__device__ __forceinline__ int wgrad_lpt_group(const int32_t *off, int G, int u, int lane) {
  if (G > 64 || (G & 7)) return u;
  const int cnt = lane < G ? off[lane + 1] - off[lane] : -1;
  int rank = 0;
  for (int j = 0; j < G; ++j) {
    const int cj = __shfl(cnt, j, 64);
    rank += (cj > cnt || (cj == cnt && j < lane)) ? 1 : 0;
  }
  const int per = G >> 3;
  const int want = 8 * (u % per) + u / per;
  const unsigned long long m = __ballot(lane < G && rank == want);
  return __ffsll((long long)m) - 1;
}

template <typename T> struct WgLds;
// ROWS: contraction rows per barrier step.  (fp16 with 64 rows - 32 MFMAs per wave per barrier instead of 16 - was
// built in round 2: the second pair of staging registers spills, 256 VGPRs + 172 B scratch, and the step got 1-2 %
// slower; -DM3_WGRAD_F16_ROWS=64 rebuilds it.)
#ifndef M3_WGRAD_F16_ROWS
#define M3_WGRAD_F16_ROWS 32
#endif
#ifndef M3_WGRAD_WIDE_ROWS
#define M3_WGRAD_WIDE_ROWS 32        // contraction rows per barrier step of the wide-tile kernel
#endif
template <> struct WgLds<half_t> { static constexpr int STRIDE = 288; static constexpr int ROWS = M3_WGRAD_F16_ROWS; };
template <> struct WgLds<bf16_t> { static constexpr int STRIDE = 288; static constexpr int ROWS = M3_WGRAD_F16_ROWS; };
template <> struct WgLds<float> { static constexpr int STRIDE = 528; static constexpr int ROWS = 32; };

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

// fragment for 16 columns starting at byte column offset colb (col*sizeof(T)) of the
// LDS image `base`, contraction rows rb .. rb+KC-1
template <typename T>
__device__ __forceinline__ typename Mma<T>::frag read_tr_frag(const char *base, int rb, int col, int li, int lg);

template <>
__device__ __forceinline__ f16x8 read_tr_frag<half_t>(const char *base, int rb, int col, int li, int lg) {
  // lane li of group lg supplies the address of (row 4*lg + (li>>2) [+16], cols col + 4*(li&3))
  const char *p0 = base + (rb + 4 * lg + (li >> 2)) * WgLds<half_t>::STRIDE + (col + 4 * (li & 3)) * 2;
  const char *p1 = p0 + 16 * WgLds<half_t>::STRIDE;
  fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t *)p0);
  fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t *)p1);
  f16x8 f;
  f[0] = (half_t)lo[0]; f[1] = (half_t)lo[1]; f[2] = (half_t)lo[2]; f[3] = (half_t)lo[3];
  f[4] = (half_t)hi[0]; f[5] = (half_t)hi[1]; f[6] = (half_t)hi[2]; f[7] = (half_t)hi[3];
  return f;
}

template <>
__device__ __forceinline__ bf16x8 read_tr_frag<bf16_t>(const char *base, int rb, int col, int li, int lg) {
  // the 16-bit transposed read does not look at the element format: same addressing as f16, bits re-labelled
  return __builtin_bit_cast(bf16x8, read_tr_frag<half_t>(base, rb, col, li, lg));
}

template <>
__device__ __forceinline__ f32x4 read_tr_frag<float>(const char *base, int rb, int col, int li, int lg) {
  const char *p = base + (rb + 4 * lg) * WgLds<float>::STRIDE + (col + li) * 4;
  f32x4 f;
  f[0] = *(const float *)(p);
  f[1] = *(const float *)(p + WgLds<float>::STRIDE);
  f[2] = *(const float *)(p + 2 * WgLds<float>::STRIDE);
  f[3] = *(const float *)(p + 3 * WgLds<float>::STRIDE);
  return f;
}

__device__ __forceinline__ void wgrad_reduce_block(int64_t blk, int tid, const float *ws, int splits, int64_t elems4, float *dW,
                                                   int beta, int nb_w, const float *bias_ws, int64_t belems4, float *db, int beta_db,
                                                   int cols);
__device__ __forceinline__ void wgrad_reduce_grouped_block(int64_t blk, int g, int tid, const float *ws, const int32_t *off, int G,
                                                           int chunk, int64_t elems4, float *dW, int beta, int nb_w,
                                                           const float *bias_ws, int64_t belems4, float *db, int beta_db);
// The previous call's slab reduction riding in front of a weight-gradient launch: the grid's first rd_zslices z slices
// are reduce blocks (dispatched first; a few microseconds of streaming), the rest is the launch proper with its z index
// shifted down.  Returns true for a reduce block (which is then done).  What it replaces: one extra launch per
// weight-gradient GEMM (110 per step) whose ~8 us were mostly launch boundary and ramp.
__device__ __forceinline__ bool wgrad_ride_along(const WgradDev &p, int tid, int &bz, int &gz) {
  bz = blockIdx.z; gz = gridDim.z;
  if (p.rd_blocks <= 0) return false;
  if (bz < p.rd_zslices) {
    const int rid = blockIdx.x + (int)gridDim.x * (blockIdx.y + (int)gridDim.y * bz);
    if (rid < p.rd_blocks) {
      const int g = rid / p.rd_nbx, bx = rid - g * p.rd_nbx;
      if (p.rd_chunk)
        wgrad_reduce_grouped_block(bx, g, tid, p.rd_ws, p.rd_off, p.rd_G, p.rd_chunk, p.rd_e4, p.rd_dW, p.rd_beta, p.rd_nbw,
                                   p.rd_bws, p.rd_b4, p.rd_db, p.rd_beta_db);
      else
        wgrad_reduce_block(bx, tid, p.rd_ws, p.rd_splits, p.rd_e4, p.rd_dW, p.rd_beta, p.rd_nbw, p.rd_bws, p.rd_b4, p.rd_db,
                           p.rd_beta_db, p.rd_cols);
    }
    return true;
  }
  bz -= p.rd_zslices; gz -= p.rd_zslices;
  return false;
}

// a 16-byte chunk of T times a per-row factor (the gate score of a routed row: the combine's backward d y = score * d out
// applied where the row enters the LDS image, so that the scaled [T*k, D] copy never exists in memory)
template <typename T> __device__ __forceinline__ u32x4 scale_chunk(u32x4 v, float s);
template <> __device__ __forceinline__ u32x4 scale_chunk<half_t>(u32x4 v, float s) {
  // fp32 product, ONE rounding - the value the dgrad GEMM's fp32 epilogue forms for the same row (a fp16 multiply would round
  // the score to 11 bits first: the two consumers of d y = score * d out would then see different rows)
  f16x8 f = __builtin_bit_cast(f16x8, v);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (half_t)((float)f[j] * s);
  return __builtin_bit_cast(u32x4, f);
}
template <> __device__ __forceinline__ u32x4 scale_chunk<bf16_t>(u32x4 v, float s) {
  bf16x8 f = __builtin_bit_cast(bf16x8, v);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (bf16_t)((float)f[j] * s);
  return __builtin_bit_cast(u32x4, f);
}
template <> __device__ __forceinline__ u32x4 scale_chunk<float>(u32x4 v, float s) {
  return __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, v) * s);
}

template <typename T, bool GC, bool GA, bool SC = false>
__global__ __launch_bounds__(WG_THREADS, 2) void wgrad_tn_kernel(const WgradDev p) {
  typedef Mma<T> MM;
  typedef typename MM::frag frag;
  constexpr int ES = (int)sizeof(T);
  constexpr int EPC = 16 / ES;
  constexpr int STRIDE = WgLds<T>::STRIDE;
  constexpr int ROWS = WgLds<T>::ROWS;                      // contraction rows per step
  constexpr int CPR = WG_T * ES / 16;                       // 16-byte chunks per tile row
  constexpr int NLD = ROWS * CPR / WG_THREADS;           // chunks per thread per operand
  constexpr int RSTEP = WG_THREADS / CPR;                   // tile rows between a thread's chunks
  constexpr int OPB = ROWS * STRIDE;                     // bytes per operand image
  constexpr int KCH = ROWS / MM::KC;                     // fragment chunks per step

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware order: all tiles of one (group, split) read the SAME rows of dC and A (each byte is
  // needed by tiles_k resp. tiles_n workgroups), so they are given consecutive logical ids, which the
  // remap places on one XCD: the re-reads hit that XCD's L2 instead of the fabric.
  const int tiles = gridDim.x;
  int bz, gz;
  if (wgrad_ride_along(p, tid, bz, gz)) return;
  const int lin = blockIdx.x + tiles * (blockIdx.y + gridDim.y * bz);
  int tile, gs, g, sp, nst;
  int64_t r0, r1, s_begin;
  if (p.chunk_rows) {                          // gs = work unit; its slab is ws[gs]
    if (!wgrad_unit(p.group_offsets, p.G, p.chunk_rows, lin, tiles, lane, tile, gs, g, r0, r1)) return;
    sp = gs; s_begin = 0;
    nst = (int)((r1 - r0 + ROWS - 1) / ROWS);
  } else {
    const int log_id = xcd_remap(lin, tiles * gridDim.y * gz);
    tile = log_id % tiles; gs = log_id / tiles;
    g = gs % (int)gridDim.y; sp = gs / (int)gridDim.y;
    if (p.group_offsets && p.lpt) g = wgrad_lpt_group(p.group_offsets, p.G, g, lane);
    if (p.group_offsets) { r0 = p.group_offsets[g]; r1 = p.group_offsets[g + 1]; }
    else { r0 = 0; r1 = p.M; }
    const int64_t nsteps_all = (r1 - r0 + ROWS - 1) / ROWS;
    const int64_t per = (nsteps_all + p.splits - 1) / p.splits;
    s_begin = (int64_t)sp * per;
    int64_t s_end = s_begin + per;
    if (s_end > nsteps_all) s_end = nsteps_all;
    nst = (int)(s_end > s_begin ? s_end - s_begin : 0);
  }
  const int tn = tile / p.tiles_k, tk = tile - tn * p.tiles_k;
  const int n0 = tn * WG_T, k0 = tk * WG_T;
  // slab / bias slab of this workgroup
  const int64_t slab_id = p.chunk_rows ? (int64_t)sp : (int64_t)sp * p.G + g;

  f32x4 acc[4][4];   // [ki][ni]: MFMA rows = k, cols = n
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging assignment: chunk i of this thread is tile row (tid / CPR) + RSTEP*i, 16-byte column c.
  // Columns beyond N / K are clamped (they only feed outputs that are never stored); rows beyond the
  // end of the group are clamped for the load and ZEROED at the LDS store (they would otherwise add
  // into every output).  Loads are unconditional so hipcc's counted vmcnt waits stay exact.
  const int srow = tid / CPR, c = tid - srow * CPR;
  int ncol = n0 + c * EPC, kcol = k0 + c * EPC;
  if (ncol > p.N - EPC) ncol = p.N - EPC;
  if (kcol > p.K - EPC) kcol = p.K - EPC;
  const char *c_base = p.dC + (int64_t)ncol * ES;
  const char *a_base = p.A + (int64_t)kcol * ES;
  const int st_off = srow * STRIDE + c * 16;                // + i*RSTEP*STRIDE
  const int64_t rbase = r0 + s_begin * ROWS + srow;      // row of chunk 0 in local step 0

  auto row_of = [&](int step, int i) -> int64_t {           // clamped slot row
    int64_t m = rbase + (int64_t)step * ROWS + i * RSTEP;
    return m < r1 ? m : r1 - 1;
  };
  auto load_index = [&](int step, int32_t(&ic)[NLD], int32_t(&ia)[NLD]) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int64_t m = row_of(step, i);
      if (GC) ic[i] = p.c_row_idx[m];
      if (GA) ia[i] = p.a_row_idx[m];
    }
  };
  // SC: the per-row factor travels with the row's data (loaded next to it, applied at the LDS store)
  auto load_global = [&](int step, const int32_t(&ic)[NLD], const int32_t(&ia)[NLD], u32x4(&rc)[NLD], u32x4(&ra)[NLD],
                         float(&rs)[NLD]) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int64_t m = row_of(step, i);
      const int64_t cr = GC ? (int64_t)div_by(ic[i], p.c_row_div, p.c_row_sh) : m;
      const int64_t ar = GA ? (int64_t)div_by(ia[i], p.a_row_div, p.a_row_sh) : m;
      rc[i] = *(const u32x4 *)(c_base + cr * p.lddc_b);
      ra[i] = *(const u32x4 *)(a_base + ar * p.lda_b);
      if (SC) rs[i] = p.c_row_scale[ic[i]];
    }
  };
  auto store_lds = [&](int buf, int step, const u32x4(&rc)[NLD], const u32x4(&ra)[NLD], const float(&rs)[NLD]) {
    char *base = smem + buf * (2 * OPB) + st_off;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const bool ok = rbase + (int64_t)step * ROWS + i * RSTEP < r1;
      const u32x4 cv = SC ? scale_chunk<T>(rc[i], rs[i]) : rc[i];
      *(u32x4 *)(base + i * RSTEP * STRIDE) = ok ? cv : u32x4{0u, 0u, 0u, 0u};
      *(u32x4 *)(base + i * RSTEP * STRIDE + OPB) = ok ? ra[i] : u32x4{0u, 0u, 0u, 0u};
    }
  };
  // Bias gradient fused as one extra MFMA row: with an all-ones A operand the product is the column
  // sum of dC over the contraction rows.  Done once per n-tile (k-tile 0, waves wr == 0).
  const bool do_bias = (p.bias_ws || p.direct_db) && tk == 0 && wr == 0;
  f32x4 acc_b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc_b[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  frag ones;
#pragma unroll
  for (int j = 0; j < MM::EPL; ++j) ones[j] = (T)1.0f;

  auto compute = [&](int buf) {
    const char *sC = smem + buf * (2 * OPB), *sA = sC + OPB;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) {
      frag fk[4], fn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fk[i] = read_tr_frag<T>(sA, kc * MM::KC, wr * 64 + i * 16, li, lg);
        fn[i] = read_tr_frag<T>(sC, kc * MM::KC, wc * 64 + i * 16, li, lg);
      }
#pragma unroll
      for (int ki = 0; ki < 4; ++ki)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ki][ni] = MM::mma(fk[ki], fn[ni], acc[ki][ni]);
      if (do_bias) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc_b[ni] = MM::mma(ones, fn[ni], acc_b[ni]);
      }
    }
  };

  if (nst > 0) {
    // prefetch distance 2 for the data (two register sets), 3 for the gather indices; steps past the
    // end re-load the last step (clamped) so that nothing in the steady-state loop is conditional.
    const int last = nst - 1;
    auto cl = [&](int s_) { return s_ < last ? s_ : last; };
    u32x4 rc0[NLD], ra0[NLD], rc1[NLD], ra1[NLD];
    float sc0[NLD], sc1[NLD];
    int32_t ic[NLD], ia[NLD];
    load_index(0, ic, ia);
    load_global(0, ic, ia, rc1, ra1, sc1);
    load_index(cl(1), ic, ia);
    load_global(cl(1), ic, ia, rc0, ra0, sc0);
    load_index(cl(2), ic, ia);
    store_lds(0, 0, rc1, ra1, sc1);
    __syncthreads();
    // entry of even local step t: buf0 = tile t, set0 = tile t+1, (ic, ia) = indices of tile t+2
    int t = 0;
    for (; t + 3 < nst; t += 2) {
      load_global(t + 2, ic, ia, rc1, ra1, sc1);
      load_index(cl(t + 3), ic, ia);
      __builtin_amdgcn_sched_barrier(0);
      compute(0);
      store_lds(1, t + 1, rc0, ra0, sc0);
      __syncthreads();
      load_global(t + 3, ic, ia, rc0, ra0, sc0);
      load_index(cl(t + 4), ic, ia);
      __builtin_amdgcn_sched_barrier(0);
      compute(1);
      store_lds(0, t + 2, rc1, ra1, sc1);
      __syncthreads();
    }
    const int rem = nst - t;
    if (rem == 3) {
      load_global(t + 2, ic, ia, rc1, ra1, sc1);
      compute(0);
      store_lds(1, t + 1, rc0, ra0, sc0);
      __syncthreads();
      compute(1);
      store_lds(0, t + 2, rc1, ra1, sc1);
      __syncthreads();
      compute(0);
    } else if (rem == 2) {
      compute(0);
      store_lds(1, t + 1, rc0, ra0, sc0);
      __syncthreads();
      compute(1);
    } else {
      compute(0);
    }
  }

  if (do_bias && lg == 0) {                       // every row of the ones-product is the column sum: take row 0
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wc * 64 + ni * 16 + li;
      if (n < p.N) wgrad_store_bias(p, acc_b[ni][0], slab_id, g, n);
    }
  }
  wgrad_store_tile(p, acc, slab_id, g, n0, k0, wr, wc, li, lg);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (16-bit operands; round 5): the same 128 x 128 tile, wave layout, work units, slabs and ride-along
// reduce, but the operands go global -> LDS directly (global_load_lds_dwordx4), as in gemm_nt_dma_kernel:
//   - no staging registers, no ds_write pass, no second register set to spill: <= 128 VGPRs and ONE 32 KiB buffer (64
//     contraction rows x 128 columns of each operand), so FOUR workgroups share a CU instead of two and their DMA / MFMA
//     phases interleave; 32 MFMAs per wave between barriers instead of 16;
//   - the image rows are unpadded (a wave's DMA instruction fills 1 KiB = four 256-byte rows, lane-linear), so the
//     transposed reads are made conflict free by an XOR swizzle instead of the 288-byte stride: the 32-byte granule g of row
//     r sits at granule g ^ (r & 7) - the eight rows a half-wave's ds_read_b64_tr_b16 touches land on eight different
//     8-bank windows - applied to the per-lane SOURCE address on the way in and to the fragment addresses on the way out
//     (four address registers per operand, one per 16-column tile of the wave: an XOR does not fold into an offset field);
//   - rows past the end of a unit must contribute nothing: their source is a zero row in device memory (a DMA cannot be
//     masked into zeros at the LDS store the way the register-staged kernel does it);
//   - gather indices of step t + 1 are loaded under step t's MFMAs.
// Not for c_row_scale (the per-row factor is applied in registers on the way into LDS): those launches keep
// wgrad_tn_kernel<.., SC = true>.
__device__ __attribute__((aligned(256))) const uint32_t g_wgrad_zero_row[64] = {0};      // 256 bytes of zeros: the source of rows past a unit's end

template <typename T, bool GC, bool GA, bool SC = false>
__global__ __launch_bounds__(WG_THREADS, SC ? 3 : 4) void wgrad_dma_kernel(const WgradDev p) {
  typedef Mma<T> MM;
  typedef typename MM::frag frag;
  constexpr int ES = (int)sizeof(T);                        // 2 (f16 / bf16) or 4 (f32)
  constexpr int ROWS = 128 / ES;                            // contraction rows per step: 64 (16-bit) / 32 (f32)
  constexpr int RS = WG_T * ES;                             // image row: 128 columns = 256 / 512 bytes
  constexpr int OPB = ROWS * RS;                            // one operand image: 16 KiB
  constexpr int RPP = 1024 / RS;                            // image rows per DMA piece (1 KiB): 4 / 2
  constexpr int LPR = 64 / RPP;                             // lanes (= 16-byte chunks) per image row: 16 / 32
  constexpr int EPC = 16 / ES;                              // elements per 16-byte chunk
  constexpr int NPC = ROWS / RPP / 4;                       // DMA pieces per wave per operand per step: 4
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [dC image | A image]
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;

  const int tiles = gridDim.x;
  int bz, gz;
  if (wgrad_ride_along(p, tid, bz, gz)) return;
  const int lin = blockIdx.x + tiles * (blockIdx.y + gridDim.y * bz);
  int tile, gs, g, sp, nst;
  int64_t r0, r1, s_begin;
  if (p.chunk_rows) {                          // gs = work unit; its slab is ws[gs]
    if (!wgrad_unit(p.group_offsets, p.G, p.chunk_rows, lin, tiles, lane, tile, gs, g, r0, r1)) return;
    sp = gs; s_begin = 0;
    nst = (int)((r1 - r0 + ROWS - 1) / ROWS);
  } else {
    const int log_id = xcd_remap(lin, tiles * gridDim.y * gz);
    tile = log_id % tiles; gs = log_id / tiles;
    g = gs % (int)gridDim.y; sp = gs / (int)gridDim.y;
    if (p.group_offsets && p.lpt) g = wgrad_lpt_group(p.group_offsets, p.G, g, lane);
    if (p.group_offsets) { r0 = p.group_offsets[g]; r1 = p.group_offsets[g + 1]; }
    else { r0 = 0; r1 = p.M; }
    const int64_t nsteps_all = (r1 - r0 + ROWS - 1) / ROWS;
    const int64_t per = (nsteps_all + p.splits - 1) / p.splits;
    s_begin = (int64_t)sp * per;
    int64_t s_end = s_begin + per;
    if (s_end > nsteps_all) s_end = nsteps_all;
    nst = (int)(s_end > s_begin ? s_end - s_begin : 0);
  }
  const int tn = tile / p.tiles_k, tk = tile - tn * p.tiles_k;
  const int n0 = tn * WG_T, k0 = tk * WG_T;
  const int64_t slab_id = p.chunk_rows ? (int64_t)sp : (int64_t)sp * p.G + g;

  f32x4 acc[4][4];   // [ki][ni]: MFMA rows = k, cols = n
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DMA assignment: wave w, piece j fills image rows (4 w + j) * 4 .. + 3; lane l -> row + (l >> 4), physical 16-byte chunk
  // l & 15, which holds logical chunk (l & 15) ^ ((row & 7) << 1).  (row & 7) only depends on the parity of j and on the
  // lane, so a lane has two column offsets per operand.  Columns beyond N / K are clamped (outputs never stored).
  // Every step but a unit's last is 64 whole rows: its sources are (wave-uniform operand base) + (32-bit per-lane offset:
  // the host checks the 4 GiB reach), one address register per piece and no 64-bit arithmetic.  The last step (rows past the
  // end read a zero row, which lives in another buffer) takes 64-bit addresses picked by a bit mask - a `ok ? a : b` between
  // the two becomes a branch around each load, a basic block per piece with its own vmcnt(0).
  // fp32: a piece is two 512-byte rows, lane l -> row + (l >> 5), chunk l & 31, which holds logical chunk
  // (l & 31) ^ (((row >> 2) & 1) << 2): rows r and r + 4 - what a half-wave's ds_read_b32 touches - sit in different
  // 64-byte halves of the 128-byte bank window; ((row >> 2) & 1) = (j >> 1) & 1 for the wave's piece j.
  const int prow = lane / LPR;                               // row of this lane inside a piece
  uint32_t colC[2], colA[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = ES == 2 ? ((lane & 15) ^ ((((4 * q + prow) & 7)) << 1)) : ((lane & 31) ^ (q << 2));
    int nc = n0 + c * EPC, kc = k0 + c * EPC;
    if (nc > p.N - EPC) nc = p.N - EPC;
    if (kc > p.K - EPC) kc = p.K - EPC;
    colC[q] = (uint32_t)nc * ES; colA[q] = (uint32_t)kc * ES;
  }
  auto colq = [](int j) { return ES == 2 ? (j & 1) : ((j >> 1) & 1); };      // which of the two column offsets piece j takes
  const int rbase = (int)(r0 + s_begin * ROWS) + (NPC * wave) * RPP + prow;    // row of piece 0 in local step 0 (M < 2^31)
  const int rlast = (int)r1 - 1;
  const uint32_t ldc = (uint32_t)p.lddc_b, lda = (uint32_t)p.lda_b;
  int32_t ic[NPC], ia[NPC];
  auto load_index = [&](int step) {
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int m = min(rbase + step * ROWS + RPP * j, rlast);
      if (GC) ic[j] = p.c_row_idx[m];
      if (GA) ia[j] = p.a_row_idx[m];
    }
  };
  char *const dma_dst = smem + (NPC * wave) * 1024;
  // SC (the combine's backward without d y: dC row of slot m = c_row_scale[c_row_idx[m]] * d out[c_row_idx[m] / div]): the
  // step's ROWS per-row factors go into a small LDS table behind the images - ONE 4-byte LDS-DMA of wave 0, lane l fetching
  // row l's factor through the index it loaded a step ahead - and multiply the dC fragments on their way into the MFMAs
  // (v_pk_mul_f16 with the factor rounded to fp16: one more 2^-11 rounding than the register-staged kernel's fp32 product,
  // inside the fp16 bound).  The zero rows of a unit's last step make their factors irrelevant.
  float *const s_sc = (float *)(smem + 2 * OPB);
  static_assert(!SC || (GC && !std::is_same<T, bf16_t>::value), "per-row factors: gathered dC rows, fp16 or fp32");
  int32_t sc_ix = 0;
  auto load_sc_index = [&](int step) {
    if (SC && wave == 0) sc_ix = p.c_row_idx[min((int)(r0 + s_begin * ROWS) + step * ROWS + (lane & (ROWS - 1)), rlast)];
  };
  auto dma_scores = [&]() {
    if (SC && wave == 0 && lane < ROWS)
      __builtin_amdgcn_global_load_lds((glb_void *)(p.c_row_scale + sc_ix), (lds_void *)s_sc, 4, 0, 0);
  };
  auto dma_full = [&](int step) {
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int m = rbase + step * ROWS + RPP * j;
      // (gather divisors are powers of two here: the host sends anything else to the register-staged kernel)
      const uint32_t cr = GC ? (uint32_t)(ic[j] >> p.c_row_sh) : (uint32_t)m;
      const uint32_t ar = GA ? (uint32_t)(ia[j] >> p.a_row_sh) : (uint32_t)m;
      __builtin_amdgcn_global_load_lds((glb_void *)(p.dC + (cr * ldc + colC[colq(j)])), (lds_void *)(dma_dst + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void *)(p.A + (ar * lda + colA[colq(j)])), (lds_void *)(dma_dst + j * 1024 + OPB), 16, 0, 0);
    }
  };
  auto dma_tail = [&](int step) {
    const uint64_t zero_row = (uint64_t)(uintptr_t)g_wgrad_zero_row + (lane & 15) * 16;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int m = rbase + step * ROWS + RPP * j;
      const uint64_t ok = m <= rlast ? ~(uint64_t)0 : (uint64_t)0;
      const uint32_t cr = GC ? (uint32_t)(ic[j] >> p.c_row_sh) : (uint32_t)min(m, rlast);
      const uint32_t ar = GA ? (uint32_t)(ia[j] >> p.a_row_sh) : (uint32_t)min(m, rlast);
      const uint64_t sc = (((uint64_t)(uintptr_t)p.dC + (cr * ldc + colC[colq(j)])) & ok) | (zero_row & ~ok);
      const uint64_t sa = (((uint64_t)(uintptr_t)p.A + (ar * lda + colA[colq(j)])) & ok) | (zero_row & ~ok);
      __builtin_amdgcn_global_load_lds((glb_void *)(uintptr_t)sc, (lds_void *)(dma_dst + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void *)(uintptr_t)sa, (lds_void *)(dma_dst + j * 1024 + OPB), 16, 0, 0);
    }
  };

  // fragment addresses.  16-bit: lane (li, lg) supplies row 4 lg + (li >> 2) (+ 16 for the second half of a fragment, + 32 for the
  // second chunk of a step), columns col + 4 (li & 3) .. + 3 of the 16-column tile starting at col (ds_read_b64_tr_b16).
  // fp32: lane (li, lg) reads the elements (row 4 lg + r, column col + li), r = 0..3, one ds_read_b32 each (+ 16 rows for
  // the second chunk of a step); the swizzle swaps the 16-column tiles i and i ^ 1 for odd lg, so a lane has one base for
  // even and one for odd tiles per operand and the tile index stays an offset.
  int adK[4], adN[4];
  if constexpr (ES == 2) {
    const int s3 = (4 * (lg & 1) + (li >> 2)) & 7;
    const int frow = (4 * lg + (li >> 2)) * RS + 8 * (li & 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ck = (wr * 8 + 2 * i + ((li & 3) >> 1)) ^ (s3 << 1);
      const int cn = (wc * 8 + 2 * i + ((li & 3) >> 1)) ^ (s3 << 1);
      adK[i] = OPB + frow + ck * 16;
      adN[i] = frow + cn * 16;
    }
  } else {
    const int frow = 4 * lg * RS + (li & 3) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ck = (wr * 16 + 4 * i + (li >> 2)) ^ ((lg & 1) << 2);
      const int cn = (wc * 16 + 4 * i + (li >> 2)) ^ ((lg & 1) << 2);
      adK[i] = OPB + frow + ck * 16;
      adN[i] = frow + cn * 16;
    }
  }
  typedef __attribute__((address_space(3))) fp16x4_t lds_h4;
  auto read_frag = [&](int ad, int rb) -> frag {
    if constexpr (ES == 2) {
      const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4 *)(smem + ad + rb * RS));
      const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4 *)(smem + ad + (rb + 16) * RS));
      f16x8 f;
      f[0] = (half_t)lo[0]; f[1] = (half_t)lo[1]; f[2] = (half_t)lo[2]; f[3] = (half_t)lo[3];
      f[4] = (half_t)hi[0]; f[5] = (half_t)hi[1]; f[6] = (half_t)hi[2]; f[7] = (half_t)hi[3];
      return __builtin_bit_cast(frag, f);
    } else {
      const char *q = smem + ad + rb * RS;
      f32x4 f;
      f[0] = *(const float *)(q);
      f[1] = *(const float *)(q + RS);
      f[2] = *(const float *)(q + 2 * RS);
      f[3] = *(const float *)(q + 3 * RS);
      return f;
    }
  };

  // Bias gradient (column sums of dC over the contraction rows; once per n-tile: k-tile 0, waves wr == 0): a lane's dC
  // fragment holds 8 (fp32: 4) contraction rows of ITS column, so four v_dot2 with a pair of ones (fp32: three adds) sum
  // them - one fp32 register per 16-column tile instead of the register-staged kernel's extra MFMA row (16 accumulator
  // registers + a ones fragment: at 128 registers they spilled); the four lane groups' partial sums meet in two shuffles.
  const bool do_bias = (p.bias_ws || p.direct_db) && tk == 0 && wr == 0;
  float acc_b[4] = {0.f, 0.f, 0.f, 0.f};
  typedef T t2 __attribute__((ext_vector_type(2)));
  auto colsum8 = [&](const frag &f, float a) -> float {
    if constexpr (ES == 4) {
      return a + ((f[0] + f[1]) + (f[2] + f[3]));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const t2 pr = t2{f[2 * j], f[2 * j + 1]};
        if constexpr (std::is_same<T, half_t>::value)
          a = __builtin_amdgcn_fdot2(pr, t2{(T)1, (T)1}, a, false);
        else if constexpr (std::is_same<T, bf16_t>::value)
          a = __builtin_amdgcn_fdot2_f32_bf16(pr, t2{(T)1.f, (T)1.f}, a, false);
      }
      return a;
    }
  };

  auto compute = [&]() {
#pragma unroll
    for (int kc = 0; kc < ROWS / MM::KC; ++kc) {
      frag fk[4], fn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fk[i] = read_frag(adK[i], kc * MM::KC);
        fn[i] = read_frag(adN[i], kc * MM::KC);
      }
      if constexpr (SC) {
        if constexpr (ES == 4) {
          const f32x4 sv = *(const f32x4 *)(s_sc + kc * 16 + 4 * lg);
#pragma unroll
          for (int i = 0; i < 4; ++i) fn[i] *= sv;
        } else {
          const f32x4 s0 = *(const f32x4 *)(s_sc + kc * 32 + 4 * lg), s1 = *(const f32x4 *)(s_sc + kc * 32 + 16 + 4 * lg);
          const f16x8 sh = f16x8{(half_t)s0[0], (half_t)s0[1], (half_t)s0[2], (half_t)s0[3],
                                 (half_t)s1[0], (half_t)s1[1], (half_t)s1[2], (half_t)s1[3]};
#pragma unroll
          for (int i = 0; i < 4; ++i) fn[i] = __builtin_bit_cast(frag, __builtin_bit_cast(f16x8, fn[i]) * sh);
        }
      }
#pragma unroll
      for (int ki = 0; ki < 4; ++ki)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ki][ni] = MM::mma(fk[ki], fn[ni], acc[ki][ni]);
      if (do_bias) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc_b[ni] = colsum8(fn[ni], acc_b[ni]);
      }
    }
  };

  if (nst > 0) {
    if (GC || GA) load_index(0);
    load_sc_index(0);
    for (int t = 0; t + 1 < nst; ++t) {
      dma_full(t);
      dma_scores();
      if (GC || GA) load_index(t + 1);                       // (arrives under this step's MFMAs; the barrier's vmcnt(0) covers it)
      load_sc_index(t + 1);
      __syncthreads();          // vmcnt(0) + barrier: the step's rows have landed
      compute();
      __syncthreads();          // everyone has read them
    }
    dma_tail(nst - 1);
    dma_scores();
    __syncthreads();
    compute();
  }

  if (do_bias) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float v = acc_b[ni];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int n = n0 + wc * 64 + ni * 16 + li;
      if (lg == 0 && n < p.N) wgrad_store_bias(p, v, slab_id, g, n);
    }
  }
  wgrad_store_tile(p, acc, slab_id, g, n0, k0, wr, wc, li, lg);
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 tiles for the ViT-Base weights (16-bit; N and K multiples of 256: 768, 2304, 3072): the 128 x 128 kernels bring
// 64 FLOP per operand byte into LDS and, with every byte of dC / A needed by K / 128 resp. N / 128 workgroups, run at what
// that path delivers (configs[3]'s experts: 1.86 GB per launch, 5.3 TB/s, 340 TFLOP/s).  This tile doubles the FLOP per
// byte: eight waves (wave (wr, wc) owns 128 k x 64 n: 8 x 4 MFMA tiles, 128 accumulator registers), one workgroup per
// CU, 64 contraction rows per step in two LDS stages of [dC image | A image] (64 rows x 512 B each): the rows of step
// t + 1 are in flight (LDS-DMA) while step t's fragments are read (ds_read_b64_tr_b16, issued as inline assembly - the
// compiler would otherwise drain the DMA before every LDS read it knows of) and multiplied.  One barrier per step.
// Image geometry: a DMA piece (1 KiB) is two rows; lane l -> row + (l >> 5), physical 16-byte chunk l & 31, which holds
// logical chunk (l & 31) ^ ((row & 7) << 1): the same 8-row XOR as the 128-wide 16-bit image, so the transposed reads
// (16 rows x 32 bytes per instruction) meet the same banks as there.  Gathers, per-row factor, fused bias sums, slabs or
// direct accumulation: as wgrad_dma_kernel.  The previous call's reduction does not ride here (512 threads): own launch.
// Diagnostic build only (make CXXFLAGS+=-DM3_WGRAD_STAMPS, tools/wgrad_big_stamps.py): lane 0 of waves 0 and 4 of the first
// workgroups of wgrad_big_kernel record s_memtime per step - after the DMA issue, after the MFMAs, after the vmcnt wait, after
// the barrier.  No stamp executes in the shipped kernel.
#ifdef M3_WGRAD_STAMPS
constexpr int WSTAMP_WGS = 512, WSTAMP_N = 2 * (2 + 4 * 24);
__device__ unsigned long long g_wbig_stamps[WSTAMP_WGS][WSTAMP_N];
#define WB_STAMP(i)                                                                                                   \
  do {                                                                                                                \
    if ((threadIdx.x & 255) == 0 && blockIdx.x < WSTAMP_WGS && blockIdx.y == 0 && blockIdx.z == 0 && (i) < WSTAMP_N / 2)   \
      g_wbig_stamps[blockIdx.x][(threadIdx.x >> 8) * (WSTAMP_N / 2) + (i)] = __builtin_amdgcn_s_memtime();             \
  } while (0)
#else
#define WB_STAMP(i) do { } while (0)
#endif
constexpr int BG_T = 256, BG_THREADS = 512, BG_RS = 512;
#ifndef M3_WGRAD_BIG_ROWS
// What a 64-row step spends (in-kernel stamps, tools/wgrad_big_stamps.py, profiles/r05_wgrad_big_stamps.txt): ~800-1 550 cycles in
// which the waves sit in the ISSUE of their eight DMA instructions (a wave is held there until the CU's load path has taken
// them: the step's 64 KiB pass while nobody multiplies), ~1 800-2 300 of fragment reads + 64 MFMAs (1 024 of them MFMA), then
// the wait and the barrier: 4 250 in all, transfer and MFMA time adding up instead of overlapping.  Three re-arrangements were
// built and measured on the dense shapes, all within +-4 % of this one: a DMA instruction behind every eight MFMAs, waves 4-7
// sending theirs after multiplying instead of before, and the DMA issued in the shadow of the fragment reads.
// contraction rows per step: 64 (two stages, one step in flight ahead of the one multiplied) or 32 (four stages, three in
// flight).  Measured level to 3 % slower with 32 (profiles/r05_wgrad_big.txt): the step is not waiting for its DMA - what
// paces it is LDS traffic (48 transposed reads per wave and step next to the 64 KiB the DMA writes), as in the 128-wide kernels
#define M3_WGRAD_BIG_ROWS 64
#endif
constexpr int BG_ROWS = M3_WGRAD_BIG_ROWS;
constexpr int BG_NSTAGE = 128 / BG_ROWS;             // 128 KiB of operand stages either way
constexpr int BG_OPB = BG_ROWS * BG_RS;              // one operand image: 16 / 32 KiB
constexpr int BG_STAGE = 2 * BG_OPB;                 // [dC | A]
constexpr int BG_LDS = BG_NSTAGE * (BG_STAGE + 256); // the stages + every stage's per-row factors

template <typename T, bool GC, bool GA, bool SC = false>
__global__ __launch_bounds__(BG_THREADS, 1) void wgrad_big_kernel(const WgradDev p) {
  typedef Mma<T> MM;
  typedef typename MM::frag frag;
  static_assert(sizeof(T) == 2, "16-bit operands");
  static_assert(!SC || (GC && std::is_same<T, half_t>::value), "per-row factors: gathered dC rows, fp16");
  constexpr int ROWS = BG_ROWS, RS = BG_RS, NPC = ROWS / 16, NS = BG_NSTAGE, PD = NS - 1;      // PD: steps in flight ahead of the one multiplied
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;

  const int tiles = gridDim.x;
  const int lin = blockIdx.x + tiles * (blockIdx.y + gridDim.y * blockIdx.z);
  int tile, gs, g, sp, nst;
  int64_t r0, r1, s_begin;
  if (p.chunk_rows) {
    if (!wgrad_unit(p.group_offsets, p.G, p.chunk_rows, lin, tiles, lane, tile, gs, g, r0, r1)) return;
    sp = gs; s_begin = 0;
    nst = (int)((r1 - r0 + ROWS - 1) / ROWS);
  } else {
    const int log_id = xcd_remap(lin, tiles * gridDim.y * gridDim.z);
    tile = log_id % tiles; gs = log_id / tiles;
    g = gs % (int)gridDim.y; sp = gs / (int)gridDim.y;
    if (p.group_offsets && p.lpt) g = wgrad_lpt_group(p.group_offsets, p.G, g, lane);
    if (p.group_offsets) { r0 = p.group_offsets[g]; r1 = p.group_offsets[g + 1]; }
    else { r0 = 0; r1 = p.M; }
    const int64_t nsteps_all = (r1 - r0 + ROWS - 1) / ROWS;
    const int64_t per = (nsteps_all + p.splits - 1) / p.splits;
    s_begin = (int64_t)sp * per;
    int64_t s_end = s_begin + per;
    if (s_end > nsteps_all) s_end = nsteps_all;
    nst = (int)(s_end > s_begin ? s_end - s_begin : 0);
  }
  const int tn = tile / p.tiles_k, tk = tile - tn * p.tiles_k;
  const int n0 = tn * BG_T, k0 = tk * BG_T;
  const int64_t slab_id = p.chunk_rows ? (int64_t)sp : (int64_t)sp * p.G + g;

  f32x4 acc[8][4];   // [ki][ni]: MFMA rows = k, cols = n
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DMA: wave w, piece j fills image rows 2 (NPC w + j) + (lane >> 5) of both operands
  const int prow = lane >> 5;
  uint32_t colC[NPC], colA[NPC];
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const int irow = 2 * (NPC * wave + j) + prow;
    const int c = (lane & 31) ^ ((irow & 7) << 1);
    colC[j] = (uint32_t)(n0 + c * 8) * 2; colA[j] = (uint32_t)(k0 + c * 8) * 2;
  }
  const int rbase = (int)(r0 + s_begin * ROWS) + 2 * NPC * wave + prow;
  const int rlast = (int)r1 - 1;
  const uint32_t ldc = (uint32_t)p.lddc_b, lda = (uint32_t)p.lda_b;
  int32_t ic[NPC], ia[NPC];
  auto load_index = [&](int step) {
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int m = min(rbase + step * ROWS + 2 * j, rlast);
      if (GC) ic[j] = p.c_row_idx[m];
      if (GA) ia[j] = p.a_row_idx[m];
    }
  };
  char *const s_sc = smem + NS * BG_STAGE;                        // [stage][64 floats]
  int32_t sc_ix = 0;
  auto load_sc_index = [&](int step) {
    if (SC && wave == 0) sc_ix = p.c_row_idx[min((int)(r0 + s_begin * ROWS) + step * ROWS + lane, rlast)];
  };
  auto dma = [&](int step, int stage, auto tail_c) {
    constexpr bool TAIL = decltype(tail_c)::value;
    char *const dst = smem + stage * BG_STAGE + (NPC * wave) * 1024;
    const uint64_t zero_row = (uint64_t)(uintptr_t)g_wgrad_zero_row + (lane & 15) * 16;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int m = rbase + step * ROWS + 2 * j;
      const uint32_t cr = GC ? (uint32_t)(ic[j] >> p.c_row_sh) : (uint32_t)(TAIL ? min(m, rlast) : m);
      const uint32_t ar = GA ? (uint32_t)(ia[j] >> p.a_row_sh) : (uint32_t)(TAIL ? min(m, rlast) : m);
      if constexpr (!TAIL) {
        __builtin_amdgcn_global_load_lds((glb_void *)(p.dC + (cr * ldc + colC[j])), (lds_void *)(dst + j * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)(p.A + (ar * lda + colA[j])), (lds_void *)(dst + j * 1024 + BG_OPB), 16, 0, 0);
      } else {                                        // rows past the unit's end read the zero row (bit-mask select: no branches)
        const uint64_t ok = m <= rlast ? ~(uint64_t)0 : (uint64_t)0;
        const uint64_t sc_ = (((uint64_t)(uintptr_t)p.dC + (cr * ldc + colC[j])) & ok) | (zero_row & ~ok);
        const uint64_t sa_ = (((uint64_t)(uintptr_t)p.A + (ar * lda + colA[j])) & ok) | (zero_row & ~ok);
        __builtin_amdgcn_global_load_lds((glb_void *)(uintptr_t)sc_, (lds_void *)(dst + j * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)(uintptr_t)sa_, (lds_void *)(dst + j * 1024 + BG_OPB), 16, 0, 0);
      }
    }
    if (SC && wave == 0)
      __builtin_amdgcn_global_load_lds((glb_void *)(p.c_row_scale + sc_ix), (lds_void *)(s_sc + stage * 256), 4, 0, 0);
  };

  // transposed fragment reads: lane (li, lg) supplies row 4 lg + (li >> 2) (+ 16: second half of a fragment, + 32: second
  // chunk of a 64-row step), 8 bytes at columns 4 (li & 3) .. + 3 of the 16-column tile
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void *)smem;
  const int s3 = (4 * (lg & 1) + (li >> 2)) & 7;
  const uint32_t frow = lds0 + (uint32_t)((4 * lg + (li >> 2)) * RS + 8 * (li & 1));
  uint32_t adK[8], adN[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) adK[i] = frow + BG_OPB + (uint32_t)(((wr * 16 + 2 * i + ((li & 3) >> 1)) ^ (s3 << 1)) * 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) adN[i] = frow + (uint32_t)(((wc * 8 + 2 * i + ((li & 3) >> 1)) ^ (s3 << 1)) * 16);
  const uint32_t ad_sc = lds0 + NS * BG_STAGE + 16 * lg;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define BG_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))

  const bool do_bias = (p.bias_ws || p.direct_db) && tk == 0 && wr == 0;
  float acc_b[4] = {0.f, 0.f, 0.f, 0.f};
  typedef T t2 __attribute__((ext_vector_type(2)));
  auto colsum8 = [&](const frag &f, float a) -> float {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const t2 pr = t2{f[2 * j], f[2 * j + 1]};
      if constexpr (std::is_same<T, half_t>::value) a = __builtin_amdgcn_fdot2(pr, t2{(T)1, (T)1}, a, false);
      else a = __builtin_amdgcn_fdot2_f32_bf16(pr, t2{(T)1.f, (T)1.f}, a, false);
    }
    return a;
  };

  auto compute = [&](int stage) {
    const uint32_t so = (uint32_t)stage * BG_STAGE;
#pragma unroll
    for (int kc = 0; kc < ROWS / 32; ++kc) {
      u32x2 rk[8][2], rn[4][2];
      f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (kc == 0) { BG_TR(rn[i][0], adN[i] + so, 0); BG_TR(rn[i][1], adN[i] + so, 16 * RS); }
        else { BG_TR(rn[i][0], adN[i] + so, 32 * RS); BG_TR(rn[i][1], adN[i] + so, 48 * RS); }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (kc == 0) { BG_TR(rk[i][0], adK[i] + so, 0); BG_TR(rk[i][1], adK[i] + so, 16 * RS); }
        else { BG_TR(rk[i][0], adK[i] + so, 32 * RS); BG_TR(rk[i][1], adK[i] + so, 48 * RS); }
      }
      if constexpr (SC) {
        const uint32_t a_ = ad_sc + (uint32_t)stage * 256;
        if (kc == 0) {
          asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(s0) : "v"(a_));
          asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(s1) : "v"(a_));
        } else {
          asm volatile("ds_read_b128 %0, %1 offset:128" : "=v"(s0) : "v"(a_));
          asm volatile("ds_read_b128 %0, %1 offset:192" : "=v"(s1) : "v"(a_));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      frag fk[8], fn[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) fk[i] = __builtin_bit_cast(frag, u32x4{rk[i][0][0], rk[i][0][1], rk[i][1][0], rk[i][1][1]});
#pragma unroll
      for (int i = 0; i < 4; ++i) fn[i] = __builtin_bit_cast(frag, u32x4{rn[i][0][0], rn[i][0][1], rn[i][1][0], rn[i][1][1]});
      if constexpr (SC) {
        const f16x8 sh = f16x8{(half_t)s0[0], (half_t)s0[1], (half_t)s0[2], (half_t)s0[3],
                               (half_t)s1[0], (half_t)s1[1], (half_t)s1[2], (half_t)s1[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) fn[i] = __builtin_bit_cast(frag, __builtin_bit_cast(f16x8, fn[i]) * sh);
      }
#pragma unroll
      for (int ki = 0; ki < 8; ++ki)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ki][ni] = MM::mma(fk[ki], fn[ni], acc[ki][ni]);
      if (do_bias) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc_b[ni] = colsum8(fn[ni], acc_b[ni]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#undef BG_TR

  // Pipeline: steps t + 1 .. t + PD are in flight (or landed) while step t is multiplied.  A step's issue slot is
  // [its 2 NPC DMA pieces, the gather indices of the step after it]; VMEM operations retire in order, so "at most
  // (PD - 1) issue slots + one index batch outstanding" means step t + 1 has landed.  Wave 0's extra per-row-factor
  // operations only make its wait stricter; the last PD steps wait for everything.
  constexpr int IDX = NPC * ((GC ? 1 : 0) + (GA ? 1 : 0));
  constexpr int KEEP = (PD - 1) * (2 * NPC + IDX) + IDX;
  const std::true_type is_tail; const std::false_type not_tail;
  auto issue = [&](int step) {                 // the DMA of `step` (its indices are in registers), then the indices of step + 1
    if (step < nst) {
      if (step + 1 == nst) dma(step, step % NS, is_tail); else dma(step, step % NS, not_tail);
      if (step + 1 < nst) { if (GC || GA) load_index(step + 1); load_sc_index(step + 1); }
    }
  };
  WB_STAMP(0);
  if (nst > 0) {
    if (GC || GA) load_index(0);
    load_sc_index(0);
#pragma unroll
    for (int q = 0; q < PD; ++q) issue(q);
    if (PD > 1 && nst > PD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // step 0 has landed for every wave
    WB_STAMP(1);
    for (int t = 0; t < nst; ++t) {
      issue(t + PD);                            // into the stage step t - 1 was multiplied from
      WB_STAMP(2 + 4 * t);
      compute(t % NS);
      WB_STAMP(3 + 4 * t);
      if (PD > 1 && t + PD + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WB_STAMP(4 + 4 * t);
      __builtin_amdgcn_s_barrier();            // step t + 1 has landed for every wave, and every wave is done reading step t
      WB_STAMP(5 + 4 * t);
    }
  }

  if (do_bias) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float v = acc_b[ni];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int n = n0 + wc * 64 + ni * 16 + li;
      if (lg == 0) wgrad_store_bias(p, v, slab_id, g, n);
    }
  }
  float *out = p.direct_dW ? p.direct_dW + (int64_t)g * p.N * p.K : p.ws + slab_id * (int64_t)p.N * p.K;
  const bool add = p.direct_dW && p.direct_beta;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    float *row = out + (int64_t)(n0 + wc * 64 + ni * 16 + li) * p.K + k0 + wr * 128 + 4 * lg;
    f32x4 old[8];
    if (add) {
#pragma unroll
      for (int ki = 0; ki < 8; ++ki) old[ki] = *(const f32x4 *)(row + ki * 16);
    }
#pragma unroll
    for (int ki = 0; ki < 8; ++ki) *(f32x4 *)(row + ki * 16) = add ? acc[ki][ni] + old[ki] : acc[ki][ni];
  }
}

// ------------------------------------------------------------------------------------------------
// Skinny weight gradient: dW [N, K] with K = 16 or 32 - the router's w_gate (custom_moe_layer.py:213-217:
// dW_gate = h^T d_logits, K = num_experts), no gathers, no bias, one group.  A 128 x 128 MFMA tile pads K to 128: seven of
// eight MFMAs multiply zeros and every A-side DMA piece takes the clamped tail path (25 us fp16 / 112 us fp32 per launch at
// M = 25 216, N = 384 for 0.3 GFLOP).  Here the call is what it is, a stream over dC: lane = two columns n of a 128-wide
// column tile, the 16 k of the wave's slice in registers (32 fp32 accumulators), the A row - the same for every lane - read
// by scalar loads, one fma per (row, n, k).  The four waves of a workgroup take interleaved rows (K = 16) or two k slices x
// two row phases (K = 32) and add up through LDS; a workgroup's [128, K] block goes to its slab exactly like a 128 x 128
// kernel's (same layout, same reduction riding on the next launch).
template <typename T, int KP>
__global__ __launch_bounds__(WG_THREADS) void wgrad_skinny_kernel(const WgradDev p) {
  constexpr int KS = KP / 16, PH = 4 / KS, UNR = 16;
  constexpr int ES = (int)sizeof(T);
  __shared__ float sred[4][64][2][16];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bz, gz;
  if (wgrad_ride_along(p, tid, bz, gz)) return;
  const int ks = wave % KS, ph = wave / KS;
  const int n0 = blockIdx.x * WG_T;
  int n = n0 + 2 * lane;
  if (n > p.N - 2) n = p.N - 2;                       // clamped columns are computed and never stored
  const int64_t per = (p.M + p.splits - 1) / p.splits;
  const int64_t r0 = (int64_t)bz * per, r1 = (r0 + per < p.M) ? r0 + per : p.M;
  typedef T t2 __attribute__((ext_vector_type(2)));
  typedef T t16 __attribute__((ext_vector_type(16)));
  float acc[2][16];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
  const char *cb = p.dC + (int64_t)n * ES;
  const char *ab = p.A + (int64_t)ks * 16 * ES;
  // batches of 16 rows (row u of a batch: m + u * PH), every load of a batch issued before its first use.  The A rows - the
  // same for every lane - are loaded by lanes 0..15 (lane u: row u of the batch, converted to fp32 there) and handed out by
  // v_readlane: 16 scalar operands per row.  Addresses: one 64-bit base per batch + 32-bit row offsets; the last, partial
  // batch clamps its rows to the last valid one and zeroes their dC values.
  const uint32_t ldc = (uint32_t)p.lddc_b, lda = (uint32_t)p.lda_b;
  auto batch = [&](const char *cm, const char *am, int cnt, auto full) {
    constexpr bool FULL = decltype(full)::value;
    t2 c[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int uu = FULL ? u : (u < cnt ? u : cnt - 1);
      c[u] = *(const t2 *)(cm + (uint32_t)(uu * PH) * ldc);
    }
    const int ua = FULL ? (lane & 15) : ((lane & 15) < cnt ? (lane & 15) : cnt - 1);
    const t16 ar = *(const t16 *)(am + (uint32_t)(ua * PH) * lda);
    float af[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) af[k] = (float)ar[k];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool ok = FULL || u < cnt;                                   // wave-uniform
      const float c0 = ok ? (float)c[u][0] : 0.f, c1 = ok ? (float)c[u][1] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, af[k]), u));
        acc[0][k] = __builtin_fmaf(c0, a, acc[0][k]);
        acc[1][k] = __builtin_fmaf(c1, a, acc[1][k]);
      }
    }
  };
  int64_t m = r0 + ph;
  const char *cm = cb + m * p.lddc_b, *am = ab + m * p.lda_b;
  for (; m + (int64_t)(UNR - 1) * PH < r1; m += (int64_t)UNR * PH) {
    batch(cm, am, UNR, std::true_type{});
    cm += (int64_t)UNR * PH * p.lddc_b; am += (int64_t)UNR * PH * p.lda_b;
  }
  if (m < r1) batch(cm, am, (int)((r1 - m + PH - 1) / PH), std::false_type{});
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 16; ++k) sred[wave][lane][j][k] = acc[j][k];
  __syncthreads();
  float *out = p.ws + (int64_t)bz * p.N * p.K;
  for (int i = tid; i < WG_T * KP; i += WG_THREADS) {
    const int nl = i / KP, k = i - nl * KP;
    if (n0 + nl >= p.N) continue;
    const int sl = k >> 4, kk = k & 15;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < PH; ++q) v += sred[q * KS + sl][nl >> 1][nl & 1][kk];
    out[(int64_t)(n0 + nl) * p.K + k] = v;
  }
}

#ifdef M3_EXPERIMENTAL      // the wide-tile kernel: built only by `make EXPERIMENTAL=1` (measured no faster in the step; the engine never takes it)
// ------------------------------------------------------------------------------------------------
// Wide tiles (fp16): 128(n) x 384(k) for the K = 384 weights (qkv, fc1, expert FC1) and 384(n) x 128(k) for the
// N = 384 ones (fc2, expert FC2, patch embedding), eight waves per workgroup, one workgroup per CU.
// A 128 x 128 tile moves 16 KiB through global loads, ds_write and transposed ds_reads per MFLOP; every one of
// those paths is within 1.5x of the MFMA time, and with four waves per workgroup the phases of a 32-row step
// (loads, LDS stores, barrier, transposed reads, 16 MFMAs) largely serialise: 20 % of the MFMA peak.  The wide tile
// has 1.5x the FLOP per staged byte (96 instead of 64), covers the whole 384-wide side (that operand is read from
// memory once per column tile instead of three times) and gives a wave 24 MFMAs per barrier with a second wave on
// its SIMD to run under its LDS phases.
// Staging: one wave loads one contraction row - lanes 0..CPR_C-1 the 16-byte chunks of the dC row, the others the
// chunks of the A row (64 chunks either way) - so a wave's load instruction reads two contiguous runs and the
// gather index of a row is one broadcast load.
template <bool WIDE_K> struct WwCfg {
  static constexpr int TN = WIDE_K ? 128 : 384, TK = WIDE_K ? 384 : 128;
  static constexpr int NWR = WIDE_K ? 4 : 2, NWC = 8 / NWR;            // waves along k, along n
  static constexpr int KI = TK / 16 / NWR, NI = TN / 16 / NWC;         // 16 x 16 tiles per wave: 6 x 4 or 4 x 6
  static constexpr int SC = TN * 2 + 32, SA = TK * 2 + 32;             // LDS row strides: 32 B past a multiple of 256 B
  static constexpr int CPR_C = TN / 8;                                 // 16-byte chunks of a dC row; the A row has 64 - CPR_C
};
constexpr int WW_THREADS = 512;

__device__ __forceinline__ f16x8 read_tr_frag16(const char *base, int stride, int rb, int col, int li, int lg) {
  const char *p0 = base + (rb + 4 * lg + (li >> 2)) * stride + (col + 4 * (li & 3)) * 2;
  const char *p1 = p0 + 16 * stride;
  fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t *)p0);
  fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t *)p1);
  f16x8 f;
  f[0] = (half_t)lo[0]; f[1] = (half_t)lo[1]; f[2] = (half_t)lo[2]; f[3] = (half_t)lo[3];
  f[4] = (half_t)hi[0]; f[5] = (half_t)hi[1]; f[6] = (half_t)hi[2]; f[7] = (half_t)hi[3];
  return f;
}

// Diagnostic build (-DM3_WGRAD_CLOCK, tools/wgrad_clock.py): workgroup 0 records how many shader cycles (s_memtime) and
// how many 100 MHz reference ticks (s_memrealtime) its life took - their ratio is the clock the kernel really ran at.
#ifdef M3_WGRAD_CLOCK
__device__ unsigned long long g_wgrad_clock[4];
#endif

template <bool GC, bool GA, bool WIDE_K, int ROWS>
__global__ __launch_bounds__(WW_THREADS, 2) void wgrad_wide_kernel(const WgradDev p) {
#ifdef M3_WGRAD_CLOCK
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  typedef Mma<half_t> MM;
  typedef MM::frag frag;
  typedef WwCfg<WIDE_K> C;
  constexpr int KI = C::KI, NI = C::NI, SC = C::SC, SA = C::SA;
  constexpr int NLD = ROWS / 8;                             // rows (= chunks) per thread per step
  constexpr int OPC = ROWS * SC, BUF = ROWS * (SC + SA);    // dC image, whole buffer
  constexpr int KCH = ROWS / 32;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wr = wave / C::NWC, wc = wave - wr * C::NWC;

  const int tiles = gridDim.x;
  const int lin = blockIdx.x + tiles * (blockIdx.y + gridDim.y * blockIdx.z);
  int tile, gs, g, sp, nst;
  int64_t r0, r1, s_begin;
  if (p.chunk_rows) {                          // gs = work unit; its slab is ws[gs]
    if (!wgrad_unit(p.group_offsets, p.G, p.chunk_rows, lin, tiles, lane, tile, gs, g, r0, r1)) return;
    sp = gs; s_begin = 0;
    nst = (int)((r1 - r0 + ROWS - 1) / ROWS);
  } else {
    const int log_id = xcd_remap(lin, tiles * gridDim.y * gridDim.z);
    tile = log_id % tiles; gs = log_id / tiles;
    g = gs % (int)gridDim.y; sp = gs / (int)gridDim.y;
    if (p.group_offsets && p.lpt) g = wgrad_lpt_group(p.group_offsets, p.G, g, lane);
    if (p.group_offsets) { r0 = p.group_offsets[g]; r1 = p.group_offsets[g + 1]; }
    else { r0 = 0; r1 = p.M; }
    const int64_t nsteps_all = (r1 - r0 + ROWS - 1) / ROWS;
    const int64_t per = (nsteps_all + p.splits - 1) / p.splits;
    s_begin = (int64_t)sp * per;
    int64_t s_end = s_begin + per;
    if (s_end > nsteps_all) s_end = nsteps_all;
    nst = (int)(s_end > s_begin ? s_end - s_begin : 0);
  }
  const int tn = tile / p.tiles_k, tk = tile - tn * p.tiles_k;
  const int n0 = tn * C::TN, k0 = tk * C::TK;
  const int64_t slab_id = p.chunk_rows ? (int64_t)sp : (int64_t)sp * p.G + g;

  f32x4 acc[KI][NI];   // MFMA rows = k, cols = n
#pragma unroll
  for (int a = 0; a < KI; ++a)
#pragma unroll
    for (int b = 0; b < NI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: wave w loads rows w, w + 8, ... of a step; lane j the j-th 16-byte chunk of (dC row | A row)
  const bool is_c = lane < C::CPR_C;
  const char *g_base = is_c ? p.dC + (int64_t)n0 * 2 + lane * 16 : p.A + (int64_t)k0 * 2 + (lane - C::CPR_C) * 16;
  const int64_t g_stride = is_c ? p.lddc_b : p.lda_b;
  // gather: every lane reads an index table (its own side's, or - value unused - the other side's), so the index
  // load is unconditional and the compiler's vmcnt counts stay exact; a_row_div is a power of two here (host check)
  const int32_t *g_idx = (is_c ? GC : GA) ? (is_c ? p.c_row_idx : p.a_row_idx) : (GC ? p.c_row_idx : p.a_row_idx);
  const int32_t g_mask = (is_c ? GC : GA) ? -1 : 0;        // (a mask, not a branch: the compiler would sink the load into it)
  const int g_sh = is_c ? 0 : __builtin_ctz((unsigned)p.a_row_div);
  const int st_off = is_c ? wave * SC + lane * 16 : OPC + wave * SA + (lane - C::CPR_C) * 16;      // + 8 i * stride
  const int st_step = 8 * (is_c ? SC : SA);
  const int64_t rbase = r0 + s_begin * ROWS + wave;        // row of chunk 0 in local step 0

  auto row_of = [&](int step, int i) -> int64_t {           // clamped row
    const int64_t m = rbase + (int64_t)step * ROWS + i * 8;
    return m < r1 ? m : r1 - 1;
  };
  auto load_index = [&](int step, int32_t(&ix)[NLD]) {
    if (!(GC || GA)) return;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int64_t m = row_of(step, i);
      const int32_t v = g_idx[m] >> g_sh;
      ix[i] = (int32_t)m ^ ((v ^ (int32_t)m) & g_mask);
    }
  };
  auto load_global = [&](int step, const int32_t(&ix)[NLD], u32x4(&rq)[NLD]) {
#ifdef WW_ABL_NOLOAD                               // diagnostic builds (profiles/README.md): one phase of the step removed
    if (step > 1) return;
#endif
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int64_t row = (GC || GA) ? (int64_t)ix[i] : row_of(step, i);
      rq[i] = *(const u32x4 *)(g_base + row * g_stride);
    }
  };
  auto store_lds = [&](int buf, int step, const u32x4(&rq)[NLD]) {
    char *base = smem + buf * BUF + st_off;
#ifdef WW_ABL_NOLDSW
    if (step > 1) return;
#endif
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const bool ok = rbase + (int64_t)step * ROWS + i * 8 < r1;       // rows past the group's end add zeros
      *(u32x4 *)(base + i * st_step) = ok ? rq[i] : u32x4{0u, 0u, 0u, 0u};
    }
  };
  const bool do_bias = (p.bias_ws || p.direct_db) && tk == 0 && wr == 0;    // column sums of dC as one extra MFMA row of ones
  f32x4 acc_b[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc_b[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  frag ones;
#pragma unroll
  for (int j = 0; j < MM::EPL; ++j) ones[j] = (half_t)1.0f;

  auto compute = [&](int buf) {
    const char *sC = smem + buf * BUF, *sA = sC + OPC;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) {
      // the n fragments stay for the whole chunk; the k fragments are read two ahead of their MFMA row (sched_barriers
      // keep the compiler from hoisting all ten reads to the top: with them all live the kernel spills)
      frag fn[NI], fk[KI];
#ifdef WW_ABL_NOTR
#define read_tr_frag16(a_, b_, c_, d_, e_, f_) ones
#endif
#pragma unroll
      for (int i = 0; i < NI; ++i) fn[i] = read_tr_frag16(sC, SC, kc * 32, (wc * NI + i) * 16, li, lg);
      fk[0] = read_tr_frag16(sA, SA, kc * 32, (wr * KI + 0) * 16, li, lg);
      fk[1] = read_tr_frag16(sA, SA, kc * 32, (wr * KI + 1) * 16, li, lg);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ki = 0; ki < KI; ++ki) {
        if (ki + 2 < KI) fk[ki + 2] = read_tr_frag16(sA, SA, kc * 32, (wr * KI + ki + 2) * 16, li, lg);
#pragma unroll
#ifdef WW_ABL_NOMFMA
        for (int ni = 0; ni < NI; ++ni) acc[ki][ni][0] += fk[ki][0] * fn[ni][0];
#else
        for (int ni = 0; ni < NI; ++ni) acc[ki][ni] = MM::mma(fk[ki], fn[ni], acc[ki][ni]);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
#ifdef WW_ABL_NOTR
#undef read_tr_frag16
#endif
      if (do_bias) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc_b[ni] = MM::mma(ones, fn[ni], acc_b[ni]);
      }
    }
  };

  if (nst > 0) {
    // Ping-pong: waves 0-3 and waves 4-7 (one of each on every SIMD) run half a step apart.  In the first half of
    // step t the first group multiplies buffer t while the second group requests the rows of step t + 2 and writes
    // its share of step t + 1 into the other buffer; in the second half they swap.  So one group's transposed reads
    // and MFMAs always run against the other group's address arithmetic, global loads and LDS stores, instead of all
    // eight waves reading, multiplying and storing in lockstep (measured on the lockstep version: the skeleton
    // without MFMAs, loads and slab stores still took 0.67 us per 32-row step).
    // Steps past the end re-load the last step (clamped) and are never multiplied, so the loop body is unconditional.
    const int last = nst - 1;
    auto cl = [&](int s_) { return s_ < last ? s_ : last; };
    const bool first_grp = wave < 4;
    // two register sets, each written to LDS and then re-used at once for the rows two steps further on: a load has
    // two whole steps to arrive (one step was not enough for rows coming from the Infinity Cache / HBM)
    u32x4 qa[NLD], qb[NLD];
    int32_t ip[NLD], iq[NLD];                     // gather indices, two sets: a refill requests the next refill's rows'
    load_index(0, ip);                             // indices BEFORE it issues its own data loads, so that using them a
    load_global(0, ip, qb);                        // step later does not wait for those data loads (vmcnt is in-order)
    load_index(cl(1), iq);
    load_global(cl(1), iq, qa);
    store_lds(0, 0, qb);
    load_index(cl(2), ip);
    load_global(cl(2), ip, qb);
    load_index(cl(3), ip);
    __syncthreads();
    // entry of even local step t: buf0 = step t, qa = step t + 1, qb = step t + 2, ip = indices of step t + 3
    auto refill_a = [&](int t) { store_lds(1, t + 1, qa); load_index(cl(t + 4), iq); load_global(cl(t + 3), ip, qa); };
    auto refill_b = [&](int t) { store_lds(0, t + 2, qb); load_index(cl(t + 5), ip); load_global(cl(t + 4), iq, qb); };
    for (int t = 0; t < nst; t += 2) {
      if (first_grp) compute(0); else refill_a(t);
      __syncthreads();
      if (first_grp) refill_a(t); else compute(0);
      __syncthreads();
      const bool odd = t + 1 < nst;
      if (first_grp) { if (odd) compute(1); } else refill_b(t);
      __syncthreads();
      if (first_grp) refill_b(t); else { if (odd) compute(1); }
      __syncthreads();
    }
  }

  if (do_bias && lg == 0) {                       // every row of the ones-product is the column sum: take row 0
    float *bs = p.bias_ws + slab_id * p.N;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bs[n0 + (wc * NI + ni) * 16 + li] = acc_b[ni][0];
  }
  // slab[n][k]; lane holds k = kb + 4*lg + r, n = nb + li
  float *slab = p.ws + slab_id * (int64_t)p.N * p.K;
#ifdef WW_ABL_NOSLAB
  if (acc[0][0][0] != 12345.f) return;
#endif
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + (wc * NI + ni) * 16 + li;
#pragma unroll
    for (int ki = 0; ki < KI; ++ki) {
      const int k = k0 + (wr * KI + ki) * 16 + 4 * lg;
      *(f32x4 *)(slab + (int64_t)n * p.K + k) = acc[ki][ni];
    }
  }
#ifdef M3_WGRAD_CLOCK
  if (lin == 0 && tid == 0) {
    g_wgrad_clock[0] = __builtin_amdgcn_s_memtime() - clk_t0;
    g_wgrad_clock[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    g_wgrad_clock[2] = (unsigned long long)nst;
  }
#endif
}

#endif  // M3_EXPERIMENTAL

// slabs -> dW (blocks [0, nb_w)) and, in the same launch, bias slabs -> db (blocks [nb_w, ...)); splits in order
__device__ __forceinline__ f32x4 wgrad_sum_slabs(const f32x4 *w, int64_t elems4, int lo, int hi, f32x4 s) {
  // eight slabs' loads in flight before the first add (a thread owns ONE 16-byte column of its slabs: issued one dependent
  // load at a time the reduction was latency-bound); slabs are added in index order
  int sp = lo;
  for (; sp + 8 <= hi; sp += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = w[(int64_t)(sp + j) * elems4];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; sp < hi; ++sp) s += w[(int64_t)sp * elems4];
  return s;
}
// cols = 256: a thread per 16-byte column, the slabs one after the other.  cols = 64 (many slabs - m3_wgrad_reduce_cols: a
// column's chain of dependent load batches was the whole duration of a launch with 100+ parts): four threads per column,
// each sums a quarter of the slabs (contiguous ranges), the quarters are added in order by the first; deterministic, another
// association than cols = 256.  Every thread of the block must call (a barrier inside).
__device__ __forceinline__ void wgrad_reduce_block(int64_t blk, int tid, const float *ws, int splits, int64_t elems4, float *dW,
                                                   int beta, int nb_w, const float *bias_ws, int64_t belems4, float *db, int beta_db,
                                                   int cols) {
  if (blk >= nb_w) {
    blk -= nb_w; ws = bias_ws; elems4 = belems4; dW = db; beta = beta_db;
  }
  if (cols == 256) {
    const int64_t i = blk * 256 + tid;
    if (i >= elems4) return;
    f32x4 s = beta ? ((const f32x4 *)dW)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    ((f32x4 *)dW)[i] = wgrad_sum_slabs((const f32x4 *)ws + i, elems4, 0, splits, s);
    return;
  }
  __shared__ f32x4 spart[3][64];
  const int col = tid & 63, part = tid >> 6;
  const int64_t i = blk * 64 + col;
  const bool in = i < elems4;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (in) {
    if (part == 0 && beta) s = ((const f32x4 *)dW)[i];
    s = wgrad_sum_slabs((const f32x4 *)ws + i, elems4, splits * part / 4, splits * (part + 1) / 4, s);
  }
  if (part > 0) spart[part - 1][col] = s;
  __syncthreads();
  if (part == 0 && in) {
    s += spart[0][col]; s += spart[1][col]; s += spart[2][col];
    ((f32x4 *)dW)[i] = s;
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *ws, int splits, int64_t elems4, float *dW, int beta, int nb_w,
                                                           const float *bias_ws, int64_t belems4, float *db, int beta_db, int cols) {
  wgrad_reduce_block(blockIdx.x, threadIdx.x, ws, splits, elems4, dW, beta, nb_w, bias_ws, belems4, db, beta_db, cols);
}

// balanced grouped mode: dW[g] (+)= sum of the slabs of group g's units, in unit order; g = group,
// blocks [0, nb_w) the weight elements, [nb_w, ..) the bias elements
__device__ __forceinline__ void wgrad_reduce_grouped_block(int64_t blk, int g, int tid, const float *ws, const int32_t *off, int G,
                                                           int chunk, int64_t elems4, float *dW, int beta, int nb_w,
                                                           const float *bias_ws, int64_t belems4, float *db, int beta_db) {
  if (blk >= nb_w) {
    blk -= nb_w; ws = bias_ws; elems4 = belems4; dW = db; beta = beta_db;
  }
  const int64_t i = blk * 256 + tid;
  int rows, n, first;
  wgrad_unit_scan(off, G, chunk, tid & 63, rows, n, first);       // all lanes take part in the scan
  n = __shfl(n, g, 64); first = __shfl(first, g, 64);
  if (i >= elems4) return;
  f32x4 *out = (f32x4 *)dW + (int64_t)g * elems4 + i;
  f32x4 s = beta ? *out : f32x4{0.f, 0.f, 0.f, 0.f};
  // as in wgrad_reduce_block: up to eight slabs' loads in flight before the first add (one dependent load per unit made the
  // reduction riding on a short launch - the router's weight gradient - the longest part of it); same summation order
  const f32x4 *w = (const f32x4 *)ws + (int64_t)first * elems4 + i;
  int u = 0;
  for (; u + 8 <= n; u += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = w[(int64_t)(u + j) * elems4];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  if (u + 4 <= n) {
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = w[(int64_t)(u + j) * elems4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s += v[j];
    u += 4;
  }
  if (u + 2 <= n) {
    const f32x4 v0 = w[(int64_t)u * elems4], v1 = w[(int64_t)(u + 1) * elems4];
    s += v0; s += v1;
    u += 2;
  }
  if (u < n) s += w[(int64_t)u * elems4];
  *out = s;
}
__global__ __launch_bounds__(256) void wgrad_reduce_grouped_kernel(const float *ws, const int32_t *off, int G, int chunk, int64_t elems4,
                                                                   float *dW, int beta, int nb_w, const float *bias_ws, int64_t belems4,
                                                                   float *db, int beta_db) {
  wgrad_reduce_grouped_block(blockIdx.x, blockIdx.y, threadIdx.x, ws, off, G, chunk, elems4, dW, beta, nb_w, bias_ws, belems4, db, beta_db);
}

// ------------------------------------------------------------------ column sums
// db[g][n] = sum over rows of group g of dC[crow(m), n].  Stage 1: grid (column blocks,
// strips per group, groups); a workgroup is 64 column chunks (16 B each) x 4 row lanes and
// streams its strip of rows with coalesced 16-byte loads; stage 2 (reduce.hip) adds the strip
// partials in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void colsum_part_kernel(const char *__restrict__ dC, int64_t lddc_b,
                                                          const int32_t *__restrict__ c_row_idx, int64_t M, int N,
                                                          const int32_t *__restrict__ group_offsets, int spg,
                                                          float *__restrict__ part) {
  constexpr int ES = (int)sizeof(T), EPC = 16 / ES;
  __shared__ float sred[4][64][EPC + 1];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + cl) * EPC;
  const int strip = blockIdx.y, g = blockIdx.z;
  int64_t r0 = 0, r1 = M;
  if (group_offsets) { r0 = group_offsets[g]; r1 = group_offsets[g + 1]; }
  const int64_t per = ((r1 - r0) + spg - 1) / spg;
  const int64_t b0 = r0 + (int64_t)strip * per;
  int64_t b1 = b0 + per;
  if (b1 > r1) b1 = r1;
  float acc[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) acc[j] = 0.f;
  if (col < N) {
#pragma unroll 4
    for (int64_t m = b0 + rl; m < b1; m += 4) {
      const int64_t cr = c_row_idx ? (int64_t)c_row_idx[m] : m;
      const u32x4 raw = *(const u32x4 *)(dC + cr * lddc_b + (int64_t)col * ES);
      const T *e = (const T *)&raw;
#pragma unroll
      for (int j = 0; j < EPC; ++j) acc[j] += (float)e[j];
    }
  }
#pragma unroll
  for (int j = 0; j < EPC; ++j) sred[rl][cl][j] = acc[j];
  __syncthreads();
  if (rl == 0 && col < N) {
    float *o = part + ((int64_t)g * spg + strip) * N + col;
#pragma unroll
    for (int j = 0; j < EPC; ++j) o[j] = sred[0][cl][j] + sred[1][cl][j] + sred[2][cl][j] + sred[3][cl][j];
  }
}

static int colsum_strips_per_group(int64_t M, int G) {
  int64_t per_group = (M + G - 1) / G;
  int64_t s = (per_group + 127) / 128;
  if (s < 1) s = 1;
  if (s > 512) s = 512;
  return (int)s;
}

}  // namespace m3

using namespace m3;

// the output tile (n x k) m3_wgrad_tn uses for a shape: callers size `splits` / `units` (and with them the slab
// workspace) for ceil(N / tn) * ceil(K / tk) tiles per group.  The wide tiles are opt-in (m3_wgrad_set_wide(1) or
// M3_WGRAD_WIDE=1): 5-20 % faster per launch by themselves, not faster inside the two-stream training step.
#ifdef M3_WGRAD_CLOCK
extern "C" int m3_debug_wgrad_clock(unsigned long long *dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wgrad_clock), sizeof(g_wgrad_clock)) == hipSuccess ? M3_OK : M3_ERR_LAUNCH;
}
#endif

static int g_wgrad_dma = -1;
extern "C" int m3_wgrad_set_dma(int on) {
  M3_REQUIRE(on >= -1 && on <= 2, "m3_wgrad_set_dma: %d", on);
  g_wgrad_dma = on;
  return M3_OK;
}

static int g_wgrad_wide = -1;
extern "C" int m3_wgrad_set_wide(int on) {
  M3_REQUIRE(on >= -1 && on <= 1, "m3_wgrad_set_wide: %d", on);
#ifndef M3_EXPERIMENTAL
  M3_REQUIRE(on <= 0, "m3_wgrad_set_wide: the wide-tile kernel is only in EXPERIMENTAL builds (make EXPERIMENTAL=1)");
#endif
  g_wgrad_wide = on;
  return M3_OK;
}

static int g_wgrad_big = -1;
extern "C" int m3_wgrad_set_big(int on) {
  M3_REQUIRE(on >= -1 && on <= 1, "m3_wgrad_set_big: %d", on);
  g_wgrad_big = on;
  return M3_OK;
}
static bool wgrad_big_shape(int N, int K, int dtype) {
  if (g_wgrad_big < 0) { const char *e = getenv("M3_WGRAD_BIG"); g_wgrad_big = e ? (atoi(e) ? 1 : 0) : 1; }
  return g_wgrad_big && dtype != M3_F32 && N % BG_T == 0 && K % BG_T == 0;
}

extern "C" int m3_wgrad_tile(int N, int K, int dtype, int *tn, int *tk) {
  M3_REQUIRE(tn && tk, "m3_wgrad_tile: null output");
  if (wgrad_big_shape(N, K, dtype)) { *tn = BG_T; *tk = BG_T; return M3_OK; }
  if (g_wgrad_wide < 0) { const char *e = getenv("M3_WGRAD_WIDE"); g_wgrad_wide = e ? (atoi(e) ? 1 : 0) : 0; }
  *tn = WG_T; *tk = WG_T;
#ifndef M3_EXPERIMENTAL
  g_wgrad_wide = 0;
#endif
  if (g_wgrad_wide && dtype == M3_F16) {
    if (K == 384 && N % 128 == 0 && N >= 768) { *tn = 128; *tk = 384; }
    else if (N == 384 && K % 128 == 0 && K >= 768) { *tn = 384; *tk = 128; }
  }
  return M3_OK;
}

// 16-byte columns per block of the dense slab reduction: four threads per column from 32 slabs on
static inline int m3_wgrad_reduce_cols(int splits) { return splits >= 32 ? 64 : 256; }
#ifdef M3_WGRAD_STAMPS
extern "C" int m3_debug_wbig_stamps(unsigned long long *dst, int wgs) {
  if (wgs > WSTAMP_WGS) wgs = WSTAMP_WGS;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wbig_stamps), (size_t)wgs * WSTAMP_N * sizeof(unsigned long long)) == hipSuccess ? M3_OK : M3_ERR_LAUNCH;
}
#endif
static inline bool sc_any(const m3_wgrad_args *a) { return a->c_row_scale != nullptr; }

extern "C" int m3_wgrad_skinny(int N, int K, int G) {
  static int on = -1;
  if (on < 0) { const char *e = getenv("M3_WGRAD_SKINNY"); on = e ? (atoi(e) ? 1 : 0) : 1; }
  return on && G == 1 && (K == 16 || K == 32) && N % 2 == 0 && N >= 2;
}

extern "C" int m3_wgrad_tn(const m3_wgrad_args *a, void *stream) {
  M3_REQUIRE(a && a->dC && a->A && (a->ws || a->direct_dW), "m3_wgrad_tn: null operand");
  M3_REQUIRE(dtype_ok(a->dtype), "m3_wgrad_tn: bad dtype");
  const int es = dtype_size(a->dtype);
  M3_REQUIRE(a->N > 0 && a->K > 0 && a->M >= 0 && a->G >= 1 && a->splits >= 1, "m3_wgrad_tn: bad shape");
  M3_REQUIRE(a->M < ((int64_t)1 << 31), "m3_wgrad_tn: M exceeds the 32-bit row indices");
  M3_REQUIRE((a->N * es) % 16 == 0 && (a->K * es) % 16 == 0, "m3_wgrad_tn: N*elem and K*elem must be multiples of 16 bytes");
  M3_REQUIRE((a->lddc * es) % 16 == 0 && (a->lda * es) % 16 == 0, "m3_wgrad_tn: rows must be 16-byte aligned");
  M3_REQUIRE(((uintptr_t)a->dC % 16) == 0 && ((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->ws % 16) == 0, "m3_wgrad_tn: alignment");
  M3_REQUIRE(!a->direct_dW || (a->splits == 1 && a->chunk_rows == 0 && ((uintptr_t)a->direct_dW % 16) == 0 && !a->bias_ws),
             "m3_wgrad_tn: direct mode needs splits == 1, no balanced units, a 16-byte aligned dW and no bias slabs");
  M3_REQUIRE(!a->direct_db || a->direct_dW, "m3_wgrad_tn: direct_db goes with direct_dW");
  M3_REQUIRE(a->G == 1 || a->group_offsets, "m3_wgrad_tn: grouped call needs group_offsets");
  M3_REQUIRE(!a->a_row_idx || a->a_row_div >= 1, "m3_wgrad_tn: a_row_div");
  WgradDev d;
  d.dC = (const char *)a->dC; d.lddc_b = a->lddc * es; d.c_row_idx = a->c_row_idx;
  M3_REQUIRE(a->c_row_idx || (!a->c_row_scale && a->c_row_div <= 1), "m3_wgrad_tn: c_row_div / c_row_scale need c_row_idx");
  M3_REQUIRE(a->c_row_div >= 0, "m3_wgrad_tn: c_row_div");
  d.c_row_div = (a->c_row_idx && a->c_row_div >= 1) ? a->c_row_div : 1;
  d.c_row_scale = a->c_row_scale;
  d.c_row_sh = div_shift(d.c_row_div);
  d.A = (const char *)a->A; d.lda_b = a->lda * es; d.a_row_idx = a->a_row_idx; d.a_row_div = a->a_row_idx ? a->a_row_div : 1;
  d.a_row_sh = div_shift(d.a_row_div);
  d.M = a->M; d.N = a->N; d.K = a->K; d.G = a->G; d.group_offsets = a->group_offsets;
  d.splits = a->splits; d.ws = a->ws; d.bias_ws = a->bias_ws;
  d.direct_dW = a->direct_dW; d.direct_db = a->direct_dW ? a->direct_db : nullptr;
  d.direct_beta = a->direct_beta; d.direct_beta_db = a->direct_beta_db;
  M3_REQUIRE(a->chunk_rows >= 0 && (a->chunk_rows == 0 || (a->group_offsets && a->chunk_rows % WG_ROWS == 0 && a->units >= 1 && a->G <= 64)),
             "m3_wgrad_tn: balanced mode needs group_offsets, G <= 64, chunk_rows a multiple of %d and units >= 1", WG_ROWS);
  d.chunk_rows = a->chunk_rows;
  hipStream_t s = (hipStream_t)stream;
  const bool gc = a->c_row_idx != nullptr, ga = a->a_row_idx != nullptr;
  // wide tiles (fp16): the shapes m3_wgrad_tile() names; the caller sized `splits` / `units` for that tile count
  int tn_w = 0, tk_w = 0;
  m3_wgrad_tile(a->N, a->K, a->dtype, &tn_w, &tk_w);
  // 256 x 256 tiles (16-bit ViT-Base weights): needs what the LDS-DMA form needs - power-of-two gather divisors, 32-bit lane
  // offsets, a per-row factor only on gathered fp16 rows; a call the tile rule names but the kernel cannot run falls through
  // to the 128 x 128 kernels (any kernel works with the caller's `splits`)
  const bool big_tile = tn_w == BG_T && tk_w == BG_T;
  const bool big = big_tile && d.a_row_sh >= 0 && d.c_row_sh >= 0 && (!a->c_row_scale || (a->c_row_idx && a->dtype == M3_F16)) &&
                   (a->M + 1) * d.lddc_b < ((int64_t)1 << 32) && (a->M + 1) * d.lda_b < ((int64_t)1 << 32);
  const bool wide = !big_tile && (tn_w != WG_T || tk_w != WG_T) && (d.a_row_div & (d.a_row_div - 1)) == 0 && !d.c_row_scale && d.c_row_div == 1 &&
                    !d.direct_dW;
  // the previous call's slab reduction (a->prev): in front of this launch (128 x 128 kernel), or as its own launch
  d.rd_blocks = 0; d.rd_zslices = 0; d.rd_cols = 256;
  static int lpt = -1;
  if (lpt < 0) { const char *e = getenv("M3_WGRAD_LPT"); lpt = e ? (atoi(e) ? 1 : 0) : 1; }
  d.lpt = lpt;
  if (a->prev) {
    const m3_wgrad_reduce_desc *r = a->prev;
    M3_REQUIRE(r->ws && r->dW && r->elems >= 0 && r->elems % 4 == 0 && (r->chunk_rows == 0 ? r->splits >= 1 : (r->group_offsets && r->G >= 1 && r->G <= 64)),
               "m3_wgrad_tn: bad prev reduce descriptor");
    M3_REQUIRE(!r->bias_ws || (r->db && r->bias_elems > 0 && r->bias_elems % 4 == 0), "m3_wgrad_tn: prev bias slabs need db");
    M3_REQUIRE(r->ws != a->ws, "m3_wgrad_tn: prev slabs and this call's slabs must be different buffers");
    if (r->elems > 0) {
      if (wide || big || a->M == 0) {
        int rc = r->chunk_rows ? m3_wgrad_reduce_grouped(r->ws, r->group_offsets, r->G, r->chunk_rows, r->elems, r->dW, r->beta, r->bias_ws,
                                                         r->bias_elems, r->db, r->beta_db, stream)
                               : m3_wgrad_reduce(r->ws, r->splits, r->elems, r->dW, r->beta, r->bias_ws, r->bias_elems, r->db, r->beta_db, stream);
        if (rc) return rc;
      } else {
        const int64_t e4 = r->elems / 4, b4 = r->bias_ws ? r->bias_elems / 4 : 0;
        const int rc_ = r->chunk_rows ? 256 : m3_wgrad_reduce_cols(r->splits);
        d.rd_cols = rc_;
        d.rd_nbw = (int)((e4 + rc_ - 1) / rc_);
        d.rd_nbx = d.rd_nbw + (int)((b4 + rc_ - 1) / rc_);
        d.rd_blocks = d.rd_nbx * (r->chunk_rows ? r->G : 1);
        d.rd_ws = r->ws; d.rd_splits = r->splits; d.rd_e4 = e4; d.rd_off = r->group_offsets; d.rd_G = r->G; d.rd_chunk = r->chunk_rows;
        d.rd_dW = r->dW; d.rd_beta = r->beta; d.rd_bws = r->bias_ws; d.rd_b4 = b4; d.rd_db = r->db; d.rd_beta_db = r->beta_db;
      }
    }
  }
#ifdef M3_EXPERIMENTAL
  if (wide) {
    constexpr int WR = M3_WGRAD_WIDE_ROWS;
    const bool wide_k = tk_w == 384;
    d.tiles_k = a->K / tk_w;
    const dim3 wgrid((a->N / tn_w) * d.tiles_k, a->chunk_rows ? a->units : a->G, a->chunk_rows ? 1 : a->splits), wblock(WW_THREADS);
    const size_t wl = 2 * WR * (WwCfg<true>::SC + WwCfg<true>::SA);
    static bool wattr = false;
#define M3_WW_ALL(F) F(true, true, true) F(true, true, false) F(true, false, true) F(true, false, false) \
                     F(false, true, true) F(false, true, false) F(false, false, true) F(false, false, false)
    if (!wattr) {
#define M3_WW_ATTR(GC_, GA_, WK_) (void)hipFuncSetAttribute((const void *)wgrad_wide_kernel<GC_, GA_, WK_, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl);
      M3_WW_ALL(M3_WW_ATTR)
#undef M3_WW_ATTR
      wattr = true;
    }
#define M3_WW_LAUNCH(GC_, GA_, WK_) if (gc == GC_ && ga == GA_ && wide_k == WK_) hipLaunchKernelGGL((wgrad_wide_kernel<GC_, GA_, WK_, WR>), wgrid, wblock, wl, s, d);
    M3_WW_ALL(M3_WW_LAUNCH)
#undef M3_WW_LAUNCH
#undef M3_WW_ALL
    return check_launch("m3_wgrad_tn");
  }
#endif
  if (big) {
    d.tiles_k = a->K / BG_T;
    const dim3 bgrid((a->N / BG_T) * d.tiles_k, a->chunk_rows ? a->units : a->G, a->chunk_rows ? 1 : a->splits), bblock(BG_THREADS);
    const bool bsc = a->c_row_scale != nullptr;
    static bool battr = false;
#define M3_BG_ALL(F) F(half_t, true, true, false) F(half_t, true, false, false) F(half_t, false, true, false) F(half_t, false, false, false) \
                     F(bf16_t, true, true, false) F(bf16_t, true, false, false) F(bf16_t, false, true, false) F(bf16_t, false, false, false) \
                     F(half_t, true, true, true) F(half_t, true, false, true)
    if (!battr) {
#define M3_BG_ATTR(TT, GC_, GA_, SC_) (void)hipFuncSetAttribute((const void *)wgrad_big_kernel<TT, GC_, GA_, SC_>, hipFuncAttributeMaxDynamicSharedMemorySize, BG_LDS);
      M3_BG_ALL(M3_BG_ATTR)
#undef M3_BG_ATTR
      battr = true;
    }
    const bool f16 = a->dtype == M3_F16;
#define M3_BG_LAUNCH(TT, GC_, GA_, SC_)                                                                          \
    if (f16 == std::is_same<TT, half_t>::value && gc == GC_ && ga == GA_ && bsc == SC_)                          \
      hipLaunchKernelGGL((wgrad_big_kernel<TT, GC_, GA_, SC_>), bgrid, bblock, BG_LDS, s, d);
    M3_BG_ALL(M3_BG_LAUNCH)
#undef M3_BG_LAUNCH
#undef M3_BG_ALL
    return check_launch("m3_wgrad_tn");
  }
  const int tiles_n = (a->N + WG_T - 1) / WG_T;
  d.tiles_k = (a->K + WG_T - 1) / WG_T;
  dim3 grid(tiles_n * d.tiles_k, a->chunk_rows ? a->units : a->G, a->chunk_rows ? 1 : a->splits), block(WG_THREADS);
  if (d.rd_blocks > 0) {                         // leading z slices for the previous call's reduce blocks
    const int per_slice = (int)(grid.x * grid.y);
    d.rd_zslices = (d.rd_blocks + per_slice - 1) / per_slice;
    grid.z += d.rd_zslices;
  }
  // the router's weight (K = 16 / 32, plain rows): the streaming kernel (m3_wgrad_skinny reports the rule to the caller,
  // who sizes `splits` for it)
  if (m3_wgrad_skinny(a->N, a->K, a->G) && !gc && !ga && !sc_any(a) && !a->bias_ws && !a->chunk_rows && !a->direct_dW) {
#define M3_WSK(TT)                                                                                  \
    do {                                                                                            \
      if (a->K == 16) hipLaunchKernelGGL((wgrad_skinny_kernel<TT, 16>), grid, block, 0, s, d);      \
      else hipLaunchKernelGGL((wgrad_skinny_kernel<TT, 32>), grid, block, 0, s, d);                 \
    } while (0)
    if (a->dtype == M3_F16) M3_WSK(half_t);
    else if (a->dtype == M3_BF16) M3_WSK(bf16_t);
    else M3_WSK(float);
#undef M3_WSK
    return check_launch("m3_wgrad_tn");
  }
  M3_REQUIRE(a->N * es >= 16 && a->K * es >= 16, "m3_wgrad_tn: N, K too small");
  const size_t lds16 = 4 * WgLds<half_t>::ROWS * WgLds<half_t>::STRIDE, lds32 = 4 * WgLds<float>::ROWS * WgLds<float>::STRIDE;
  const bool sc = a->c_row_scale != nullptr;
  // LDS-DMA variant (wgrad_dma_kernel): whole 16-byte column chunks on both sides, power-of-two gather divisors, a per-row
  // factor only with fp16 / fp32.  m3_wgrad_set_dma / M3_WGRAD_DMA: 0 never, 2 whenever the kernel can run the call, 1
  // (default) where it measured faster with operands streamed from HBM as inside the training step
  // (tools/wgrad_ab_bench.py, profiles/r05_wgrad_ab_streamed.txt): fp32 always (-12..-26 %); 16-bit when the launch has one
  // part per group anyway - tiles x groups fill the 1024 workgroup slots: direct accumulation, the ViT-Base experts, -35 % -
  // or the weight is large (N K >= 1.5 M elements: ViT-Base qkv / fc1 / fc2, -10..-24 %); at configs[1]'s 384-wide weights
  // the register-staged kernel is level or ahead (+-5 %) and keeps them
  if (g_wgrad_dma < 0) { const char *e = getenv("M3_WGRAD_DMA"); g_wgrad_dma = e ? atoi(e) : 1; }
  const bool dma_can = (!sc || (gc && a->dtype != M3_BF16)) && a->N * es >= 16 && a->K * es >= 16 && d.a_row_sh >= 0 && d.c_row_sh >= 0 &&
                       (a->M + 1) * d.lddc_b < ((int64_t)1 << 32) && (a->M + 1) * d.lda_b < ((int64_t)1 << 32);       // 32-bit lane offsets
  const bool dma_pays = es == 4 || (int64_t)a->N * a->K >= 1500000 || (int64_t)tiles_n * d.tiles_k * a->G >= 1024;
  if (dma_can && (g_wgrad_dma == 2 || (g_wgrad_dma == 1 && dma_pays))) {
    const size_t ldsd = 2 * 64 * WG_T * 2 + 256;       // 32 KiB + the step's per-row factors
#define M3_WD(TT)                                                                                    \
    do {                                                                                             \
      if (gc && ga) hipLaunchKernelGGL((wgrad_dma_kernel<TT, true, true>), grid, block, ldsd, s, d); \
      else if (gc) hipLaunchKernelGGL((wgrad_dma_kernel<TT, true, false>), grid, block, ldsd, s, d); \
      else if (ga) hipLaunchKernelGGL((wgrad_dma_kernel<TT, false, true>), grid, block, ldsd, s, d); \
      else hipLaunchKernelGGL((wgrad_dma_kernel<TT, false, false>), grid, block, ldsd, s, d);        \
    } while (0)
#define M3_WDS(TT)                                                                                          \
    do {                                                                                                    \
      if (ga) hipLaunchKernelGGL((wgrad_dma_kernel<TT, true, true, true>), grid, block, ldsd, s, d);         \
      else hipLaunchKernelGGL((wgrad_dma_kernel<TT, true, false, true>), grid, block, ldsd, s, d);          \
    } while (0)
    if (sc) {
      if (a->dtype == M3_F16) M3_WDS(half_t);
      else M3_WDS(float);
    } else if (a->dtype == M3_F16) M3_WD(half_t);
    else if (a->dtype == M3_BF16) M3_WD(bf16_t);
    else M3_WD(float);
#undef M3_WD
#undef M3_WDS
    return check_launch("m3_wgrad_tn");
  }
#define M3_WG(TT, LDS)                                                                               \
  do {                                                                                               \
    if (sc && ga) hipLaunchKernelGGL((wgrad_tn_kernel<TT, true, true, true>), grid, block, LDS, s, d);    \
    else if (sc) hipLaunchKernelGGL((wgrad_tn_kernel<TT, true, false, true>), grid, block, LDS, s, d);    \
    else if (gc && ga) hipLaunchKernelGGL((wgrad_tn_kernel<TT, true, true>), grid, block, LDS, s, d); \
    else if (gc) hipLaunchKernelGGL((wgrad_tn_kernel<TT, true, false>), grid, block, LDS, s, d);     \
    else if (ga) hipLaunchKernelGGL((wgrad_tn_kernel<TT, false, true>), grid, block, LDS, s, d);     \
    else hipLaunchKernelGGL((wgrad_tn_kernel<TT, false, false>), grid, block, LDS, s, d);            \
  } while (0)
  static bool attr_set = false;
  if (!attr_set) {                               // both images exceed the 64 KiB a launch gets without asking
#define M3_WG_ATTR(TT, LDS)                                                                                                             \
    (void)hipFuncSetAttribute((const void *)wgrad_tn_kernel<TT, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);      \
    (void)hipFuncSetAttribute((const void *)wgrad_tn_kernel<TT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);     \
    (void)hipFuncSetAttribute((const void *)wgrad_tn_kernel<TT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);     \
    (void)hipFuncSetAttribute((const void *)wgrad_tn_kernel<TT, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);    \
    (void)hipFuncSetAttribute((const void *)wgrad_tn_kernel<TT, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); \
    (void)hipFuncSetAttribute((const void *)wgrad_tn_kernel<TT, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    M3_WG_ATTR(float, lds32)
    M3_WG_ATTR(bf16_t, lds16)
    M3_WG_ATTR(half_t, lds16)
#undef M3_WG_ATTR
    attr_set = true;
  }
  if (a->dtype == M3_F16) M3_WG(half_t, lds16);
  else if (a->dtype == M3_BF16) M3_WG(bf16_t, lds16);
  else M3_WG(float, lds32);
#undef M3_WG
  return check_launch("m3_wgrad_tn");
}

extern "C" int m3_wgrad_reduce(const float *ws, int splits, int64_t elems, float *dW, int beta, const float *bias_ws,
                               int64_t bias_elems, float *db, int beta_db, void *stream) {
  M3_REQUIRE(ws && dW && splits >= 1 && elems >= 0 && elems % 4 == 0, "m3_wgrad_reduce: bad args");
  M3_REQUIRE(((uintptr_t)ws % 16) == 0 && ((uintptr_t)dW % 16) == 0, "m3_wgrad_reduce: alignment");
  M3_REQUIRE(!bias_ws || (db && bias_elems > 0 && bias_elems % 4 == 0 && ((uintptr_t)bias_ws % 16) == 0 &&
                          ((uintptr_t)db % 16) == 0),
             "m3_wgrad_reduce: bias slabs need db, 16-byte alignment and a multiple of 4 elements");
  if (elems == 0) return M3_OK;
  const int64_t e4 = elems / 4, b4 = bias_ws ? bias_elems / 4 : 0;
  const int cols = m3_wgrad_reduce_cols(splits);
  const int nb_w = (int)((e4 + cols - 1) / cols), nb_b = (int)((b4 + cols - 1) / cols);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(nb_w + nb_b)), dim3(256), 0, (hipStream_t)stream, ws,
                     splits, e4, dW, beta, nb_w, bias_ws, b4, db, beta_db, cols);
  return check_launch("m3_wgrad_reduce");
}

extern "C" int m3_wgrad_reduce_grouped(const float *ws, const int32_t *group_offsets, int G, int chunk_rows, int64_t elems,
                                       float *dW, int beta, const float *bias_ws, int64_t bias_elems, float *db,
                                       int beta_db, void *stream) {
  M3_REQUIRE(ws && dW && group_offsets && G >= 1 && G <= 64 && chunk_rows >= 1 && elems >= 0 && elems % 4 == 0,
             "m3_wgrad_reduce_grouped: bad args");
  M3_REQUIRE(((uintptr_t)ws % 16) == 0 && ((uintptr_t)dW % 16) == 0, "m3_wgrad_reduce_grouped: alignment");
  M3_REQUIRE(!bias_ws || (db && bias_elems > 0 && bias_elems % 4 == 0 && ((uintptr_t)bias_ws % 16) == 0 &&
                          ((uintptr_t)db % 16) == 0),
             "m3_wgrad_reduce_grouped: bias slabs need db, 16-byte alignment and a multiple of 4 elements");
  if (elems == 0) return M3_OK;
  const int64_t e4 = elems / 4, b4 = bias_ws ? bias_elems / 4 : 0;
  const int nb_w = (int)((e4 + 255) / 256), nb_b = (int)((b4 + 255) / 256);
  hipLaunchKernelGGL(wgrad_reduce_grouped_kernel, dim3((unsigned)(nb_w + nb_b), (unsigned)G), dim3(256), 0,
                     (hipStream_t)stream, ws, group_offsets, G, chunk_rows, e4, dW, beta, nb_w, bias_ws, b4, db, beta_db);
  return check_launch("m3_wgrad_reduce_grouped");
}

extern "C" int m3_wgrad_bias_reduce(const float *bias_ws, int splits, int64_t elems, float *db, int beta, void *stream) {
  M3_REQUIRE(bias_ws && db && splits >= 1 && elems > 0 && elems < ((int64_t)1 << 31), "m3_wgrad_bias_reduce: bad args");
  return launch_reduce_rows_f32(bias_ws, splits, (int)elems, 1, 0, db, beta, (hipStream_t)stream);
}

extern "C" int64_t m3_colsum_ws_elems(int64_t M, int N, int G) {
  return (int64_t)G * colsum_strips_per_group(M, G) * N;
}

extern "C" int m3_colsum(const void *dC, int dtype, int64_t lddc, const int32_t *c_row_idx, int64_t M, int N, int G,
                         const int32_t *group_offsets, float *ws, float *db, int beta, void *stream) {
  M3_REQUIRE(dC && ws && db, "m3_colsum: null operand");
  M3_REQUIRE(dtype_ok(dtype), "m3_colsum: bad dtype");
  M3_REQUIRE(N % 4 == 0 && lddc % 4 == 0 && G >= 1, "m3_colsum: N, lddc must be multiples of 4");
  M3_REQUIRE(G == 1 || group_offsets, "m3_colsum: grouped call needs group_offsets");
  M3_REQUIRE((N * dtype_size(dtype)) % 16 == 0 && (lddc * dtype_size(dtype)) % 16 == 0 && ((uintptr_t)dC % 16) == 0,
             "m3_colsum: rows must be 16-byte aligned and N*elem a multiple of 16");
  const int spg = colsum_strips_per_group(M, G);
  const int es = dtype_size(dtype);
  hipStream_t s = (hipStream_t)stream;
  const int chunks = N * es / 16;
  const dim3 grid((chunks + 63) / 64, spg, G), block(256);
  if (dtype == M3_F16)
    hipLaunchKernelGGL(colsum_part_kernel<half_t>, grid, block, 0, s, (const char *)dC, lddc * es, c_row_idx, M, N,
                       group_offsets, spg, ws);
  else if (dtype == M3_BF16)
    hipLaunchKernelGGL(colsum_part_kernel<bf16_t>, grid, block, 0, s, (const char *)dC, lddc * es, c_row_idx, M, N,
                       group_offsets, spg, ws);
  else
    hipLaunchKernelGGL(colsum_part_kernel<float>, grid, block, 0, s, (const char *)dC, lddc * es, c_row_idx, M, N,
                       group_offsets, spg, ws);
  int rc = check_launch("m3_colsum(part)");
  if (rc) return rc;
  return launch_reduce_rows_f32(ws, spg, N, G, (int64_t)spg * N, db, beta, s);
}
