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

namespace m3 {

constexpr int WG_T = 128;        // tile edge (n and k)
constexpr int WG_ROWS = 32;      // contraction rows per step
constexpr int WG_THREADS = 256;

struct WgradDev {
  const char *dC; int64_t lddc_b; const int32_t *c_row_idx;
  const char *A; int64_t lda_b; const int32_t *a_row_idx; int32_t a_row_div;
  int64_t M; int32_t N; int32_t K; int32_t G;
  const int32_t *group_offsets;
  int32_t splits;
  float *ws;
  int32_t tiles_k;
};

template <typename T> struct WgLds;
template <> struct WgLds<half_t> { static constexpr int STRIDE = 288; };
template <> struct WgLds<float> { static constexpr int STRIDE = 528; };

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
__device__ __forceinline__ f32x4 read_tr_frag<float>(const char *base, int rb, int col, int li, int lg) {
  const char *p = base + (rb + 4 * lg) * WgLds<float>::STRIDE + (col + li) * 4;
  f32x4 f;
  f[0] = *(const float *)(p);
  f[1] = *(const float *)(p + WgLds<float>::STRIDE);
  f[2] = *(const float *)(p + 2 * WgLds<float>::STRIDE);
  f[3] = *(const float *)(p + 3 * WgLds<float>::STRIDE);
  return f;
}

template <typename T>
__global__ __launch_bounds__(WG_THREADS, 2) void wgrad_tn_kernel(const WgradDev p) {
  typedef Mma<T> MM;
  typedef typename MM::frag frag;
  constexpr int ES = (int)sizeof(T);
  constexpr int STRIDE = WgLds<T>::STRIDE;
  constexpr int CPR = WG_T * ES / 16;                       // 16-byte chunks per tile row
  constexpr int NLD = WG_ROWS * CPR / WG_THREADS;           // chunks per thread per operand
  constexpr int OPB = WG_ROWS * STRIDE;                     // bytes per operand image
  constexpr int KCH = WG_ROWS / MM::KC;                     // fragments chunks per step

  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto sC = [&](int buf) -> char * { return smem + buf * (2 * OPB); };
  auto sA = [&](int buf) -> char * { return smem + buf * (2 * OPB) + OPB; };

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;

  const int tn = blockIdx.x / p.tiles_k, tk = blockIdx.x - tn * p.tiles_k;
  const int g = blockIdx.y, sp = blockIdx.z;
  const int n0 = tn * WG_T, k0 = tk * WG_T;

  int64_t r0, r1;
  if (p.group_offsets) { r0 = p.group_offsets[g]; r1 = p.group_offsets[g + 1]; }
  else { r0 = 0; r1 = p.M; }
  const int64_t nsteps_all = (r1 - r0 + WG_ROWS - 1) / WG_ROWS;
  const int64_t per = (nsteps_all + p.splits - 1) / p.splits;
  const int64_t s_begin = (int64_t)sp * per;
  int64_t s_end = s_begin + per;
  if (s_end > nsteps_all) s_end = nsteps_all;

  f32x4 acc[4][4];   // [ki][ni]: MFMA rows = k, cols = n
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 rc[NLD], ra[NLD];
  auto load_global = [&](int64_t step) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = tid + WG_THREADS * i;
      const int row = q / CPR, c = q - row * CPR;
      const int64_t m = r0 + step * WG_ROWS + row;
      const bool mok = m < r1;
      const int ncol = n0 + c * (16 / ES), kcol = k0 + c * (16 / ES);
      u32x4 vc = u32x4{0u, 0u, 0u, 0u}, va = u32x4{0u, 0u, 0u, 0u};
      if (mok) {
        if (ncol < p.N) {
          const int64_t cr = p.c_row_idx ? (int64_t)p.c_row_idx[m] : m;
          vc = *(const u32x4 *)(p.dC + cr * p.lddc_b + (int64_t)ncol * ES);
        }
        if (kcol < p.K) {
          const int64_t ar = p.a_row_idx ? (int64_t)(p.a_row_idx[m] / p.a_row_div) : m;
          va = *(const u32x4 *)(p.A + ar * p.lda_b + (int64_t)kcol * ES);
        }
      }
      rc[i] = vc; ra[i] = va;
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = tid + WG_THREADS * i;
      const int row = q / CPR, c = q - row * CPR;
      *(u32x4 *)(sC(buf) + row * STRIDE + c * 16) = rc[i];
      *(u32x4 *)(sA(buf) + row * STRIDE + c * 16) = ra[i];
    }
  };

  if (s_begin < s_end) {
    load_global(s_begin);
    store_lds(0);
    __syncthreads();
    int buf = 0;
    for (int64_t st = s_begin; st < s_end; ++st) {
      if (st + 1 < s_end) load_global(st + 1);
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) {
        frag fk[4], fn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fk[i] = read_tr_frag<T>(sA(buf), kc * MM::KC, wr * 64 + i * 16, li, lg);
          fn[i] = read_tr_frag<T>(sC(buf), kc * MM::KC, wc * 64 + i * 16, li, lg);
        }
#pragma unroll
        for (int ki = 0; ki < 4; ++ki)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[ki][ni] = MM::mma(fk[ki], fn[ni], acc[ki][ni]);
      }
      if (st + 1 < s_end) store_lds(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // slab[sp][g][n][k]; lane holds k = kb + 4*lg + r, n = nb + li
  float *slab = p.ws + ((int64_t)sp * p.G + g) * (int64_t)p.N * p.K;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = n0 + wc * 64 + ni * 16 + li;
    if (n >= p.N) continue;
#pragma unroll
    for (int ki = 0; ki < 4; ++ki) {
      const int k = k0 + wr * 64 + ki * 16 + 4 * lg;
      if (k >= p.K) continue;
      *(f32x4 *)(slab + (int64_t)n * p.K + k) = acc[ki][ni];
    }
  }
}

__global__ void wgrad_reduce_kernel(const float *ws, int splits, int64_t elems4, float *dW, int beta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems4) return;
  f32x4 s = beta ? ((const f32x4 *)dW)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int sp = 0; sp < splits; ++sp) s += ((const f32x4 *)ws)[(int64_t)sp * elems4 + i];
  ((f32x4 *)dW)[i] = s;
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

extern "C" int m3_wgrad_tn(const m3_wgrad_args *a, void *stream) {
  M3_REQUIRE(a && a->dC && a->A && a->ws, "m3_wgrad_tn: null operand");
  M3_REQUIRE(a->dtype == M3_F32 || a->dtype == M3_F16, "m3_wgrad_tn: bad dtype");
  const int es = dtype_size(a->dtype);
  M3_REQUIRE(a->N > 0 && a->K > 0 && a->M >= 0 && a->G >= 1 && a->splits >= 1, "m3_wgrad_tn: bad shape");
  M3_REQUIRE((a->N * es) % 16 == 0 && (a->K * es) % 16 == 0, "m3_wgrad_tn: N*elem and K*elem must be multiples of 16 bytes");
  M3_REQUIRE((a->lddc * es) % 16 == 0 && (a->lda * es) % 16 == 0, "m3_wgrad_tn: rows must be 16-byte aligned");
  M3_REQUIRE(((uintptr_t)a->dC % 16) == 0 && ((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->ws % 16) == 0, "m3_wgrad_tn: alignment");
  M3_REQUIRE(a->G == 1 || a->group_offsets, "m3_wgrad_tn: grouped call needs group_offsets");
  M3_REQUIRE(!a->a_row_idx || a->a_row_div >= 1, "m3_wgrad_tn: a_row_div");
  WgradDev d;
  d.dC = (const char *)a->dC; d.lddc_b = a->lddc * es; d.c_row_idx = a->c_row_idx;
  d.A = (const char *)a->A; d.lda_b = a->lda * es; d.a_row_idx = a->a_row_idx; d.a_row_div = a->a_row_idx ? a->a_row_div : 1;
  d.M = a->M; d.N = a->N; d.K = a->K; d.G = a->G; d.group_offsets = a->group_offsets;
  d.splits = a->splits; d.ws = a->ws;
  const int tiles_n = (a->N + WG_T - 1) / WG_T;
  d.tiles_k = (a->K + WG_T - 1) / WG_T;
  const dim3 grid(tiles_n * d.tiles_k, a->G, a->splits), block(WG_THREADS);
  hipStream_t s = (hipStream_t)stream;
  if (a->dtype == M3_F16) {
    hipLaunchKernelGGL(wgrad_tn_kernel<half_t>, grid, block, 4 * WG_ROWS * WgLds<half_t>::STRIDE, s, d);
  } else {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void *)wgrad_tn_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                4 * WG_ROWS * WgLds<float>::STRIDE);
      attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_tn_kernel<float>, grid, block, 4 * WG_ROWS * WgLds<float>::STRIDE, s, d);
  }
  return check_launch("m3_wgrad_tn");
}

extern "C" int m3_wgrad_reduce(const float *ws, int splits, int64_t elems, float *dW, int beta, void *stream) {
  M3_REQUIRE(ws && dW && splits >= 1 && elems >= 0 && elems % 4 == 0, "m3_wgrad_reduce: bad args");
  M3_REQUIRE(((uintptr_t)ws % 16) == 0 && ((uintptr_t)dW % 16) == 0, "m3_wgrad_reduce: alignment");
  if (elems == 0) return M3_OK;
  const int64_t e4 = elems / 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((e4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws,
                     splits, e4, dW, beta);
  return check_launch("m3_wgrad_reduce");
}

extern "C" int64_t m3_colsum_ws_elems(int64_t M, int N, int G) {
  return (int64_t)G * colsum_strips_per_group(M, G) * N;
}

extern "C" int m3_colsum(const void *dC, int dtype, int64_t lddc, const int32_t *c_row_idx, int64_t M, int N, int G,
                         const int32_t *group_offsets, float *ws, float *db, int beta, void *stream) {
  M3_REQUIRE(dC && ws && db, "m3_colsum: null operand");
  M3_REQUIRE(dtype == M3_F32 || dtype == M3_F16, "m3_colsum: bad dtype");
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
  else
    hipLaunchKernelGGL(colsum_part_kernel<float>, grid, block, 0, s, (const char *)dC, lddc * es, c_row_idx, M, N,
                       group_offsets, spg, ws);
  int rc = check_launch("m3_colsum(part)");
  if (rc) return rc;
  return launch_reduce_rows_f32(ws, spg, N, G, (int64_t)spg * N, db, beta, s);
}
