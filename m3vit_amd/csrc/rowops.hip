// HBM-bound row kernels of the MoE-ViT block for gfx950: weighted combine of the k expert
// outputs (+ residual), LayerNorm forward/backward on the fp32 residual stream, and the
// small cast / patchify helpers.  One wave owns one token row; every access is a
// 16-byte (fp32) or 8-byte (f16) vector per lane, fully coalesced along D.
//
//   combine : bmm(gate_score[T,1,k], moe_outp[T,k,D]), models/moe/ckpt/custom_moe_layer.py:298-305,
//             fused with x = x + moe_output, models/moe/ckpt/vision_transformer_moe.py:450
//   layernorm: norm1/norm2 = nn.LayerNorm(eps 1e-6), vision_transformer_moe.py:441-442,567
//   im2row / assemble_tokens: PatchEmbed + cls/pos, vision_transformer_moe.py:330-341,782-791
#include "common.h"

namespace m3 {

constexpr int ROW_THREADS = 256;   // 4 waves = 4 rows per workgroup

// ------------------------------------------------------------------- combine
// KT: top-k as a template constant (1, 2, 4, 8; 0 = run-time k).  With a run-time trip count the k row loads of a token
// are issued one dependent iteration at a time; unrolled, all k rows of a 16-byte column are in flight together.
template <typename T, int KT>
__global__ __launch_bounds__(ROW_THREADS) void combine_fwd_kernel(const T *__restrict__ y, const float *__restrict__ score,
                                                                  const float *__restrict__ residual, int64_t T_,
                                                                  int k_rt, int D, float *__restrict__ out) {
  const int k = KT ? KT : k_rt;
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * (ROW_THREADS / 64) + (threadIdx.x >> 6);
  if (t >= T_) return;
  const float *sc = score + t * k;
  if constexpr (KT > 0) {
    float sv[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) sv[j] = sc[j];
    for (int d = lane * 4; d < D; d += 256) {
      f32x4 v[KT];
#pragma unroll
      for (int j = 0; j < KT; ++j) v[j] = Vec4<T>::load(y + (t * KT + j) * D + d);
      f32x4 acc = residual ? *(const f32x4 *)(residual + t * D + d) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < KT; ++j) {                       // (same order as the run-time loop: j = 0, 1, ...)
        acc[0] = __builtin_fmaf(sv[j], v[j][0], acc[0]); acc[1] = __builtin_fmaf(sv[j], v[j][1], acc[1]);
        acc[2] = __builtin_fmaf(sv[j], v[j][2], acc[2]); acc[3] = __builtin_fmaf(sv[j], v[j][3], acc[3]);
      }
      *(f32x4 *)(out + t * D + d) = acc;
    }
  } else {
    for (int d = lane * 4; d < D; d += 256) {
      f32x4 acc = residual ? *(const f32x4 *)(residual + t * D + d) : f32x4{0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < k; ++j) {
        const f32x4 v = Vec4<T>::load(y + (t * k + j) * D + d);
        const float s = sc[j];
        acc[0] = __builtin_fmaf(s, v[0], acc[0]); acc[1] = __builtin_fmaf(s, v[1], acc[1]);
        acc[2] = __builtin_fmaf(s, v[2], acc[2]); acc[3] = __builtin_fmaf(s, v[3], acc[3]);
      }
      *(f32x4 *)(out + t * D + d) = acc;
    }
  }
}

template <typename T, int KT>
__global__ __launch_bounds__(ROW_THREADS) void combine_bwd_kernel(const float *__restrict__ dout, const T *__restrict__ y,
                                                                  const float *__restrict__ score, int64_t T_, int k_rt,
                                                                  int D, T *__restrict__ dy, float *__restrict__ dscore) {
  const int k = KT ? KT : k_rt;
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * (ROW_THREADS / 64) + (threadIdx.x >> 6);
  if (t >= T_) return;
  if constexpr (KT > 0) {
    // one pass over the token's gradient row for all k routed rows: dout is read once, the k dot products run side by side
    float dot[KT], sv[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) { dot[j] = 0.f; sv[j] = score[t * KT + j]; }
    for (int d = lane * 4; d < D; d += 256) {
      const f32x4 g = *(const f32x4 *)(dout + t * D + d);
      f32x4 v[KT];
#pragma unroll
      for (int j = 0; j < KT; ++j) v[j] = Vec4<T>::load(y + (t * KT + j) * D + d);
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        dot[j] += g[0] * v[j][0] + g[1] * v[j][1] + g[2] * v[j][2] + g[3] * v[j][3];
        if (dy) Vec4<T>::store(dy + (t * KT + j) * D + d, f32x4{sv[j] * g[0], sv[j] * g[1], sv[j] * g[2], sv[j] * g[3]});
      }
    }
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const float s = wave_sum(dot[j]);
      if (lane == 0) dscore[t * KT + j] = s;
    }
  } else {
    for (int j = 0; j < k; ++j) {
      const float s = score[t * k + j];
      float dot = 0.f;
      for (int d = lane * 4; d < D; d += 256) {
        const f32x4 g = *(const f32x4 *)(dout + t * D + d);
        const f32x4 v = Vec4<T>::load(y + (t * k + j) * D + d);
        dot += g[0] * v[0] + g[1] * v[1] + g[2] * v[2] + g[3] * v[3];
        if (dy) Vec4<T>::store(dy + (t * k + j) * D + d, f32x4{s * g[0], s * g[1], s * g[2], s * g[3]});
      }
      dot = wave_sum(dot);
      if (lane == 0) dscore[t * k + j] = dot;
    }
  }
}

// ----------------------------------------------------------------- layernorm
template <typename T, int NCH>
__global__ __launch_bounds__(ROW_THREADS) void layernorm_fwd_kernel(const float *__restrict__ x, int64_t T_, int D,
                                                                    const float *__restrict__ gamma,
                                                                    const float *__restrict__ beta, float eps,
                                                                    T *__restrict__ y, float *__restrict__ mean,
                                                                    float *__restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * (ROW_THREADS / 64) + (threadIdx.x >> 6);
  if (t >= T_) return;
  const float *xr = x + t * D;
  // NCH = ceil(D / 256) 16-byte vectors per lane, kept in registers between the passes.  Every load of the row - and gamma /
  // beta, which are only needed after the two reductions - is unconditional (lanes past D re-read column 0 and contribute
  // zeros) and issued up front: a load under `if (d < D)` is its own basic block with its own wait (see layernorm_bwd_kernel).
  f32x4 v[NCH], g[NCH], b[NCH];
  bool on[NCH];
  int col[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    on[i] = lane * 4 + i * 256 < D;
    col[i] = on[i] ? lane * 4 + i * 256 : 0;
    v[i] = *(const f32x4 *)(xr + col[i]);
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    g[i] = *(const f32x4 *)(gamma + col[i]);
    b[i] = *(const f32x4 *)(beta + col[i]);
  }
  __builtin_amdgcn_sched_barrier(0);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (!on[i]) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (on[i]) {
      const f32x4 c = v[i] - mu;
      q += c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
    }
  }
  const float var = wave_sum(q) / (float)D;
  const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (on[i]) {
      const f32x4 o = (v[i] - mu) * rs * g[i] + b[i];
      Vec4<T>::store(y + t * D + col[i], o);
    }
  }
  if (lane == 0) { mean[t] = mu; rstd[t] = rs; }
}

// dx = dx_res + rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)); per-block partial
// dgamma/dbeta (rows of a block summed in row order).
// Geometry: ln_bwd_waves() waves per workgroup (8; M3_LN_WAVES overrides it for tuning), each wave owns `rpw` consecutive
// rows (default LNB_RPW; M3_LN_ROWS).  Measured at T = 25 216, D = 384 (tools/ln_bench.py, operands streamed; then the
// two-stream training step, same box): 4 waves x 1 / 2 / 4 / 8 / 16 rows 34.7 / 36.7 / 41.7 / 39.7 / 60.5 us per launch,
// 16 waves x 1 / 2 / 4 rows 42.1 / 41.4 / 40.5 us, 8 waves x 8 rows 38.0 us.  Fewer rows per wave stream faster by
// themselves but leave one dgamma / dbeta partial row per workgroup (6304 rows x 2 x D floats at one row per wave: 12 % of
// the kernel's bytes), and 1024-thread workgroups cannot be placed beside the other task stream's kernels (the step went
// 18.16 -> 18.57 ms with them); 8 x 8 is the fastest INSIDE the step.  The partials are summed by ONE batched launch for
// many layers (m3_layernorm_bwd_reduce) instead of a 24-workgroup reduce launch behind every LayerNorm backward.
constexpr int LNB_RPW = 8;                 // default rows per wave
static inline int ln_bwd_waves(int D) {
  static int forced = -1;                    // M3_LN_WAVES = 4 / 8 / 16 (tuning; 16 only while the image fits 64 KiB)
  if (forced < 0) { const char *e = getenv("M3_LN_WAVES"); forced = e ? atoi(e) : 0; }
  if (forced == 4 || forced == 8 || (forced == 16 && D <= 512)) return forced;
  return 8;
}
static int ln_rows_per_wave() {
  static int rpw = 0;
  if (rpw == 0) {
    const char *e = getenv("M3_LN_ROWS");
    const int v = e ? atoi(e) : LNB_RPW;
    rpw = (v >= 1 && v <= 64) ? v : LNB_RPW;
  }
  return rpw;
}

// NCH = ceil(D / 256): 16-byte chunks per lane (registers are sized for the row width in use: D = 384 -> 2, not 4,
// which takes the kernel from 116 to ~70 VGPRs and from 4 to 7 waves per SIMD)
template <typename T, typename TA, int NCH, int LNB_WAVES>
__global__ __launch_bounds__(LNB_WAVES * 64) void layernorm_bwd_kernel(const T *__restrict__ dy, const float *__restrict__ x,
                                                                    const float *__restrict__ mean,
                                                                    const float *__restrict__ rstd,
                                                                    const float *__restrict__ gamma,
                                                                    const float *__restrict__ dx_res, int64_t T_, int D,
                                                                    float *__restrict__ dx, float *__restrict__ part,
                                                                    TA *__restrict__ dx_act, int rpw) {
  extern __shared__ float sred[];   // [LNB_WAVES][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nblk = gridDim.x;
  f32x4 dg[NCH], db[NCH], gam[NCH];
  // lanes past D in the last 256-column chunk read column 0 again, contribute zeros and never store: every load of a row is
  // unconditional and issued before the first use.  (A load under `if (d < D)` is its own basic block with its own wait: the
  // row's chunks, and the residual behind them, then cost one memory latency EACH - four per row at D = 384 - and a wave's
  // rows run one after the other.)
  bool on[NCH];
  int col[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    on[i] = lane * 4 + i * 256 < D;
    col[i] = on[i] ? lane * 4 + i * 256 : 0;
    dg[i] = f32x4{0.f, 0.f, 0.f, 0.f}; db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    gam[i] = *(const f32x4 *)(gamma + col[i]);
  }
  // (two rows side by side - their loads in flight together, their shuffle chains interleaved - was measured in round 3:
  // 41.2 us against 38.0 us for this loop at T = 25 216, D = 384: the second row's registers cost more occupancy than the
  // overlap wins)
  for (int r = 0; r < rpw; ++r) {
    const int64_t t = ((int64_t)blockIdx.x * LNB_WAVES + wave) * rpw + r;
    if (t >= T_) break;
    const float mu = mean[t], rs = rstd[t];
    typename Vec4<T>::type dyr[NCH];
    f32x4 xr[NCH], rr[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      dyr[i] = *(const typename Vec4<T>::type *)(dy + t * D + col[i]);
      xr[i] = *(const f32x4 *)(x + t * D + col[i]);
    }
    if (dx_res) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) rr[i] = *(const f32x4 *)(dx_res + t * D + col[i]);
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) rr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 gdy[NCH], xh[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      f32x4 dyv = f32x4{(float)dyr[i][0], (float)dyr[i][1], (float)dyr[i][2], (float)dyr[i][3]};
      if (!on[i]) dyv = f32x4{0.f, 0.f, 0.f, 0.f};
      xh[i] = (xr[i] - mu) * rs;
      gdy[i] = dyv * gam[i];
      s1 += gdy[i][0] + gdy[i][1] + gdy[i][2] + gdy[i][3];
      s2 += gdy[i][0] * xh[i][0] + gdy[i][1] * xh[i][1] + gdy[i][2] * xh[i][2] + gdy[i][3] * xh[i][3];
      dg[i] += dyv * xh[i];
      db[i] += dyv;
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (on[i]) {
        const f32x4 o = (gdy[i] - m1 - xh[i] * m2) * rs + rr[i];
        *(f32x4 *)(dx + t * D + col[i]) = o;
        if (dx_act) Vec4<TA>::store(dx_act + t * D + col[i], o);   // activation-dtype copy for the next GEMMs
      }
    }
  }
  // block partials: waves in order
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int d = lane * 4 + i * 256;
    if (d < D) {
      *(f32x4 *)(sred + (wave * 2 + 0) * D + d) = dg[i];
      *(f32x4 *)(sred + (wave * 2 + 1) * D + d) = db[i];
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += LNB_WAVES * 64) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < LNB_WAVES; ++w) { a += sred[(w * 2 + 0) * D + d]; b += sred[(w * 2 + 1) * D + d]; }
    part[(int64_t)blockIdx.x * D + d] = a;
    part[((int64_t)nblk + blockIdx.x) * D + d] = b;
  }
}

// ------------------------------------------------------------------ cast helpers
template <typename T>
__global__ void cast_matrix_kernel(const float *__restrict__ src, int rows, int cols, int transpose, T *__restrict__ dst) {
  // grid.z = group; tile 32x32 through LDS when transposing
  __shared__ float tile[32][33];
  const int64_t goff = (int64_t)blockIdx.z * rows * cols;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  if (!transpose) {
    for (int i = ty; i < 32; i += 8) {
      const int r = r0 + i, c = c0 + tx;
      if (r < rows && c < cols) dst[goff + (int64_t)r * cols + c] = (T)src[goff + (int64_t)r * cols + c];
    }
    return;
  }
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[goff + (int64_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;   // dst[c][r]
    if (r < rows && c < cols) dst[goff + (int64_t)c * rows + r] = (T)tile[tx][i];
  }
}

// many matrices in one launch: block -> descriptor by binary search over the tile prefix; a job may ask for the
// plain copy, the transposed copy, or both from ONE read of the fp32 tile
struct CastDesc {            // = m3_cast_desc
  const float *src; void *dst; void *dst_t;
  int32_t G, rows, cols;
  int32_t tile_start, flags, pad1;
};

template <typename T>
__global__ __launch_bounds__(256) void cast_batch_kernel(const CastDesc *__restrict__ descs, int n_desc) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n_desc - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_start <= b) lo = mid; else hi = mid - 1;
  }
  const CastDesc d = descs[lo];
  const int tcols = (d.cols + 31) / 32, trows = (d.rows + 31) / 32;
  int rest = b - d.tile_start;
  const int g = rest / (tcols * trows);
  rest -= g * tcols * trows;
  const int r0 = (rest / tcols) * 32, c0 = (rest % tcols) * 32;
  const int64_t goff = (int64_t)g * d.rows * d.cols;
  const float *src = d.src + goff;
  if (((d.rows | d.cols) & 3) == 0) {
    // rows and columns multiples of 4 (every weight of the model): one 16-byte load per thread (8 threads x 32 rows), 8-byte
    // stores for the plain copy and - through the LDS tile - for the transposed one.  M3_CAST_PERM32 moves whole groups
    // of four (source group 4b + a of an aligned 32 -> position group 2a + b), so it acts on the group index.
    const int tr = threadIdx.x >> 3, tg = threadIdx.x & 7;
    const int tgp = 2 * (tg & 3) + (tg >> 2);
    const int r = r0 + tr, c = c0 + 4 * tg;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (r < d.rows && c < d.cols) v = *(const f32x4 *)(src + (int64_t)r * d.cols + c);
    if (d.dst && r < d.rows && c < d.cols)
      Vec4<T>::store((T *)d.dst + goff + (int64_t)r * d.cols + c0 + 4 * ((d.flags & 1) ? tgp : tg), v);
    tile[tr][4 * tg + 0] = v[0]; tile[tr][4 * tg + 1] = v[1]; tile[tr][4 * tg + 2] = v[2]; tile[tr][4 * tg + 3] = v[3];
    if (!d.dst_t) return;
    __syncthreads();
    const int cc = c0 + tr, rg = r0 + 4 * tg;                        // dst_t[cc][rg .. rg + 3] = src[rg .. rg + 3][cc]
    if (cc < d.cols && rg < d.rows) {
      const f32x4 w = f32x4{tile[4 * tg + 0][tr], tile[4 * tg + 1][tr], tile[4 * tg + 2][tr], tile[4 * tg + 3][tr]};
      Vec4<T>::store((T *)d.dst_t + goff + (int64_t)cc * d.rows + r0 + 4 * ((d.flags & 2) ? tgp : tg), w);
    }
    return;
  }
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  // M3_CAST_PERM32: source index q = 16b + 4a + c of an aligned 32-group goes to position 8a + 4b + c
  const int txp = ((tx >> 2) & 3) * 8 + (tx >> 4) * 4 + (tx & 3);
  const int tx_d = (d.flags & 1) ? txp : tx, tx_t = (d.flags & 2) ? txp : tx;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    const float v = (r < d.rows && c < d.cols) ? src[(int64_t)r * d.cols + c] : 0.f;
    if (d.dst && r < d.rows && c < d.cols) ((T *)d.dst + goff)[(int64_t)r * d.cols + c0 + tx_d] = (T)v;
    tile[i][tx] = v;
  }
  if (!d.dst_t) return;
  __syncthreads();
  T *dst_t = (T *)d.dst_t + goff;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;   // dst_t[c][r]
    if (r < d.rows && c < d.cols) dst_t[(int64_t)c * d.rows + r0 + tx_t] = (T)tile[tx][i];
  }
}

__global__ void add_f32_kernel(float *__restrict__ dst, const float *__restrict__ src, int64_t n4, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) {
    ((f32x4 *)dst)[i] += ((const f32x4 *)src)[i];
  } else if (i == n4) {
    for (int64_t j = n4 * 4; j < n; ++j) dst[j] += src[j];
  }
}

template <typename T>
__global__ void cast_f32_kernel(const float *__restrict__ src, int64_t n4, T *__restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  Vec4<T>::store(dst + i * 4, *(const f32x4 *)(src + i * 4));
}

template <typename T>
__global__ void scale_rows_cast_kernel(const float *__restrict__ src, int64_t n4, int cols4, const float *__restrict__ scale,
                                       int div, T *__restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float sc = scale[(i / cols4) / div];
  Vec4<T>::store(dst + i * 4, *(const f32x4 *)(src + i * 4) * sc);
}

// rows[(b*hp + py)*wp + px][c*P*P + iy*P + ix] = img[b][c][py*P+iy][px*P+ix]
template <typename T>
__global__ void im2row_kernel(const float *__restrict__ img, int B, int Cin, int H, int W, int P, T *__restrict__ rows) {
  const int hp = H / P, wp = W / P;
  const int K = Cin * P * P;
  const int64_t total4 = (int64_t)B * hp * wp * K / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i * 4;
    const int64_t row = e / K;
    const int kk = (int)(e - row * K);
    const int c = kk / (P * P), rem = kk - c * P * P, iy = rem / P, ix = rem - iy * P;   // ix % 4 == 0 (P % 4 == 0)
    const int px = (int)(row % wp), py = (int)((row / wp) % hp), b = (int)(row / ((int64_t)wp * hp));
    const float *s = img + (((int64_t)b * Cin + c) * H + (py * P + iy)) * W + px * P + ix;
    Vec4<T>::store(rows + e, *(const f32x4 *)s);
  }
}

__global__ void assemble_tokens_kernel(const float *__restrict__ patch, const float *__restrict__ cls,
                                       const float *__restrict__ pos, int B, int np_, int D, float *__restrict__ tok) {
  const int N = np_ + 1;
  const int64_t total4 = (int64_t)B * N * D / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i * 4;
    const int d = (int)(e % D);
    const int64_t r = e / D;
    const int n = (int)(r % N), b = (int)(r / N);
    const f32x4 p = *(const f32x4 *)(pos + (int64_t)n * D + d);
    const f32x4 v = (n == 0) ? *(const f32x4 *)(cls + d) : *(const f32x4 *)(patch + ((int64_t)b * np_ + (n - 1)) * D + d);
    *(f32x4 *)(tok + e) = v + p;
  }
}

// Input gradient of an MoE layer's branch point: the k routed copies of a token (MOEScatter's backward: a gather-sum,
// custom_moe_layer.py:254-259 / fmoe functions.py MOEScatter.backward) plus the gate's share d logits @ w_gate^T
// (the backward of `inp @ w_gate`, noisy_gate_vmoe.py:91).  A wave takes CG_ROWS rows per pass; the E rows of w_gate^T live
// in LDS as [E][D + 4] fp32 (a lane reads its 4 columns of every expert row as one 16-byte access, shared by the pass's
// rows), the tokens' E logit gradients are wave-uniform scalars.  Replaces a [T,k,D] -> [T,D] sum pass plus a K = E GEMM pass that re-read and re-wrote the
// fp32 [T,D] result.
constexpr int CG_THREADS = 512;    // 8 waves share one LDS image of w_gate^T
constexpr int CG_ROWS = 2;         // rows per wave and pass (4: 142 VGPRs, 3 waves per SIMD)
template <typename T, typename TO, int KT, int NCH>
__global__ __launch_bounds__(CG_THREADS) void combine_gate_bwd_kernel(const T *__restrict__ dxe, int64_t T_, int k_rt, int D,
                                                                      const float *__restrict__ dl,
                                                                      const float *__restrict__ wg, int E,
                                                                      TO *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];        // [E][D + 4]: the transposing fill below walks e
  const int DP = D + 4;     // fastest, and a row stride of D (a multiple of the 32 banks) would put all 64 lanes on one bank
  const int k = KT ? KT : k_rt;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int q = threadIdx.x * 4; q < D * E; q += CG_THREADS * 4) {    // w_gate is [D][E]: transpose on the way in
    const f32x4 v = *(const f32x4 *)(wg + q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = (q + i) / E, e = (q + i) - d * E;
      s_w[e * DP + d] = v[i];
    }
  }
  __syncthreads();
  constexpr int WPB = CG_THREADS / 64, RW = CG_ROWS; // RW rows per wave and pass: every LDS read of w_gate^T feeds RW rows
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  // lanes past D in the last column chunk read column 0 again and never store: every load stays unconditional (a
  // predicated load becomes its own basic block with its own wait - the row's loads would go out one latency at a time)
  bool on[NCH];
  int col[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) { on[c] = c * 256 + lane * 4 < D; col[c] = on[c] ? c * 256 + lane * 4 : 0; }
  const int64_t ngrp = (T_ + RW - 1) / RW;
  for (int64_t gq = (int64_t)blockIdx.x * WPB + wave; gq < ngrp; gq += (int64_t)gridDim.x * WPB) {
    const int64_t t0 = gq * RW;
    // every load of the RW rows first (RW x k rows x NCH column chunks in flight), then the E-step fma chains side by side
    f32x4 acc[RW][NCH];
    if constexpr (KT > 0) {
      typename Vec4<T>::type raw[RW][NCH][KT];                       // as loaded: converted only after all are on their way
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int64_t t = t0 + r < T_ ? t0 + r : T_ - 1;           // (rows past T: a valid row again, never stored)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int j = 0; j < KT; ++j)
            raw[r][c][j] = *(const typename Vec4<T>::type *)(dxe + (t * KT + j) * D + col[c]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          acc[r][c] = f32x4{(float)raw[r][c][0][0], (float)raw[r][c][0][1], (float)raw[r][c][0][2], (float)raw[r][c][0][3]};
#pragma unroll
          for (int j = 1; j < KT; ++j)
            acc[r][c] = acc[r][c] + f32x4{(float)raw[r][c][j][0], (float)raw[r][c][j][1], (float)raw[r][c][j][2], (float)raw[r][c][j][3]};
        }
    } else {
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int64_t t = t0 + r < T_ ? t0 + r : T_ - 1;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
          for (int j = 0; j < k; ++j) acc[r][c] = acc[r][c] + Vec4<T>::load(dxe + (t * k + j) * D + col[c]);
        }
      }
    }
    const float *u[RW];                                              // wave-uniform addresses: scalar loads
#pragma unroll
    for (int r = 0; r < RW; ++r) u[r] = dl + (t0 + r < T_ ? t0 + r : T_ - 1) * E;
#pragma unroll 4
    for (int e = 0; e < E; ++e) {
      float ue[RW];
#pragma unroll
      for (int r = 0; r < RW; ++r) ue[r] = u[r][e];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const f32x4 w = *(const f32x4 *)(s_w + e * DP + col[c]);
        const f32x2 wl = f32x2{w[0], w[1]}, wh = f32x2{w[2], w[3]};
#pragma unroll
        for (int r = 0; r < RW; ++r) {                              // packed fp32 fma: two columns per instruction
          const f32x2 uu = f32x2{ue[r], ue[r]};
          f32x2 lo = f32x2{acc[r][c][0], acc[r][c][1]}, hi = f32x2{acc[r][c][2], acc[r][c][3]};
          lo = __builtin_elementwise_fma(uu, wl, lo);
          hi = __builtin_elementwise_fma(uu, wh, hi);
          acc[r][c] = f32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (on[c] && t0 + r < T_) Vec4<TO>::store(out + (t0 + r) * D + c * 256 + lane * 4, acc[r][c]);
  }
}

}  // namespace m3

using namespace m3;

static inline unsigned row_blocks(int64_t T) { return (unsigned)((T + 3) / 4); }

extern "C" int m3_combine_fwd(const void *y, int dtype, const float *score, const float *residual, int64_t T, int k,
                              int D, float *out, void *stream) {
  M3_REQUIRE(y && score && out, "m3_combine_fwd: null operand");
  M3_REQUIRE(dtype_ok(dtype), "m3_combine_fwd: bad dtype");
  M3_REQUIRE(D % 4 == 0 && D > 0 && k >= 1, "m3_combine_fwd: D must be a multiple of 4");
  if (T == 0) return M3_OK;
  hipStream_t s = (hipStream_t)stream;
#define M3_CF(TT, KT_) hipLaunchKernelGGL((combine_fwd_kernel<TT, KT_>), dim3(row_blocks(T)), dim3(ROW_THREADS), 0, s, (const TT *)y, score, residual, T, k, D, out)
#define M3_CF_K(TT) do { if (k == 4) M3_CF(TT, 4); else if (k == 2) M3_CF(TT, 2); else if (k == 1) M3_CF(TT, 1); else if (k == 8) M3_CF(TT, 8); else M3_CF(TT, 0); } while (0)
  if (dtype == M3_F16) M3_CF_K(half_t); else if (dtype == M3_BF16) M3_CF_K(bf16_t); else M3_CF_K(float);
#undef M3_CF_K
#undef M3_CF
  return check_launch("m3_combine_fwd");
}

extern "C" int m3_combine_gate_bwd(const void *dxe, int dtype, int64_t T, int k, int D, const float *d_logits,
                                   const float *w_gate, int E, void *dh, int dh_dtype, void *stream) {
  M3_REQUIRE(dxe && d_logits && w_gate && dh, "m3_combine_gate_bwd: null operand");
  M3_REQUIRE(dtype_ok(dtype), "m3_combine_gate_bwd: bad dtype");
  M3_REQUIRE(dh_dtype == M3_F32 || dh_dtype == dtype, "m3_combine_gate_bwd: dh is fp32 or the activation dtype");
  M3_REQUIRE(D % 4 == 0 && D > 0 && k >= 1 && E >= 1, "m3_combine_gate_bwd: D must be a multiple of 4");
  const size_t lds = (size_t)E * (D + 4) * sizeof(float);
  M3_REQUIRE(lds <= 64 * 1024, "m3_combine_gate_bwd: w_gate [D=%d][E=%d] does not fit the 64 KB LDS image", D, E);
  if (T == 0) return M3_OK;
  hipStream_t s = (hipStream_t)stream;
  // grid-stride over the rows: two 16-wave workgroups per CU (all 32 wave slots), each filling its LDS image of w_gate once
  const int64_t rb = ((T + CG_ROWS - 1) / CG_ROWS + CG_THREADS / 64 - 1) / (CG_THREADS / 64);
  const unsigned grid = (unsigned)(rb < 1024 ? rb : 1024);
  const int nch = (D + 255) / 256;
  M3_REQUIRE(nch <= 4 && (D * E) % 4 == 0 && ((uintptr_t)w_gate % 16) == 0, "m3_combine_gate_bwd: D <= 1024, w_gate 16-byte aligned");
  const bool f32out = dh_dtype == M3_F32;
#define M3_CG(TT, TO_, KT_, NC_) hipLaunchKernelGGL((combine_gate_bwd_kernel<TT, TO_, KT_, NC_>), dim3(grid), dim3(CG_THREADS), lds, s, (const TT *)dxe, T, k, D, d_logits, w_gate, E, (TO_ *)dh)
#define M3_CG_O(TT, KT_, NC_) do { if (f32out) M3_CG(TT, float, KT_, NC_); else M3_CG(TT, TT, KT_, NC_); } while (0)
#define M3_CG_N(TT, KT_) do { if (nch == 1) M3_CG_O(TT, KT_, 1); else if (nch == 2) M3_CG_O(TT, KT_, 2); else if (nch == 3) M3_CG_O(TT, KT_, 3); else M3_CG_O(TT, KT_, 4); } while (0)
#define M3_CG_K(TT) do { if (k == 4) M3_CG_N(TT, 4); else if (k == 2) M3_CG_N(TT, 2); else M3_CG_N(TT, 0); } while (0)
  if (dtype == M3_F16) M3_CG_K(half_t); else if (dtype == M3_BF16) M3_CG_K(bf16_t); else M3_CG_K(float);
#undef M3_CG_K
#undef M3_CG_N
#undef M3_CG_O
#undef M3_CG
  return check_launch("m3_combine_gate_bwd");
}

extern "C" int m3_combine_bwd(const float *dout, const void *y, int dtype, const float *score, int64_t T, int k, int D,
                              void *dy, float *dscore, void *stream) {
  M3_REQUIRE(dout && y && score && dscore, "m3_combine_bwd: null operand");
  M3_REQUIRE(dtype_ok(dtype), "m3_combine_bwd: bad dtype");
  M3_REQUIRE(D % 4 == 0 && D > 0 && k >= 1, "m3_combine_bwd: D must be a multiple of 4");
  if (T == 0) return M3_OK;
  hipStream_t s = (hipStream_t)stream;
#define M3_CB(TT, KT_) hipLaunchKernelGGL((combine_bwd_kernel<TT, KT_>), dim3(row_blocks(T)), dim3(ROW_THREADS), 0, s, dout, (const TT *)y, score, T, k, D, (TT *)dy, dscore)
#define M3_CB_K(TT) do { if (k == 4) M3_CB(TT, 4); else if (k == 2) M3_CB(TT, 2); else if (k == 1) M3_CB(TT, 1); else if (k == 8) M3_CB(TT, 8); else M3_CB(TT, 0); } while (0)
  if (dtype == M3_F16) M3_CB_K(half_t); else if (dtype == M3_BF16) M3_CB_K(bf16_t); else M3_CB_K(float);
#undef M3_CB_K
#undef M3_CB
  return check_launch("m3_combine_bwd");
}

extern "C" int m3_layernorm_fwd(const float *x, int64_t T, int D, const float *gamma, const float *beta, float eps,
                                void *y, int y_dtype, float *mean, float *rstd, void *stream) {
  M3_REQUIRE(x && gamma && beta && y && mean && rstd, "m3_layernorm_fwd: null operand");
  M3_REQUIRE(dtype_ok(y_dtype), "m3_layernorm_fwd: bad dtype");
  M3_REQUIRE(D % 4 == 0 && D > 0 && D <= 1024, "m3_layernorm_fwd: D must be a multiple of 4 and <= 1024 (got %d)", D);
  if (T == 0) return M3_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nch = (D + 255) / 256;
#define M3_LNF(TT, NC) hipLaunchKernelGGL((layernorm_fwd_kernel<TT, NC>), dim3(row_blocks(T)), dim3(ROW_THREADS), 0, s, x, T, D, gamma, beta, eps, (TT *)y, mean, rstd)
#define M3_LNF_N(TT) do { if (nch == 1) M3_LNF(TT, 1); else if (nch == 2) M3_LNF(TT, 2); else if (nch == 3) M3_LNF(TT, 3); else M3_LNF(TT, 4); } while (0)
  if (y_dtype == M3_F16) M3_LNF_N(half_t); else if (y_dtype == M3_BF16) M3_LNF_N(bf16_t); else M3_LNF_N(float);
#undef M3_LNF_N
#undef M3_LNF
  return check_launch("m3_layernorm_fwd");
}

extern "C" int m3_ln_bwd_blocks(int64_t T, int D) {
  const int64_t rows = (int64_t)ln_bwd_waves(D) * ln_rows_per_wave();
  return (int)((T + rows - 1) / rows);
}

extern "C" int m3_layernorm_bwd(const void *dy, int dy_dtype, const float *x, const float *mean, const float *rstd,
                                const float *gamma, const float *dx_res, int64_t T, int D, float *dx, float *ws,
                                float *dgamma, float *dbeta, int beta, void *dx_act, int dx_act_dtype, void *stream) {
  M3_REQUIRE(dy && x && mean && rstd && gamma && dx && ws, "m3_layernorm_bwd: null operand");
  M3_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "m3_layernorm_bwd: dgamma and dbeta go together");
  M3_REQUIRE(dtype_ok(dy_dtype), "m3_layernorm_bwd: bad dtype");
  M3_REQUIRE(!dx_act || dtype_ok(dx_act_dtype), "m3_layernorm_bwd: bad dx_act dtype");
  M3_REQUIRE(D % 4 == 0 && D > 0 && D <= 1024, "m3_layernorm_bwd: D must be a multiple of 4 and <= 1024");
  if (T == 0) return M3_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = m3_ln_bwd_blocks(T, D), rpw = ln_rows_per_wave(), nw = ln_bwd_waves(D);
  const size_t lds = (size_t)2 * nw * D * sizeof(float);
  const bool a16 = dx_act && dx_act_dtype == M3_F16;
  const int nch = (D + 255) / 256;
#define M3_LNB_N(TT, TA, NC, NW)                                                                                     \
  hipLaunchKernelGGL((layernorm_bwd_kernel<TT, TA, NC, NW>), dim3(nblk), dim3(NW * 64), lds, s, (const TT *)dy, x,  \
                     mean, rstd, gamma, dx_res, T, D, dx, ws, (TA *)dx_act, rpw)
#define M3_LNB_W(TT, TA, NC)                                                              \
  do {                                                                                   \
    if (nw == 16) M3_LNB_N(TT, TA, NC, 16); else if (nw == 8) M3_LNB_N(TT, TA, NC, 8);   \
    else M3_LNB_N(TT, TA, NC, 4);                                                        \
  } while (0)
#define M3_LNB(TT, TA)                                                                   \
  do {                                                                                   \
    if (nch == 1) M3_LNB_W(TT, TA, 1); else if (nch == 2) M3_LNB_W(TT, TA, 2);           \
    else if (nch == 3) M3_LNB_W(TT, TA, 3); else M3_LNB_W(TT, TA, 4);                    \
  } while (0)
  const bool ab16 = dx_act && dx_act_dtype == M3_BF16;
  // (the activation-dtype copy of dx has the dtype of the incoming gradient or is fp32; mixed 16-bit pairs are not built)
  M3_REQUIRE(!(a16 && dy_dtype == M3_BF16) && !(ab16 && dy_dtype == M3_F16), "m3_layernorm_bwd: dy fp16 with dx_act bf16 (or the reverse) is not supported");
  if (dy_dtype == M3_F16) { if (a16) M3_LNB(half_t, half_t); else M3_LNB(half_t, float); }
  else if (dy_dtype == M3_BF16) { if (ab16) M3_LNB(bf16_t, bf16_t); else M3_LNB(bf16_t, float); }
  else { if (a16) M3_LNB(float, half_t); else if (ab16) M3_LNB(float, bf16_t); else M3_LNB(float, float); }
#undef M3_LNB
#undef M3_LNB_W
#undef M3_LNB_N
  int rc = check_launch("m3_layernorm_bwd");
  if (rc || !dgamma) return rc;                  // no dgamma / dbeta: the partials stay in ws for m3_layernorm_bwd_reduce
  return launch_reduce_rows2_f32(ws, nblk, D, dgamma, dbeta, beta, s);
}

extern "C" int m3_layernorm_bwd_reduce(const float *ws, int64_t layer_stride, int nblk, int D,
                                       const m3_ln_param_grads *grads_dev, int first, int count, int beta, void *stream) {
  M3_REQUIRE(ws && grads_dev && nblk >= 1 && D > 0 && first >= 0 && count >= 0 && layer_stride >= (int64_t)2 * nblk * D,
             "m3_layernorm_bwd_reduce: bad args");
  if (count == 0) return M3_OK;
  return launch_reduce_rows2_batch_f32(ws, layer_stride, nblk, D, grads_dev, first, count, beta, (hipStream_t)stream);
}

extern "C" int m3_cast_matrix(const float *src, int G, int rows, int cols, int transpose, void *dst, int dst_dtype,
                              void *stream) {
  M3_REQUIRE(src && dst && G >= 1 && rows > 0 && cols > 0, "m3_cast_matrix: bad args");
  M3_REQUIRE(dtype_ok(dst_dtype), "m3_cast_matrix: bad dtype");
  const dim3 grid((cols + 31) / 32, (rows + 31) / 32, G), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dst_dtype == M3_F16) hipLaunchKernelGGL(cast_matrix_kernel<half_t>, grid, block, 0, s, src, rows, cols, transpose, (half_t *)dst);
  else if (dst_dtype == M3_BF16) hipLaunchKernelGGL(cast_matrix_kernel<bf16_t>, grid, block, 0, s, src, rows, cols, transpose, (bf16_t *)dst);
  else hipLaunchKernelGGL(cast_matrix_kernel<float>, grid, block, 0, s, src, rows, cols, transpose, (float *)dst);
  return check_launch("m3_cast_matrix");
}

extern "C" int m3_cast_batch(const m3_cast_desc *descs_dev, int n_desc, int total_tiles, int dst_dtype, void *stream) {
  static_assert(sizeof(m3_cast_desc) == sizeof(CastDesc), "descriptor layout");
  M3_REQUIRE(descs_dev && n_desc >= 1 && total_tiles >= 1, "m3_cast_batch: bad args");
  M3_REQUIRE(dtype_ok(dst_dtype), "m3_cast_batch: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  const CastDesc *d = (const CastDesc *)descs_dev;
  if (dst_dtype == M3_F16) hipLaunchKernelGGL(cast_batch_kernel<half_t>, dim3(total_tiles), dim3(256), 0, s, d, n_desc);
  else if (dst_dtype == M3_BF16) hipLaunchKernelGGL(cast_batch_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, s, d, n_desc);
  else hipLaunchKernelGGL(cast_batch_kernel<float>, dim3(total_tiles), dim3(256), 0, s, d, n_desc);
  return check_launch("m3_cast_batch");
}

extern "C" int m3_add_f32(float *dst, const float *src, int64_t n, void *stream) {
  M3_REQUIRE(dst && src && n >= 0, "m3_add_f32: bad args");
  M3_REQUIRE(((uintptr_t)dst % 16) == 0 && ((uintptr_t)src % 16) == 0, "m3_add_f32: 16-byte alignment");
  if (n == 0) return M3_OK;
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(add_f32_kernel, dim3((unsigned)((n4 + 1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, src,
                     n4, n);
  return check_launch("m3_add_f32");
}

extern "C" int m3_cast_f32(const float *src, int64_t n, void *dst, int dst_dtype, void *stream) {
  M3_REQUIRE(src && dst && n >= 0 && n % 4 == 0, "m3_cast_f32: n must be a multiple of 4");
  M3_REQUIRE(dtype_ok(dst_dtype), "m3_cast_f32: bad dtype");
  if (n == 0) return M3_OK;
  const int64_t n4 = n / 4;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
  if (dst_dtype == M3_F16) hipLaunchKernelGGL(cast_f32_kernel<half_t>, grid, block, 0, s, src, n4, (half_t *)dst);
  else if (dst_dtype == M3_BF16) hipLaunchKernelGGL(cast_f32_kernel<bf16_t>, grid, block, 0, s, src, n4, (bf16_t *)dst);
  else hipLaunchKernelGGL(cast_f32_kernel<float>, grid, block, 0, s, src, n4, (float *)dst);
  return check_launch("m3_cast_f32");
}

extern "C" int m3_scale_rows_cast(const float *src, int64_t rows, int cols, const float *row_scale, int div, void *dst,
                                  int dst_dtype, void *stream) {
  M3_REQUIRE(src && dst && row_scale && rows >= 0 && cols > 0 && cols % 4 == 0 && div >= 1, "m3_scale_rows_cast: bad args");
  M3_REQUIRE(dtype_ok(dst_dtype), "m3_scale_rows_cast: bad dtype");
  if (rows == 0) return M3_OK;
  const int64_t n4 = rows * cols / 4;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
  if (dst_dtype == M3_F16) hipLaunchKernelGGL(scale_rows_cast_kernel<half_t>, grid, block, 0, s, src, n4, cols / 4, row_scale, div, (half_t *)dst);
  else if (dst_dtype == M3_BF16) hipLaunchKernelGGL(scale_rows_cast_kernel<bf16_t>, grid, block, 0, s, src, n4, cols / 4, row_scale, div, (bf16_t *)dst);
  else hipLaunchKernelGGL(scale_rows_cast_kernel<float>, grid, block, 0, s, src, n4, cols / 4, row_scale, div, (float *)dst);
  return check_launch("m3_scale_rows_cast");
}

extern "C" int m3_im2row(const float *img, int B, int Cin, int H, int W, int P, void *rows, int dtype, void *stream) {
  M3_REQUIRE(img && rows, "m3_im2row: null operand");
  M3_REQUIRE(dtype_ok(dtype), "m3_im2row: bad dtype");
  M3_REQUIRE(P % 4 == 0 && H % P == 0 && W % P == 0 && W % 4 == 0, "m3_im2row: P, W must be multiples of 4; H, W multiples of P");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == M3_F16) hipLaunchKernelGGL(im2row_kernel<half_t>, dim3(2048), dim3(256), 0, s, img, B, Cin, H, W, P, (half_t *)rows);
  else if (dtype == M3_BF16) hipLaunchKernelGGL(im2row_kernel<bf16_t>, dim3(2048), dim3(256), 0, s, img, B, Cin, H, W, P, (bf16_t *)rows);
  else hipLaunchKernelGGL(im2row_kernel<float>, dim3(2048), dim3(256), 0, s, img, B, Cin, H, W, P, (float *)rows);
  return check_launch("m3_im2row");
}

extern "C" int m3_assemble_tokens(const float *patch, const float *cls, const float *pos, int B, int np_, int D,
                                  float *tokens, void *stream) {
  M3_REQUIRE(patch && cls && pos && tokens && D % 4 == 0, "m3_assemble_tokens: bad args");
  hipLaunchKernelGGL(assemble_tokens_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, patch, cls, pos, B, np_, D,
                     tokens);
  return check_launch("m3_assemble_tokens");
}

// ---- backward of assemble_tokens: dpatch (act dtype) = dtok[:,1:,:] ; dpos (+)= sum_b dtok ;
// dcls (+)= sum_b dtok[:,0,:]
namespace m3 {
// thread (e, bl): 16-byte element e of a [N, D] token image, batch lane bl of TB_BL: images bl, bl + TB_BL, ... summed in that
// order, then the lanes' sums added in lane order through LDS (fixed order: deterministic).  (One thread per element
// walking all B images by itself - 74 workgroups for ViT-S - ran at 1 TB/s.)
constexpr int TB_BL = 8, TB_EL = 32;                 // batch lanes x elements per 256-thread workgroup
template <typename T>
__global__ __launch_bounds__(TB_BL * TB_EL) void tokens_bwd_kernel(const float *__restrict__ dtok, int B, int np_, int D, T *__restrict__ dpatch,
                                  float *__restrict__ dpos, float *__restrict__ dcls, int beta) {
  __shared__ f32x4 part[TB_BL][TB_EL];
  const int N = np_ + 1;
  const int64_t total4 = (int64_t)N * D / 4;
  const int el = threadIdx.x % TB_EL, bl = threadIdx.x / TB_EL;
  const int64_t i = (int64_t)blockIdx.x * TB_EL + el;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  int n = 0, d = 0;
  if (i < total4) {
    const int64_t e = i * 4;
    n = (int)(e / D); d = (int)(e - (int64_t)n * D);
    for (int b = bl; b < B; b += TB_BL) {
      const f32x4 g = *(const f32x4 *)(dtok + ((int64_t)b * N + n) * D + d);
      s += g;
      if (n > 0 && dpatch) Vec4<T>::store(dpatch + ((int64_t)b * np_ + (n - 1)) * D + d, g);
    }
  }
  part[bl][el] = s;
  __syncthreads();
  if (bl == 0 && i < total4) {
#pragma unroll
    for (int j = 1; j < TB_BL; ++j) s += part[j][el];
    f32x4 *pp = (f32x4 *)(dpos + i * 4);
    *pp = beta ? (*pp + s) : s;
    if (n == 0) {
      f32x4 *pc = (f32x4 *)(dcls + d);
      *pc = beta ? (*pc + s) : s;
    }
  }
}
}  // namespace m3

extern "C" int m3_tokens_bwd(const float *dtok, int B, int np_, int D, void *dpatch, int dtype, float *dpos,
                             float *dcls, int beta, void *stream) {
  M3_REQUIRE(dtok && dpos && dcls && D % 4 == 0, "m3_tokens_bwd: bad args");
  M3_REQUIRE(dtype_ok(dtype), "m3_tokens_bwd: bad dtype");
  const int64_t total4 = (int64_t)(np_ + 1) * D / 4;
  const dim3 grid((unsigned)((total4 + m3::TB_EL - 1) / m3::TB_EL)), block(m3::TB_BL * m3::TB_EL);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == M3_F16) hipLaunchKernelGGL(m3::tokens_bwd_kernel<m3::half_t>, grid, block, 0, s, dtok, B, np_, D, (m3::half_t *)dpatch, dpos, dcls, beta);
  else if (dtype == M3_BF16) hipLaunchKernelGGL(m3::tokens_bwd_kernel<m3::bf16_t>, grid, block, 0, s, dtok, B, np_, D, (m3::bf16_t *)dpatch, dpos, dcls, beta);
  else hipLaunchKernelGGL(m3::tokens_bwd_kernel<float>, grid, block, 0, s, dtok, B, np_, D, (float *)dpatch, dpos, dcls, beta);
  return m3::check_launch("m3_tokens_bwd");
}

// ---- row gather with optional k-way sum: dst[i,:] = sum_{j<k} src[idx[i*k+j] / div, :]
// (k = 1: MOEScatter / MOEGather row movement of fastmoe behind custom_moe_layer.py:263-265;
//  k > 1: the backward of MOEScatter, which sums the k routed copies of a token)
namespace m3 {
template <typename T>
__global__ __launch_bounds__(ROW_THREADS) void gather_rows_kernel(const T *__restrict__ src, const int32_t *__restrict__ idx,
                                                                  int div, int64_t nout, int k, int D,
                                                                  T *__restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * (ROW_THREADS / 64) + (threadIdx.x >> 6);
  if (i >= nout) return;
  for (int d = lane * 4; d < D; d += 256) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < k; ++j) {
      const int64_t r = idx[i * k + j] / div;
      acc += Vec4<T>::load(src + r * D + d);
    }
    Vec4<T>::store(dst + i * D + d, acc);
  }
}
}  // namespace m3

extern "C" int m3_gather_rows(const void *src, int dtype, const int32_t *idx, int div, int64_t nout, int k, int D,
                              void *dst, void *stream) {
  M3_REQUIRE(src && idx && dst && D % 4 == 0 && D > 0 && k >= 1 && div >= 1, "m3_gather_rows: bad args");
  M3_REQUIRE(dtype_ok(dtype), "m3_gather_rows: bad dtype");
  if (nout == 0) return M3_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((nout + 3) / 4)), block(m3::ROW_THREADS);
  if (dtype == M3_F16) hipLaunchKernelGGL(m3::gather_rows_kernel<m3::half_t>, grid, block, 0, s, (const m3::half_t *)src, idx, div, nout, k, D, (m3::half_t *)dst);
  else if (dtype == M3_BF16) hipLaunchKernelGGL(m3::gather_rows_kernel<m3::bf16_t>, grid, block, 0, s, (const m3::bf16_t *)src, idx, div, nout, k, D, (m3::bf16_t *)dst);
  else hipLaunchKernelGGL(m3::gather_rows_kernel<float>, grid, block, 0, s, (const float *)src, idx, div, nout, k, D, (float *)dst);
  return m3::check_launch("m3_gather_rows");
}
