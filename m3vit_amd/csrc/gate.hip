// Fused task-aware router for gfx950: logits + noise + softmax + top-(k+1) + gates +
// load-balance partials in one pass over the token rows.
//
// Replaces the 5-6 ATen kernels of NoisyGate_VMoE.forward
// (models/moe/ckpt/noisy_gate_vmoe.py:91-93,168,197-207), the multi-gate select
// (custom_moe_layer.py:213-217: the caller just passes that task's w_gate) and the
// task-conditioning cat (custom_moe_layer.py:176-179: folded into logit_bias), plus the
// importance / load summaries of vision_transformer_moe.py:453-459.
//
// HBM-bound by design: each token row is read once (coalesced 16-byte loads, staged
// through LDS so that every LANE then owns one TOKEN), w_gate is wave-uniform and comes
// through the scalar cache, all E logits of a token live in that lane's registers, so
// softmax and the top-(k+1) selection need no cross-lane traffic at all.
//
// Arithmetic order is pinned (and mirrored by oracle/gate_route.c) so that expert
// indices are bit-exact: sequential fmaf chain over d starting from the bias, selection
// on the noisy logits with ties -> lowest index.
#include "common.h"

#pragma clang fp contract(off)

namespace m3 {

constexpr int GATE_TOK = 64;       // tokens per workgroup (one wave)
constexpr int GATE_DW_TOK = 128;   // tokens per workgroup in the dW kernel
constexpr int GATE_ROWB = 64;      // bytes of a token row staged per step

template <typename T, int EPAD, bool EXACT>
__global__ __launch_bounds__(GATE_TOK) void gate_fwd_kernel(
    const char *__restrict__ x, int64_t T_, int D, int64_t ldx_b, const float *__restrict__ w, int E,
    const float *__restrict__ bias, const float *__restrict__ noise, float noise_std, int k, int64_t *idx,
    int32_t *idx32, float *score, float *top_logits, float *clean, float *noisy_out, float *gates,
    float *part_imp, int32_t *part_load) {
  constexpr int DC = GATE_ROWB / (int)sizeof(T);   // d per step
  constexpr int LDS_STRIDE = DC + 1;               // floats; odd -> conflict-free per-lane rows
  __shared__ float sx[GATE_TOK * LDS_STRIDE];

  const int lane = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * GATE_TOK;
  const int64_t t = t0 + lane;
  const bool tok_ok = t < T_;

  float acc[EPAD];
#pragma unroll
  for (int e = 0; e < EPAD; ++e) acc[e] = (bias && (EXACT || e < E)) ? bias[e] : 0.f;

  constexpr int CPR = GATE_ROWB / 16;              // 16-byte chunks per row per step = 8
  constexpr int EPC = 16 / (int)sizeof(T);         // elements per chunk
  const int dbytes = D * (int)sizeof(T);
  for (int d0b = 0; d0b < dbytes; d0b += GATE_ROWB) {
    // stage 64 rows x GATE_ROWB bytes: chunk q = lane + 64*i -> row q / CPR, c = q % CPR
#pragma unroll
    for (int i = 0; i < CPR; ++i) {
      const int q = lane + 64 * i;
      const int row = q / CPR, c = q % CPR;
      const int64_t tr = t0 + row;
      const int cb = d0b + c * 16;
      float v[EPC];
      if (tr < T_ && cb < dbytes) {
        const u32x4 raw = *(const u32x4 *)(x + tr * ldx_b + cb);
        if constexpr (sizeof(T) == 2) {
          const f16x8 h = __builtin_bit_cast(f16x8, raw);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
        } else {
          const f32x4 f = __builtin_bit_cast(f32x4, raw);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = f[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < EPC; ++j) v[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < EPC; ++j) sx[row * LDS_STRIDE + c * EPC + j] = v[j];
    }
    __syncthreads();
    const int d0 = d0b / (int)sizeof(T);
    const int dn = (D - d0 < DC) ? (D - d0) : DC;
    int dd = 0;
    if constexpr (EXACT && EPAD <= 16) {
      // 4 rows of w_gate fetched by back-to-back scalar loads, then 4 x E fmas (chain order kept)
      for (; dd + 4 <= dn; dd += 4) {
        float wv[4][EPAD];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float *wr = w + (int64_t)(d0 + dd + u) * E;
#pragma unroll
          for (int e = 0; e < EPAD; ++e) wv[u][e] = wr[e];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float xv = sx[lane * LDS_STRIDE + dd + u];
#pragma unroll
          for (int e = 0; e < EPAD; ++e) acc[e] = __builtin_fmaf(xv, wv[u][e], acc[e]);
        }
      }
    }
    for (; dd < dn; ++dd) {
      const float xv = sx[lane * LDS_STRIDE + dd];
      const float *wr = w + (int64_t)(d0 + dd) * E;   // wave-uniform -> scalar loads
#pragma unroll
      for (int e = 0; e < EPAD; ++e)
        if (EXACT || e < E) acc[e] = __builtin_fmaf(xv, wr[e], acc[e]);
    }
    __syncthreads();
  }

  // noisy logits
  float nz[EPAD];
#pragma unroll
  for (int e = 0; e < EPAD; ++e) {
    float n = acc[e];
    if (noise && noise_std != 0.f && (EXACT || e < E) && tok_ok) {
      const float scaled = noise[t * E + e] * noise_std;
      n = acc[e] + scaled;
    }
    nz[e] = n;
  }
  // softmax (pinned order)
  float m = nz[0];
#pragma unroll
  for (int e = 1; e < EPAD; ++e)
    if (EXACT || e < E) m = nz[e] > m ? nz[e] : m;
  float q[EPAD];
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < EPAD; ++e) {
    q[e] = (EXACT || e < E) ? expf(nz[e] - m) : 0.f;
    if (EXACT || e < E) s = s + q[e];
  }
  // top-(k+1) on the noisy logits, ties -> lowest index
  const int kp = (k + 1 < E) ? k + 1 : E;
  unsigned long long taken = 0ull, sel_k = 0ull;
  for (int j = 0; j < kp; ++j) {
    int best = -1;
    float bv = 0.f, bq = 0.f;
#pragma unroll
    for (int e = 0; e < EPAD; ++e) {
      if ((EXACT || e < E) && !((taken >> e) & 1ull) && (best < 0 || nz[e] > bv)) {
        best = e; bv = nz[e]; bq = q[e];
      }
    }
    taken |= 1ull << best;
    const float p = bq / s;
    if (tok_ok) {
      top_logits[t * kp + j] = p;
      if (j < k) {
        idx[t * k + j] = best;
        if (idx32) idx32[t * k + j] = best;
        score[t * k + j] = p;
      }
    }
    if (j < k) sel_k |= 1ull << best;
  }
  // dense outputs + load-balance partials
  float imp_part = 0.f;
#pragma unroll
  for (int e = 0; e < EPAD; ++e) {
    if (EXACT || e < E) {
      const bool sel = (sel_k >> e) & 1ull;
      const float p = q[e] / s;
      const float gv = (sel && tok_ok) ? p : 0.f;
      if (tok_ok) {
        if (clean) clean[t * E + e] = acc[e];
        if (noisy_out) noisy_out[t * E + e] = nz[e];
        if (gates) gates[t * E + e] = gv;
      }
      const float wsum = wave_sum(gv);
      const unsigned long long b = __ballot(gv > 0.f);
      if (lane == 0) {
        part_imp[(int64_t)blockIdx.x * E + e] = wsum;
        part_load[(int64_t)blockIdx.x * E + e] = __popcll(b);
      }
    }
  }
  (void)imp_part;
}

__global__ void gate_reduce_kernel(const float *part_imp, const int32_t *part_load, int nblk, int E, float *imp,
                                   int64_t *load) {
  const int e = threadIdx.x;
  if (e >= E) return;
  float s = 0.f;
  int64_t c = 0;
  for (int b = 0; b < nblk; ++b) {
    s += part_imp[(int64_t)b * E + e];
    c += part_load[(int64_t)b * E + e];
  }
  imp[e] = s;
  load[e] = c;
}

// d_logits through scatter + softmax: thread per token
__global__ void gate_bwd_logits_kernel(const float *noisy, const int64_t *idx, const float *d_score,
                                       const float *d_imp, int64_t T_, int E, int k, float *d_logits) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T_) return;
  const float *nz = noisy + t * E;
  float m = nz[0];
  for (int e = 1; e < E; ++e) m = nz[e] > m ? nz[e] : m;
  float s = 0.f;
  for (int e = 0; e < E; ++e) s = s + expf(nz[e] - m);
  // sum_j dp_j p_j over the selected experts
  float dot = 0.f;
  for (int j = 0; j < k; ++j) {
    const int e = (int)idx[t * k + j];
    const float p = expf(nz[e] - m) / s;
    const float dp = (d_score ? d_score[t * k + j] : 0.f) + (d_imp ? d_imp[e] : 0.f);
    dot += dp * p;
  }
  for (int e = 0; e < E; ++e) {
    const float p = expf(nz[e] - m) / s;
    float dp = 0.f;
    for (int j = 0; j < k; ++j)
      if ((int)idx[t * k + j] == e) dp = (d_score ? d_score[t * k + j] : 0.f) + (d_imp ? d_imp[e] : 0.f);
    d_logits[t * E + e] = p * (dp - dot);
  }
}

// dW partials: lanes own d, tokens are walked; d_logits rows are wave-uniform scalars.
template <typename T, int EPAD>
__global__ void gate_bwd_dw_kernel(const char *__restrict__ x, int64_t T_, int D, int64_t ldx_b,
                                   const float *__restrict__ dl, int E, float *part_dw) {
  const int d = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * GATE_DW_TOK;
  int64_t t1 = t0 + GATE_DW_TOK;
  if (t1 > T_) t1 = T_;
  float acc[EPAD];
#pragma unroll
  for (int e = 0; e < EPAD; ++e) acc[e] = 0.f;
  if (d < D) {
    for (int64_t t = t0; t < t1; ++t) {
      const float xv = (float)(*(const T *)(x + t * ldx_b + (int64_t)d * sizeof(T)));
      const float *dr = dl + t * E;
#pragma unroll
      for (int e = 0; e < EPAD; ++e)
        if (e < E) acc[e] = __builtin_fmaf(xv, dr[e], acc[e]);
    }
    float *out = part_dw + ((int64_t)blockIdx.x * D + d) * E;
#pragma unroll
    for (int e = 0; e < EPAD; ++e)
      if (e < E) out[e] = acc[e];
  }
}

__global__ void gate_dw_reduce_kernel(const float *part, int nblk, int64_t elems, float *dw, int beta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems) return;
  float s = beta ? dw[i] : 0.f;
  for (int b = 0; b < nblk; ++b) s += part[(int64_t)b * elems + i];
  dw[i] = s;
}

// dx[t,d] (+)= sum_e dl[t,e] * w[d,e]: one wave per token row pair; w through LDS.
template <int EPAD>
__global__ void gate_bwd_dx_kernel(const float *__restrict__ dl, const float *__restrict__ w, int64_t T_, int D, int E,
                                   float *dx, int64_t lddx, int beta) {
  extern __shared__ float sw[];   // transposed [E][D]: lane d reads consecutive addresses
  for (int i = threadIdx.x; i < D * E; i += blockDim.x) {
    const int d = i / E, e = i - d * E;
    sw[e * D + d] = w[i];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int64_t t = (int64_t)blockIdx.x * wpb + wave; t < T_; t += (int64_t)gridDim.x * wpb) {
    float d_l[EPAD];
#pragma unroll
    for (int e = 0; e < EPAD; ++e) d_l[e] = (e < E) ? dl[t * E + e] : 0.f;
    for (int d = lane; d < D; d += 64) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < EPAD; ++e)
        if (e < E) s = __builtin_fmaf(d_l[e], sw[e * D + d], s);
      float *o = dx + t * lddx + d;
      *o = beta ? (*o + s) : s;
    }
  }
}

}  // namespace m3

using namespace m3;

extern "C" int m3_gate_num_blocks(int64_t T) { return (int)((T + GATE_TOK - 1) / GATE_TOK); }
extern "C" int m3_gate_dw_blocks(int64_t T) { return (int)((T + GATE_DW_TOK - 1) / GATE_DW_TOK); }

template <typename T>
static int launch_gate_fwd(int epad, dim3 grid, hipStream_t s, const char *x, int64_t T_, int D, int64_t ldx_b,
                           const float *w, int E, const float *bias, const float *noise, float std, int k, int64_t *idx,
                           int32_t *idx32, float *score, float *top, float *clean, float *noisy, float *gates,
                           float *pi, int32_t *pl) {
#define M3_GATE_CASE(EP)                                                                                      \
  case EP:                                                                                                    \
    if (E == EP)                                                                                              \
      hipLaunchKernelGGL((gate_fwd_kernel<T, EP, true>), grid, dim3(GATE_TOK), 0, s, x, T_, D, ldx_b, w, E,    \
                         bias, noise, std, k, idx, idx32, score, top, clean, noisy, gates, pi, pl);          \
    else                                                                                                      \
      hipLaunchKernelGGL((gate_fwd_kernel<T, EP, false>), grid, dim3(GATE_TOK), 0, s, x, T_, D, ldx_b, w, E,   \
                         bias, noise, std, k, idx, idx32, score, top, clean, noisy, gates, pi, pl);          \
    break;
  switch (epad) {
    M3_GATE_CASE(4)
    M3_GATE_CASE(8)
    M3_GATE_CASE(16)
    M3_GATE_CASE(32)
    M3_GATE_CASE(64)
    default: return M3_ERR_UNSUPPORTED;
  }
#undef M3_GATE_CASE
  return check_launch("m3_gate_fwd");
}

static int epad_of(int E) { return E <= 4 ? 4 : E <= 8 ? 8 : E <= 16 ? 16 : E <= 32 ? 32 : 64; }
static int epad8_of(int E) { return E <= 8 ? 8 : E <= 16 ? 16 : E <= 32 ? 32 : 64; }

extern "C" int m3_gate_fwd(const void *x, int x_dtype, int64_t T, int D, int64_t ldx, const float *w_gate, int E,
                           const float *logit_bias, const float *noise, float noise_std, int k, int64_t *idx,
                           int32_t *idx32, float *score, float *top_logits, float *clean, float *noisy, float *gates,
                           float *part_importance, int32_t *part_load, void *stream) {
  M3_REQUIRE(x && w_gate && idx && score && top_logits && part_importance && part_load, "m3_gate_fwd: null operand");
  M3_REQUIRE(x_dtype == M3_F32 || x_dtype == M3_F16, "m3_gate_fwd: bad dtype");
  M3_REQUIRE(E >= 2 && E <= 64, "m3_gate_fwd: E=%d outside [2,64]", E);
  M3_REQUIRE(k >= 1 && k <= 8 && k <= E, "m3_gate_fwd: k=%d invalid for E=%d", k, E);
  const int es = dtype_size(x_dtype);
  M3_REQUIRE(T >= 0 && D > 0 && (D * es) % 16 == 0 && (ldx * es) % 16 == 0 && ((uintptr_t)x % 16) == 0,
             "m3_gate_fwd: rows must be 16-byte aligned and D*elem a multiple of 16");
  if (T == 0) return M3_OK;
  const dim3 grid((unsigned)m3_gate_num_blocks(T));
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == M3_F16)
    return launch_gate_fwd<half_t>(epad_of(E), grid, s, (const char *)x, T, D, ldx * es, w_gate, E, logit_bias, noise,
                                   noise_std, k, idx, idx32, score, top_logits, clean, noisy, gates, part_importance,
                                   part_load);
  return launch_gate_fwd<float>(epad_of(E), grid, s, (const char *)x, T, D, ldx * es, w_gate, E, logit_bias, noise,
                                noise_std, k, idx, idx32, score, top_logits, clean, noisy, gates, part_importance,
                                part_load);
}

extern "C" int m3_gate_reduce(const float *part_importance, const int32_t *part_load, int nblk, int E,
                              float *importance, int64_t *load, void *stream) {
  M3_REQUIRE(part_importance && part_load && importance && load && E >= 1 && E <= 64 && nblk >= 0,
             "m3_gate_reduce: bad args");
  int rc = launch_reduce_rows_f32(part_importance, nblk, E, 1, 0, importance, 0, (hipStream_t)stream);
  if (rc) return rc;
  return launch_reduce_rows_i32(part_load, nblk, E, 1, 0, load, 0, (hipStream_t)stream);
}

extern "C" int m3_gate_bwd_logits(const float *noisy, const int64_t *idx, const float *d_score,
                                  const float *d_importance, int64_t T, int E, int k, float *d_logits, void *stream) {
  M3_REQUIRE(noisy && idx && d_logits && E >= 2 && E <= 64 && k >= 1 && k <= E, "m3_gate_bwd_logits: bad args");
  if (T == 0) return M3_OK;
  hipLaunchKernelGGL(gate_bwd_logits_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     noisy, idx, d_score, d_importance, T, E, k, d_logits);
  return check_launch("m3_gate_bwd_logits");
}

extern "C" int m3_gate_bwd_params(const void *x, int x_dtype, int64_t T, int D, int64_t ldx, const float *w_gate,
                                  int E, const float *d_logits, float *part_dw, float *d_w_gate, int beta_dw,
                                  float *dx, int64_t lddx, int beta_dx, void *stream) {
  M3_REQUIRE(x && w_gate && d_logits, "m3_gate_bwd_params: null operand");
  M3_REQUIRE(x_dtype == M3_F32 || x_dtype == M3_F16, "m3_gate_bwd_params: bad dtype");
  M3_REQUIRE(E >= 2 && E <= 64 && D > 0 && D <= 1024, "m3_gate_bwd_params: E in [2,64], D <= 1024");
  M3_REQUIRE((d_w_gate == nullptr) == (part_dw == nullptr), "m3_gate_bwd_params: part_dw and d_w_gate go together");
  if (T == 0) return M3_OK;
  hipStream_t s = (hipStream_t)stream;
  const int es = dtype_size(x_dtype);
  const int ep = epad8_of(E);
  if (d_w_gate) {
    const int nblk = m3_gate_dw_blocks(T);
    const dim3 grid(nblk), block(((D + 63) / 64) * 64);
#define M3_DW_CASE(TT, EP)                                                                                     \
  hipLaunchKernelGGL((gate_bwd_dw_kernel<TT, EP>), grid, block, 0, s, (const char *)x, T, D, ldx * es, d_logits, \
                     E, part_dw)
    if (x_dtype == M3_F16) {
      if (ep == 8) M3_DW_CASE(half_t, 8); else if (ep == 16) M3_DW_CASE(half_t, 16);
      else if (ep == 32) M3_DW_CASE(half_t, 32); else M3_DW_CASE(half_t, 64);
    } else {
      if (ep == 8) M3_DW_CASE(float, 8); else if (ep == 16) M3_DW_CASE(float, 16);
      else if (ep == 32) M3_DW_CASE(float, 32); else M3_DW_CASE(float, 64);
    }
#undef M3_DW_CASE
    int rc = check_launch("m3_gate_bwd_params(dw)");
    if (rc) return rc;
    const int64_t elems = (int64_t)D * E;
    hipLaunchKernelGGL(gate_dw_reduce_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, part_dw, nblk,
                       elems, d_w_gate, beta_dw);
    rc = check_launch("m3_gate_bwd_params(dw reduce)");
    if (rc) return rc;
  }
  if (dx) {
    const size_t lds = (size_t)D * E * sizeof(float);
    M3_REQUIRE(lds <= 64 * 1024, "m3_gate_bwd_params: D*E too large for the dx kernel");
    int64_t blocks = (T + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks), block(256);
#define M3_DX_CASE(EP) \
  hipLaunchKernelGGL((gate_bwd_dx_kernel<EP>), grid, block, lds, s, d_logits, w_gate, T, D, E, dx, lddx, beta_dx)
    if (ep == 8) M3_DX_CASE(8); else if (ep == 16) M3_DX_CASE(16); else if (ep == 32) M3_DX_CASE(32); else M3_DX_CASE(64);
#undef M3_DX_CASE
    return check_launch("m3_gate_bwd_params(dx)");
  }
  return M3_OK;
}
