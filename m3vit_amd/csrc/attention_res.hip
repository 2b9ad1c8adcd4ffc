// Attention forward / backward for SHORT sequences (N <= 256 tokens, fp16, head dim 32 or 64): the case of
// the 224x224 configurations (N = 197; ViT-S heads of 32, ViT-B heads of 64), where one (image, head) pair is
// only 197 x 64 / 128 bytes per operand.  Same maths and operand plan as attention.hip (Attention.forward,
// models/moe/ckpt/vision_transformer_moe.py:299-313), but everything a workgroup needs is staged
// ONCE: K / V (forward) or Q / dO / K (backward) live in LDS for the whole kernel, so the loops over
// the query tiles run without global-memory latency and with at most one barrier per step.
//
// LDS images are row-major rows of dh halves (64 or 128 bytes), XOR-swizzled at 16-byte granularity
// (64-byte rows: chunk ^= (row >> 1) & 3; 128-byte rows: chunk ^= row & 7): conflict-free both for
// ds_read_b128 row fragments (contraction along the row) and for ds_read_b64_tr_b16 transposed fragments
// (contraction along the rows), so no transposed copies are kept.
//   fwd : wave owns query tiles {w, w+4, ..}: S^T[key][q] = K Q^T for ALL keys (<= 16 tiles in
//         registers), plain softmax (no online rescaling), O^T[d][q] = V^T P^T.
//   bwd : wave owns key tiles {w, w+NW, ..} (NW = 4 waves for dh 32, 8 for dh 64; dK^T, dV^T in registers),
//         sweeps 32-row query steps:
//         S = Q K^T, dP = dO V^T, P = exp(S - lse), dS = P (dP - delta);  dV^T += dO^T P,
//         dK^T += Q^T dS;  dS^T goes to a double-buffered LDS image for dQ^T = K^T dS^T.
#include "common.h"
#include <type_traits>

namespace m3 {

constexpr int AR_THREADS = 256;
constexpr int AR_MAXN = 256;
constexpr float AR_LOG2E = 1.4426950408889634f;

typedef __fp16 ar_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// byte offset of (row, byte-in-row) in a swizzled image whose rows hold W halves (W = 32: 64-byte rows, W = 64: 128)
template <int W> __device__ __forceinline__ int swo(int row, int byte) {
  if (W == 32) return row * 64 + ((((byte >> 4) ^ (row >> 1)) & 3) << 4) + (byte & 15);
  return row * 128 + ((((byte >> 4) ^ row) & 7) << 4) + (byte & 15);
}

// row fragment: lane (li, lg) <- 8 halves of row (r0 + li), elements 32*ch + 8*lg .. +7
template <typename T, int W> __device__ __forceinline__ typename Mma<T>::frag row_frag(const char *img, int r0, int ch, int li, int lg) {
  return *(const typename Mma<T>::frag *)(img + swo<W>(r0 + li, ch * 64 + lg * 16));
}

// transposed fragment: lane (li, lg) <- column (col + li), contraction rows rb + {4*lg + r, 16 + 4*lg + r}
// (the k order of Mma<T>::from_tiles, i.e. of an accumulator pair used as the other operand)
template <typename T, int W> __device__ __forceinline__ typename Mma<T>::frag tr_frag(const char *img, int rb, int col, int li, int lg) {
  const int r = rb + 4 * lg + (li >> 2), byte = (col + 4 * (li & 3)) * 2;
  const ar_fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) ar_fp16x4 *)(img + swo<W>(r, byte)));
  const ar_fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) ar_fp16x4 *)(img + swo<W>(r + 16, byte)));
  f16x8 f;
  f[0] = (half_t)lo[0]; f[1] = (half_t)lo[1]; f[2] = (half_t)lo[2]; f[3] = (half_t)lo[3];
  f[4] = (half_t)hi[0]; f[5] = (half_t)hi[1]; f[6] = (half_t)hi[2]; f[7] = (half_t)hi[3];
  return __builtin_bit_cast(typename Mma<T>::frag, f);   // (the 16-bit transposed read is format-agnostic: fp16 or bf16 bits)
}

// (MFMA through Mma<T>::mma: v_mfma_f32_16x16x32_f16 or _bf16)

// ------------------------------------------------------------------------------ forward
// NKT: key tiles, rounded up to a multiple of 4 - a TEMPLATE constant, so that the loops over the key tiles are
// straight-line code (with a run-time `if (kt < nkt)` every key tile was its own basic block and the 13 independent
// S^T products + running max of N = 197 ran strictly one after the other).  Keys past the sequence (only in the last four
// tiles) are masked to -inf; the K / V images are zero-padded to NKT * 16 rows.
template <typename T, int DH, int NKT>
__global__ __launch_bounds__(AR_THREADS, DH == 32 ? 4 : 2) void attention_fwd_res_kernel(const T *__restrict__ qkv, int N, int heads,
                                                                           T *__restrict__ o,
                                                                           float *__restrict__ lse, float scale) {
  constexpr int RBY = DH * 2, CPR = RBY / 16, NCH = DH / 32, NDT = DH / 16;
  constexpr int NKEY = NKT * 16, NQW = NKT / 4;            // padded keys; query tiles per wave (q tiles == key tiles)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *sK = smem, *sV = smem + NKEY * RBY;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / heads, h = bh - b * heads;   // heads of an image share lines
  const int C = heads * DH;
  const int64_t ld = 3 * (int64_t)C;
  const T *qbase = qkv + (int64_t)b * N * ld + h * DH;
  const T *kbase = qbase + C, *vbase = qbase + 2 * C;

  // this wave's query tiles {wave, wave + 4, ...}: Q fragments straight from global, all issued up front
  const int nkt = (N + 15) >> 4;                 // key tiles == query tiles
  typename Mma<T>::frag qfs[NQW][NCH];
#pragma unroll
  for (int i = 0; i < NQW; ++i) {
    const int qrow = (i * 4 + wave) * 16 + li;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      qfs[i][ch] = Mma<T>::zero();
      if (qrow < N) qfs[i][ch] = *(const typename Mma<T>::frag *)(qbase + (int64_t)qrow * ld + ch * 32 + 8 * lg);
    }
  }
  // K / V images: ALL loads of the thread are issued before the first LDS store (a rolled loop waits for each
  // iteration's loads in turn: in-kernel stamps showed a third of a workgroup's life in this phase)
  {
    constexpr int SIT = NKEY * CPR / AR_THREADS;                              // (NKEY * CPR is a multiple of 256)
    constexpr int SB = SIT % 4 == 0 ? 4 : (SIT % 3 == 0 ? 3 : (SIT % 2 == 0 ? 2 : 1));   // batches of <= 4 iterations (registers) that tile SIT exactly
#pragma unroll
    for (int it0 = 0; it0 < SIT; it0 += SB) {
      u32x4 kv[SB], vv[SB];
#pragma unroll
      for (int it = 0; it < SB; ++it) {
        const int q = (it0 + it) * AR_THREADS + tid, row = q / CPR, c = q % CPR;
        kv[it] = u32x4{0u, 0u, 0u, 0u}; vv[it] = kv[it];
        if (row < N) {
          kv[it] = *(const u32x4 *)((const char *)(kbase + (int64_t)row * ld) + c * 16);
          vv[it] = *(const u32x4 *)((const char *)(vbase + (int64_t)row * ld) + c * 16);
        }
      }
#pragma unroll
      for (int it = 0; it < SB; ++it) {
        const int q = (it0 + it) * AR_THREADS + tid, row = q / CPR, c = q % CPR;
        *(u32x4 *)(sK + swo<DH>(row, c * 16)) = kv[it];             // (row < NKEY by construction)
        *(u32x4 *)(sV + swo<DH>(row, c * 16)) = vv[it];
      }
    }
  }
  __syncthreads();

  const float c1 = scale * AR_LOG2E;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NQW; ++i) {
    const int qt = i * 4 + wave;
    if (qt >= nkt) break;
    const int qrow = qt * 16 + li;
    // S^T tiles: st[kt][r] = S[q = li][key = 16*kt + 4*lg + r]; keys >= N can only sit in the last four tiles
    f32x4 st[NKT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4 sv = zero4;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) sv = Mma<T>::mma(row_frag<T, DH>(sK, kt * 16, ch, li, lg), qfs[i][ch], sv);
      if (kt >= NKT - 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = (kt * 16 + 4 * lg + r < N) ? sv[r] : -INFINITY;
      }
      mx = fmaxf(fmaxf(mx, fmaxf(sv[0], sv[1])), fmaxf(sv[2], sv[3]));
      st[kt] = sv;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * c1;
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(st[kt][r] * c1 - mc);      // exp(scale * (s - max)); exp2(-inf) = 0 for masked keys
        st[kt][r] = p;
        psum += p;
      }
    }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    // O^T[d = 16*dt + 4*lg + r][q = li] = sum_key V^T[d][key] P^T[key][q]
    f32x4 oacc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) oacc[dt] = zero4;
#pragma unroll
    for (int cc = 0; cc < NKT / 2; ++cc) {
      const typename Mma<T>::frag pf = Mma<T>::from_tiles(&st[cc * 2]);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) oacc[dt] = Mma<T>::mma(tr_frag<T, DH>(sV, cc * 32, dt * 16, li, lg), pf, oacc[dt]);
    }
    if (qrow < N) {
      const float inv = 1.0f / psum;
      T *orow = o + ((int64_t)b * N + qrow) * C + h * DH;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) Vec4<T>::store(orow + dt * 16 + 4 * lg, oacc[dt] * inv);
      if (lg == 0) lse[((int64_t)b * heads + h) * N + qrow] = mx * scale + __logf(psum);
    }
  }
}

// Diagnostic build only (-DM3_ATTN_STAMPS, tools/attn_stamps.py): lane 0 of wave 0 of every workgroup of the
// short-sequence backward records s_memtime at its phase boundaries; no stamp executes in the shipped kernel.
#ifdef M3_ATTN_STAMPS
constexpr int ASTAMP_WGS = 2048, ASTAMP_N = 40;
__device__ unsigned long long g_attn_stamps[ASTAMP_WGS][ASTAMP_N];
#define ATTN_STAMP(i)                                                                              \
  do {                                                                                             \
    if (threadIdx.x == 0 && blockIdx.x < ASTAMP_WGS && (i) < ASTAMP_N) g_attn_stamps[blockIdx.x][(i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define ATTN_STAMP(i) do { } while (0)
#endif

// ----------------------------------------------------------------------------- backward
// NW waves; wave owns key tiles {w, w + NW, ...}, KTE of them - a TEMPLATE constant (ceil(number of key tiles / NW)):
// every loop over key tiles, dQ chunks and query sub-tiles is then straight-line code without a branch.  (With run-time
// `if (tile < nkt)` guards every (query tile, key tile) pair was its own basic block: the MFMA -> exp2 -> multiply chains
// of the pairs ran one after the other, 2400 cycles for 16 MFMAs and 32 exp2 per step.)  Key tiles past the sequence have
// all-zero K / V fragments and zeroed probabilities - they only ever sit in a wave's LAST tile, and the workgroup is paced
// by the wave whose last tile is real anyway.  K image and dS^T images cover KTE * NW * 16 keys (zero padded).
// dQ pieces (2 query tiles x NDT d tiles) = NW.
template <typename T, int DH, int KTE>
__global__ __launch_bounds__(DH == 32 ? 256 : 512, DH == 32 ? 2 : 1) void attention_bwd_res_kernel(
    const T *__restrict__ qkv, const T *__restrict__ o, const T *__restrict__ d_o,
    const float *__restrict__ lse, int N, int heads, T *__restrict__ dqkv, float scale) {
  constexpr int RBY = DH * 2, CPR = RBY / 16, NCH = DH / 32, NDT = DH / 16;
  constexpr int NW = DH == 32 ? 4 : 8, NT = NW * 64;
  constexpr int NKEY = KTE * NW * 16, NKC = NKEY / 32;      // keys covered by the wave tiles; 32-key chunks of the dQ contraction
  static_assert(2 * NDT == NW, "one dQ piece per wave");
  static_assert(NKEY <= AR_MAXN, "at most 256 keys");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NP = (N + 31) & ~31;                       // query rows, padded to the 32-row steps (NP <= NKEY)
  char *sQ = smem, *sdO = sQ + NP * RBY, *sK = sdO + NP * RBY;        // sK: [NKEY keys][RBY], rows >= N zero
  char *sdS = sK + NKEY * RBY;                         // two [NKEY keys][32 q] images (64-byte rows)
  float *sLse = (float *)(sdS + 2 * NKEY * 64);        // lse * log2(e); +inf-like for the padded rows
  float *sDelta = sLse + NP;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / heads, h = bh - b * heads;
  const int C = heads * DH;
  const int64_t ld = 3 * (int64_t)C;
  const T *qbase = qkv + (int64_t)b * N * ld + h * DH;
  const T *kbase = qbase + C, *vbase = qbase + 2 * C;
  const T *obase = o + (int64_t)b * N * C + h * DH;
  const T *dobase = d_o + (int64_t)b * N * C + h * DH;
  T *dqbase = dqkv + (int64_t)b * N * ld + h * DH;
  const float *lbase = lse + ((int64_t)b * heads + h) * N;
  ATTN_STAMP(0);

  // ---- stage Q, dO, K; delta[q] = sum_d dO O (CPR chunk-threads per row).  ALL loads of the thread are issued before
  // the first LDS store (stamps: the rolled loop, which waits for each iteration's four loads in turn, was 34 % of
  // a workgroup's life)
  {
    constexpr int SIT = NKEY * CPR / NT;
    u32x4 qv[SIT], dv[SIT], kv[SIT], ov[SIT];
    float lv[SIT];
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int q = it * NT + tid, row = q / CPR, c = q % CPR;
      qv[it] = u32x4{0u, 0u, 0u, 0u}; dv[it] = qv[it]; kv[it] = qv[it]; ov[it] = qv[it];
      lv[it] = 1e30f;                                              // padded query rows: P = exp2(-huge) = 0
      if (row < N) {
        qv[it] = *(const u32x4 *)((const char *)(qbase + (int64_t)row * ld) + c * 16);
        kv[it] = *(const u32x4 *)((const char *)(kbase + (int64_t)row * ld) + c * 16);
        dv[it] = *(const u32x4 *)((const char *)(dobase + (int64_t)row * C) + c * 16);
        ov[it] = *(const u32x4 *)((const char *)(obase + (int64_t)row * C) + c * 16);
        if (c == 0) lv[it] = lbase[row] * AR_LOG2E;
      }
    }
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int q = it * NT + tid, row = q / CPR, c = q % CPR;
      const typename Mma<T>::frag dh8 = __builtin_bit_cast(typename Mma<T>::frag, dv[it]), oh8 = __builtin_bit_cast(typename Mma<T>::frag, ov[it]);
      float sd = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) sd += (float)dh8[j] * (float)oh8[j];
#pragma unroll
      for (int m = 1; m < CPR; m <<= 1) sd += __shfl_xor(sd, m, 64);
      *(u32x4 *)(sK + swo<DH>(row, c * 16)) = kv[it];               // (row < NKEY by construction)
      if (row < NP) {
        *(u32x4 *)(sQ + swo<DH>(row, c * 16)) = qv[it];
        *(u32x4 *)(sdO + swo<DH>(row, c * 16)) = dv[it];
        if (c == 0) {
          sDelta[row] = sd;
          sLse[row] = lv[it];
        }
      }
    }
  }
  // this wave's key tiles: K / V fragments (B operands) straight from global; keys >= N are zero rows
  typename Mma<T>::frag kf[KTE][NCH], vf[KTE][NCH];
#pragma unroll
  for (int kt = 0; kt < KTE; ++kt) {
    const int key = (kt * NW + wave) * 16 + li;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      kf[kt][ch] = Mma<T>::zero();
      vf[kt][ch] = Mma<T>::zero();
      if (key < N) {
        kf[kt][ch] = *(const typename Mma<T>::frag *)(kbase + (int64_t)key * ld + ch * 32 + 8 * lg);
        vf[kt][ch] = *(const typename Mma<T>::frag *)(vbase + (int64_t)key * ld + ch * 32 + 8 * lg);
      }
    }
  }
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 dkt[NDT][KTE], dvt[NDT][KTE];
#pragma unroll
  for (int a = 0; a < NDT; ++a)
#pragma unroll
    for (int c = 0; c < KTE; ++c) { dkt[a][c] = zero4; dvt[a][c] = zero4; }
  ATTN_STAMP(1);                                   // loads issued, LDS stores queued
  __syncthreads();
  ATTN_STAMP(2);                                   // operands staged

  const float c1 = scale * AR_LOG2E, inv_c1 = 1.0f / c1;
  const int nsteps = NP >> 5;
  // keys past the sequence can only sit in a wave's LAST tile: its probabilities are zeroed by this lane's flag
  const bool last_key_ok = ((KTE - 1) * NW + wave) * 16 + li < N;
  // K^T fragments of this wave's dQ piece (dt fixed per wave): the same in every step - read once, kept in registers
  const int dq_qt = wave / NDT, dq_dt = wave - dq_qt * NDT;
  typename Mma<T>::frag ka[NKC];
#pragma unroll
  for (int c = 0; c < NKC; ++c) ka[c] = tr_frag<T, DH>(sK, c * 32, dq_dt * 16, li, lg);
  for (int st = 0; st < nsteps; ++st) {
    const int qs = st * 32;
    char *dsb = sdS + (st & 1) * NKEY * 64;
    // ---- S, dP -> P, dS (unscaled) for this wave's key tiles;  D[q = 4*lg + r][key = li].  Row constants ride in as
    // the MFMA chains' initial accumulators: S' = Q K^T - lse / (scale log2 e) and dP' = dO V^T - delta leave the chains
    // ready, so P = exp2(c1 S') and dS = P dP' are two multiplies and one exp2 per element
    f32x4 pt[2][KTE], dst[2][KTE];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      typename Mma<T>::frag qfr[NCH], dof[NCH];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        qfr[ch] = row_frag<T, DH>(sQ, qs + qt * 16, ch, li, lg);
        dof[ch] = row_frag<T, DH>(sdO, qs + qt * 16, ch, li, lg);
      }
      const f32x4 l4 = *(const f32x4 *)(sLse + qs + qt * 16 + 4 * lg), d4 = *(const f32x4 *)(sDelta + qs + qt * 16 + 4 * lg);
      const f32x4 s0 = l4 * (-inv_c1), dp0 = -d4;
#pragma unroll
      for (int kt = 0; kt < KTE; ++kt) {
        f32x4 s = s0, dp = dp0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          s = Mma<T>::mma(qfr[ch], kf[kt][ch], s);
          dp = Mma<T>::mma(dof[ch], vf[kt][ch], dp);
        }
        f32x4 pv;
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(s[r] * c1);
        if (kt == KTE - 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[r] = last_key_ok ? pv[r] : 0.f;
        }
        pt[qt][kt] = pv;
        dst[qt][kt] = pv * dp;
      }
    }
    ATTN_STAMP(3 + 4 * st);                          // S, dP, P, dS of the step
    // ---- dV^T += dO^T P ; dK^T += Q^T dS   (contraction over the 32 queries of the step)
    {
      typename Mma<T>::frag aq[NDT], ado[NDT];
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        aq[dt] = tr_frag<T, DH>(sQ, qs, dt * 16, li, lg);
        ado[dt] = tr_frag<T, DH>(sdO, qs, dt * 16, li, lg);
      }
#pragma unroll
      for (int kt = 0; kt < KTE; ++kt) {
        f32x4 tp[2] = {pt[0][kt], pt[1][kt]}, td[2] = {dst[0][kt], dst[1][kt]};
        const typename Mma<T>::frag pf = Mma<T>::from_tiles(tp), dsf = Mma<T>::from_tiles(td);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          dvt[dt][kt] = Mma<T>::mma(ado[dt], pf, dvt[dt][kt]);
          dkt[dt][kt] = Mma<T>::mma(aq[dt], dsf, dkt[dt][kt]);
        }
        // dS^T[key][q]: this lane holds 4 consecutive q of one key per (qt, kt) -> one 8-byte store (zeros for padding keys)
        const int krow = (kt * NW + wave) * 16 + li;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          typename Vec4<T>::type v = typename Vec4<T>::type{(T)dst[qt][kt][0], (T)dst[qt][kt][1], (T)dst[qt][kt][2], (T)dst[qt][kt][3]};
          *(typename Vec4<T>::type *)(dsb + swo<32>(krow, (qt * 16 + 4 * lg) * 2)) = v;
        }
      }
    }
    ATTN_STAMP(4 + 4 * st);                          // dV, dK accumulated, dS^T stored
    __syncthreads();      // the other dS^T buffer is rewritten only after the NEXT barrier
    ATTN_STAMP(5 + 4 * st);                          // barrier passed
    // ---- dQ^T[d = 16*dt + 4*lg + r][q = 16*qt + li] = K^T dS^T, one (qt, dt) piece per wave
    {
      // all fragment reads first, then the MFMA chain - two interleaved accumulators, so that a product does not wait
      // for its predecessor's result
      typename Mma<T>::frag da[NKC];
#pragma unroll
      for (int c = 0; c < NKC; ++c) da[c] = tr_frag<T, 32>(dsb, c * 32, dq_qt * 16, li, lg);
      f32x4 acc = zero4, acc2 = zero4;
#pragma unroll
      for (int c = 0; c < NKC; c += 2) {
        acc = Mma<T>::mma(ka[c], da[c], acc);
        if (c + 1 < NKC) acc2 = Mma<T>::mma(ka[c + 1], da[c + 1], acc2);
      }
      acc += acc2;
      const int qr = qs + dq_qt * 16 + li;
      if (qr < N) Vec4<T>::store(dqbase + (int64_t)qr * ld + dq_dt * 16 + 4 * lg, acc * scale);
    }
    ATTN_STAMP(6 + 4 * st);                          // dQ piece of the step stored
  }

  // ---- dK, dV rows of this wave's keys
#pragma unroll
  for (int kt = 0; kt < KTE; ++kt) {
    const int key = (kt * NW + wave) * 16 + li;
    if (key < N) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        Vec4<T>::store(dqbase + C + (int64_t)key * ld + dt * 16 + 4 * lg, dkt[dt][kt] * scale);
        Vec4<T>::store(dqbase + 2 * C + (int64_t)key * ld + dt * 16 + 4 * lg, dvt[dt][kt]);
      }
    }
  }
  ATTN_STAMP(36);                                  // dK / dV stores issued
#ifdef M3_ATTN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATTN_STAMP(37);                                  // stores acknowledged
#endif
}

// -------------------------------------------------------------------- forward, long sequences
// N > 256: one workgroup per (image, head, 64-query block); wave w owns the 16 queries q0 + 16w.  K / V are streamed in
// 64-key tiles through a two-slot ring of swizzled LDS images (rows prefetched into registers one tile ahead, one
// barrier per tile), online softmax in the S^T accumulator layout, V read transposed with ds_read_b64_tr_b16.
template <typename T, int DH>
__global__ __launch_bounds__(AR_THREADS, 2) void attention_fwd_stream_kernel(const T *__restrict__ qkv, int N, int heads,
                                                                              T *__restrict__ o,
                                                                              float *__restrict__ lse, float scale) {
  constexpr int RBY = DH * 2, CPR = RBY / 16, NCH = DH / 32, NDT = DH / 16;
  constexpr int KT = 64;                              // keys per tile
  constexpr int NLD = KT * CPR / AR_THREADS;          // chunks per thread per operand per tile (1 / 2)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *sK = smem, *sV = smem + 2 * KT * RBY;         // [2][64][RBY] each

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int nqb = gridDim.x;
  const int log_id = xcd_remap(blockIdx.x + nqb * blockIdx.y, nqb * gridDim.y);
  const int qb = log_id % nqb;
  const int bh = log_id / nqb, b = bh / heads, h = bh - b * heads;
  const int C = heads * DH;
  const int64_t ld = 3 * (int64_t)C;
  const T *qbase = qkv + (int64_t)b * N * ld + h * DH;
  const T *kbase = qbase + C, *vbase = qbase + 2 * C;

  const int qrow = qb * 64 + wave * 16 + li;
  typename Mma<T>::frag qf[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    qf[ch] = Mma<T>::zero();
    if (qrow < N) qf[ch] = *(const typename Mma<T>::frag *)(qbase + (int64_t)qrow * ld + ch * 32 + 8 * lg);
  }
  u32x4 rk[NLD], rv[NLD];
  auto fetch = [&](int key0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = tid + AR_THREADS * i, row = q / CPR, c = q % CPR;
      rk[i] = u32x4{0u, 0u, 0u, 0u};
      rv[i] = u32x4{0u, 0u, 0u, 0u};
      if (key0 + row < N) {
        rk[i] = *(const u32x4 *)((const char *)(kbase + (int64_t)(key0 + row) * ld) + c * 16);
        rv[i] = *(const u32x4 *)((const char *)(vbase + (int64_t)(key0 + row) * ld) + c * 16);
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = tid + AR_THREADS * i, row = q / CPR, c = q % CPR;
      *(u32x4 *)(sK + buf * KT * RBY + swo<DH>(row, c * 16)) = rk[i];
      *(u32x4 *)(sV + buf * KT * RBY + swo<DH>(row, c * 16)) = rv[i];
    }
  };
  fetch(0);
  stash(0);
  __syncthreads();

  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) oacc[dt] = zero4;
  const float c1 = scale * AR_LOG2E;
  float m_run = -INFINITY, l_run = 0.f;             // running max of the raw scores, running sum of exp2
  const int ntiles = (N + KT - 1) / KT;
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * KT, buf = t & 1;
    const bool more = t + 1 < ntiles;
    if (more) fetch(key0 + KT);
    const char *cK = sK + buf * KT * RBY, *cV = sV + buf * KT * RBY;
    // S^T tiles: st[kt][r] = S[q = li][key = key0 + 16*kt + 4*lg + r]
    f32x4 st[4];
    float mt = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4 sv = zero4;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) sv = Mma<T>::mma(row_frag<T, DH>(cK, kt * 16, ch, li, lg), qf[ch], sv);
      if (!more) {                                   // only the last tile can hold keys >= N
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = (key0 + kt * 16 + 4 * lg + r < N) ? sv[r] : -INFINITY;
      }
      mt = fmaxf(fmaxf(mt, fmaxf(sv[0], sv[1])), fmaxf(sv[2], sv[3]));
      st[kt] = sv;
    }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c1);
    const float mc = m_new * c1;
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(st[kt][r] * c1 - mc);
        st[kt][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) oacc[dt] *= alpha;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const typename Mma<T>::frag pf = Mma<T>::from_tiles(&st[cc * 2]);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) oacc[dt] = Mma<T>::mma(tr_frag<T, DH>(cV, cc * 32, dt * 16, li, lg), pf, oacc[dt]);
    }
    if (more) stash(buf ^ 1);                        // that slot was last read before the previous barrier
    __syncthreads();
  }
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  if (qrow < N) {
    const float inv = 1.0f / l_tot;
    T *orow = o + ((int64_t)b * N + qrow) * C + h * DH;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) Vec4<T>::store(orow + dt * 16 + 4 * lg, oacc[dt] * inv);
    if (lg == 0) lse[((int64_t)b * heads + h) * N + qrow] = m_run * scale + __logf(l_tot);
  }
}

template <typename T>
static int launch_attention_fwd_stream_t(const void *qkv, int B, int N, int heads, int dh, void *o, float *lse, float scale,
                                hipStream_t s) {
  const dim3 grid((N + 63) / 64, B * heads);
  const size_t lds = (size_t)4 * 64 * dh * 2;
  if (dh == 32)
    hipLaunchKernelGGL((attention_fwd_stream_kernel<T, 32>), grid, dim3(AR_THREADS), lds, s, (const T *)qkv, N, heads,
                       (T *)o, lse, scale);
  else
    hipLaunchKernelGGL((attention_fwd_stream_kernel<T, 64>), grid, dim3(AR_THREADS), lds, s, (const T *)qkv, N, heads,
                       (T *)o, lse, scale);
  return check_launch("m3_attention_fwd");
}

// ------------------------------------------------------------------- backward, long sequences
// N > 256: one workgroup per (image, head, 256-key block).  The key block's K image stays in LDS, its K / V
// fragments and dK^T / dV^T in registers; the queries are swept in 32-row steps whose Q / dO rows are prefetched one
// step ahead (registers -> the other half of a two-slot LDS ring, visible after the step's single barrier).
// delta[q] = <dO[q], O[q]> comes from a row kernel run once per call; the key block's dQ contribution goes to an fp32
// slab summed in key-block order by attention_dq_reduce_kernel (attention.hip).
template <typename T, int DH>
__global__ __launch_bounds__(64) void attention_delta_kernel(const T *__restrict__ o, const T *__restrict__ d_o,
                                                             int64_t rows, int heads, float *__restrict__ delta) {
  // one wave per 64 (token, head) rows; delta laid out [token][head]
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= rows * heads) return;
  const T *po = o + i * DH, *pd = d_o + i * DH;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < DH / 8; ++c) {
    const typename Mma<T>::frag a = *(const typename Mma<T>::frag *)(po + c * 8), b = *(const typename Mma<T>::frag *)(pd + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (float)a[j] * (float)b[j];
  }
  delta[i] = s;
}

template <typename T, int DH>
__global__ __launch_bounds__(DH == 32 ? 256 : 512, DH == 32 ? 2 : 1) void attention_bwd_stream_kernel(
    const T *__restrict__ qkv, const T *__restrict__ d_o, const float *__restrict__ lse,
    const float *__restrict__ delta, int N, int heads, T *__restrict__ dqkv, float *__restrict__ dq_ws, float scale) {
  constexpr int RBY = DH * 2, CPR = RBY / 16, NCH = DH / 32, NDT = DH / 16;
  constexpr int NW = DH == 32 ? 4 : 8, NT = NW * 64, KTW = 16 / NW, KB = 256;
  static_assert(2 * NDT == NW && 32 * CPR * 2 == NT, "one dQ piece and one staged chunk per thread");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *sK = smem;                                     // [256 keys][RBY]
  char *sQ = sK + KB * RBY;                            // [2][32 rows][RBY]
  char *sdO = sQ + 2 * 32 * RBY;
  char *sdS = sdO + 2 * 32 * RBY;                      // [2][256 keys][64 B]
  float *sLse = (float *)(sdS + 2 * KB * 64);          // [2][32]: lse * log2(e)
  float *sDelta = sLse + 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int nkb = (N + KB - 1) / KB;
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = wid / nkb, kbi = wid - bh * nkb, b = bh / heads, h = bh - b * heads;
  const int kb0 = kbi * KB;
  const int C = heads * DH;
  const int64_t ld = 3 * (int64_t)C;
  const T *qbase = qkv + (int64_t)b * N * ld + h * DH;
  const T *kbase = qbase + C, *vbase = qbase + 2 * C;
  const T *dobase = d_o + (int64_t)b * N * C + h * DH;
  T *dqbase = dqkv + (int64_t)b * N * ld + h * DH;
  const float *lbase = lse + ((int64_t)b * heads + h) * N;
  const float *dbase = delta + (int64_t)b * N * heads + h;             // [token][head]
  float *dqw = dq_ws + ((int64_t)kbi * (gridDim.x / nkb) + bh) * N * DH;
  const int nkeys = (N - kb0 < KB) ? N - kb0 : KB;
  const int nkt = (nkeys + 15) >> 4;                  // valid key tiles of this block
  const int nkc = (nkeys + 31) >> 5;                  // 32-key chunks that matter for dQ

  // ---- K image of the key block, zero the dS^T images
  for (int q = tid; q < KB * CPR; q += NT) {
    const int row = q / CPR, c = q % CPR;
    u32x4 kv = u32x4{0u, 0u, 0u, 0u};
    if (row < nkeys) kv = *(const u32x4 *)((const char *)(kbase + (int64_t)(kb0 + row) * ld) + c * 16);
    *(u32x4 *)(sK + swo<DH>(row, c * 16)) = kv;
  }
  for (int q = tid; q < 2 * KB * 4; q += NT) *(u32x4 *)(sdS + q * 16) = u32x4{0u, 0u, 0u, 0u};
  typename Mma<T>::frag kf[KTW][NCH], vf[KTW][NCH];
#pragma unroll
  for (int kt = 0; kt < KTW; ++kt) {
    const int key = kb0 + (kt * NW + wave) * 16 + li;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      kf[kt][ch] = Mma<T>::zero();
      vf[kt][ch] = Mma<T>::zero();
      if (key < N) {
        kf[kt][ch] = *(const typename Mma<T>::frag *)(kbase + (int64_t)key * ld + ch * 32 + 8 * lg);
        vf[kt][ch] = *(const typename Mma<T>::frag *)(vbase + (int64_t)key * ld + ch * 32 + 8 * lg);
      }
    }
  }
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 dkt[NDT][KTW], dvt[NDT][KTW];
#pragma unroll
  for (int a = 0; a < NDT; ++a)
#pragma unroll
    for (int c = 0; c < KTW; ++c) { dkt[a][c] = zero4; dvt[a][c] = zero4; }

  // staging assignment: thread -> one 16-byte chunk of the step: first half of the threads Q, second half dO
  const int half_t_ = NT / 2;
  const bool is_do = tid >= half_t_;
  const int sq = is_do ? tid - half_t_ : tid;          // 0 .. 32*CPR-1
  const int srow = sq / CPR, sc = sq % CPR;
  auto fetch = [&](int qs, u32x4 &v, float &l2, float &dl) {
    const int qr = qs + srow;
    v = u32x4{0u, 0u, 0u, 0u};
    if (qr < N) {
      if (is_do) v = *(const u32x4 *)((const char *)(dobase + (int64_t)qr * C) + sc * 16);
      else v = *(const u32x4 *)((const char *)(qbase + (int64_t)qr * ld) + sc * 16);
    }
    l2 = 1e30f; dl = 0.f;                               // padded query rows: P = exp2(-huge) = 0
    if (tid < 32 && qs + tid < N) { l2 = lbase[qs + tid] * AR_LOG2E; dl = dbase[(int64_t)(qs + tid) * heads]; }
  };
  auto stash = [&](int buf, const u32x4 &v, float l2, float dl) {
    *(u32x4 *)((is_do ? sdO : sQ) + buf * 32 * RBY + swo<DH>(srow, sc * 16)) = v;
    if (tid < 32) { sLse[buf * 32 + tid] = l2; sDelta[buf * 32 + tid] = dl; }
  };
  {
    u32x4 v; float l2, dl;
    fetch(0, v, l2, dl);
    stash(0, v, l2, dl);
  }
  __syncthreads();

  const float c1 = scale * AR_LOG2E, inv_c1 = 1.0f / c1;
  const int nsteps = (N + 31) >> 5;
  // a full key block (every block but a sequence's last) runs without the per-tile guards: straight-line MFMA chains
  // (a uniform run-time guard is a basic-block boundary the scheduler does not move MFMAs across)
  auto run = [&](auto full_c) {
  constexpr bool FULL = decltype(full_c)::value;
  for (int st = 0; st < nsteps; ++st) {
    const int qs = st * 32, buf = st & 1;
    const char *cQ = sQ + buf * 32 * RBY, *cdO = sdO + buf * 32 * RBY;
    char *dsb = sdS + buf * KB * 64;
    u32x4 nv; float nl2, ndl;
    const bool more = st + 1 < nsteps;
    if (more) fetch(qs + 32, nv, nl2, ndl);            // rows of the next step: in flight under this step's MFMAs
    // ---- S, dP -> P, dS (unscaled) for this wave's key tiles;  D[q = 4*lg + r][key = li]
    f32x4 pt[2][KTW], dst[2][KTW];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      typename Mma<T>::frag qfr[NCH], dof[NCH];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        qfr[ch] = row_frag<T, DH>(cQ, qt * 16, ch, li, lg);
        dof[ch] = row_frag<T, DH>(cdO, qt * 16, ch, li, lg);
      }
      f32x4 s0, dp0;                             // row constants as the chains' initial accumulators (see attention_bwd_res_kernel)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = -sLse[buf * 32 + qt * 16 + 4 * lg + r] * inv_c1;
        dp0[r] = -sDelta[buf * 32 + qt * 16 + 4 * lg + r];
      }
#pragma unroll
      for (int kt = 0; kt < KTW; ++kt) {
        const int tix = kt * NW + wave;
        if (FULL || tix < nkt) {
          f32x4 s = s0, dp = dp0;
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) {
            s = Mma<T>::mma(qfr[ch], kf[kt][ch], s);
            dp = Mma<T>::mma(dof[ch], vf[kt][ch], dp);
          }
          f32x4 pv;
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(s[r] * c1);
          if (!FULL && tix == nkt - 1 && tix * 16 + li >= nkeys) pv = zero4;      // keys past N live in the last tile only
          pt[qt][kt] = pv;
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[qt][kt][r] = pv[r] * dp[r];
        } else {
          pt[qt][kt] = zero4;
          dst[qt][kt] = zero4;
        }
      }
    }
    ATTN_STAMP(3 + 4 * st);                          // S, dP, P, dS of the step
    // ---- dV^T += dO^T P ; dK^T += Q^T dS   (contraction over the 32 queries of the step)
    {
      typename Mma<T>::frag aq[NDT], ado[NDT];
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        aq[dt] = tr_frag<T, DH>(cQ, 0, dt * 16, li, lg);
        ado[dt] = tr_frag<T, DH>(cdO, 0, dt * 16, li, lg);
      }
#pragma unroll
      for (int kt = 0; kt < KTW; ++kt) {
        if (FULL || kt * NW + wave < nkt) {
          f32x4 tp[2] = {pt[0][kt], pt[1][kt]}, td[2] = {dst[0][kt], dst[1][kt]};
          const typename Mma<T>::frag pf = Mma<T>::from_tiles(tp), dsf = Mma<T>::from_tiles(td);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            dvt[dt][kt] = Mma<T>::mma(ado[dt], pf, dvt[dt][kt]);
            dkt[dt][kt] = Mma<T>::mma(aq[dt], dsf, dkt[dt][kt]);
          }
          const int krow = (kt * NW + wave) * 16 + li;
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            typename Vec4<T>::type v = typename Vec4<T>::type{(T)dst[qt][kt][0], (T)dst[qt][kt][1], (T)dst[qt][kt][2], (T)dst[qt][kt][3]};
            *(typename Vec4<T>::type *)(dsb + swo<32>(krow, (qt * 16 + 4 * lg) * 2)) = v;
          }
        }
      }
    }
    if (more) stash(buf ^ 1, nv, nl2, ndl);     // the other ring slot was last read before the previous barrier
    __syncthreads();                            // dS^T of this step and the next step's rows are visible
    // ---- dQ^T[d][q] contribution of this key block, one (qt, dt) piece per wave -> fp32 slab
    {
      const int qt = wave / NDT, dt = wave - qt * NDT;
      // all fragment reads first (clamped chunk index: always inside the images), then the dependent MFMA chain
      typename Mma<T>::frag ka[AR_MAXN / 32], da[AR_MAXN / 32];
#pragma unroll
      for (int c = 0; c < AR_MAXN / 32; ++c) {
        const int cc = (FULL || c < nkc) ? c : nkc - 1;
        ka[c] = tr_frag<T, DH>(sK, cc * 32, dt * 16, li, lg);
        da[c] = tr_frag<T, 32>(dsb, cc * 32, qt * 16, li, lg);
      }
      f32x4 acc = zero4;
#pragma unroll
      for (int c = 0; c < AR_MAXN / 32; ++c)
        if (FULL || c < nkc) acc = Mma<T>::mma(ka[c], da[c], acc);
      const int qr = qs + qt * 16 + li;
      if (qr < N) *(f32x4 *)(dqw + (int64_t)qr * DH + dt * 16 + 4 * lg) = acc * scale;
    }
  }
  };
  if (nkeys == KB) run(std::true_type{}); else run(std::false_type{});

  // ---- dK, dV rows of this wave's keys
#pragma unroll
  for (int kt = 0; kt < KTW; ++kt) {
    const int key = kb0 + (kt * NW + wave) * 16 + li;
    if (key < N) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        Vec4<T>::store(dqbase + C + (int64_t)key * ld + dt * 16 + 4 * lg, dkt[dt][kt] * scale);
        Vec4<T>::store(dqbase + 2 * C + (int64_t)key * ld + dt * 16 + 4 * lg, dvt[dt][kt]);
      }
    }
  }
}

size_t attn_stream_bwd_lds(int dh) { return (size_t)256 * dh * 2 + 4 * 32 * dh * 2 + 2 * 256 * 64 + 128 * sizeof(float); }

// dq_ws: [nkb][B*heads][N][dh] slabs followed by delta [B*N*heads]
template <typename T>
static int launch_attention_bwd_stream_t(const void *qkv, const void *o, const void *d_o, const float *lse, int B, int N, int heads,
                                int dh, void *dqkv, float *dq_ws, float scale, hipStream_t s) {
  const int nkb = (N + 255) / 256;
  float *delta = dq_ws + (int64_t)nkb * B * heads * N * dh;
  const int64_t rows = (int64_t)B * N;
  const unsigned dblocks = (unsigned)((rows * heads + 63) / 64);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)attention_bwd_stream_kernel<T, 32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)attn_stream_bwd_lds(32));
    (void)hipFuncSetAttribute((const void *)attention_bwd_stream_kernel<T, 64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)attn_stream_bwd_lds(64));
    attr_set = true;
  }
  if (dh == 32) {
    hipLaunchKernelGGL((attention_delta_kernel<T, 32>), dim3(dblocks), dim3(64), 0, s, (const T *)o, (const T *)d_o, rows,
                       heads, delta);
    hipLaunchKernelGGL((attention_bwd_stream_kernel<T, 32>), dim3(B * heads * nkb), dim3(256), attn_stream_bwd_lds(32), s,
                       (const T *)qkv, (const T *)d_o, lse, delta, N, heads, (T *)dqkv, dq_ws, scale);
  } else {
    hipLaunchKernelGGL((attention_delta_kernel<T, 64>), dim3(dblocks), dim3(64), 0, s, (const T *)o, (const T *)d_o, rows,
                       heads, delta);
    hipLaunchKernelGGL((attention_bwd_stream_kernel<T, 64>), dim3(B * heads * nkb), dim3(512), attn_stream_bwd_lds(64), s,
                       (const T *)qkv, (const T *)d_o, lse, delta, N, heads, (T *)dqkv, dq_ws, scale);
  }
  return check_launch("m3_attention_bwd");
}

static inline int attn_res_fwd_nkt(int N) { return ((N + 15) / 16 + 3) / 4 * 4; }      // key tiles rounded up to 4, 8, 12, 16
size_t attn_res_fwd_lds(int N, int dh) { return (size_t)2 * attn_res_fwd_nkt(N) * 16 * dh * 2; }
// tiles per wave of the short-sequence backward for N keys: ceil(key tiles / waves); LDS of that instance
static inline int attn_res_bwd_kte(int N, int dh) { const int nw = dh == 32 ? 4 : 8; return ((N + 15) / 16 + nw - 1) / nw; }
size_t attn_res_bwd_lds(int N, int dh) {
  const size_t np = (N + 31) & ~31;
  const size_t nkey = (size_t)attn_res_bwd_kte(N, dh) * (dh == 32 ? 4 : 8) * 16;
  return 2 * np * dh * 2 + nkey * dh * 2 + 2 * nkey * 64 + 2 * np * sizeof(float);
}

template <typename T>
static int launch_attention_fwd_res_t(const void *qkv, int B, int N, int heads, int dh, void *o, float *lse, float scale,
                             hipStream_t s) {
  const size_t lds = attn_res_fwd_lds(N, dh);
  const int nkt = attn_res_fwd_nkt(N);
#define M3_AF_GO(DH_, NK_) hipLaunchKernelGGL((attention_fwd_res_kernel<T, DH_, NK_>), dim3(B * heads), dim3(AR_THREADS), lds, s, \
                                             (const T *)qkv, N, heads, (T *)o, lse, scale)
  if (dh == 32) {
    if (nkt == 4) M3_AF_GO(32, 4); else if (nkt == 8) M3_AF_GO(32, 8); else if (nkt == 12) M3_AF_GO(32, 12); else M3_AF_GO(32, 16);
  } else {
    if (nkt == 4) M3_AF_GO(64, 4); else if (nkt == 8) M3_AF_GO(64, 8); else if (nkt == 12) M3_AF_GO(64, 12); else M3_AF_GO(64, 16);
  }
#undef M3_AF_GO
  return check_launch("m3_attention_fwd");
}

template <typename T>
static int launch_attention_bwd_res_t(const void *qkv, const void *o, const void *d_o, const float *lse, int B, int N, int heads,
                             int dh, void *dqkv, float scale, hipStream_t s) {
  const size_t lds = attn_res_bwd_lds(N, dh);
  const int kte = attn_res_bwd_kte(N, dh);
  static bool attr_set = false;
#define M3_AB_ATTR(DH_, K_) (void)hipFuncSetAttribute((const void *)attention_bwd_res_kernel<T, DH_, K_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                      (int)attn_res_bwd_lds(K_ * (DH_ == 32 ? 64 : 128), DH_))
  if (!attr_set) {
    M3_AB_ATTR(32, 1); M3_AB_ATTR(32, 2); M3_AB_ATTR(32, 3); M3_AB_ATTR(32, 4); M3_AB_ATTR(64, 1); M3_AB_ATTR(64, 2);
    attr_set = true;
  }
#undef M3_AB_ATTR
#define M3_AB_GO(DH_, K_, NT_) hipLaunchKernelGGL((attention_bwd_res_kernel<T, DH_, K_>), dim3(B * heads), dim3(NT_), lds, s, (const T *)qkv, \
                                                 (const T *)o, (const T *)d_o, lse, N, heads, (T *)dqkv, scale)
  if (dh == 32) {
    if (kte == 1) M3_AB_GO(32, 1, 256); else if (kte == 2) M3_AB_GO(32, 2, 256); else if (kte == 3) M3_AB_GO(32, 3, 256); else M3_AB_GO(32, 4, 256);
  } else {
    if (kte == 1) M3_AB_GO(64, 1, 512); else M3_AB_GO(64, 2, 512);
  }
#undef M3_AB_GO
  return check_launch("m3_attention_bwd");
}

// fp16 / bf16 front doors (attention.hip picks these kernels for every 16-bit dtype)
int launch_attention_fwd_stream(int dtype, const void *qkv, int B, int N, int heads, int dh, void *o, float *lse, float scale, hipStream_t s) {
  return dtype == M3_BF16 ? launch_attention_fwd_stream_t<bf16_t>(qkv, B, N, heads, dh, o, lse, scale, s) : launch_attention_fwd_stream_t<half_t>(qkv, B, N, heads, dh, o, lse, scale, s);
}

int launch_attention_bwd_stream(int dtype, const void *qkv, const void *o, const void *d_o, const float *lse, int B, int N, int heads, int dh, void *dqkv, float *dq_ws, float scale, hipStream_t s) {
  return dtype == M3_BF16 ? launch_attention_bwd_stream_t<bf16_t>(qkv, o, d_o, lse, B, N, heads, dh, dqkv, dq_ws, scale, s) : launch_attention_bwd_stream_t<half_t>(qkv, o, d_o, lse, B, N, heads, dh, dqkv, dq_ws, scale, s);
}

int launch_attention_fwd_res(int dtype, const void *qkv, int B, int N, int heads, int dh, void *o, float *lse, float scale, hipStream_t s) {
  return dtype == M3_BF16 ? launch_attention_fwd_res_t<bf16_t>(qkv, B, N, heads, dh, o, lse, scale, s) : launch_attention_fwd_res_t<half_t>(qkv, B, N, heads, dh, o, lse, scale, s);
}

int launch_attention_bwd_res(int dtype, const void *qkv, const void *o, const void *d_o, const float *lse, int B, int N, int heads, int dh, void *dqkv, float scale, hipStream_t s) {
  return dtype == M3_BF16 ? launch_attention_bwd_res_t<bf16_t>(qkv, o, d_o, lse, B, N, heads, dh, dqkv, scale, s) : launch_attention_bwd_res_t<half_t>(qkv, o, d_o, lse, B, N, heads, dh, dqkv, scale, s);
}

}  // namespace m3

#ifdef M3_ATTN_STAMPS
extern "C" int m3_debug_attn_stamps(unsigned long long *dst, int wgs) {
  if (wgs > m3::ASTAMP_WGS) wgs = m3::ASTAMP_WGS;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(m3::g_attn_stamps), (size_t)wgs * m3::ASTAMP_N * sizeof(unsigned long long)) == hipSuccess ? M3_OK : M3_ERR_LAUNCH;
}
#endif
