// Fused (flash-style) multi-head attention forward / backward for gfx950 on the packed
// qkv activations of the ViT block:  o = softmax(q k^T * dh^-0.5) v.
//
// Replaces Attention.forward's q@k^T -> softmax -> @v chain and its autograd
// (models/moe/ckpt/vision_transformer_moe.py:299-313; dense twin
// models/backbones/vit.py:177-207) without materialising the [B,h,N,N] scores.
// qkv is exactly what the qkv Linear wrote: [token][3][head][dh]; o is [token][head*dh].
//
// MFMA operand plan (16x16 tiles; "key on the lane" so that no accumulator ever has to
// move between lanes, cf. cdna_hip_programming.md App. B):
//   fwd : S^T[key][q] = K Q^T     (A = K rows from LDS, B = Q rows held in registers)
//         O^T[d][q]  += V^T P^T   (A = V^T from a transposed LDS image, B = the lane's own
//                                  exp'd S^T values - C layout == B-operand layout)
//   bwd : one workgroup per (image, head), each wave owns 64 keys and keeps dK^T / dV^T
//         for them in registers while sweeping 32-row query blocks:
//         S[q][key] = Q K^T, dP = dO V^T (B = K / V fragments resident in registers);
//         dV^T += dO^T P, dK^T += Q^T dS (A from transposed LDS images of dO / Q, B = own
//         registers); dS goes through LDS once for dQ^T[d][q] = K^T dS^T.
// f32 uses the exact 16x16x4 f32 MFMA, f16 the 16x16x32 f16 MFMA; softmax statistics,
// accumulators and LSE are fp32.
#include "common.h"

namespace m3 {

constexpr int AT_THREADS = 256;
constexpr int AT_KT = 64;       // keys per LDS tile (fwd)
constexpr int AT_QB = 64;       // query rows per workgroup (fwd): 4 waves x 16

// 4 consecutive contraction elements starting at p, for each of the CT sub-tiles 16 apart
template <typename T>
__device__ __forceinline__ typename Mma<T>::frag load_slots(const T *p);
template <>
__device__ __forceinline__ f32x4 load_slots<float>(const float *p) { return *(const f32x4 *)p; }
template <>
__device__ __forceinline__ f16x8 load_slots<half_t>(const half_t *p) {
  const f16x4 a = *(const f16x4 *)p, b = *(const f16x4 *)(p + 16);
  return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

template <typename T, int DH>
__global__ __launch_bounds__(AT_THREADS, 2) void attention_fwd_kernel(const T *__restrict__ qkv, int B, int N, int heads,
                                                                       T *__restrict__ o, float *__restrict__ lse,
                                                                       float scale) {
  typedef Mma<T> MM;
  typedef typename MM::frag frag;
  constexpr int ES = (int)sizeof(T);
  constexpr int KC = MM::KC, EPL = MM::EPL, CT = MM::CT;
  constexpr int NCH = DH / KC;              // d chunks
  constexpr int NDT = DH / 16;              // d tiles of the output
  constexpr int KSTR = DH * ES + 16;        // bytes, sK row stride
  constexpr int VSTR = AT_KT * ES + 16;     // bytes, sVt row stride
  constexpr int CPRK = DH * ES / 16;        // 16-byte chunks per K/V row
  constexpr int EPC = 16 / ES;

  __shared__ __attribute__((aligned(16))) char sK[AT_KT * KSTR];
  __shared__ __attribute__((aligned(16))) char sVt[DH * VSTR];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  // XCD-aware order: the query blocks of one (image, head) and the neighbouring heads of the same
  // image (which share 128-byte lines of the packed qkv rows) get consecutive logical ids -> one XCD.
  const int nqb = gridDim.x;
  const int log_id = xcd_remap(blockIdx.x + nqb * blockIdx.y, nqb * gridDim.y);
  const int qb = log_id % nqb;
  const int bh = log_id / nqb, b = bh / heads, h = bh - b * heads;
  const int C = heads * DH;
  const int64_t ld = 3 * (int64_t)C;
  const T *qbase = qkv + (int64_t)b * N * ld + h * DH;
  const T *kbase = qbase + C, *vbase = qbase + 2 * C;

  const int q0 = qb * AT_QB + wave * 16;
  const int qrow = q0 + li;
  frag qf[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (qrow < N) qf[ch] = *(const frag *)(qbase + (int64_t)qrow * ld + ch * KC + EPL * lg);
    else qf[ch] = MM::zero();
  }

  f32x4 oacc[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  for (int key0 = 0; key0 < N; key0 += AT_KT) {
    __syncthreads();
    // ---- stage K rows and V^T
    for (int q = tid; q < AT_KT * CPRK; q += AT_THREADS) {
      const int row = q / CPRK, c = q - row * CPRK;
      const int key = key0 + row;
      u32x4 kv = u32x4{0u, 0u, 0u, 0u}, vv = u32x4{0u, 0u, 0u, 0u};
      if (key < N) {
        kv = *(const u32x4 *)((const char *)(kbase + (int64_t)key * ld) + c * 16);
        vv = *(const u32x4 *)((const char *)(vbase + (int64_t)key * ld) + c * 16);
      }
      *(u32x4 *)(sK + row * KSTR + c * 16) = kv;
      const T *ve = (const T *)&vv;
#pragma unroll
      for (int j = 0; j < EPC; ++j) *(T *)(sVt + (c * EPC + j) * VSTR + row * ES) = ve[j];
    }
    __syncthreads();

    // ---- S^T tiles: [key = 16*kt + 4*lg + r][q = li]
    f32x4 st[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const frag kf = *(const frag *)(sK + (kt * 16 + li) * KSTR + (ch * KC + EPL * lg) * ES);
        st[kt] = MM::mma(kf, qf[ch], st[kt]);
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + kt * 16 + 4 * lg + r;
        const float s = (key < N) ? st[kt][r] * scale : -INFINITY;
        st[kt][r] = s;
        mt = fmaxf(mt, s);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(st[kt][r] - m_new);
        st[kt][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < NDT; ++i) oacc[i] *= alpha;

    // ---- O^T += V^T P^T
#pragma unroll
    for (int cc = 0; cc < AT_KT / KC; ++cc) {
      const frag pf = MM::from_tiles(&st[cc * CT]);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const frag vf = load_slots<T>((const T *)(sVt + (dt * 16 + li) * VSTR) + cc * KC + 4 * lg);
        oacc[dt] = MM::mma(vf, pf, oacc[dt]);
      }
    }
  }

  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  if (qrow < N) {
    const float inv = 1.0f / l_tot;
    T *orow = o + ((int64_t)b * N + qrow) * C + h * DH;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) Vec4<T>::store(orow + dt * 16 + 4 * lg, oacc[dt] * inv);
    if (lg == 0) lse[((int64_t)b * heads + h) * N + qrow] = m_run + __logf(l_tot);
  }
}

// ------------------------------------------------------------------------ backward
constexpr int AB_QB = 32;        // query rows per step
constexpr int AB_KEYS = 256;     // keys per workgroup (4 waves x 64)

template <typename T, int DH>
__global__ __launch_bounds__(AT_THREADS, ((DH == 32 && sizeof(T) == 2) ? 2 : 1)) void attention_bwd_kernel(const T *__restrict__ qkv, const T *__restrict__ o,
                                                                       const T *__restrict__ d_o,
                                                                       const float *__restrict__ lse, int B, int N,
                                                                       int heads, T *__restrict__ dqkv, float *__restrict__ dq_ws, float scale) {
  typedef Mma<T> MM;
  typedef typename MM::frag frag;
  constexpr int ES = (int)sizeof(T);
  constexpr int KC = MM::KC, EPL = MM::EPL, CT = MM::CT;
  constexpr int NCH = DH / KC;
  constexpr int NDT = DH / 16;
  constexpr int QCH = AB_QB / KC;                 // q chunks per step (f16: 1, f32: 2)
  constexpr int RSTR = DH * ES + 16;              // row-major Q / dO images
  constexpr int TSTR = AB_QB * ES + 16;           // transposed Q^T / dO^T images
  constexpr int SSTR = AB_KEYS * ES + 16;         // K^T and dS images (keys contiguous)
  constexpr int CPR = DH * ES / 16;
  constexpr int EPC = 16 / ES;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *sKt = smem;                               // [DH][SSTR]
  char *sdS = sKt + DH * SSTR;                    // [AB_QB][SSTR]
  char *sQ = sdS + AB_QB * SSTR;                  // [AB_QB][RSTR]
  char *sdO = sQ + AB_QB * RSTR;
  char *sQt = sdO + AB_QB * RSTR;                 // [DH][TSTR]
  char *sdOt = sQt + DH * TSTR;
  float *sLse = (float *)(sdOt + DH * TSTR);      // [AB_QB]
  float *sDelta = sLse + AB_QB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  // one workgroup per (image, head, 256-key block); the key blocks of one (image, head) read the same Q / dO
  // rows and neighbouring heads share lines: consecutive logical ids -> one XCD
  const int nkb = (N + AB_KEYS - 1) / AB_KEYS;
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = wid / nkb, kbi = wid - bh * nkb, b = bh / heads, h = bh - b * heads;
  const int C = heads * DH;
  const int64_t ld = 3 * (int64_t)C;
  const T *qbase = qkv + (int64_t)b * N * ld + h * DH;
  const T *kbase = qbase + C, *vbase = qbase + 2 * C;
  const T *obase = o + (int64_t)b * N * C + h * DH;
  const T *dobase = d_o + (int64_t)b * N * C + h * DH;
  T *dqbase = dqkv + (int64_t)b * N * ld + h * DH;
  const float *lbase = lse + ((int64_t)b * heads + h) * N;
  const int kw0 = wave * 64;

  // Sequences longer than AB_KEYS keys: each 256-key block has its own workgroup.  dK/dV of a key block are
  // complete after its sweep over the queries; its dQ contribution goes to an fp32 slab
  // dq_ws[key block][image, head][N][DH], summed in key-block order by attention_dq_reduce_kernel (deterministic).
  const int64_t nbh = gridDim.x / nkb;
  float *dqw = dq_ws ? dq_ws + ((int64_t)kbi * nbh + bh) * N * DH : nullptr;
  const int kb0 = kbi * AB_KEYS;
  {
  __syncthreads();
  // ---- K^T image (all keys) + this wave's K / V fragments
  for (int q = tid; q < AB_KEYS * CPR; q += AT_THREADS) {
    const int row = q / CPR, c = q - row * CPR;
    u32x4 kv = u32x4{0u, 0u, 0u, 0u};
    if (kb0 + row < N) kv = *(const u32x4 *)((const char *)(kbase + (int64_t)(kb0 + row) * ld) + c * 16);
    const T *ke = (const T *)&kv;
#pragma unroll
    for (int j = 0; j < EPC; ++j) *(T *)(sKt + (c * EPC + j) * SSTR + row * ES) = ke[j];
  }
  frag kf[4][NCH], vf[4][NCH];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int key = kb0 + kw0 + kt * 16 + li;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if (key < N) {
        kf[kt][ch] = *(const frag *)(kbase + (int64_t)key * ld + ch * KC + EPL * lg);
        vf[kt][ch] = *(const frag *)(vbase + (int64_t)key * ld + ch * KC + EPL * lg);
      } else {
        kf[kt][ch] = MM::zero();
        vf[kt][ch] = MM::zero();
      }
    }
  }
  f32x4 dkt[NDT][4], dvt[NDT][4];
#pragma unroll
  for (int a = 0; a < NDT; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) { dkt[a][c] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[a][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nkeys = (N - kb0 < AB_KEYS) ? N - kb0 : AB_KEYS;
  const int nkc = (nkeys + KC - 1) / KC;   // key chunks that matter for dQ

  for (int qs = 0; qs < N; qs += AB_QB) {
    __syncthreads();
    // ---- stage Q, dO (row-major + transposed), lse, delta
    for (int q = tid; q < AB_QB * CPR; q += AT_THREADS) {
      const int row = q / CPR, c = q - row * CPR;
      const int qr = qs + row;
      u32x4 qv = u32x4{0u, 0u, 0u, 0u}, dv = u32x4{0u, 0u, 0u, 0u};
      if (qr < N) {
        qv = *(const u32x4 *)((const char *)(qbase + (int64_t)qr * ld) + c * 16);
        dv = *(const u32x4 *)((const char *)(dobase + (int64_t)qr * C) + c * 16);
      }
      *(u32x4 *)(sQ + row * RSTR + c * 16) = qv;
      *(u32x4 *)(sdO + row * RSTR + c * 16) = dv;
      const T *qe = (const T *)&qv, *de = (const T *)&dv;
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        *(T *)(sQt + (c * EPC + j) * TSTR + row * ES) = qe[j];
        *(T *)(sdOt + (c * EPC + j) * TSTR + row * ES) = de[j];
      }
    }
    {
      // delta[row] = sum_d dO*O ; 8 threads per row
      const int row = tid >> 3, part = tid & 7;
      const int qr = qs + row;
      float s = 0.f;
      if (qr < N) {
        constexpr int PER = DH / 8;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const int d = part * PER + j;
          s += (float)dobase[(int64_t)qr * C + d] * (float)obase[(int64_t)qr * C + d];
        }
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if (part == 0) {
        sDelta[row] = s;
        sLse[row] = (qr < N) ? lbase[qr] : 0.f;
      }
    }
    __syncthreads();

    // ---- S, dP -> P, dS for this wave's 64 keys
    f32x4 pt[2][4], dst[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      frag qfr[NCH], dof[NCH];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        qfr[ch] = *(const frag *)(sQ + (qt * 16 + li) * RSTR + (ch * KC + EPL * lg) * ES);
        dof[ch] = *(const frag *)(sdO + (qt * 16 + li) * RSTR + (ch * KC + EPL * lg) * ES);
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          s = MM::mma(qfr[ch], kf[kt][ch], s);      // D[q = 4*lg + r][key = li]
          dp = MM::mma(dof[ch], vf[kt][ch], dp);
        }
        const int key = kb0 + kw0 + kt * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = qt * 16 + 4 * lg + r;
          const bool ok = (key < N) && (qs + ql < N);
          const float p = ok ? __expf(s[r] * scale - sLse[ql]) : 0.f;
          pt[qt][kt][r] = p;
          dst[qt][kt][r] = p * (dp[r] - sDelta[ql]) * scale;
        }
      }
    }
    // ---- dV^T += dO^T P ; dK^T += Q^T dS   (contraction over the 32 queries)
#pragma unroll
    for (int qc = 0; qc < QCH; ++qc) {
      frag aq[NDT], ado[NDT];
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        aq[dt] = load_slots<T>((const T *)(sQt + (dt * 16 + li) * TSTR) + qc * KC + 4 * lg);
        ado[dt] = load_slots<T>((const T *)(sdOt + (dt * 16 + li) * TSTR) + qc * KC + 4 * lg);
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 tp[2], td[2];
        tp[0] = pt[qc * CT][kt]; td[0] = dst[qc * CT][kt];
        tp[1] = pt[(qc * CT + CT - 1)][kt]; td[1] = dst[(qc * CT + CT - 1)][kt];
        const frag pf = MM::from_tiles(tp), dsf = MM::from_tiles(td);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          dvt[dt][kt] = MM::mma(ado[dt], pf, dvt[dt][kt]);
          dkt[dt][kt] = MM::mma(aq[dt], dsf, dkt[dt][kt]);
        }
      }
    }
    // ---- dS -> LDS [q][key]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *(T *)(sdS + (qt * 16 + 4 * lg + r) * SSTR + (kw0 + kt * 16 + li) * ES) = MM::from_float(dst[qt][kt][r]);
    __syncthreads();
    // ---- dQ^T[d][q] = K^T dS^T : pieces (qt, dt) dealt to the waves
    for (int piece = wave; piece < 2 * NDT; piece += 4) {
      const int qt = piece / NDT, dt = piece - qt * NDT;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < nkc; ++c) {
        const frag a = *(const frag *)(sKt + (dt * 16 + li) * SSTR + (c * KC + EPL * lg) * ES);
        const frag bq = *(const frag *)(sdS + (qt * 16 + li) * SSTR + (c * KC + EPL * lg) * ES);
        acc = MM::mma(a, bq, acc);                   // D[d = 4*lg + r][q = li]
      }
      const int qr = qs + qt * 16 + li;
      if (qr < N) {
        if (dqw) *(f32x4 *)(dqw + (int64_t)qr * DH + dt * 16 + 4 * lg) = acc;
        else Vec4<T>::store(dqbase + (int64_t)qr * ld + dt * 16 + 4 * lg, acc);
      }
    }
  }

  // ---- dK, dV rows of this wave's keys
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int key = kb0 + kw0 + kt * 16 + li;
    if (key < N) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        Vec4<T>::store(dqbase + C + (int64_t)key * ld + dt * 16 + 4 * lg, dkt[dt][kt]);
        Vec4<T>::store(dqbase + 2 * C + (int64_t)key * ld + dt * 16 + 4 * lg, dvt[dt][kt]);
      }
    }
  }
  }
}

// dq[b, n, h, :] = sum over key blocks of the fp32 slabs (key-block order), written into the q slot of dqkv
template <typename T>
__global__ void attention_dq_reduce_kernel(const float *__restrict__ ws, int nkb, int B, int N, int heads, int DH,
                                           T *__restrict__ dqkv) {
  const int64_t per = (int64_t)B * heads * N * DH;
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= per) return;
  f32x4 s = *(const f32x4 *)(ws + i4);
  for (int kb = 1; kb < nkb; ++kb) s += *(const f32x4 *)(ws + kb * per + i4);
  const int d = (int)(i4 % DH);
  const int64_t row = i4 / DH;                  // (b*heads + h)*N + n
  const int n = (int)(row % N);
  const int64_t bh = row / N;
  const int h = (int)(bh % heads);
  const int64_t b = bh / heads;
  Vec4<T>::store(dqkv + ((b * N + n) * 3) * (int64_t)(heads * DH) + h * DH + d, s);
}

template <typename T, int DH>
static size_t attn_bwd_lds() {
  constexpr int ES = (int)sizeof(T);
  return (size_t)DH * (AB_KEYS * ES + 16) + (size_t)AB_QB * (AB_KEYS * ES + 16) + 2 * (size_t)AB_QB * (DH * ES + 16) +
         2 * (size_t)DH * (AB_QB * ES + 16) + 2 * AB_QB * sizeof(float);
}

// short-sequence fp16 variants (attention_res.hip)
int launch_attention_fwd_res(int dtype, const void *qkv, int B, int N, int heads, int dh, void *o, float *lse, float scale,
                             hipStream_t s);
int launch_attention_fwd_stream(int dtype, const void *qkv, int B, int N, int heads, int dh, void *o, float *lse, float scale,
                                hipStream_t s);
int launch_attention_bwd_res(int dtype, const void *qkv, const void *o, const void *d_o, const float *lse, int B, int N, int heads,
                             int dh, void *dqkv, float scale, hipStream_t s);
int launch_attention_bwd_stream(int dtype, const void *qkv, const void *o, const void *d_o, const float *lse, int B, int N, int heads,
                                int dh, void *dqkv, float *dq_ws, float scale, hipStream_t s);
constexpr int ATTN_RES_MAXN = 256;

}  // namespace m3

using namespace m3;

extern "C" int m3_attention_fwd(const void *qkv, int dtype, int B, int N, int heads, int dh, void *o, float *lse,
                                void *stream) {
  M3_REQUIRE(qkv && o && lse, "m3_attention_fwd: null operand");
  M3_REQUIRE(dtype_ok(dtype), "m3_attention_fwd: bad dtype");
  M3_REQUIRE(dh == 32 || dh == 64, "m3_attention_fwd: head dim %d not in {32, 64}", dh);
  M3_REQUIRE(B > 0 && N > 0 && heads > 0, "m3_attention_fwd: bad shape");
  M3_REQUIRE(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)o % 16) == 0, "m3_attention_fwd: alignment");
  const dim3 grid((N + AT_QB - 1) / AT_QB, B * heads), block(AT_THREADS);
  const float scale = 1.0f / sqrtf((float)dh);
  hipStream_t s = (hipStream_t)stream;
  const bool b16 = dtype == M3_F16 || dtype == M3_BF16;
  if (b16 && N <= ATTN_RES_MAXN) return launch_attention_fwd_res(dtype, qkv, B, N, heads, dh, o, lse, scale, s);
  if (b16) return launch_attention_fwd_stream(dtype, qkv, B, N, heads, dh, o, lse, scale, s);
#define M3_AF(TT, DD) \
  hipLaunchKernelGGL((attention_fwd_kernel<TT, DD>), grid, block, 0, s, (const TT *)qkv, B, N, heads, (TT *)o, lse, scale)
  if (dtype == M3_F16) { if (dh == 32) M3_AF(half_t, 32); else M3_AF(half_t, 64); }
  else { if (dh == 32) M3_AF(float, 32); else M3_AF(float, 64); }
#undef M3_AF
  return check_launch("m3_attention_fwd");
}

template <typename T, int DH>
static int launch_attn_bwd(const void *qkv, const void *o, const void *d_o, const float *lse, int B, int N, int heads,
                           void *dqkv, float *dq_ws, float scale, hipStream_t s) {
  const size_t lds = attn_bwd_lds<T, DH>();
  static bool attr_set = false;
  if (!attr_set) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void *)attention_bwd_kernel<T, DH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int nkb = (N + AB_KEYS - 1) / AB_KEYS;
  hipLaunchKernelGGL((attention_bwd_kernel<T, DH>), dim3(B * heads * nkb), dim3(AT_THREADS), lds, s, (const T *)qkv,
                     (const T *)o, (const T *)d_o, lse, B, N, heads, (T *)dqkv, dq_ws, scale);
  int rc = check_launch("m3_attention_bwd");
  if (rc || nkb == 1) return rc;
  const int64_t n4 = (int64_t)B * heads * N * DH / 4;
  hipLaunchKernelGGL((attention_dq_reduce_kernel<T>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, dq_ws, nkb, B, N,
                     heads, DH, (T *)dqkv);
  return check_launch("m3_attention_bwd(dq reduce)");
}

extern "C" int64_t m3_attention_bwd_ws_elems(int B, int N, int heads, int dh) {
  // dQ slabs [key block][B*heads][N][dh] + delta [B*N*heads] (fp16 long-sequence kernel)
  return N > AB_KEYS ? (int64_t)((N + AB_KEYS - 1) / AB_KEYS) * B * heads * N * dh + (int64_t)B * N * heads : 0;
}

extern "C" int m3_attention_bwd(const void *qkv, const void *o, const void *d_o, const float *lse, int dtype, int B,
                                int N, int heads, int dh, void *dqkv, float *dq_ws, void *stream) {
  M3_REQUIRE(qkv && o && d_o && lse && dqkv, "m3_attention_bwd: null operand");
  M3_REQUIRE(dtype_ok(dtype), "m3_attention_bwd: bad dtype");
  M3_REQUIRE(dh == 32 || dh == 64, "m3_attention_bwd: head dim %d not in {32, 64}", dh);
  M3_REQUIRE(N > 0, "m3_attention_bwd: bad N");
  M3_REQUIRE(N <= AB_KEYS || dq_ws, "m3_attention_bwd: N=%d > %d needs the fp32 dQ workspace (m3_attention_bwd_ws_elems)", N, AB_KEYS);
  if (N <= AB_KEYS) dq_ws = nullptr;
  const float scale = 1.0f / sqrtf((float)dh);
  hipStream_t s = (hipStream_t)stream;
  const bool b16 = dtype == M3_F16 || dtype == M3_BF16;
  if (b16 && N <= ATTN_RES_MAXN)
    return launch_attention_bwd_res(dtype, qkv, o, d_o, lse, B, N, heads, dh, dqkv, scale, s);
  if (b16) {
    int rc = launch_attention_bwd_stream(dtype, qkv, o, d_o, lse, B, N, heads, dh, dqkv, dq_ws, scale, s);
    if (rc) return rc;
    const int nkb = (N + AB_KEYS - 1) / AB_KEYS;
    const int64_t n4 = (int64_t)B * heads * N * dh / 4;
    if (dtype == M3_BF16)
      hipLaunchKernelGGL((attention_dq_reduce_kernel<bf16_t>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, dq_ws, nkb, B,
                         N, heads, dh, (bf16_t *)dqkv);
    else
      hipLaunchKernelGGL((attention_dq_reduce_kernel<half_t>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, dq_ws, nkb, B,
                         N, heads, dh, (half_t *)dqkv);
    return check_launch("m3_attention_bwd(dq reduce)");
  }
  if (dtype == M3_F16) {
    if (dh == 32) return launch_attn_bwd<half_t, 32>(qkv, o, d_o, lse, B, N, heads, dqkv, dq_ws, scale, s);
    return launch_attn_bwd<half_t, 64>(qkv, o, d_o, lse, B, N, heads, dqkv, dq_ws, scale, s);
  }
  if (dh == 32) return launch_attn_bwd<float, 32>(qkv, o, d_o, lse, B, N, heads, dqkv, dq_ws, scale, s);
  return launch_attn_bwd<float, 64>(qkv, o, d_o, lse, B, N, heads, dqkv, dq_ws, scale, s);
}
