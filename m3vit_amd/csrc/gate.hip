// Fused task-aware router for gfx950: logits + noise + softmax + top-(k+1) + gates +
// load-balance partials in one pass over the token rows, and its backward.
//
// Replaces the 5-6 ATen kernels of NoisyGate_VMoE.forward
// (models/moe/ckpt/noisy_gate_vmoe.py:91-93,168,197-207), the multi-gate select
// (custom_moe_layer.py:213-217: the caller just passes that task's w_gate) and the
// task-conditioning cat (custom_moe_layer.py:176-179: folded into logit_bias), plus the
// importance / load summaries and the cv^2 balance loss of
// vision_transformer_moe.py:23-87,453-459,540.
//
// HBM-bound by design: each token row is read once (coalesced 16-byte loads, staged
// through LDS so that every LANE then owns one TOKEN).  A workgroup is 64 tokens x up to
// 4 waves; wave w accumulates the logits of experts [w*EW, (w+1)*EW) for all 64 tokens
// (the w_gate rows of a step are staged in LDS too and read as wave-wide broadcasts), then
// wave 0 gathers the E logits of each token into that lane's registers, so softmax and the
// top-(k+1) selection need no cross-lane traffic at all.
//
// Arithmetic order is pinned (and mirrored by oracle/gate_route.c) so that expert
// indices are bit-exact: sequential fmaf chain over d starting from the bias, selection
// on the noisy logits with ties -> lowest index.
#include "common.h"

#pragma clang fp contract(off)

namespace m3 {

constexpr int GATE_TOK = 64;       // tokens per workgroup
constexpr int GATE_DW_TOK = 64;    // tokens per workgroup in the dW kernels
constexpr int GATE_ROWB = 128;     // bytes of a token row staged per step (one cache line)

__device__ __forceinline__ float normal_cdf(float z) { return 0.5f * erfcf(-z * 0.70710678118654752f); }
__device__ __forceinline__ float normal_pdf(float z) { return 0.3989422804014327f * expf(-0.5f * z * z); }

struct GateFwdDev {
  const char *x; int64_t T; int D; int64_t ldx_b;
  const float *w; int E;
  const float *bias; const float *noise; float noise_std; int k;
  int64_t *idx; int32_t *idx32; int32_t *idx_next; float *score; float *top_logits;
  float *clean; float *noisy; float *gates;
  float *part_imp; int32_t *part_load; float *part_load_prob;
  int32_t *part_count;      // optional [nblk, E]: tokens of the block whose top-k holds expert e (the routing histogram)
};

// EW = experts per wave, NW = EPAD / EW waves per workgroup
// NS > 0: the row is exactly NS steps long and is fetched whole up front; NS = 0: any D, one step in flight
template <typename T, int EPAD, int EW, bool EXACT, int NS>
__global__ __launch_bounds__(GATE_TOK *(EPAD / EW)) void gate_fwd_kernel(const GateFwdDev p) {
  constexpr int NW = EPAD / EW;
  constexpr int NT = GATE_TOK * NW;
  constexpr int DC = GATE_ROWB / (int)sizeof(T);   // d per step
  constexpr int LDS_STRIDE = DC + 4;               // floats; 16-byte aligned rows, 17 * 16 B apart: eight consecutive lanes' 16-byte
                                                   // accesses cover all 32 banks (conflict-free ds_read_b128 / ds_write_b128)
  constexpr int CPR = GATE_ROWB / 16;              // 16-byte chunks per row per step
  constexpr int EPC = 16 / (int)sizeof(T);         // elements per chunk
  constexpr int NCH = GATE_TOK * CPR / NT;         // chunks per thread per step
  constexpr int NWE = (DC * EPAD + NT - 1) / NT;   // w_gate slice elements per thread per step
  // w_gate rows by SCALAR loads (round 5, exact E only): every lane of a wave multiplies its token by the same EW weights,
  // so the row slice w[d][e0 .. e0+EW) belongs in SGPRs (s_load_dwordx{4,8,16}, v_pk_fma with an SGPR-pair operand), not in
  // LDS: read from LDS as wave-wide broadcasts it cost one 16-byte ds_read per two packed fmas, in four waves at once - the
  // LDS pipe, not the fma chain, set the time (220 us at E = 64, D = 768).  The fma chain per logit is unchanged (over d in
  // order), so the results are bit for bit the staged form's.
  // Only where a wave owns 16 experts (E = 64): with 4 experts per wave the scalar loads are too small to pay (E = 16,
  // D = 384: 28 us staged, 38 us scalar - the scalar cache misses per 64-byte line and nothing hides it).
  constexpr bool SW = EXACT && EW >= 16;
  // LDS: [sx: the step's token rows as fp32, double-buffered | sw: the step's w_gate rows (staged form only)] during the
  // main loop; afterwards the noisy logits and gates of the epilogue lie over sx (every wave is past its last read of sx at
  // the barrier behind the clean logits), the clean logits have a region of their own (written while other waves may
  // still be in the main loop).  52 KiB at E = 64 (three workgroups per CU) where separate arrays took 132 KiB (one).
  constexpr int SXF = 2 * GATE_TOK * LDS_STRIDE;                 // floats
  constexpr int SWF = SW ? 0 : 2 * DC * EPAD;
  constexpr int SDF = GATE_TOK * (EPAD + 1);                     // one dense [token][expert] array, +1: lane = token reads
  constexpr int OVER = (2 * SDF > SXF + SWF) ? 2 * SDF : SXF + SWF;
  __shared__ __attribute__((aligned(16))) float smem[OVER + SDF];
  float (*sx)[GATE_TOK * LDS_STRIDE] = reinterpret_cast<float (*)[GATE_TOK * LDS_STRIDE]>(smem);
  float (*sw)[DC * EPAD] = reinterpret_cast<float (*)[DC * EPAD]>(smem + SXF);
  // epilogue staging: the token-per-lane epilogue of wave 0 leaves the dense rows (clean, noisy, gates), the token's
  // top-k mask and its two CDF thresholds here; ALL waves then store the rows coalesced and share the per-expert partial
  // sums.  (Written from registers by lane = token, a dense [T, E] output was 64 four-byte stores to 64 different lines
  // per instruction, E times per array, and the E butterfly sums ran in one wave.)
  float *const sdense[3] = {smem + OVER, smem, smem + SDF};      // [0] clean (also the waves' hand-over to wave 0)
  float *const slog = sdense[0];
  __shared__ unsigned long long ssel[GATE_TOK];
  __shared__ float sthr[2][GATE_TOK];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int e0 = wave * EW;
  const int E = EXACT ? EPAD : p.E;
  const int64_t T_ = p.T;
  const int64_t t0 = (int64_t)blockIdx.x * GATE_TOK;
  const int D = p.D;
  const int dbytes = D * (int)sizeof(T);
  const int nsteps = (dbytes + GATE_ROWB - 1) / GATE_ROWB;

  float acc[EW];
#pragma unroll
  for (int e = 0; e < EW; ++e) acc[e] = (p.bias && (EXACT || e0 + e < E)) ? p.bias[e0 + e] : 0.f;

  // staging: chunk q = tid + NT*i -> row q / CPR, c = q % CPR; rows past T and bytes past D read as zero
  auto fetch_w = [&](int step, float (&prew)[NWE]) {
    if constexpr (SW) return;
#pragma unroll
    for (int i = 0; i < NWE; ++i) {
      const int q = tid + NT * i;                  // element (dd, e) of the slice
      const int dd = q / EPAD, e = q - dd * EPAD;
      const int d = step * DC + dd;
      prew[i] = (q < DC * EPAD && d < D && e < E) ? p.w[(int64_t)d * E + e] : 0.f;
    }
  };
  auto fetch_x = [&](int step, u32x4 (&pre)[NCH]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int q = tid + NT * i;
      const int row = q / CPR, c = q % CPR;
      const int64_t tr = t0 + row;
      const int cb = step * GATE_ROWB + c * 16;
      pre[i] = (tr < T_ && cb < dbytes) ? *(const u32x4 *)(p.x + tr * p.ldx_b + cb) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto stash = [&](int buf, const u32x4 (&pre)[NCH], const float (&prew)[NWE]) {
    if constexpr (!SW) {
#pragma unroll
      for (int i = 0; i < NWE; ++i) {
        const int q = tid + NT * i;
        if (q < DC * EPAD) sw[buf][q] = prew[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int q = tid + NT * i;
      const int row = q / CPR, c = q % CPR;
      float *dst = &sx[buf][row * LDS_STRIDE + c * EPC];
      if constexpr (sizeof(T) == 2) {
        typedef T t16x8 __attribute__((ext_vector_type(8)));          // fp16 or bf16 rows
        const t16x8 h = __builtin_bit_cast(t16x8, pre[i]);
        *(f32x4 *)dst = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        *(f32x4 *)(dst + 4) = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
      } else {
        *(f32x4 *)dst = __builtin_bit_cast(f32x4, pre[i]);
      }
    }
  };
  // the step's w_gate rows come from LDS (all lanes read the same address: broadcast), x from the
  // lane's own LDS row; per logit the fma chain runs over d in order, as the oracle's
  auto compute = [&](int buf, int d0, int dn) {
    const float *xs = &sx[buf][lane * LDS_STRIDE];
    // SW: the rows of the step straight from w_gate, wave-uniform addresses (scalar loads); else the staged copy
    const float *ws = SW ? p.w + (int64_t)d0 * EPAD + e0 : &sw[buf][e0];
    // four d per 16-byte read of the lane's row (the fma chain per logit still runs over d in order)
    int dd = 0;
#pragma unroll 4
    for (; dd + 4 <= dn; dd += 4) {
      const f32x4 xv = *(const f32x4 *)(xs + dd);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < EW; ++e) acc[e] = __builtin_fmaf(xv[j], ws[(dd + j) * EPAD + e], acc[e]);
    }
    for (; dd < dn; ++dd) {
      const float xv = xs[dd];
#pragma unroll
      for (int e = 0; e < EW; ++e) acc[e] = __builtin_fmaf(xv, ws[dd * EPAD + e], acc[e]);
    }
  };

  if constexpr (NS > 0) {
    // rows of exactly NS steps (host-checked): every load of the workgroup's rows is issued before the first use, so
    // the HBM latency is paid once and not once per step (the ring below pays it nsteps times: with about 1.5
    // workgroups per CU at T = 25k there is nothing else resident to hide it behind)
    u32x4 px[NS][NCH];
    float pw[NS][NWE];
#pragma unroll
    for (int st = 0; st < NS; ++st) fetch_x(st, px[st]);
#pragma unroll
    for (int st = 0; st < NS; ++st) fetch_w(st, pw[st]);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      stash(st & 1, px[st], pw[st]);
      __syncthreads();          // one barrier per step: the other buffer is only rewritten after the next one
      compute(st & 1, st * DC, DC);
    }
  } else {
    u32x4 pre[NCH];
    float prew[NWE];
    fetch_w(0, prew);
    fetch_x(0, pre);
    for (int step = 0; step < nsteps; ++step) {
      const int buf = step & 1;
      stash(buf, pre, prew);
      __syncthreads();            // one barrier per step: the other buffer is only rewritten after the next one
      if (step + 1 < nsteps) { fetch_w(step + 1, prew); fetch_x(step + 1, pre); }
      const int d0 = step * DC;
      compute(buf, d0, (D - d0 < DC) ? (D - d0) : DC);
    }
  }

  // gather the E clean logits of token `lane` into wave 0
  float cl[EPAD];
  if constexpr (NW > 1) {
#pragma unroll
    for (int e = 0; e < EW; ++e) slog[lane * (EPAD + 1) + e0 + e] = acc[e];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int e = 0; e < EPAD; ++e) cl[e] = slog[lane * (EPAD + 1) + e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < EPAD; ++e) cl[e] = acc[e];
  }

  const int64_t t = t0 + lane;
  const bool tok_ok = t < T_;
  const int k = p.k;
  const bool prob_load = p.part_load_prob != nullptr;   // host: only when noise_std != 0 and k < E
  if (wave == 0) {
  // noisy logits
  float nz[EPAD];
#pragma unroll
  for (int e = 0; e < EPAD; ++e) {
    float n = cl[e];
    if (p.noise && p.noise_std != 0.f && (EXACT || e < E) && tok_ok) {
      const float scaled = p.noise[t * E + e] * p.noise_std;
      n = cl[e] + scaled;
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
  float thr_in = 0.f, thr_out = 0.f;       // top_logits[t][k], top_logits[t][k-1]
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
    const float pr = bq / s;
    if (j == k) thr_in = pr;
    if (j == k - 1) thr_out = pr;
    if (tok_ok) {
      p.top_logits[t * kp + j] = pr;
      if (j < k) {
        p.idx[t * k + j] = best;
        if (p.idx32) p.idx32[t * k + j] = best;
        p.score[t * k + j] = pr;
      } else if (p.idx_next) {
        p.idx_next[t] = best;
      }
    }
    if (j < k) sel_k |= 1ull << best;
  }
  // dense values, top-k mask and thresholds of this token -> LDS
#pragma unroll
  for (int e = 0; e < EPAD; ++e) {
    const bool sel = (sel_k >> e) & 1ull;
    const float pr = q[e] / s;
    if constexpr (NW == 1) sdense[0][lane * (EPAD + 1) + e] = cl[e];       // (NW > 1: the waves' hand-over already lies there)
    sdense[1][lane * (EPAD + 1) + e] = nz[e];
    sdense[2][lane * (EPAD + 1) + e] = ((EXACT || e < E) && sel && tok_ok) ? pr : 0.f;
  }
  ssel[lane] = tok_ok ? sel_k : 0ull;
  sthr[0][lane] = thr_in;
  sthr[1][lane] = thr_out;
  }   // wave 0
  __syncthreads();

  // ---- all waves: the block's dense rows, coalesced (the 64 tokens' rows are one contiguous run of E floats each)
  {
    const int nvalid = (int)((T_ - t0 < GATE_TOK) ? (T_ - t0) : GATE_TOK);
    const int nel = nvalid * E;
    float *const outs[3] = {p.clean, p.noisy, p.gates};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (!outs[a]) continue;
      float *dst = outs[a] + t0 * E;
      for (int i = tid; i < nel; i += NT) {
        const int row = EXACT ? i / EPAD : i / E, e = i - row * E;
        dst[i] = sdense[a][row * (EPAD + 1) + e];
      }
    }
  }
  // ---- all waves: load-balance partials, expert e on wave e % NW (lane = token; the same butterfly sums as before)
  const float inv_std = prob_load ? 1.f / p.noise_std : 0.f;
  const unsigned long long my_sel = ssel[lane];
  const float thr_in = sthr[0][lane], thr_out = sthr[1][lane];
  for (int e = wave; e < E; e += NW) {
    const float gv = sdense[2][lane * (EPAD + 1) + e];
    const float wsum = wave_sum(gv);
    const unsigned long long b = __ballot(gv > 0.f);
    if (p.part_count) {
      // routed entries per expert of this 64-token block: what m3_route_build's histogram pass would count from idx
      // (NOT the load: a selected expert whose probability underflowed to 0 is routed all the same)
      const unsigned long long bs = __ballot((my_sel >> e) & 1ull);
      if (lane == 0) p.part_count[(int64_t)blockIdx.x * E + e] = __popcll(bs);
    }
    float psum = 0.f;
    if (prob_load) {
      // _prob_in_top_k, vision_transformer_moe.py:33-71: thresholds are PROBABILITIES, clean/noisy are logits
      const float cle = sdense[0][lane * (EPAD + 1) + e], nze = sdense[1][lane * (EPAD + 1) + e];
      const bool is_in = nze > thr_in;
      const float z = (cle - (is_in ? thr_in : thr_out)) * inv_std;
      psum = wave_sum(tok_ok ? normal_cdf(z) : 0.f);
    }
    if (lane == 0) {
      p.part_imp[(int64_t)blockIdx.x * E + e] = wsum;
      p.part_load[(int64_t)blockIdx.x * E + e] = __popcll(b);
      if (prob_load) p.part_load_prob[(int64_t)blockIdx.x * E + e] = psum;
    }
  }
}

// Reduce the per-block balance partials in fixed order and evaluate
//   cv_loss = cv^2(importance) + cv^2(load)     (vision_transformer_moe.py:73-87,540)
// and its gradient.  One workgroup of 64*E.. threads: thread (r, e) sums rows r, r+RL, ... of column e.
struct BalanceDev {
  const float *part_imp; const int32_t *part_load; const float *part_load_prob;
  int nblk; int E;
  float *importance; int64_t *load; float *load_prob;
  float *loss_acc; float *loss_out;
  float *d_importance; float *d_load_prob;
  // routing scan riding along as workgroup 1 (m3_balance_route): per-block exclusive prefix of part_count along the block
  // axis + the expert totals / offsets / 128-row tile prefix - what m3_route_build's second launch computes
  const int32_t *part_count; int32_t *blk_base; int32_t *counts; int32_t *offsets; int32_t *tile_starts; int64_t *counts64;
};

// one wave per expert (64 blocks per step, shuffle scan), then offsets / tile prefix by one thread
__device__ __forceinline__ void route_scan_blocks(const BalanceDev &p) {
  __shared__ int32_t tot[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, E = p.E;
  for (int e = wave; e < E; e += 1024 / 64) {
    int32_t carry = 0;
    for (int b0 = 0; b0 < p.nblk; b0 += 64) {
      const int b = b0 + lane;
      const int32_t v = b < p.nblk ? p.part_count[(int64_t)b * E + e] : 0;
      int32_t inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int32_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
      }
      if (b < p.nblk) p.blk_base[(int64_t)b * E + e] = carry + inc - v;
      carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) {
      tot[e] = carry;
      p.counts[e] = carry;
      if (p.counts64) p.counts64[e] = carry;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t o = 0, ts = 0;
    for (int i = 0; i < E; ++i) {
      p.offsets[i] = o;
      p.tile_starts[i] = ts;
      o += tot[i];
      ts += (tot[i] + 127) / 128;
    }
    p.offsets[E] = o;
    p.tile_starts[E] = ts;
  }
}

__device__ __forceinline__ void cv2_and_grad(const float *v, int E, float &cv, float *grad_scratch) {
  // unbiased var / (mean^2 + 1e-10); zero for a single expert
  float mean = 0.f;
  for (int e = 0; e < E; ++e) mean += v[e];
  mean /= (float)E;
  float var = 0.f;
  for (int e = 0; e < E; ++e) var += (v[e] - mean) * (v[e] - mean);
  var /= (float)(E - 1);
  const float den = mean * mean + 1e-10f;
  cv = var / den;
  for (int e = 0; e < E; ++e)
    grad_scratch[e] = 2.f * (v[e] - mean) / ((float)(E - 1) * den) - var * (2.f * mean / (float)E) / (den * den);
}

__global__ __launch_bounds__(1024) void balance_kernel(const BalanceDev p) {
  if (blockIdx.x == 1) { route_scan_blocks(p); return; }
  // thread (r, e): e = tid % E, r = tid / E sums rows r, r + RL, ... (4 independent loads in flight),
  // then the RL row-lanes are added in lane order by thread (0, e)
  __shared__ float s_imp[1024], s_prob[1024];
  __shared__ int s_load[1024];
  __shared__ float v_imp[64], v_load[64], g_imp[64], g_load[64];
  const int E = p.E;
  const int RL = 1024 / E;
  const int e = threadIdx.x % E, r = threadIdx.x / E;
  float a = 0.f, c = 0.f;
  int l = 0;
  if (r < RL) {
    int b = r;
    for (; b + 3 * RL < p.nblk; b += 4 * RL) {
      const int64_t i0 = (int64_t)b * E + e, i1 = i0 + (int64_t)RL * E, i2 = i1 + (int64_t)RL * E, i3 = i2 + (int64_t)RL * E;
      const float a0 = p.part_imp[i0], a1 = p.part_imp[i1], a2 = p.part_imp[i2], a3 = p.part_imp[i3];
      const int l0 = p.part_load[i0], l1 = p.part_load[i1], l2 = p.part_load[i2], l3 = p.part_load[i3];
      a += a0; a += a1; a += a2; a += a3;
      l += l0 + l1 + l2 + l3;
      if (p.part_load_prob) {
        const float c0 = p.part_load_prob[i0], c1 = p.part_load_prob[i1], c2 = p.part_load_prob[i2], c3 = p.part_load_prob[i3];
        c += c0; c += c1; c += c2; c += c3;
      }
    }
    for (; b < p.nblk; b += RL) {
      a += p.part_imp[(int64_t)b * E + e];
      l += p.part_load[(int64_t)b * E + e];
      if (p.part_load_prob) c += p.part_load_prob[(int64_t)b * E + e];
    }
  }
  s_imp[threadIdx.x] = a; s_load[threadIdx.x] = l; s_prob[threadIdx.x] = c;
  __syncthreads();
  // tree over the row-lanes (fixed shape -> deterministic)
  int top = 1;
  while (top < RL) top <<= 1;
  for (int stride = top >> 1; stride >= 1; stride >>= 1) {
    if (r < stride && r + stride < RL) {
      const int i = r * E + e, j = (r + stride) * E + e;
      s_imp[i] += s_imp[j]; s_load[i] += s_load[j]; s_prob[i] += s_prob[j];
    }
    __syncthreads();
  }
  if (r == 0) {
    const float imp = s_imp[e], pr = s_prob[e];
    const int ld = s_load[e];
    p.importance[e] = imp;
    p.load[e] = ld;
    if (p.load_prob) p.load_prob[e] = pr;
    v_imp[e] = imp;
    v_load[e] = p.part_load_prob ? pr : (float)ld;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float cv_i = 0.f, cv_l = 0.f;
    if (E > 1) {
      cv2_and_grad(v_imp, E, cv_i, g_imp);
      cv2_and_grad(v_load, E, cv_l, g_load);
    } else {
      g_imp[0] = 0.f; g_load[0] = 0.f;
    }
    const float loss = cv_i + cv_l;
    if (p.loss_out) *p.loss_out = loss;
    if (p.loss_acc) *p.loss_acc += loss;
  }
  __syncthreads();
  if (r == 0) {
    if (p.d_importance) p.d_importance[e] = g_imp[e];
    if (p.d_load_prob) p.d_load_prob[e] = p.part_load_prob ? g_load[e] : 0.f;
  }
}

// d_logits through scatter + softmax (+ the Normal-CDF load term): thread per token
struct GateBwdDev {
  const float *noisy; const float *clean; const float *top_logits;
  const int64_t *idx; const int32_t *idx_next;
  const float *d_score; const float *d_top; const float *d_importance; const float *d_load_prob;
  float balance_scale; float noise_std;
  int64_t T; int E; int k;
  float *d_logits;
  const float *balance_scale_dev;
  void *d_logits_act; int act_dtype;     // optional second copy in the activation dtype (the operand of the d w_gate GEMM)
};

template <int EPAD>
__global__ __launch_bounds__(256) void gate_bwd_logits_kernel(const GateBwdDev p) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.T) return;
  const int E = p.E, k = p.k;
  const int kp = (k + 1 < E) ? k + 1 : E;
  const float *nzp = p.noisy + t * E;
  // the loss weight: a launch constant, times an optional device-resident factor (the upstream gradient of the loss when the
  // caller is an autograd graph replayed from a hipGraph: its value is not known on the host at launch time)
  const float bscale = p.balance_scale_dev ? p.balance_scale * p.balance_scale_dev[0] : p.balance_scale;
  float pr[EPAD], dp[EPAD], g[EPAD];
  float m = nzp[0];
#pragma unroll
  for (int e = 0; e < EPAD; ++e) {
    pr[e] = e < E ? nzp[e] : 0.f;
    dp[e] = 0.f; g[e] = 0.f;
    if (e < E && e > 0) m = pr[e] > m ? pr[e] : m;
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < EPAD; ++e) {
    const float nz = pr[e];
    pr[e] = e < E ? expf(nz - m) : 0.f;
    s = s + pr[e];
    // keep the noisy logit for the is_in test of the load term
    g[e] = nz;
  }
  const float inv_s = 1.f / s;
#pragma unroll
  for (int e = 0; e < EPAD; ++e) pr[e] *= inv_s;

  // Normal-CDF load term (vision_transformer_moe.py:33-71,456-457): d load_prob[e] -> clean logits and,
  // through the two probability thresholds, the softmax
  float d_thr_in = 0.f, d_thr_out = 0.f;
  const bool prob_load = p.d_load_prob != nullptr;
  if (prob_load) {
    const float thr_in = p.top_logits[t * kp + k], thr_out = p.top_logits[t * kp + k - 1];
    const float inv_std = 1.f / p.noise_std;
    const float *clp = p.clean + t * E;
#pragma unroll
    for (int e = 0; e < EPAD; ++e) {
      if (e < E) {
        const bool is_in = g[e] > thr_in;
        const float z = (clp[e] - (is_in ? thr_in : thr_out)) * inv_std;
        const float ge = p.d_load_prob[e] * bscale * normal_pdf(z) * inv_std;
        g[e] = ge;
        if (is_in) d_thr_in -= ge; else d_thr_out -= ge;
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < EPAD; ++e) g[e] = 0.f;
  }
  // d p at the selected experts
  for (int j = 0; j < k; ++j) {
    const int ej = (int)p.idx[t * k + j];
    float v = (p.d_score ? p.d_score[t * k + j] : 0.f) + (p.d_top ? p.d_top[t * kp + j] : 0.f) +
              (p.d_importance ? p.d_importance[ej] * bscale : 0.f);
    if (j == k - 1) v += d_thr_out;
#pragma unroll
    for (int e = 0; e < EPAD; ++e)
      if (e == ej) dp[e] += v;
  }
  if (kp > k && p.idx_next) {
    const int en = p.idx_next[t];
    const float v = (p.d_top ? p.d_top[t * kp + k] : 0.f) + d_thr_in;
#pragma unroll
    for (int e = 0; e < EPAD; ++e)
      if (e == en) dp[e] += v;
  }
  float dot = 0.f;
#pragma unroll
  for (int e = 0; e < EPAD; ++e) dot += dp[e] * pr[e];
#pragma unroll
  for (int e = 0; e < EPAD; ++e) {
    if (e < E) {
      const float v = pr[e] * (dp[e] - dot) + g[e];
      p.d_logits[t * E + e] = v;
      if (p.d_logits_act) {
        if (p.act_dtype == M3_F16) ((half_t *)p.d_logits_act)[t * E + e] = (half_t)v;
        else if (p.act_dtype == M3_BF16) ((bf16_t *)p.d_logits_act)[t * E + e] = (bf16_t)v;
        else ((float *)p.d_logits_act)[t * E + e] = v;
      }
    }
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

// dW partials for E <= 16 (round 3): a thread owns FOUR adjacent d (one 8-byte fp16 load per token) x all E experts = 4*E
// accumulators, RP (4, or 2 for D > 512) threads share a column group and walk the workgroup's tokens RP apart; the tokens' d_logits rows sit
// in LDS (one 16-byte read per 4 experts).  64 FMAs per 8 bytes loaded, eight loads in flight per thread; the row-parity
// partials meet in LDS.  The kernel above (one d per thread, a 2-byte load and E scalar loads per token, no unrolling) ran the
// 0.3 GFLOP product at the speed of its load chain; the TN MFMA GEMM that replaced it in round 1 pads E = 16 to a 128-wide tile.
template <typename T, int EPAD>
__global__ __launch_bounds__(512) void gate_bwd_dw4_kernel(const char *__restrict__ x, int64_t T_, int D, int64_t ldx_b,
                                                            const float *__restrict__ dl, int E, float *part_dw) {
  extern __shared__ __attribute__((aligned(16))) float sm[];        // [GATE_DW_TOK][EPAD] d_logits, then [RP][D/4][4*EPAD+1] partials
  float *sdl = sm, *spart = sm + GATE_DW_TOK * EPAD;
  const int CG = D / 4;
  const int RP = blockDim.x / CG;                                    // 4 or 2
  const int tid = threadIdx.x, rp = tid / CG, cg = tid - rp * CG;
  const int64_t t0 = (int64_t)blockIdx.x * GATE_DW_TOK;
  const int ntok = (int)((T_ - t0 < GATE_DW_TOK) ? T_ - t0 : GATE_DW_TOK);
  for (int q = tid; q < GATE_DW_TOK * EPAD; q += blockDim.x) {
    const int t = q / EPAD, e = q - t * EPAD;
    sdl[q] = (t < ntok && e < E) ? dl[(t0 + t) * E + e] : 0.f;       // rows past the end: zeros (their x rows are clamped)
  }
  __syncthreads();
  float acc[4][EPAD];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < EPAD; ++e) acc[c][e] = 0.f;
  const char *xc = x + (int64_t)cg * 4 * sizeof(T);
  constexpr int UN = 8;                                              // tokens per thread: GATE_DW_TOK / RP, in batches of UN
  static_assert((GATE_DW_TOK / 4) % UN == 0, "token batches");
  const int per = GATE_DW_TOK / RP;
  for (int i0 = 0; i0 < per; i0 += UN) {
    typename Vec4<T>::type raw[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      int t = rp + RP * (i0 + u);
      if (t >= ntok) t = ntok - 1;                                   // clamped: multiplied by the zero d_logits row
      raw[u] = *(const typename Vec4<T>::type *)(xc + (t0 + t) * ldx_b);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int t = rp + RP * (i0 + u);
      const float xv[4] = {(float)raw[u][0], (float)raw[u][1], (float)raw[u][2], (float)raw[u][3]};
#pragma unroll
      for (int e4 = 0; e4 < EPAD; e4 += 4) {
        const f32x4 d4 = *(const f32x4 *)(sdl + t * EPAD + e4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[c][e4 + 0] = __builtin_fmaf(xv[c], d4[0], acc[c][e4 + 0]); acc[c][e4 + 1] = __builtin_fmaf(xv[c], d4[1], acc[c][e4 + 1]);
          acc[c][e4 + 2] = __builtin_fmaf(xv[c], d4[2], acc[c][e4 + 2]); acc[c][e4 + 3] = __builtin_fmaf(xv[c], d4[3], acc[c][e4 + 3]);
        }
      }
    }
  }
  // every row parity leaves its partials in LDS, slab [rp][cg][4*EPAD + 1] (the odd stride keeps both the per-thread
  // writes - lanes = column groups - and the linear read below conflict-free); then all threads add the slabs in parity
  // order and write [D][E] in output order (coalesced)
  constexpr int GS = 4 * EPAD + 1;
  float *mine = spart + ((int64_t)rp * CG + cg) * GS;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < EPAD; ++e) mine[c * EPAD + e] = acc[c][e];
  __syncthreads();
  float *out = part_dw + (int64_t)blockIdx.x * D * E;
  for (int q = tid; q < D * EPAD; q += blockDim.x) {
    const int g = q / (4 * EPAD), r4 = q - g * (4 * EPAD), e = r4 % EPAD, d = g * 4 + r4 / EPAD;
    float v = 0.f;
    for (int r = 0; r < RP; ++r) v += spart[((int64_t)r * CG + g) * GS + r4];
    if (e < E) out[(int64_t)d * E + e] = v;
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

template <typename T, int EP, int EW>
static void launch_gate_fwd_e(bool exact, dim3 grid, hipStream_t s, const GateFwdDev &d) {
  constexpr int NT = GATE_TOK * (EP / EW);
  const int rowb = d.D * (int)sizeof(T);
  if (exact && rowb == 6 * GATE_ROWB) hipLaunchKernelGGL((gate_fwd_kernel<T, EP, EW, true, 6>), grid, dim3(NT), 0, s, d);
  else if (exact && rowb == 12 * GATE_ROWB) hipLaunchKernelGGL((gate_fwd_kernel<T, EP, EW, true, 12>), grid, dim3(NT), 0, s, d);
  else if (exact) hipLaunchKernelGGL((gate_fwd_kernel<T, EP, EW, true, 0>), grid, dim3(NT), 0, s, d);
  else hipLaunchKernelGGL((gate_fwd_kernel<T, EP, EW, false, 0>), grid, dim3(NT), 0, s, d);
}

template <typename T>
static int launch_gate_fwd(int epad, dim3 grid, hipStream_t s, const GateFwdDev &d) {
  const bool exact = d.E == epad;
  switch (epad) {
    case 4: launch_gate_fwd_e<T, 4, 4>(exact, grid, s, d); break;
    case 8: launch_gate_fwd_e<T, 8, 4>(exact, grid, s, d); break;
    case 16: launch_gate_fwd_e<T, 16, 4>(exact, grid, s, d); break;
    case 32: launch_gate_fwd_e<T, 32, 8>(exact, grid, s, d); break;
    case 64: launch_gate_fwd_e<T, 64, 16>(exact, grid, s, d); break;
    default: return M3_ERR_UNSUPPORTED;
  }
  return check_launch("m3_gate_fwd");
}

static int epad_of(int E) { return E <= 4 ? 4 : E <= 8 ? 8 : E <= 16 ? 16 : E <= 32 ? 32 : 64; }
static int epad8_of(int E) { return E <= 8 ? 8 : E <= 16 ? 16 : E <= 32 ? 32 : 64; }

extern "C" int m3_gate_fwd(const m3_gate_fwd_args *a, void *stream) {
  M3_REQUIRE(a && a->x && a->w_gate && a->idx && a->score && a->top_logits && a->part_importance && a->part_load,
             "m3_gate_fwd: null operand");
  M3_REQUIRE(dtype_ok(a->x_dtype), "m3_gate_fwd: bad dtype");
  M3_REQUIRE(a->E >= 2 && a->E <= 64, "m3_gate_fwd: E=%d outside [2,64]", a->E);
  M3_REQUIRE(a->k >= 1 && a->k <= 8 && a->k <= a->E, "m3_gate_fwd: k=%d invalid for E=%d", a->k, a->E);
  const int es = dtype_size(a->x_dtype);
  M3_REQUIRE(a->T >= 0 && a->D > 0 && (a->D * es) % 16 == 0 && (a->ldx * es) % 16 == 0 && ((uintptr_t)a->x % 16) == 0,
             "m3_gate_fwd: rows must be 16-byte aligned and D*elem a multiple of 16");
  const bool noisy_run = a->noise && a->noise_std != 0.f;
  M3_REQUIRE(!a->part_load_prob || (noisy_run && a->k < a->E && a->clean && a->noisy),
             "m3_gate_fwd: part_load_prob (Normal-CDF load) needs noise, noise_std != 0, k < E and the dense outputs");
  if (a->T == 0) return M3_OK;
  GateFwdDev d;
  d.x = (const char *)a->x; d.T = a->T; d.D = a->D; d.ldx_b = a->ldx * es;
  d.w = a->w_gate; d.E = a->E; d.bias = a->logit_bias; d.noise = a->noise; d.noise_std = a->noise_std; d.k = a->k;
  d.idx = a->idx; d.idx32 = a->idx32; d.idx_next = a->idx_next; d.score = a->score; d.top_logits = a->top_logits;
  d.clean = a->clean; d.noisy = a->noisy; d.gates = a->gates;
  d.part_imp = a->part_importance; d.part_load = a->part_load; d.part_load_prob = a->part_load_prob;
  d.part_count = a->part_count;
  const dim3 grid((unsigned)m3_gate_num_blocks(a->T));
  hipStream_t s = (hipStream_t)stream;
  if (a->x_dtype == M3_F16) return launch_gate_fwd<half_t>(epad_of(a->E), grid, s, d);
  else if (a->x_dtype == M3_BF16) return launch_gate_fwd<bf16_t>(epad_of(a->E), grid, s, d);
  return launch_gate_fwd<float>(epad_of(a->E), grid, s, d);
}

extern "C" int m3_gate_reduce(const float *part_importance, const int32_t *part_load, int nblk, int E,
                              float *importance, int64_t *load, void *stream) {
  M3_REQUIRE(part_importance && part_load && importance && load && E >= 1 && E <= 64 && nblk >= 0,
             "m3_gate_reduce: bad args");
  int rc = launch_reduce_rows_f32(part_importance, nblk, E, 1, 0, importance, 0, (hipStream_t)stream);
  if (rc) return rc;
  return launch_reduce_rows_i32(part_load, nblk, E, 1, 0, load, 0, (hipStream_t)stream);
}

extern "C" int m3_balance_loss(const float *part_importance, const int32_t *part_load, const float *part_load_prob,
                               int nblk, int E, float *importance, int64_t *load, float *load_prob, float *loss_out,
                               float *loss_acc, float *d_importance, float *d_load_prob, void *stream) {
  M3_REQUIRE(part_importance && part_load && importance && load && E >= 1 && E <= 64 && nblk >= 0,
             "m3_balance_loss: bad args");
  M3_REQUIRE(!part_load_prob || load_prob, "m3_balance_loss: load_prob output needed with part_load_prob");
  BalanceDev d;
  d.part_imp = part_importance; d.part_load = part_load; d.part_load_prob = part_load_prob;
  d.nblk = nblk; d.E = E; d.importance = importance; d.load = load; d.load_prob = load_prob;
  d.loss_acc = loss_acc; d.loss_out = loss_out; d.d_importance = d_importance; d.d_load_prob = d_load_prob;
  d.part_count = nullptr; d.blk_base = nullptr; d.counts = nullptr; d.offsets = nullptr; d.tile_starts = nullptr; d.counts64 = nullptr;
  hipLaunchKernelGGL(balance_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, d);
  return check_launch("m3_balance_loss");
}

extern "C" int m3_balance_route(const float *part_importance, const int32_t *part_load, const float *part_load_prob,
                                int nblk, int E, float *importance, int64_t *load, float *load_prob, float *loss_out,
                                float *loss_acc, float *d_importance, float *d_load_prob, const int32_t *part_count,
                                int32_t *blk_base, int32_t *counts, int32_t *offsets, int32_t *tile_starts,
                                int64_t *counts64, void *stream) {
  M3_REQUIRE(part_importance && part_load && importance && load && E >= 1 && E <= 64 && nblk >= 1,
             "m3_balance_route: bad args");
  M3_REQUIRE(!part_load_prob || load_prob, "m3_balance_route: load_prob output needed with part_load_prob");
  M3_REQUIRE(part_count && blk_base && counts && offsets && tile_starts, "m3_balance_route: null routing operand");
  BalanceDev d;
  d.part_imp = part_importance; d.part_load = part_load; d.part_load_prob = part_load_prob;
  d.nblk = nblk; d.E = E; d.importance = importance; d.load = load; d.load_prob = load_prob;
  d.loss_acc = loss_acc; d.loss_out = loss_out; d.d_importance = d_importance; d.d_load_prob = d_load_prob;
  d.part_count = part_count; d.blk_base = blk_base; d.counts = counts; d.offsets = offsets; d.tile_starts = tile_starts;
  d.counts64 = counts64;
  hipLaunchKernelGGL(balance_kernel, dim3(2), dim3(1024), 0, (hipStream_t)stream, d);      // workgroup 0: balance loss, 1: routing scan
  return check_launch("m3_balance_route");
}

extern "C" int m3_gate_bwd_logits(const m3_gate_bwd_args *a, void *stream) {
  M3_REQUIRE(a && a->noisy && a->idx && a->d_logits && a->E >= 2 && a->E <= 64 && a->k >= 1 && a->k <= a->E,
             "m3_gate_bwd_logits: bad args");
  const bool has_next = a->k < a->E;
  M3_REQUIRE(!a->d_top || !has_next || a->idx_next, "m3_gate_bwd_logits: d_top needs idx_next");
  M3_REQUIRE(!a->d_load_prob || (has_next && a->idx_next && a->clean && a->top_logits && a->noise_std != 0.f),
             "m3_gate_bwd_logits: d_load_prob needs k < E, idx_next, clean, top_logits and noise_std != 0");
  if (a->T == 0) return M3_OK;
  GateBwdDev d;
  d.noisy = a->noisy; d.clean = a->clean; d.top_logits = a->top_logits; d.idx = a->idx; d.idx_next = a->idx_next;
  d.d_score = a->d_score; d.d_top = a->d_top; d.d_importance = a->d_importance; d.d_load_prob = a->d_load_prob;
  d.balance_scale = a->balance_scale; d.noise_std = a->noise_std; d.T = a->T; d.E = a->E; d.k = a->k;
  d.d_logits = a->d_logits;
  d.balance_scale_dev = a->balance_scale_dev;
  d.d_logits_act = a->d_logits_act; d.act_dtype = a->act_dtype;
  M3_REQUIRE(!a->d_logits_act || dtype_ok(a->act_dtype), "m3_gate_bwd_logits: bad dtype of the second copy");
  const dim3 grid((unsigned)((a->T + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  switch (epad_of(a->E)) {
    case 4: hipLaunchKernelGGL(gate_bwd_logits_kernel<4>, grid, dim3(256), 0, s, d); break;
    case 8: hipLaunchKernelGGL(gate_bwd_logits_kernel<8>, grid, dim3(256), 0, s, d); break;
    case 16: hipLaunchKernelGGL(gate_bwd_logits_kernel<16>, grid, dim3(256), 0, s, d); break;
    case 32: hipLaunchKernelGGL(gate_bwd_logits_kernel<32>, grid, dim3(256), 0, s, d); break;
    default: hipLaunchKernelGGL(gate_bwd_logits_kernel<64>, grid, dim3(256), 0, s, d); break;
  }
  return check_launch("m3_gate_bwd_logits");
}

extern "C" int m3_gate_bwd_params(const void *x, int x_dtype, int64_t T, int D, int64_t ldx, const float *w_gate,
                                  int E, const float *d_logits, float *part_dw, float *d_w_gate, int beta_dw,
                                  float *dx, int64_t lddx, int beta_dx, void *stream) {
  M3_REQUIRE(x && w_gate && d_logits, "m3_gate_bwd_params: null operand");
  M3_REQUIRE(dtype_ok(x_dtype), "m3_gate_bwd_params: bad dtype");
  M3_REQUIRE(E >= 2 && E <= 64 && D > 0 && D <= 1024, "m3_gate_bwd_params: E in [2,64], D <= 1024");
  M3_REQUIRE((d_w_gate == nullptr) == (part_dw == nullptr), "m3_gate_bwd_params: part_dw and d_w_gate go together");
  if (T == 0) return M3_OK;
  hipStream_t s = (hipStream_t)stream;
  const int es = dtype_size(x_dtype);
  const int ep = epad8_of(E);
  if (d_w_gate) {
    const int nblk = m3_gate_dw_blocks(T);
    const int64_t elems = (int64_t)D * E;
    // E <= 16, 4-column groups that fill a 1024-thread workgroup, 8-byte-aligned rows: the four-d-per-thread kernel and the
    // two-stage row reduction (8 row lanes per column block instead of one thread walking all nblk partial rows)
    const int rpn = (D / 4) * 4 <= 512 ? 4 : 2;
    const size_t lds4 = ((size_t)GATE_DW_TOK * ep + (size_t)rpn * (D / 4) * (4 * ep + 1)) * sizeof(float);
    if (ep <= 16 && D % 4 == 0 && (D / 4) * rpn <= 512 && ((uintptr_t)x % 8) == 0 && (ldx * es) % 8 == 0 &&
        lds4 <= 144 * 1024) {
      const dim3 grid4(nblk), block4((D / 4) * rpn);
#define M3_DW4_CASE(TT, EP)                                                                                                  \
  do {                                                                                                                       \
    static bool attr = false;                                                                                                \
    if (!attr) { (void)hipFuncSetAttribute((const void *)gate_bwd_dw4_kernel<TT, EP>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024); attr = true; } \
    hipLaunchKernelGGL((gate_bwd_dw4_kernel<TT, EP>), grid4, block4, lds4, s, (const char *)x, T, D, ldx * es, d_logits, E, part_dw); \
  } while (0)
      if (x_dtype == M3_F16) { if (ep == 8) M3_DW4_CASE(half_t, 8); else M3_DW4_CASE(half_t, 16); }
      else if (x_dtype == M3_BF16) { if (ep == 8) M3_DW4_CASE(bf16_t, 8); else M3_DW4_CASE(bf16_t, 16); }
      else { if (ep == 8) M3_DW4_CASE(float, 8); else M3_DW4_CASE(float, 16); }
#undef M3_DW4_CASE
      int rc4 = check_launch("m3_gate_bwd_params(dw4)");
      if (rc4) return rc4;
      rc4 = launch_reduce_rows_f32(part_dw, nblk, (int)elems, 1, 0, d_w_gate, beta_dw, s);
      if (rc4) return rc4;
    } else {
    const dim3 grid(nblk), block(((D + 63) / 64) * 64);
#define M3_DW_CASE(TT, EP)                                                                                     \
  hipLaunchKernelGGL((gate_bwd_dw_kernel<TT, EP>), grid, block, 0, s, (const char *)x, T, D, ldx * es, d_logits, \
                     E, part_dw)
    if (x_dtype == M3_F16) {
      if (ep == 8) M3_DW_CASE(half_t, 8); else if (ep == 16) M3_DW_CASE(half_t, 16);
      else if (ep == 32) M3_DW_CASE(half_t, 32); else M3_DW_CASE(half_t, 64);
    } else if (x_dtype == M3_BF16) {
      if (ep == 8) M3_DW_CASE(bf16_t, 8); else if (ep == 16) M3_DW_CASE(bf16_t, 16);
      else if (ep == 32) M3_DW_CASE(bf16_t, 32); else M3_DW_CASE(bf16_t, 64);
    } else {
      if (ep == 8) M3_DW_CASE(float, 8); else if (ep == 16) M3_DW_CASE(float, 16);
      else if (ep == 32) M3_DW_CASE(float, 32); else M3_DW_CASE(float, 64);
    }
#undef M3_DW_CASE
    int rc = check_launch("m3_gate_bwd_params(dw)");
    if (rc) return rc;
    hipLaunchKernelGGL(gate_dw_reduce_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, part_dw, nblk,
                       elems, d_w_gate, beta_dw);
    rc = check_launch("m3_gate_bwd_params(dw reduce)");
    if (rc) return rc;
    }
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
