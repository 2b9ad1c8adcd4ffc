// Fused FFN forward for gfx950:  Y[crow(m), :] = epi( GELU(X[arow(m), :] W1[g]^T + b1[g]) W2[g]^T + b2[g] ).
//
// One launch replaces FC1 (+bias+GELU) and FC2 (+bias) of
//   - the expert FFN `_Expert.forward` (models/moe/ckpt/custom_moe_layer.py:36-44: htoh4 -> activation -> h4toh,
//     FMoELinear :32-33) together with the row gather / token-major scatter of MOEScatter / MOEGather behind
//     `_fmoe_general_global_forward` (:263-265), and
//   - the dense `Mlp.forward` (models/moe/ckpt/vision_transformer_moe.py:255-261: fc1 -> GELU -> fc2) with the
//     residual add of Block (:450),
// so that the hidden activations [rows, H] are never READ back from HBM in the forward; whether they are WRITTEN
// (pre_out / act_out, for a backward that keeps them) or recomputed in backward (the reference's default
// activation-checkpointing mode, vision_transformer_moe.py:495-524) is the caller's choice.
//
// Why this shape.  The unfused 128x128 GEMM tiles of gemm.hip stream BOTH operands through LDS (one operand byte per
// 64 FLOP) and at K = 384 that stream - not the MFMAs - bounds the loop (profiles/README.md).  Here a wave owns 16*MT
// token rows for the whole FFN and keeps everything that belongs to those rows in registers:
//   - its X rows as MFMA B-operand fragments (D/32 x MT x 4 VGPRs), loaded once;
//   - per 64-wide hidden chunk: hid^T = W1[chunk] X^T accumulated with the WEIGHT as the MFMA A operand, so the
//     accumulator of lane (g, i) holds hid[row i][4 consecutive h].  After bias + GELU + f16 conversion those
//     registers ARE the B-operand fragment of the second product (contraction over h) - no LDS round trip and no
//     lane movement; the k order inside a 32-group is permuted, and the W2 operand copy is stored with the same
//     permutation (m3_cast_batch flag M3_CAST_PERM32), which a contraction allows;
//   - the output accumulator Y^T[D x rows] (D/16 x MT x 4 VGPRs), summed over all hidden chunks.
// Only the weights move through LDS (half the LDS-fill bytes per FLOP of the 128x128 tile, no intermediate prologue /
// epilogue): 16 KiB slices ([128 rows] x [128 B = one cache line], XOR-swizzled 16-byte chunks) in a 3-slot ring.
// Staging is global -> registers -> LDS: a wave's four 16-byte loads per slice are issued TWO slices ahead (two
// register sets), written to the ring one barrier before the slice is read, with ONE s_barrier per slice placed in
// the MIDDLE of a slice's MFMAs so that the fragment reads of the next slice issue under the current slice's MFMAs.
// (A first version used LDS-DMA for the slices: with one wave per SIMD nothing hides the ~200 cycles a wave spends
// ISSUING each global_load_lds - in-kernel stamps, tools/ffn_stamps.py - and the loop ran at 1400 cycles per slice
// against 512 of MFMA work.)  Fragment reads are double-buffered in registers ([8 reads][16 MFMAs] pinned by
// sched_barrier).  One workgroup = 4 waves (one per SIMD, up to 512 registers each) = 64*MT rows; grouped calls take
// the per-expert row ranges from the device-resident offsets (no host sync) and never mix experts in a tile.
#include "common.h"

namespace m3 {

constexpr int FFN_SLICE = 16384;                 // bytes per weight slice: 128 image rows x 128 B
constexpr int FFN_NSLOT = 3;
constexpr int FFN_RING = FFN_NSLOT * FFN_SLICE;  // 48 KiB
constexpr int FFN_HC = 64;                       // hidden columns per chunk

struct FfnDev {
  const char *X; int64_t ldx_b;                  // byte stride of an X row
  const int32_t *x_row_idx; int32_t x_row_div;
  const char *W1; const char *W2p;               // [G][H][D], [G][D][H perm32]
  const float *b1; const float *b2;              // [G][H], [G][D] or null
  char *Y; int64_t ldy; int32_t y_f32;           // ldy in elements
  const int32_t *y_row_idx;
  const float *residual; int64_t ld_res;
  half_t *pre_out; half_t *act_out;              // [slot rows][H] or null
  int64_t M; int32_t H; int32_t G;
  const int32_t *group_offsets;
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ void glds16(const char *src, char *dst) {
  __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)dst, 16, 0, 0);
}

// Diagnostic build only (-DM3_FFN_STAMPS, tools/ffn_stamps.py): lane 0 of wave 0 records s_memtime at the phase
// boundaries of its workgroup; m3_debug_ffn_stamps copies them out.  No stamp executes in the shipped kernel.
#ifdef M3_FFN_STAMPS
constexpr int FSTAMP_WGS = 2048, FSTAMP_N = 16;
__device__ unsigned long long g_ffn_stamps[FSTAMP_WGS][FSTAMP_N];
#define FFN_STAMP(i)                                                                               \
  do {                                                                                             \
    if (threadIdx.x == 0 && blockIdx.x < FSTAMP_WGS) g_ffn_stamps[blockIdx.x][(i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define FFN_STAMP(i) do { } while (0)
#endif

// D: model width (contraction of FC1, output width of FC2), MT: 16-row tiles per wave, NW: waves per workgroup
// (4: one wave per SIMD, up to 512 registers; 8: two per SIMD, 256 registers - the partner wave covers the issue time
// of memory instructions, LDS latency and the GELU's VALU work, at twice the LDS fragment reads per MFMA when MT = 1)
// ABL (diagnostic builds only, M3_FFN_ABL): bit 0 no GELU, bit 1 no weight loads inside the loop, bit 2 no MFMAs
template <int D, int MT, int NW, bool OUT_F32, int ABL = 0>
__global__ __launch_bounds__(NW * 64, NW / 4) void ffn_fwd_kernel(const FfnDev p) {
  typedef Mma<half_t> MM;
  typedef MM::frag frag;
  constexpr int KS = D / 32;              // k steps of FC1
  constexpr int DT = D / 16;              // output tiles
  constexpr int WROWS = 16 * MT;          // rows per wave
  constexpr int ROWS = NW * WROWS;        // rows per workgroup
  constexpr int RU = ROWS / 64;           // row-index registers per lane (rows lane, lane + 64, ..)
  constexpr int NT = NW * 64;             // threads
  constexpr int PPW = 16 / NW;            // 1-KiB pieces of a weight slice per wave
  constexpr bool FA2 = NW == 4;           // double-buffered fragment registers (one wave per SIMD only: 256 registers do not hold a second set)
  static_assert(ROWS % 64 == 0 && RU <= 2 && 16 % NW == 0, "geometry");
  constexpr int XROWB = D * 2;            // bytes of an X row
  constexpr int XCH = XROWB / 16;         // 16-byte chunks per X row (multiple of 16)
  constexpr int NP = D / 128;             // slices per phase per hidden chunk
  constexpr int SPC = 2 * NP;             // slices per hidden chunk (even: the register set of a slice is static)
  static_assert(D % 128 == 0 && XCH % 16 == 0, "shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *const ring = smem;
  char *const xreg = smem + FFN_RING;     // X image (ROWS x XROWB); the biases sit behind it

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;

  FFN_STAMP(0);
  // ---- which tile: prefix of ceil(rows_g / ROWS) over the groups, from the device-resident offsets
  int g = 0, nwg, tile0 = 0;
  int64_t m_begin, m_end;
  if (p.group_offsets) {
    // lane e holds group e's tile count; wave-wide inclusive scan (G <= 64), every wave computes the same
    int n_e = 0;
    if (lane < p.G) n_e = (p.group_offsets[lane + 1] - p.group_offsets[lane] + ROWS - 1) / ROWS;
    int incl = n_e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    nwg = __shfl(incl, 63, 64);
    if ((int)blockIdx.x >= nwg) return;
    const int t = xcd_remap(blockIdx.x, nwg);
    g = __ffsll((long long)__ballot(incl > t)) - 1;
    tile0 = t - __shfl(incl - n_e, g, 64);
    g = __builtin_amdgcn_readfirstlane(g);
    tile0 = __builtin_amdgcn_readfirstlane(tile0);
    m_begin = (int64_t)p.group_offsets[g] + (int64_t)tile0 * ROWS;
    m_end = p.group_offsets[g + 1];
  } else {
    nwg = (int)gridDim.x;
    const int t = xcd_remap(blockIdx.x, nwg);
    m_begin = (int64_t)t * ROWS;
    m_end = p.M;
    if (m_begin >= m_end) return;
  }
  const int H = p.H;
  const int NC = H / FFN_HC;
  const int NSL = NC * SPC;
  const char *const w1g = p.W1 + (int64_t)g * H * XROWB;
  const char *const w2g = p.W2p + (int64_t)g * D * H * 2;

  // ---- weight slices: wave w owns image rows (4w + pc) * 8 .. + 7 (pc = 0..3), lane -> row + lane / 8, LDS chunk
  // slot lane % 8, which receives source chunk (lane % 8) ^ swz(row), swz(row) = (row >> 1) & 7 (lane-linear image).
  // Phase-1 slice j of chunk hc: image row r = lcl * 64 + hl  <-  W1[hc*64 + hl][k bytes (2j + lcl) * 128 ..]
  // Phase-2 slice j of chunk hc: image row r (= d - 128 j)     <-  W2p[128 j + r][h bytes hc * 128 ..]
  uint32_t off1[PPW], off2[PPW];
#pragma unroll
  for (int pc = 0; pc < PPW; ++pc) {
    const int r = (PPW * wave + pc) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    off1[pc] = (uint32_t)((r & 63) * XROWB + (r >> 6) * 128 + c * 16);
    off2[pc] = (uint32_t)(r * H * 2 + c * 16);
  }
  u32x4 wq[2][PPW];                             // two slices in flight in registers
  auto load_slice = [&](int sl, u32x4(&q)[PPW]) {
    const int hc = sl / SPC, j = sl - hc * SPC;
    if (j < NP) {
      const char *src = w1g + (int64_t)hc * (FFN_HC * XROWB) + j * 256;
#pragma unroll
      for (int pc = 0; pc < PPW; ++pc) q[pc] = *(const u32x4 *)(src + off1[pc]);
    } else {
      const char *src = w2g + (int64_t)(j - NP) * (128 * 2) * H + hc * 128;
#pragma unroll
      for (int pc = 0; pc < PPW; ++pc) q[pc] = *(const u32x4 *)(src + off2[pc]);
    }
  };
  auto slot_of = [&](int sl) { return ring + (sl % FFN_NSLOT) * FFN_SLICE; };
  auto write_slice = [&](int sl, const u32x4(&q)[PPW]) {
    char *dst = slot_of(sl) + wave * (PPW * 1024) + lane * 16;
#pragma unroll
    for (int pc = 0; pc < PPW; ++pc) *(u32x4 *)(dst + pc * 1024) = q[pc];
  };

  // ---- row indices of the tile: lane holds rows `lane` (and `lane + 64`); everything else gets them by shuffle
  int32_t xrow[RU], yrow[RU];
#pragma unroll
  for (int u = 0; u < RU; ++u) {
    int64_t m = m_begin + u * 64 + lane;
    if (m >= m_end) m = m_end - 1;              // clamp: valid memory, never stored
    xrow[u] = p.x_row_idx ? p.x_row_idx[m] / p.x_row_div : (int32_t)m;
    yrow[u] = p.y_row_idx ? p.y_row_idx[m] : (int32_t)m;
  }

  load_slice(0, wq[0]);
  load_slice(1, wq[1]);

  // ---- X image by LDS-DMA (once per tile): row r at xreg + r * XROWB, chunk c of the row stored at position
  // (c & ~15) | ((c & 15) ^ (r & 15))  (conflict-free ds_read_b128 of 16 rows at one k chunk).
  // Instruction q (wave w: q = w * XI + jj) fills flat positions 64 q .. 64 q + 63.
  constexpr int XI = ROWS * XCH / 64 / NW;      // X DMA instructions per wave
#pragma unroll
  for (int jj = 0; jj < XI; ++jj) {
    const int f = (wave * XI + jj) * 64 + lane;
    const int r = f / XCH, cp = f - r * XCH;
    const int c = (cp & ~15) | ((cp & 15) ^ (r & 15));
    int src = __shfl(xrow[0], r & 63, 64);
    if (RU > 1) { const int s1 = __shfl(xrow[RU - 1], r & 63, 64); if (r >= 64) src = s1; }
    glds16(p.X + (int64_t)src * p.ldx_b + c * 16, xreg + (wave * XI + jj) * 1024);
  }
  FFN_STAMP(1);                                  // loads issued

  // biases -> LDS (fp32): b1 [H] behind the X image, b2 [D] behind it; zeros when absent
  float *const b1s = (float *)(xreg + ROWS * XROWB);
  float *const b2s = b1s + H;
  {
    const int nb = H + D;
    for (int i0 = wave * 64; i0 < nb; i0 += NT) {
      const int i = i0 + lane;
      if (i < nb) {
        const float *src = (i < H) ? (p.b1 ? p.b1 + (int64_t)g * H + i : nullptr) : (p.b2 ? p.b2 + (int64_t)g * D + (i - H) : nullptr);
        b1s[i] = src ? *src : 0.f;
      }
    }
  }
  write_slice(0, wq[0]);
  if (2 < NSL) load_slice(2, wq[0]);
  __syncthreads();                               // vmcnt(0) + barrier: X image, biases and slice 0 are in LDS
  FFN_STAMP(2);

  // X fragments (B operand): lane (lg, li) holds X[16 mt + li][32 s + 8 lg .. + 7]
  frag xf[MT][KS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int r = wave * WROWS + mt * 16 + li;
      const int c = 4 * s + lg;
      xf[mt][s] = *(const frag *)(xreg + r * XROWB + (((c & ~15) | ((c & 15) ^ (r & 15))) << 4));
    }
  // rows of this lane's accumulator columns (tile row 16 mt + li): output row, validity, slot row
  int64_t crow[MT];
  bool rok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = wave * WROWS + mt * 16 + li;
    int v = __shfl(yrow[0], r & 63, 64);
    if (RU > 1) { const int v1 = __shfl(yrow[RU - 1], r & 63, 64); if (r >= 64) v = v1; }
    crow[mt] = v;
    rok[mt] = m_begin + r < m_end;
  }
  FFN_STAMP(3);

  f32x4 yacc[DT][MT];
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) yacc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int swl = (li >> 1) & 7;
  // fragment read offsets inside a slice (A operand: 16 image rows at one 16-byte chunk)
  //   phase 1, k step sloc (0..3) of the slice: rows (sloc >> 1) * 64 + 16 ht + li, chunk ((sloc & 1) * 4 + lg) ^ swl
  //   phase 2, k step ks2 (0..1):               rows 16 dtl + li,                  chunk (ks2 * 4 + lg) ^ swl
  int rd1[4], rd2[2];
#pragma unroll
  for (int s = 0; s < 4; ++s) rd1[s] = ((s >> 1) * 64 + li) * 128 + ((((s & 1) * 4 + lg) ^ swl) << 4);
#pragma unroll
  for (int s = 0; s < 2; ++s) rd2[s] = li * 128 + (((s * 4 + lg) ^ swl) << 4);

  // Sync point S(n), between the two halves of slice n - 1 (n >= 1):
  //   a. this wave's quarter of slice n (in registers since S(n - 2)) goes to ring slot n % 3 - the slot of slice
  //      n - 3, which every wave finished before it passed S(n - 1);
  //   b. barrier: slice n is readable (and every wave is done with the first half of slice n - 1);
  //   c. the loads of slice n + 2 are issued into the register set that step a freed.
  // The compiler counts these plain loads (and the optional pre / act stores) itself: no hand-written vmcnt.
  auto sync_next = [&](int sl, u32x4(&q)[PPW]) {   // called between the halves of slice sl; q = register set of sl + 1
    const int nx = sl + 1;
    if (nx >= NSL) return;
    write_slice(nx, q);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nx + 2 < NSL && !(ABL & 2)) load_slice(nx + 2, q);
  };

  // Fragment double buffer: while the 16 MFMAs of one half slice run on fa[cur], the 8 fragment reads of the next half
  // slice are already in flight into fa[cur ^ 1] (one wave per SIMD: nobody else hides the LDS latency).  A half slice is
  // 8 fragments: f = kc * 4 + ht (phase 1: k step 2 * half + kc, hidden tile ht) or f = ks2 * 4 + d4 (phase 2: output
  // tile 4 * half + d4) - consecutive MFMAs go to different accumulators.
  frag fa[FA2 ? 2 : 1][8];
  auto load1 = [&](frag(&f)[8], const char *sb, int half) {
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int ht = 0; ht < 4; ++ht) f[kc * 4 + ht] = *(const frag *)(sb + rd1[half * 2 + kc] + ht * 16 * 128);
  };
  auto load2 = [&](frag(&f)[8], const char *sb, int half) {
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
      for (int d4 = 0; d4 < 4; ++d4) f[ks2 * 4 + d4] = *(const frag *)(sb + rd2[ks2] + (half * 4 + d4) * 16 * 128);
  };

  if (FA2) load1(fa[0], slot_of(0), 0);
  for (int hc = 0; hc < NC; ++hc) {
    const int sl0 = hc * SPC;
    f32x4 acc1[4][MT];
#pragma unroll
    for (int ht = 0; ht < 4; ++ht) {
      const f32x4 bv = *(const f32x4 *)(b1s + hc * FFN_HC + ht * 16 + 4 * lg);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc1[ht][mt] = bv;       // bias as the initial accumulator (rows of the tile = h)
    }
    // ---- phase 1: hid^T[chunk] = W1[chunk] X^T
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int sl = sl0 + j;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (FA2) {
          if (half == 0) {
            load1(fa[1], slot_of(sl), 1);
          } else {
            sync_next(sl, wq[(j + 1) & 1]);
            if (j + 1 < NP) load1(fa[0], slot_of(sl + 1), 0); else load2(fa[0], slot_of(sl + 1), 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        } else {
          if (half == 1) sync_next(sl, wq[(j + 1) & 1]);
          load1(fa[0], slot_of(sl), half);
        }
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
          for (int ht = 0; ht < 4; ++ht)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const frag a = fa[FA2 ? half : 0][kc * 4 + ht];
              if (ABL & 4) { asm volatile("" :: "v"(a)); acc1[ht][mt][0] += 1.f; }
              else acc1[ht][mt] = MM::mma(a, xf[mt][4 * j + 2 * half + kc], acc1[ht][mt]);
            }
        if (FA2) __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- optional outputs for a backward that keeps the hidden activations: pre = x W1^T + b1, act = GELU(pre),
    // rows in slot order; the lane owns 4 consecutive h of one row per (ht, mt)
    // ---- GELU, f16: the accumulators become the B fragments of phase 2 (k slot 8 lg + jj <-> h = 16 (jj / 4) + 4 lg + jj % 4)
    frag hf[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int64_t orow = (m_begin + wave * WROWS + mt * 16 + li) * H + hc * FFN_HC + 4 * lg;
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        f32x4 t[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x4 pre = acc1[2 * ks2 + u][mt];
          if (p.pre_out && rok[mt]) Vec4<half_t>::store(p.pre_out + orow + (2 * ks2 + u) * 16, pre);
#pragma unroll
          for (int r = 0; r < 4; ++r) t[u][r] = (ABL & 1) ? pre[r] : gelu_f(pre[r]);
        }
        hf[mt][ks2] = MM::from_tiles(t);
        if (p.act_out && rok[mt]) {
          *(f16x4 *)(p.act_out + orow + (2 * ks2) * 16) = __builtin_shufflevector(hf[mt][ks2], hf[mt][ks2], 0, 1, 2, 3);
          *(f16x4 *)(p.act_out + orow + (2 * ks2 + 1) * 16) = __builtin_shufflevector(hf[mt][ks2], hf[mt][ks2], 4, 5, 6, 7);
        }
      }
    }
    // ---- phase 2: Y^T += W2p[:, chunk] hid^T[chunk]
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int sl = sl0 + NP + j;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (FA2) {
          if (half == 0) {
            load2(fa[1], slot_of(sl), 1);
          } else {
            sync_next(sl, wq[(NP + j + 1) & 1]);
            if (j + 1 < NP) load2(fa[0], slot_of(sl + 1), 0); else load1(fa[0], slot_of(sl + 1), 0);   // (past the end: unused)
          }
          __builtin_amdgcn_sched_barrier(0);
        } else {
          if (half == 1) sync_next(sl, wq[(NP + j + 1) & 1]);
          load2(fa[0], slot_of(sl), half);
        }
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
          for (int d4 = 0; d4 < 4; ++d4)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const int dt = 8 * j + 4 * half + d4;
              const frag a = fa[FA2 ? half : 0][ks2 * 4 + d4];
              if (ABL & 4) { asm volatile("" :: "v"(a), "v"(hf[mt][ks2])); yacc[dt][mt][0] += 1.f; }
              else yacc[dt][mt] = MM::mma(a, hf[mt][ks2], yacc[dt][mt]);
            }
        if (FA2) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  FFN_STAMP(4);                                  // main loop done

  // ---- epilogue straight from the accumulators: the lane owns Y[row 16 mt + li][16 dt + 4 lg .. + 3]
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const f32x4 bv = *(const f32x4 *)(b2s + dt * 16 + 4 * lg);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (!rok[mt]) continue;
      f32x4 v = yacc[dt][mt] + bv;
      const int col = dt * 16 + 4 * lg;
      if (OUT_F32) {
        if (p.residual) v += *(const f32x4 *)(p.residual + crow[mt] * p.ld_res + col);
        *(f32x4 *)((float *)p.Y + crow[mt] * p.ldy + col) = v;
      } else {
        Vec4<half_t>::store((half_t *)p.Y + crow[mt] * p.ldy + col, v);
      }
    }
  }
  FFN_STAMP(5);                                  // stores issued
#ifdef M3_FFN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FFN_STAMP(6);                                  // stores acknowledged
#endif
}

}  // namespace m3

using namespace m3;

extern "C" int m3_ffn_fwd(const m3_ffn_args *a, void *stream) {
  M3_REQUIRE(a && a->X && a->W1 && a->W2p && a->Y, "m3_ffn_fwd: null operand");
  M3_REQUIRE(a->dtype == M3_F16, "m3_ffn_fwd: fp16 activations only (fp32 runs the unfused m3_gemm_nt pair)");
  M3_REQUIRE(a->D == 384 || a->D == 768, "m3_ffn_fwd: D must be 384 or 768 (got %d)", a->D);
  M3_REQUIRE(a->H >= 64 && a->H % 64 == 0 && a->H <= 8192, "m3_ffn_fwd: H must be a multiple of 64 (got %d)", a->H);
  M3_REQUIRE(a->M >= 0 && a->G >= 1, "m3_ffn_fwd: bad M / G");
  M3_REQUIRE(a->G == 1 || a->group_offsets, "m3_ffn_fwd: grouped call needs group_offsets");
  M3_REQUIRE(a->G <= 64, "m3_ffn_fwd: at most 64 groups");
  M3_REQUIRE(a->ldx >= a->D && (a->ldx * 2) % 16 == 0 && a->ldy >= a->D, "m3_ffn_fwd: leading dimensions");
  M3_REQUIRE(a->y_dtype == M3_F32 || a->y_dtype == M3_F16, "m3_ffn_fwd: bad y_dtype");
  M3_REQUIRE((a->ldy * (a->y_dtype == M3_F32 ? 4 : 2)) % 16 == 0, "m3_ffn_fwd: Y rows must be 16-byte aligned");
  M3_REQUIRE(((uintptr_t)a->X % 16) == 0 && ((uintptr_t)a->W1 % 16) == 0 && ((uintptr_t)a->W2p % 16) == 0 &&
             ((uintptr_t)a->Y % 16) == 0, "m3_ffn_fwd: operands must be 16-byte aligned");
  M3_REQUIRE(!a->residual || (a->y_dtype == M3_F32 && a->ld_res % 4 == 0 && ((uintptr_t)a->residual % 16) == 0),
             "m3_ffn_fwd: residual needs fp32 output and 16-byte aligned rows");
  M3_REQUIRE(!a->x_row_idx || a->x_row_div >= 1, "m3_ffn_fwd: x_row_div must be >= 1");
  M3_REQUIRE(((uintptr_t)a->pre_out % 8) == 0 && ((uintptr_t)a->act_out % 8) == 0, "m3_ffn_fwd: pre_out / act_out must be 8-byte aligned");
  M3_REQUIRE((int64_t)a->H * a->D * 2 < ((int64_t)1 << 31), "m3_ffn_fwd: one group's weights exceed the 32-bit lane offsets");
  M3_REQUIRE(a->M < ((int64_t)1 << 31), "m3_ffn_fwd: row count exceeds the 32-bit row indices");
  if (a->M == 0) return M3_OK;
  FfnDev d;
  d.X = (const char *)a->X; d.ldx_b = a->ldx * 2;
  d.x_row_idx = a->x_row_idx; d.x_row_div = a->x_row_idx ? a->x_row_div : 1;
  d.W1 = (const char *)a->W1; d.W2p = (const char *)a->W2p;
  d.b1 = a->b1; d.b2 = a->b2;
  d.Y = (char *)a->Y; d.ldy = a->ldy; d.y_f32 = a->y_dtype == M3_F32;
  d.y_row_idx = a->y_row_idx;
  d.residual = a->residual; d.ld_res = a->ld_res;
  d.pre_out = (half_t *)a->pre_out; d.act_out = (half_t *)a->act_out;
  d.M = a->M; d.H = a->H; d.G = a->G;
  d.group_offsets = a->group_offsets;
  hipStream_t s = (hipStream_t)stream;
  // geometry: D = 384: 128-row tiles, as 8 waves x 16 rows (two waves per SIMD; default) or 4 waves x 32 rows
  // (M3_FFN_WAVES=4); D = 768: 64-row tiles of 4 waves x 16 rows
  static int nw_env = -1, abl = -1;
  if (nw_env < 0) { const char *e = getenv("M3_FFN_WAVES"); nw_env = e ? atoi(e) : 8; }
  if (abl < 0) { const char *e = getenv("M3_FFN_ABL"); abl = e ? atoi(e) : 0; }
  const int nw = (a->D == 384 && nw_env != 4) ? 8 : 4;
  const int mt = (a->D == 384 && nw == 4) ? 2 : 1;
  const int rows = nw * 16 * mt;
  const int64_t tiles = (a->M + rows - 1) / rows + (a->group_offsets ? a->G : 0);
  M3_REQUIRE(tiles < ((int64_t)1 << 30), "m3_ffn_fwd: grid too large");
  const size_t lds = FFN_RING + (size_t)rows * a->D * 2 + (size_t)(a->H + a->D) * 4;
  M3_REQUIRE(lds <= 163840, "m3_ffn_fwd: H too large for the bias image");
  const dim3 grid((unsigned)tiles), block(nw * 64);
#define M3_FFN_LAUNCH(DD, MTT, NWW, F32, AB)                                                                          \
  do {                                                                                                                \
    static bool attr_set = false;                                                                                     \
    if (!attr_set) {                                                                                                  \
      (void)hipFuncSetAttribute((const void *)ffn_fwd_kernel<DD, MTT, NWW, F32, AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840); \
      attr_set = true;                                                                                                \
    }                                                                                                                 \
    hipLaunchKernelGGL((ffn_fwd_kernel<DD, MTT, NWW, F32, AB>), grid, block, lds, s, d);                              \
  } while (0)
  if (a->D == 384 && nw == 8) {
#ifdef M3_FFN_ABLATIONS
    if (abl == 1 && !d.y_f32) M3_FFN_LAUNCH(384, 1, 8, false, 1);
    else if (abl == 2 && !d.y_f32) M3_FFN_LAUNCH(384, 1, 8, false, 2);
    else if (abl == 4 && !d.y_f32) M3_FFN_LAUNCH(384, 1, 8, false, 4);
    else if (abl == 7 && !d.y_f32) M3_FFN_LAUNCH(384, 1, 8, false, 7);
    else
#endif
    if (d.y_f32) M3_FFN_LAUNCH(384, 1, 8, true, 0); else M3_FFN_LAUNCH(384, 1, 8, false, 0);
  } else if (a->D == 384) {
    if (d.y_f32) M3_FFN_LAUNCH(384, 2, 4, true, 0); else M3_FFN_LAUNCH(384, 2, 4, false, 0);
  } else {
    if (d.y_f32) M3_FFN_LAUNCH(768, 1, 4, true, 0); else M3_FFN_LAUNCH(768, 1, 4, false, 0);
  }
#undef M3_FFN_LAUNCH
  return check_launch("m3_ffn_fwd");
}

#ifdef M3_FFN_STAMPS
extern "C" int m3_debug_ffn_stamps(unsigned long long *dst, int wgs) {
  if (wgs > FSTAMP_WGS) wgs = FSTAMP_WGS;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ffn_stamps), (size_t)wgs * FSTAMP_N * sizeof(unsigned long long)) == hipSuccess ? M3_OK : M3_ERR_LAUNCH;
}
#endif
