// Long-contraction NT GEMM for gfx950: 256 x 256 output tiles, eight waves, operands through a two-slot LDS ring filled by
// LDS-DMA that stays in flight across barriers.  Same contract as gemm_nt_dma_kernel (gemm.hip): C[m,n] = epi(sum_k
// A[arow(m),k] * B[g][n,k]), grouped over experts, row gather on the A load, row scatter on the C store, the four
// epilogue kinds.  It takes the launches whose contraction is long enough to amortise a 256 x 256 tile's prologue and
// store phase: the ViT-Base shapes (K = 768 / 3072: expert FC1 / FC2 of BASELINE configs[3] / configs[4], reference call
// sites models/moe/ckpt/custom_moe_layer.py:24-44; qkv / fc1 / fc2 of vision_transformer_moe.py:255-261,295-313).
//
// Why another tile.  A 128 x 128 tile moves one operand byte per 64 FLOP through the CU's L2 -> LDS path (21.5 B/clk/CU
// measured): the K loop of gemm_nt_dma_kernel cannot pass ~670 TFLOP/s whatever K is.  A 256 x 256
// tile moves one byte per 128 FLOP and its 8 waves hold the whole fp32 tile in registers (128 accumulator registers per
// lane), so one workgroup owns a CU and nobody else's MFMAs cover its memory latency: the loads must run AHEAD of the
// MFMAs inside the workgroup.
//
// Schedule (per 64-deep K tile = 4 phases of 16 MFMAs 16x16x32 per wave; a wave owns 128 rows x 64 columns and
// multiplies one 64 x 32 quadrant per phase, re-using either its A or its B fragments from the phase before):
//   phase 0  quadrant (m0, n0)   reads A-m0 (8 x ds_read_b128) + B-n0 (4)     DMA: B-n1 of tile t+1
//   phase 1  quadrant (m0, n1)   reads B-n1 (4)                               DMA: A-m1 of tile t+1
//   phase 2  quadrant (m1, n1)   reads A-m1 (8)                               DMA: A-m0 of tile t+2
//   phase 3  quadrant (m1, n0)   reads nothing (B-n0 is still in registers)   DMA: B-n0 of tile t+2, then vmcnt(4)
// A K tile's slot holds four 16 KiB half-tile images in the order they are first read: [A-m0 | B-n0 | B-n1 | A-m1]
// (A-m0 = the first 64 rows of each wave row's 128, B-n0 = the first 32 columns of each wave column's 64), each filled by
// two LDS-DMA instructions per wave (8 rows x 128 B a piece, lane-linear in LDS, the XOR swizzle on the SOURCE address).
// A half-tile is re-filled two phases after its last read and is waited for five to six phases after its DMA was issued:
// six half-tiles (96 KiB) are in flight per CU.  vmcnt is waited once per K tile, never to zero inside the loop.
// Each phase is [LDS reads + DMA issue | barrier | MFMAs | barrier]; waves 4-7 run one barrier behind waves 0-3, so on
// every SIMD one wave multiplies while its partner reads and issues (MI355X_MICROARCH.md, two waves per SIMD).
// Hazards (the stagger costs one barrier each way):
//   RAW  a DMA is readable one phase after the phase whose FIRST barrier follows the issuing waves' vmcnt wait;
//   WAR  a half-tile is re-filled >= 2 phases after the phase that read it last.
// The LDS reads of the loop are inline asm: the compiler orders a ds_read it knows about behind every LDS-DMA in flight
// (vmcnt(0) in front of the first read of each phase), which would serialise the ring.
#include "gemm_dev.h"

namespace m3 {

constexpr int XB = 256;                  // tile edge
constexpr int XT = 512;                  // threads
constexpr int XH = 16384;                // one half-tile image: 128 rows x 128 B
constexpr int XSLOT = 4 * XH;            // [A-m0 | B-n0 | B-n1 | A-m1]
constexpr int X_A0 = 0, X_B0 = XH, X_B1 = 2 * XH, X_A1 = 3 * XH;
constexpr int XRING = 2 * XSLOT;         // 128 KiB
constexpr int XLDS = XRING + XB * 4;     // + the tile's 256 per-row epilogue factors

enum { X_FULL = 0, X_PENULT = 1, X_LAST = 2 };

#define X_LDSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))

template <typename T, int EPI>
__global__ __launch_bounds__(XT, 2) void gemm_nt_big_kernel(const GemmDev p) {
  typedef Mma<T> MM;
  typedef typename MM::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;

  // ---- which tile.  Grouped: 256-row tiles per expert, counted here from the device-resident group offsets (the route's
  // tile prefix is in 128-row tiles): lane l holds expert l's tile count, a shuffle scan gives the prefix (G <= 64).
  int g = 0;
  int64_t m_begin, m_end;
  int nwg, incl = 0;
  if (p.group_offsets) {
    int tl = 0;
    if (lane < p.G) tl = (p.group_offsets[lane + 1] - p.group_offsets[lane] + XB - 1) / XB;
    incl = tl;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    nwg = __builtin_amdgcn_readfirstlane(__shfl(incl, 63, 64)) * p.n_tiles;
  } else {
    nwg = (int)gridDim.x;
  }
  if ((int)blockIdx.x >= nwg) return;
  const int t = xcd_remap(blockIdx.x, nwg);
  int mt, nt;
  tile_of(t, p.n_tiles, p.m_band, nwg / p.n_tiles, mt, nt);
  if (p.group_offsets) {
    g = __popcll(__ballot(incl <= mt));
    int t0 = g ? __shfl(incl, g - 1, 64) : 0;
    g = __builtin_amdgcn_readfirstlane(g);
    t0 = __builtin_amdgcn_readfirstlane(t0);
    m_begin = (int64_t)__builtin_amdgcn_readfirstlane(p.group_offsets[g]) + (int64_t)(mt - t0) * XB;
    m_end = __builtin_amdgcn_readfirstlane(p.group_offsets[g + 1]);
  } else {
    m_begin = (int64_t)mt * XB;
    m_end = p.M;
  }
  const int n0 = nt * XB;

  // ---- DMA sources.  Wave w, piece j fills image rows (2 w + j) * 8 .. + 7 of a half-tile; lane l -> image row + l / 8,
  // LDS slot l % 8, which holds source chunk (l % 8) ^ swz(image row).  Image row r of A-m{h} is tile row
  // (r & 63) + 128 (r >> 6) + 64 h; of B-n{h} it is tile row 64 (r >> 5) + 32 h + (r & 31).  Rows past the end are
  // clamped (their outputs are never stored).  Every index load is issued before the first one is used.
  uint32_t srcA[2][2], srcB[2][2];      // byte offsets from the (wave-uniform) operand bases: the host checks the 4 GiB reach
  int64_t mrow[2][2];
  int32_t aix[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (2 * wave + j) * 8 + (lane >> 3);
      int64_t m = m_begin + (r & 63) + ((r >> 6) << 7) + 64 * h;
      if (m >= m_end) m = m_end - 1;
      mrow[h][j] = m;
    }
  if (p.a_row_idx) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) aix[h][j] = p.a_row_idx[mrow[h][j]];
  }
  // per-row epilogue factor (DropPath scale / gate score of the routed row): thread r < 256 requests row r's factor now
  const bool want_rs = p.row_scale && tid < XB;
  const int32_t *rs_idx = p.row_scale_idx ? p.row_scale_idx : p.c_row_idx;
  int64_t rs_m = m_begin + tid;
  if (rs_m >= m_end) rs_m = m_end - 1;
  int32_t rs_ix = 0;
  if (want_rs && rs_idx) rs_ix = rs_idx[rs_m];
  float my_rs = 1.0f;
  if (want_rs) {
    const int64_t srow = rs_idx ? (int64_t)rs_ix : rs_m;
    my_rs = p.row_scale[p.row_scale_div == 1 ? srow : srow / p.row_scale_div];
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (2 * wave + j) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ dma_swz(r);
      const int64_t src = p.a_row_idx ? (int64_t)div_by(aix[h][j], p.a_row_div, p.a_row_sh) : mrow[h][j];
      srcA[h][j] = (uint32_t)(src * p.lda_b) + c * 16;
      int n = n0 + ((r >> 5) << 6) + 32 * h + (r & 31);
      if (n >= p.N) n = p.N - 1;
      srcB[h][j] = (uint32_t)((int64_t)n * p.ldb_b) + c * 16;
    }
  const char *const baseA = p.A;
  const char *const baseB = p.B + (int64_t)g * p.b_group_b;
  const int nkt = (p.K * (int)sizeof(T)) / 128;      // K tiles (the host guarantees an even number >= 2)

  char *const dma_dst = smem + wave * 2048;          // + half-tile image offset + j * 1024
  auto issue = [&](const char *base, const uint32_t (&src)[2], int kt, int off) {
    const char *b = base + (int64_t)kt * 128;        // (wave-uniform: the loads take the SGPR-base form)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((glb_void *)(b + src[j]), (lds_void *)(dma_dst + off + j * 1024), 16, 0, 0);
  };

  // fragment read addresses (bytes from the start of a half-tile image): A row wr * 64 + 16 i + li, B row wc * 32 + 16 i + li
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void *)smem;
  uint32_t rdA[2], rdB[2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    const uint32_t ch = (uint32_t)(((kc * 4 + lg) ^ dma_swz(li)) << 4);
    rdA[kc] = lds0 + (wr * 64 + li) * 128 + ch;
    rdB[kc] = lds0 + (wc * 32 + li) * 128 + ch;
  }

  f32x4 acc[4][8];   // [ni][mi]: rows of an MFMA tile = n (4 lg + r), columns = m (li)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  frag fa[4][2], fb0[2][2], fb1[2][2];

  // ---- prologue: all of tile 0 and the first two half-tiles of tile 1
  issue(baseA, srcA[0], 0, X_A0);
  issue(baseB, srcB[0], 0, X_B0);
  issue(baseB, srcB[1], 0, X_B1);
  issue(baseA, srcA[1], 0, X_A1);
  issue(baseA, srcA[0], 1, XSLOT + X_A0);
  issue(baseB, srcB[0], 1, XSLOT + X_B0);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // tile 0 has landed for every wave
  if (wr == 1) __builtin_amdgcn_s_barrier();         // waves 4-7 run one barrier behind

#define X_MFMA_SEG(MH, FB, NH)                                                                           \
  do {                                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    _Pragma("unroll") for (int kc = 0; kc < 2; ++kc)                                                     \
      _Pragma("unroll") for (int n2 = 0; n2 < 2; ++n2)                                                   \
        _Pragma("unroll") for (int m4 = 0; m4 < 4; ++m4)                                                 \
          acc[(NH) * 2 + n2][(MH) * 4 + m4] = MM::mma(FB[n2][kc], fa[m4][kc], acc[(NH) * 2 + n2][(MH) * 4 + m4]); \
    __builtin_amdgcn_s_setprio(0);                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    __builtin_amdgcn_s_barrier();                                                                        \
  } while (0)

  auto tile = [&](auto slot_c, auto mode_c, const int kt) {
    constexpr int SLOT = decltype(slot_c)::value, MODE = decltype(mode_c)::value;
    constexpr int S = SLOT * XSLOT, O = (SLOT ^ 1) * XSLOT;
    // the slot base goes into the address register: ds_read's offset field is 16 bits
    const uint32_t a0 = rdA[0] + S, a1 = rdA[1] + S, b0 = rdB[0] + S, b1 = rdB[1] + S;
    // ---- phase 0: quadrant (m0, n0)
#pragma unroll
    for (int i = 0; i < 2; ++i) { X_LDSR(fb0[i][0], b0, X_B0 + i * 2048); X_LDSR(fb0[i][1], b1, X_B0 + i * 2048); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { X_LDSR(fa[i][0], a0, X_A0 + i * 2048); X_LDSR(fa[i][1], a1, X_A0 + i * 2048); }
    if constexpr (MODE != X_LAST) issue(baseB, srcB[1], kt + 1, O + X_B1);
    X_MFMA_SEG(0, fb0, 0);
    // ---- phase 1: quadrant (m0, n1)
#pragma unroll
    for (int i = 0; i < 2; ++i) { X_LDSR(fb1[i][0], b0, X_B1 + i * 2048); X_LDSR(fb1[i][1], b1, X_B1 + i * 2048); }
    if constexpr (MODE != X_LAST) issue(baseA, srcA[1], kt + 1, O + X_A1);
    X_MFMA_SEG(0, fb1, 1);
    // ---- phase 2: quadrant (m1, n1)
#pragma unroll
    for (int i = 0; i < 4; ++i) { X_LDSR(fa[i][0], a0, X_A1 + i * 2048); X_LDSR(fa[i][1], a1, X_A1 + i * 2048); }
    if constexpr (MODE == X_FULL) issue(baseA, srcA[0], kt + 2, S + X_A0);
    X_MFMA_SEG(1, fb1, 1);
    // ---- phase 3: quadrant (m1, n0); the wait that makes tile kt + 1 readable from the next phase on
    if constexpr (MODE == X_FULL) {
      issue(baseB, srcB[0], kt + 2, S + X_B0);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    if constexpr (MODE == X_PENULT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    X_MFMA_SEG(1, fb0, 0);
  };
  const std::integral_constant<int, 0> s0; const std::integral_constant<int, 1> s1;
  const std::integral_constant<int, X_FULL> full; const std::integral_constant<int, X_PENULT> pen;
  const std::integral_constant<int, X_LAST> last;
  // (the host guarantees an EVEN number of K tiles >= 2: one loop of tile pairs and one straight-line final pair - with
  // an if / else over the tail's length the accumulators of the two paths end up in different registers and are
  // shuffled through scratch)
  int kt = 0;
#pragma nounroll
  for (; kt + 2 < nkt; kt += 2) { tile(s0, full, kt); tile(s1, full, kt + 1); }
  tile(s0, pen, kt);
  tile(s1, last, kt + 1);
#undef X_MFMA_SEG
  if (wr == 0) __builtin_amdgcn_s_barrier();         // (waves 0-3 make up for the barrier waves 4-7 started with)

  // ---- epilogue: the fp32 tile goes through the (now free) ring in two 128-row halves (half h = waves wr == h, 128 KiB),
  // every thread then owns 8 consecutive n of one row: 16-byte stores, whole 512-byte row segments per 32 lanes
  const float *bias = p.bias ? p.bias + (int64_t)g * p.N : nullptr;
  const int cg = tid & 31, r16 = tid >> 5;
  const int n = n0 + cg * 8;
  const bool ncol = n < p.N;
  f32x4 bb0 = f32x4{0.f, 0.f, 0.f, 0.f}, bb1 = bb0;
  if (bias && ncol) { bb0 = *(const f32x4 *)(bias + n); bb1 = *(const f32x4 *)(bias + n + 4); }
  float *const s_rs = (float *)(smem + XRING);
  if (tid < XB) s_rs[tid] = my_rs;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    // this half's memory operands, requested ahead of the staging barriers (rows past the end are clamped)
    u32x4 gq[8];
    int32_t crow8[8];
    if constexpr (EPI != DMA_EPI_ANY) {
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        int64_t m = m_begin + h * 128 + ps * 16 + r16;
        if (m >= m_end) m = m_end - 1;
        crow8[ps] = p.c_row_idx ? p.c_row_idx[m] : (int32_t)m;
      }
      if constexpr (EPI == DMA_EPI_GPRE) {
        if (ncol) {
#pragma unroll
          for (int ps = 0; ps < 8; ++ps) gq[ps] = *(const u32x4 *)((const T *)p.gpre + (int64_t)crow8[ps] * p.ld_gpre + n);
        }
      }
    }
    __syncthreads();                                 // the ring (h = 0) / the first half's image (h = 1) is free
    if (wr == h) {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int lrow = mi * 16 + li;
          const int chunk = wc * 16 + ni * 4 + lg;
          *(f32x4 *)(smem + lrow * 1024 + ((chunk ^ (lrow & 31)) << 4)) = acc[ni][mi];
        }
    }
    __syncthreads();
    if (ncol) {
      if constexpr (EPI != DMA_EPI_ANY) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
          const int lrow = ps * 16 + r16;
          const int64_t crow = crow8[ps];
          const int sw = lrow & 31;
          f32x4 v0 = *(const f32x4 *)(smem + lrow * 1024 + (((2 * cg) ^ sw) << 4));
          f32x4 v1 = *(const f32x4 *)(smem + lrow * 1024 + (((2 * cg + 1) ^ sw) << 4));
          v0 += bb0; v1 += bb1;
          if constexpr (EPI == DMA_EPI_GELU) {
            Vec8<T>::store((T *)p.pre_out + crow * p.ld_pre + n, v0, v1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] = gelu_f(v0[j]); v1[j] = gelu_f(v1[j]); }
          }
          if constexpr (EPI == DMA_EPI_GPRE) {
            typedef T t8 __attribute__((ext_vector_type(8)));
            const t8 pr = __builtin_bit_cast(t8, gq[ps]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] *= gelu_grad_f((float)pr[j]); v1[j] *= gelu_grad_f((float)pr[4 + j]); }
          }
          const float sc = s_rs[h * 128 + lrow];
          v0 *= sc; v1 *= sc;
          if constexpr (EPI == DMA_EPI_RES) {
            if (m_begin + h * 128 + lrow < m_end) {          // (C may be the residual buffer: no duplicate read-modify-write)
              v0 += *(const f32x4 *)(p.residual + crow * p.ld_res + n);
              v1 += *(const f32x4 *)(p.residual + crow * p.ld_res + n + 4);
              *(f32x4 *)((float *)p.C + crow * p.ldc + n) = v0;
              *(f32x4 *)((float *)p.C + crow * p.ldc + n + 4) = v1;
            }
          } else {
            Vec8<T>::store((T *)p.C + crow * p.ldc + n, v0, v1);
          }
        }
      } else {
#pragma unroll 2
        for (int ps = 0; ps < 8; ++ps) {
          const int lrow = ps * 16 + r16;
          const int64_t m = m_begin + h * 128 + lrow;
          if (m >= m_end) break;
          const int64_t crow = p.c_row_idx ? (int64_t)p.c_row_idx[m] : m;
          const int sw = lrow & 31;
          f32x4 v0 = *(const f32x4 *)(smem + lrow * 1024 + (((2 * cg) ^ sw) << 4));
          f32x4 v1 = *(const f32x4 *)(smem + lrow * 1024 + (((2 * cg + 1) ^ sw) << 4));
          v0 += bb0; v1 += bb1;
          if (p.pre_out) Vec8<T>::store((T *)p.pre_out + crow * p.ld_pre + n, v0, v1);
          if (p.act == M3_ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] = gelu_f(v0[j]); v1[j] = gelu_f(v1[j]); }
          }
          if (p.gpre) {
            f32x4 p0, p1;
            Vec8<T>::load((const T *)p.gpre + crow * p.ld_gpre + n, p0, p1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] *= gelu_grad_f(p0[j]); v1[j] *= gelu_grad_f(p1[j]); }
          }
          if (p.row_scale) {
            const float sc = s_rs[h * 128 + lrow];
            v0 *= sc; v1 *= sc;
          }
          if (p.residual) {
            v0 += *(const f32x4 *)(p.residual + crow * p.ld_res + n);
            v1 += *(const f32x4 *)(p.residual + crow * p.ld_res + n + 4);
          }
          if (p.c_f32) {
            *(f32x4 *)((float *)p.C + crow * p.ldc + n) = v0;
            *(f32x4 *)((float *)p.C + crow * p.ldc + n + 4) = v1;
          } else {
            Vec8<T>::store((T *)p.C + crow * p.ldc + n, v0, v1);
          }
        }
      }
    }
  }
}

// The 256 x 256 kernel takes a call when: 16-bit operands, an even number of whole 128-byte K slices and at least 8 of them (K >= 512), the staged
// epilogue's alignment, at most 64 groups (one lane per group in the tile scan), and enough tiles to fill the chip about
// (measured rule below).
bool gemm_big_eligible(const GemmDev &d, int es, bool force) {
  if (es != 2 || !d.vec8) return false;
  const int kb = d.K * es;
  if (kb % 256 != 0 || kb / 128 < 8) return false;      // an even number of 128-byte K tiles
  if (d.group_offsets && d.G > 64) return false;
  if (force) return true;
  // measured with streamed operands (tools/vitb_gemm_bench.py, profiles/r05_vitb_gemm_streamed.txt): one workgroup owns a CU, so
  // a tile's prologue and its store phase (256 KiB of outputs, the epilogue's GELU arithmetic) overlap with nothing - at
  // K = 768 (12 K tiles) they cost what the loop gains (+-3 %, the GELU' input gradient -8..-15 %) and the 128 x 128 kernel,
  // four workgroups per CU, keeps those launches; from K = 2048 on the loop wins (K = 3072: -25 % grouped, -6 % dense even
  // with 114 tiles on 256 CUs)
  if (kb / 128 < 32) return false;
  const int64_t mt = (d.M + XB - 1) / XB;
  const int64_t ntl = (d.N + XB - 1) / XB;
  if (d.N % XB > 0 && d.N % XB <= 128 && ntl <= 2) return false;      // (a mostly empty column tile: N = 384 wastes a third of the MFMAs)
  return mt * ntl >= 96;
}

int launch_gemm_big(const GemmDev &d0, int dtype, int epi, hipStream_t s) {
  GemmDev d = d0;
  d.n_tiles = (d.N + XB - 1) / XB;
  d.m_band = 1;             // row-tile major: a 256-row tile's A rows at K >= 2048 are 1 MB and more - four of them do not sit in an L2 (counters: banded +13 % fetch)
  const int64_t mt = (d.M + XB - 1) / XB + (d.group_offsets ? d.G : 0);
  static bool attr_done = false;
  if (!attr_done) {
#define X_ATTR(TT, E) (void)hipFuncSetAttribute((const void *)gemm_nt_big_kernel<TT, E>, hipFuncAttributeMaxDynamicSharedMemorySize, XLDS)
#define X_ATTR_ALL(TT) X_ATTR(TT, DMA_EPI_ANY); X_ATTR(TT, DMA_EPI_GPRE); X_ATTR(TT, DMA_EPI_RES); X_ATTR(TT, DMA_EPI_PLAIN); X_ATTR(TT, DMA_EPI_GELU)
    X_ATTR_ALL(half_t); X_ATTR_ALL(bf16_t);
#undef X_ATTR_ALL
#undef X_ATTR
    attr_done = true;
  }
  const dim3 grid((unsigned)(mt * d.n_tiles)), block(XT);
#define X_GO(TT)                                                                                                     \
  do {                                                                                                               \
    if (epi == DMA_EPI_GPRE) hipLaunchKernelGGL((gemm_nt_big_kernel<TT, DMA_EPI_GPRE>), grid, block, XLDS, s, d);    \
    else if (epi == DMA_EPI_RES) hipLaunchKernelGGL((gemm_nt_big_kernel<TT, DMA_EPI_RES>), grid, block, XLDS, s, d);  \
    else if (epi == DMA_EPI_PLAIN) hipLaunchKernelGGL((gemm_nt_big_kernel<TT, DMA_EPI_PLAIN>), grid, block, XLDS, s, d); \
    else if (epi == DMA_EPI_GELU) hipLaunchKernelGGL((gemm_nt_big_kernel<TT, DMA_EPI_GELU>), grid, block, XLDS, s, d); \
    else hipLaunchKernelGGL((gemm_nt_big_kernel<TT, DMA_EPI_ANY>), grid, block, XLDS, s, d);                         \
  } while (0)
  if (dtype == M3_F16) X_GO(half_t);
  else X_GO(bf16_t);
#undef X_GO
  return check_launch("m3_gemm_nt");
}

}  // namespace m3
