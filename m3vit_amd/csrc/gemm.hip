// Grouped / dense "NT" GEMM for gfx950:  C[m,n] = epi( sum_k A[arow(m),k] * B[g][n,k] ).
//
// Replaces the per-expert cuBLAS loop behind FMoELinear (reference call sites
// models/moe/ckpt/custom_moe_layer.py:32-33,41,43), the row gather/scatter of
// MOEScatter/MOEGather (custom_moe_layer.py:263-265) which is fused into the operand
// load / the store, and the nn.Linear GEMMs of the attention block and the dense Mlp
// (models/moe/ckpt/vision_transformer_moe.py:255-261,295-313).
//
// Structure (one 128x128 output tile per 256-thread workgroup, 2x2 waves of 64x64):
//   - both operands are K-contiguous, staged global -> VGPR -> LDS in 128-byte row
//     slices (BK = 32 f32 / 64 f16), double-buffered, one barrier per K step;
//   - LDS image is XOR-swizzled at 16-byte granularity (chunk ^= (row>>1)&7) so that the
//     ds_read_b128 fragment reads of 16 different rows are bank-conflict free;
//   - MFMA 16x16x32 f16 / 16x16x4 f32 (exact), fp32 accumulate; the weight tile is the
//     MFMA "A" operand so that every lane ends up with 4 consecutive n of one row m
//     and the epilogue uses 8/16-byte vector accesses;
//   - grouped mode: workgroup -> (expert, m-tile) through the device-resident
//     tile_starts prefix (no host sync), rows past the expert's end are zero-filled and
//     never stored;
//   - tile ids are remapped so that the n-tiles of one m-tile run on the same XCD (A rows
//     come from HBM once, then from that XCD's L2).
#include "gemm_dev.h"
#include <type_traits>

namespace m3 {

// Diagnostic build only (make CXXFLAGS+=-DM3_GEMM_STAMPS, tools/gemm_stamps.py): lane 0 of wave 0 of every workgroup of
// the LDS-DMA kernel records s_memtime at its phase boundaries; m3_debug_gemm_stamps copies them out.
#ifdef M3_GEMM_STAMPS
constexpr int STAMP_WGS = 4096, STAMP_N = 64;      // 0 entry, 1 set-up, 2+2ks / 3+2ks K step ks (< 27), 56..58 epilogue, 63 hw id
__device__ unsigned long long g_gemm_stamps[STAMP_WGS][STAMP_N];
#define M3_STAMP(i)                                                                         \
  do {                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x < STAMP_WGS && (i) < STAMP_N)                        \
      g_gemm_stamps[blockIdx.x][(i)] = __builtin_amdgcn_s_memtime();                        \
  } while (0)
#else
#define M3_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// MI: 16-row MFMA tiles per wave along m - 4 (a 128 x 128 tile) or 5 (160 x 128: dense fp32 launches whose 128-row tiles
// leave the chip a ragged last round, e.g. M = 25 216, N = 384: 591 tiles on 256 CUs = 3 per CU for 79 of them, 474 tiles
// of 160 rows = 2 per CU at most; fp32 is MFMA-bound, so the busiest CU's rows set the time: 384 -> 320)
template <typename T, bool KTAIL, int MI = 4>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_kernel(const GemmDev p) {
  constexpr int BMT = 32 * MI;                     // rows of the tile
  typedef Mma<T> MM;
  typedef typename MM::frag frag;
  constexpr int BK = ROWB / (int)sizeof(T);
  constexpr int CHUNKS = ROWB / 64;   // 64-byte fragments groups per row slice = 2

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [buf][A: BMT rows | B: 128 rows] * 128 B

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;

  // ---- which tile
  // grouped: the grid is sized for the upper bound of row tiles; remap only the LIVE workgroups over the XCDs (a remap
  // over the whole grid would park the surplus ids, i.e. no work, on the last XCD)
  int ts_lane = 0;
  const int nwg = p.tile_starts ? grouped_live_tiles(p.tile_starts, p.G, lane, ts_lane) * p.n_tiles : (int)gridDim.x;
  if ((int)blockIdx.x >= nwg) return;
  const int t = xcd_remap(blockIdx.x, nwg);
  int mt, nt;
  tile_of(t, p.n_tiles, p.m_band, nwg / p.n_tiles, mt, nt);
  int g = 0;
  int64_t m_begin, m_end;
  if (p.tile_starts) {
    const TileOwner ow = grouped_tile_owner(p.tile_starts, p.group_offsets, p.G, mt, lane, ts_lane);
    g = ow.g; m_begin = ow.m_begin; m_end = ow.m_end;
  } else {
    m_begin = (int64_t)mt * BMT;
    m_end = p.M;
    if (m_begin >= m_end) return;
  }
  const int n0 = nt * BN;

  // ---- per-thread staging assignment: 4 x 16-byte chunks per operand per step
  // chunk q = tid + 256*i -> row = q >> 3 (0..127), c = q & 7.
  // Rows past the end of the group / of N are CLAMPED to a valid row instead of predicated: an
  // output element depends only on its own A row and B row, and those rows/columns are never
  // stored, so the loads can be unconditional (no branches -> hipcc keeps counted vmcnt waits).
  // Addresses are (wave-uniform 64-bit base that advances with k) + (32-bit per-lane byte offset):
  // the loads take the SGPR-base form and cost no per-step VALU address arithmetic.
  uint32_t a_off[MI], b_off[4];
  const int c_stage = tid & 7;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = (tid >> 3) + 32 * i;
    int64_t m = m_begin + row;
    if (m >= m_end) m = m_end - 1;
    int64_t src = m;
    if (p.a_row_idx) src = (int64_t)div_by(p.a_row_idx[m], p.a_row_div, p.a_row_sh);
    a_off[i] = (uint32_t)(src * p.lda_b) + c_stage * 16;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int n = n0 + (tid >> 3) + 32 * i;
    if (n >= p.N) n = p.N - 1;
    b_off[i] = (uint32_t)((int64_t)n * p.ldb_b) + c_stage * 16;
  }
  const char *a_base = p.A;
  const char *b_base = p.B + (int64_t)g * p.b_group_b;
  const int kbytes = p.K * (int)sizeof(T);
  const int nk = (kbytes + ROWB - 1) / ROWB;

  // LDS addressing: the XOR swizzle term (row>>1)&7 only depends on the lane (tile rows advance in
  // multiples of 16 / 32), so one base per (operand, k-chunk) plus compile-time offsets is enough.
  const int st_base = (tid >> 3) * ROWB + ((c_stage ^ ((tid >> 4) & 7)) << 4);      // + i*4096
  const int sw = (li >> 1) & 7;
  const int rdA0 = (wr * 16 * MI + li) * ROWB + ((lg ^ sw) << 4);                   // + i*2048
  const int rdA1 = (wr * 16 * MI + li) * ROWB + (((4 + lg) ^ sw) << 4);
  const int rdB0 = (wc * 64 + li) * ROWB + ((lg ^ sw) << 4) + BMT * ROWB;
  const int rdB1 = (wc * 64 + li) * ROWB + (((4 + lg) ^ sw) << 4) + BMT * ROWB;

  // Two register sets: tile t+1 waits in one set while tile t+2 is being fetched into the other,
  // so every global load has two compute phases to land (prefetch distance 2).
  u32x4 ra0[MI], rb0[4], ra1[MI], rb1[4];
  auto load_global = [&](int ks, u32x4(&ra)[MI], u32x4(&rb)[4]) {
    int kb = ks * ROWB;
    if (KTAIL && kb + c_stage * 16 >= kbytes) kb = -c_stage * 16;   // K tail: a valid chunk, zeroed at store
    const char *pa = a_base + kb, *pb = b_base + kb;
#pragma unroll
    for (int i = 0; i < MI; ++i) ra[i] = *(const u32x4 *)(pa + a_off[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = *(const u32x4 *)(pb + b_off[i]);
  };
  constexpr int BUFB = (BMT + BN) * ROWB;           // one buffer: both operand images
  auto store_lds = [&](int buf, const u32x4(&ra)[MI], const u32x4(&rb)[4], int ks) {
    const bool kin = !KTAIL || (ks * ROWB + c_stage * 16) < kbytes;
    char *base = smem + buf * BUFB + st_base;
#pragma unroll
    for (int i = 0; i < MI; ++i) *(u32x4 *)(base + i * 32 * ROWB) = kin ? ra[i] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4 *)(base + i * 32 * ROWB + BMT * ROWB) = kin ? rb[i] : u32x4{0u, 0u, 0u, 0u};
  };

  f32x4 acc[4][MI];   // [ni][mi]: rows of the MFMA tile = n, cols = m
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char *sb = smem + buf * BUFB;
#pragma unroll
    for (int kc = 0; kc < CHUNKS; ++kc) {
      frag fa[MI], fb[4];
      const char *pa = sb + (kc ? rdA1 : rdA0), *pb = sb + (kc ? rdB1 : rdB0);
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *(const frag *)(pa + i * 16 * ROWB);
#pragma unroll
      for (int i = 0; i < 4; ++i) fb[i] = *(const frag *)(pb + i * 16 * ROWB);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = MM::mma(fb[ni], fa[mi], acc[ni][mi]);
    }
  };

  load_global(0, ra1, rb1);
  load_global(nk > 1 ? 1 : 0, ra0, rb0);
  store_lds(0, ra1, rb1, 0);
  __syncthreads();

  // Steady state (entry of an even step ks): LDS buf0 holds tile ks, set 0 holds tile ks+1 (in
  // flight).  The loop body only runs while both of its loads are in range, so every load/store is
  // unconditional and hipcc's counted vmcnt waits stay exact; the last 1-3 tiles are peeled.
  int ks = 0;
  for (; ks + 3 < nk; ks += 2) {
    load_global(ks + 2, ra1, rb1);
    __builtin_amdgcn_sched_barrier(0);          // keep the prefetch ahead of the MFMA phase
    compute(0);
    store_lds(1, ra0, rb0, ks + 1);
    __syncthreads();
    load_global(ks + 3, ra0, rb0);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    store_lds(0, ra1, rb1, ks + 2);
    __syncthreads();
  }
  const int rem = nk - ks;
  if (rem == 3) {
    load_global(ks + 2, ra1, rb1);
    __builtin_amdgcn_sched_barrier(0);          // keep the prefetch ahead of the MFMA phase
    compute(0);
    store_lds(1, ra0, rb0, ks + 1);
    __syncthreads();
    compute(1);
    store_lds(0, ra1, rb1, ks + 2);
    __syncthreads();
    compute(0);
  } else if (rem == 2) {
    compute(0);
    store_lds(1, ra0, rb0, ks + 1);
    __syncthreads();
    compute(1);
  } else {
    compute(0);
  }

  // ---- epilogue.  Fast path: the 128x128 fp32 tile is transposed through the (now free) 64 KiB
  // of LDS so that every lane owns 8 consecutive n of one row: bias/residual/pre/C accesses become
  // 16/32-byte vectors and each wave store covers whole 256/512-byte row segments (matters most for
  // the scattered token-major store of the expert FC2 and for the two-output FC1).
  const float *bias = p.bias ? p.bias + (int64_t)g * p.N : nullptr;
  if (p.vec8) {
    // MI = 4: the whole tile at once (64 KiB).  MI = 5: 80 KiB would not fit the operand buffers' 72 KiB - the two 80-row
    // halves (wave rows wr = 0, 1) go through one after the other
    constexpr int NH = MI == 4 ? 1 : 2, HR = BMT / NH;      // passes, rows per pass
    const int cg = tid & 15, r16 = tid >> 4;
    const int n = n0 + cg * 8;
    f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (bias && n < p.N) { b0 = *(const f32x4 *)(bias + n); b1 = *(const f32x4 *)(bias + n + 4); }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      __syncthreads();                       // all waves are done with the operand buffers / the previous half
      if (NH == 1 || wr == h) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const int row = (NH == 1 ? wr * 64 : 0) + mi * 16 + li;
            const int chunk = wc * 16 + ni * 4 + lg;
            *(f32x4 *)(smem + row * 512 + ((chunk ^ (row & 31)) << 4)) = acc[ni][mi];
          }
      }
      __syncthreads();
      if (n < p.N) {
#pragma unroll 2
        for (int ps = 0; ps < HR / 16; ++ps) {
          const int row = ps * 16 + r16;
          const int64_t m = m_begin + h * HR + row;
          if (m >= m_end) break;
          const int64_t crow = p.c_row_idx ? (int64_t)p.c_row_idx[m] : m;
          const int sw = row & 31;
          f32x4 v0 = *(const f32x4 *)(smem + row * 512 + (((2 * cg) ^ sw) << 4));
          f32x4 v1 = *(const f32x4 *)(smem + row * 512 + (((2 * cg + 1) ^ sw) << 4));
          v0 += b0; v1 += b1;
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
            const int64_t srow = p.row_scale_idx ? (int64_t)p.row_scale_idx[m] : crow;
            const float sc = p.row_scale[srow / p.row_scale_div];
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
    return;
  }
  // Generic path (N or a leading dimension not a multiple of 8): lane holds for tile (ni, mi)
  // n = nb + 4*lg + r (r = 0..3), m = mb + li.
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int64_t m = m_begin + wr * 16 * MI + mi * 16 + li;
    if (m >= m_end) continue;
    const int64_t crow = p.c_row_idx ? (int64_t)p.c_row_idx[m] : m;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wc * 64 + ni * 16 + 4 * lg;
      if (n >= p.N) continue;
      f32x4 v = acc[ni][mi];
      if (bias) {
        const f32x4 bv = *(const f32x4 *)(bias + n);
        v += bv;
      }
      if (p.pre_out) Vec4<T>::store((T *)p.pre_out + crow * p.ld_pre + n, v);
      if (p.act == M3_ACT_GELU) {
        v[0] = gelu_f(v[0]); v[1] = gelu_f(v[1]); v[2] = gelu_f(v[2]); v[3] = gelu_f(v[3]);
      }
      if (p.gpre) {
        const f32x4 pr = Vec4<T>::load((const T *)p.gpre + crow * p.ld_gpre + n);
        v[0] *= gelu_grad_f(pr[0]); v[1] *= gelu_grad_f(pr[1]);
        v[2] *= gelu_grad_f(pr[2]); v[3] *= gelu_grad_f(pr[3]);
      }
      if (p.row_scale) v *= p.row_scale[(p.row_scale_idx ? (int64_t)p.row_scale_idx[m] : crow) / p.row_scale_div];
      if (p.residual) v += *(const f32x4 *)(p.residual + crow * p.ld_res + n);
      if (p.c_f32) *(f32x4 *)((float *)p.C + crow * p.ldc + n) = v;
      else Vec4<T>::store((T *)p.C + crow * p.ldc + n, v);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (fp16 fast path: K*elem a multiple of 128 bytes, staged epilogue shapes).
// Same tile / wave layout / epilogue as gemm_nt_kernel, but the operands go global -> LDS directly
// (global_load_lds_dwordx4: one wave instruction fills 8 rows x 128 B = 1 KiB of the image, lane-linear
// in LDS, with the XOR swizzle applied on the per-lane SOURCE address):
//   - no staging registers and no ds_write pass: <= 128 VGPRs and 32 KiB of LDS per workgroup, so FOUR
//     workgroups share a CU instead of two - their load / MFMA / store phases interleave, which is what
//     the K = 384 shapes of this model need (a tile spends more time in its prologue and in the
//     bandwidth-bound store phase than in MFMAs);
//   - ONE buffer of full 128-byte row slices per operand.  (A first version double-buffered 64-byte slices: a
//     64-byte slice uses half of each 128-byte line it touches and the other half is requested again one step
//     later - with four workgroups per CU the line has usually left the 32 KiB L1 by then, so the L2 -> L1
//     path (64 B/clk/CU) carried every operand byte twice and bounded the K loop.  Full-line slices halve
//     that traffic: -15..-20 % per launch at K = 384, -30 % at K = 1536.)  The double buffer is given up for
//     them: 32 KiB keeps four workgroups per CU, and it is the OTHER workgroups' MFMAs, not this one's,
//     that cover a DMA's latency.  Two barriers per K step; __syncthreads() after the DMA is also the
//     vmcnt(0) that retires it (hipcc drains LDS-DMA at a barrier).
constexpr int DMA_RB = 128;                // bytes of a row slice = one cache line
constexpr int DMA_LDS = 2 * BM * DMA_RB;   // A + B image: 32 KiB
constexpr int DMA_LDS_ALL = DMA_LDS + BM * 4;   // + the tile's 128 per-row epilogue factors (row_scale)


// EPI: the epilogue's kind as a template constant.  DMA_EPI_ANY keeps every option behind run-time flags: each
// `if (p.gpre)` / `if (p.residual)` / `if (m >= m_end) break` is then a basic-block boundary, the loads of a store pass are
// issued inside the pass and the eight passes of a tile run strictly one after the other, every one paying its memory
// latency in front of its stores.  The four kinds below cover every 16-bit launch of the training step with straight-line
// passes: what a thread needs from memory for a 64-row half - scatter indices, GELU' pre-activations, residual rows - is
// requested BEFORE that half's staging barriers and arrives under the LDS transposition; the per-row factor is always
// applied (1.0 without row_scale); rows past the group's end repeat the group's last row (the operand rows were clamped at
// the load, so the values are that row's own: a duplicate store of identical data) - except with the fp32 residual, where
// C may alias the residual and the store stays predicated.
//   PLAIN  C = acc (+ bias), optional scatter                      qkv, every plain input gradient, expert FC2 forward
//   GELU   pre_out = acc + bias ; C = GELU(pre_out)                fc1 / expert FC1 forward
//   GPRE   C = acc * GELU'(gpre)                                   fc2 / expert FC2 input gradient
//   RES    C(fp32) = acc (+ bias) + residual                       proj, fc2 forward
// (the kinds' enum: gemm_dev.h)

template <typename T, int EPI = DMA_EPI_ANY>
__global__ __launch_bounds__(GEMM_THREADS, 4) void gemm_nt_dma_kernel(const GemmDev p) {
  typedef Mma<T> MM;
  typedef typename MM::frag frag;
  constexpr int RB = DMA_RB;
  constexpr int OPB = BM * RB;               // one operand image (16 KiB)
  constexpr int CPR = RB / 16;               // 16-byte chunks per row slice
  constexpr int RPI = 64 / CPR;              // image rows filled by one wave instruction (1 KiB)
  constexpr int NPC = BM / RPI / 4;          // DMA pieces per wave per operand per step (4)
  constexpr int KCH = RB / 64;               // fragment groups per slice
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [A|B][128 rows * 128 B] = 32 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  M3_STAMP(0);                                                                         // entry
#ifdef M3_GEMM_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < STAMP_WGS) {                                    // which CU / XCC this workgroup ran on
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_gemm_stamps[blockIdx.x][STAMP_N - 1] = ((unsigned long long)xcc << 32) | hw;
  }
#endif

  int ts_lane = 0;
  const int nwg = p.tile_starts ? grouped_live_tiles(p.tile_starts, p.G, lane, ts_lane) * p.n_tiles : (int)gridDim.x;     // live workgroups (see above)
  if ((int)blockIdx.x >= nwg) return;
  const int t = xcd_remap(blockIdx.x, nwg);
  int mt, nt;
  tile_of(t, p.n_tiles, p.m_band, nwg / p.n_tiles, mt, nt);
  int g = 0;
  int64_t m_begin, m_end;
  if (p.tile_starts) {
    const TileOwner ow = grouped_tile_owner(p.tile_starts, p.group_offsets, p.G, mt, lane, ts_lane);
    g = ow.g; m_begin = ow.m_begin; m_end = ow.m_end;
  } else {
    m_begin = (int64_t)mt * BM;
    m_end = p.M;
    if (m_begin >= m_end) return;
  }
  const int n0 = nt * BN;

  // DMA assignment: wave w, piece j fills image rows (NPC*w + j)*RPI .. +RPI-1; lane l -> row + l / CPR,
  // LDS slot l % CPR, which holds source chunk (l % CPR) ^ swz(row).  Rows past the end are clamped (never stored).
  // Every index load of the prologue (the NPC gathered A rows of this lane; the row of the epilogue factor) is issued
  // before the first one is used: one memory latency in front of the first DMA instead of one per index (an index load
  // inside the per-piece loop, with the divide behind it, becomes its own basic block with its own wait).
  const char *a_src[NPC], *b_src[NPC];
  int64_t mrow[NPC];
  int32_t aix[NPC];
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const int row = (NPC * wave + j) * RPI + lane / CPR;
    int64_t m = m_begin + row;
    if (m >= m_end) m = m_end - 1;
    mrow[j] = m;
  }
  if (p.a_row_idx) {
#pragma unroll
    for (int j = 0; j < NPC; ++j) aix[j] = p.a_row_idx[mrow[j]];
  }
  // per-row epilogue factor (DropPath scale / gate score of the routed row): thread r < 128 requests row r's factor NOW -
  // an index load and a dependent load - so that they arrive under the K loop instead of in front of every store pass
  const bool want_rs = p.row_scale && tid < BM;
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
  for (int j = 0; j < NPC; ++j) {
    const int row = (NPC * wave + j) * RPI + lane / CPR;
    const int c = (lane % CPR) ^ dma_swz(row);
    const int64_t src = p.a_row_idx ? (int64_t)div_by(aix[j], p.a_row_div, p.a_row_sh) : mrow[j];
    a_src[j] = p.A + src * p.lda_b + c * 16;
    int n = n0 + row;
    if (n >= p.N) n = p.N - 1;
    b_src[j] = p.B + (int64_t)g * p.b_group_b + (int64_t)n * p.ldb_b + c * 16;
  }
  const int nk = (p.K * (int)sizeof(T)) / RB;

  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  auto dma = [&](int ks) {
    char *dst = smem + (NPC * wave) * 1024;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      __builtin_amdgcn_global_load_lds((glb_void *)(a_src[j] + ks * RB), (lds_void *)(dst + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void *)(b_src[j] + ks * RB), (lds_void *)(dst + j * 1024 + OPB), 16, 0, 0);
    }
  };

  int rdA[KCH], rdB[KCH];
#pragma unroll
  for (int kc = 0; kc < KCH; ++kc) {
    rdA[kc] = (wr * 64 + li) * RB + (((kc * 4 + lg) ^ dma_swz(li)) << 4);               // + i*16*RB
    rdB[kc] = (wc * 64 + li) * RB + (((kc * 4 + lg) ^ dma_swz(li)) << 4) + OPB;
  }

  f32x4 acc[4][4];   // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&]() {
    const char *sb = smem;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) {
      frag fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *(const frag *)(sb + rdA[kc] + i * 16 * RB);
        fb[i] = *(const frag *)(sb + rdB[kc] + i * 16 * RB);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = MM::mma(fb[ni], fa[mi], acc[ni][mi]);
    }
  };

  M3_STAMP(1);                                                                         // set-up done
  for (int ks = 0; ks < nk; ++ks) {
    dma(ks);
    __syncthreads();          // vmcnt(0) + barrier: the slice has landed
    if (ks < 27) M3_STAMP(2 + 2 * ks);                                                 // slice ks landed
    compute();
    __syncthreads();          // everyone has read it
    if (ks < 27) M3_STAMP(3 + 2 * ks);                                                 // slice ks multiplied
  }

  // ---- epilogue: the fp32 tile goes through the (now free) 32 KiB in two 64-row halves (half h = waves wr == h),
  // every lane then owns 8 consecutive n of one row
  const float *bias = p.bias ? p.bias + (int64_t)g * p.N : nullptr;
  const int cg = tid & 15, r16 = tid >> 4;
  const int n = n0 + cg * 8;
  f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (bias && n < p.N) { b0 = *(const f32x4 *)(bias + n); b1 = *(const f32x4 *)(bias + n + 4); }
  float *const s_rs = (float *)(smem + DMA_LDS);         // (behind the operand images: written once, read after the barriers below)
  if (tid < BM) s_rs[tid] = my_rs;                       // 1.0 without row_scale
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    // specialised epilogues: this half's memory operands, requested ahead of the staging barriers (unconditional loads:
    // rows past the end are clamped)
    u32x4 gq[4];
    f32x4 rq[4][2];
    int32_t crow4[4];                            // (rows fit 32 bits: the host checks the 4 GiB reach of an operand panel)
    // (fp32 residual rows are 8 registers a pass: two passes are requested here, two behind the barriers, while
    // this wave's accumulators are on their way out - all four at once did not fit the 128-register budget)
    constexpr int NPRE = EPI == DMA_EPI_RES ? 2 : 4;
    auto fetch_epi = [&](int ps) {
      if constexpr (EPI == DMA_EPI_GPRE) gq[ps] = *(const u32x4 *)((const T *)p.gpre + (int64_t)crow4[ps] * p.ld_gpre + n);
      if constexpr (EPI == DMA_EPI_RES) {
        rq[ps][0] = *(const f32x4 *)(p.residual + (int64_t)crow4[ps] * p.ld_res + n);
        rq[ps][1] = *(const f32x4 *)(p.residual + (int64_t)crow4[ps] * p.ld_res + n + 4);
      }
    };
    if constexpr (EPI != DMA_EPI_ANY) {
      if (n < p.N) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          int64_t m = m_begin + h * 64 + ps * 16 + r16;
          if (m >= m_end) m = m_end - 1;
          crow4[ps] = p.c_row_idx ? p.c_row_idx[m] : (int32_t)m;
        }
#pragma unroll
        for (int ps = 0; ps < NPRE; ++ps) fetch_epi(ps);
      }
    }
    if (h) __syncthreads();
    if (wr == h) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int lrow = mi * 16 + li;
          const int chunk = wc * 16 + ni * 4 + lg;
          *(f32x4 *)(smem + lrow * 512 + ((chunk ^ (lrow & 31)) << 4)) = acc[ni][mi];
        }
    }
    __syncthreads();
    if constexpr (EPI != DMA_EPI_ANY) {
      if (n < p.N) {
#pragma unroll
        for (int ps = NPRE; ps < 4; ++ps) fetch_epi(ps);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int lrow = ps * 16 + r16;
          const int64_t crow = crow4[ps];
          const int sw = lrow & 31;
          f32x4 v0 = *(const f32x4 *)(smem + lrow * 512 + (((2 * cg) ^ sw) << 4));
          f32x4 v1 = *(const f32x4 *)(smem + lrow * 512 + (((2 * cg + 1) ^ sw) << 4));
          v0 += b0; v1 += b1;
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
          const float sc = s_rs[h * 64 + lrow];
          v0 *= sc; v1 *= sc;
          if constexpr (EPI == DMA_EPI_RES) {
            v0 += rq[ps][0]; v1 += rq[ps][1];
            if (m_begin + h * 64 + lrow < m_end) {          // (C may be the residual buffer: no duplicate read-modify-write)
              *(f32x4 *)((float *)p.C + crow * p.ldc + n) = v0;
              *(f32x4 *)((float *)p.C + crow * p.ldc + n + 4) = v1;
            }
          } else {
            Vec8<T>::store((T *)p.C + crow * p.ldc + n, v0, v1);
          }
        }
      }
    } else if (n < p.N) {
#pragma unroll 2
      for (int ps = 0; ps < 4; ++ps) {
        const int lrow = ps * 16 + r16;
        const int64_t m = m_begin + h * 64 + lrow;
        if (m >= m_end) break;
        const int64_t crow = p.c_row_idx ? (int64_t)p.c_row_idx[m] : m;
        const int sw = lrow & 31;
        f32x4 v0 = *(const f32x4 *)(smem + lrow * 512 + (((2 * cg) ^ sw) << 4));
        f32x4 v1 = *(const f32x4 *)(smem + lrow * 512 + (((2 * cg + 1) ^ sw) << 4));
        v0 += b0; v1 += b1;
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
          const float sc = s_rs[h * 64 + lrow];
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
    M3_STAMP(56 + h);                                                                  // stores of half h issued
  }
#ifdef M3_GEMM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  M3_STAMP(58);                                                                        // my stores acknowledged
#endif
}


}  // namespace m3

using namespace m3;

// weight-stationary variant: mask of the epilogues it may take (bit 0 plain, 1 GELU + pre-activation, 2 GELU' multiply,
// 3 fp32 residual, bit 4: grouped calls too).  Off by default (see the kernel's header); -1 = take M3_GEMM_WS from the
// environment at the first call.
static int g_ws_mode = -1;
extern "C" int m3_gemm_set_variant(int ws_mask) {
  M3_REQUIRE(ws_mask >= -1 && ws_mask < 32, "m3_gemm_set_variant: mask %d out of range", ws_mask);
#ifndef M3_EXPERIMENTAL
  M3_REQUIRE(ws_mask <= 0, "m3_gemm_set_variant: the weight-stationary kernel is only in EXPERIMENTAL builds (make EXPERIMENTAL=1)");
#endif
  g_ws_mode = ws_mask;
  return M3_OK;
}

static int g_big_mode = -1;
extern "C" int m3_gemm_set_big(int mode) {
  M3_REQUIRE(mode >= -1 && mode <= 2, "m3_gemm_set_big: mode %d out of range", mode);
  g_big_mode = mode;
  return M3_OK;
}

extern "C" int m3_gemm_nt(const m3_gemm_args *a, void *stream) {
  M3_REQUIRE(a && a->A && a->B && a->C, "m3_gemm_nt: null operand");
  M3_REQUIRE(dtype_ok(a->dtype), "m3_gemm_nt: bad dtype %d", a->dtype);
  const int es = dtype_size(a->dtype);
  M3_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, "m3_gemm_nt: bad shape M=%lld N=%d K=%d", (long long)a->M, a->N, a->K);
  M3_REQUIRE((a->K * es) % 16 == 0, "m3_gemm_nt: K*elem (%d) must be a multiple of 16 bytes", a->K * es);
  M3_REQUIRE((a->lda * es) % 16 == 0 && (a->ldb * es) % 16 == 0, "m3_gemm_nt: lda/ldb rows must be 16-byte aligned");
  M3_REQUIRE(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->B % 16) == 0 && ((uintptr_t)a->C % 16) == 0,
             "m3_gemm_nt: operands must be 16-byte aligned");
  M3_REQUIRE(a->N % 4 == 0 && a->ldc % 4 == 0, "m3_gemm_nt: N and ldc must be multiples of 4");
  M3_REQUIRE(a->lda >= a->K && a->ldb >= a->K && a->ldc >= a->N, "m3_gemm_nt: leading dimension too small");
  M3_REQUIRE(a->c_dtype == M3_F32 || a->c_dtype == a->dtype, "m3_gemm_nt: c_dtype must be f32 or the operand dtype");
  M3_REQUIRE(a->G >= 1, "m3_gemm_nt: G must be >= 1");
  M3_REQUIRE((a->group_offsets == nullptr) == (a->tile_starts == nullptr), "m3_gemm_nt: group_offsets/tile_starts go together");
  M3_REQUIRE(a->G == 1 || a->group_offsets, "m3_gemm_nt: grouped call needs group_offsets");
  M3_REQUIRE(!a->a_row_idx || a->a_row_div >= 1, "m3_gemm_nt: a_row_div must be >= 1");
  M3_REQUIRE(!a->pre_out || a->ld_pre % 4 == 0, "m3_gemm_nt: ld_pre % 4");
  M3_REQUIRE(!a->gelu_grad_pre || a->ld_gpre % 4 == 0, "m3_gemm_nt: ld_gpre % 4");
  M3_REQUIRE(!a->residual || a->ld_res % 4 == 0, "m3_gemm_nt: ld_res % 4");
  M3_REQUIRE(!a->row_scale || a->row_scale_div >= 1, "m3_gemm_nt: row_scale_div must be >= 1");
  if (a->M == 0) return M3_OK;

  GemmDev d;
  d.A = (const char *)a->A; d.lda_b = a->lda * es;
  d.a_row_idx = a->a_row_idx; d.a_row_div = a->a_row_idx ? a->a_row_div : 1;
  d.a_row_sh = div_shift(d.a_row_div);
  d.B = (const char *)a->B; d.ldb_b = a->ldb * es; d.b_group_b = (int64_t)a->N * a->ldb * es;
  d.C = (char *)a->C; d.ldc = a->ldc; d.c_f32 = (a->c_dtype == M3_F32) ? 1 : 0;
  d.c_row_idx = a->c_row_idx;
  d.bias = a->bias;
  d.pre_out = (char *)a->pre_out; d.ld_pre = a->ld_pre;
  d.gpre = (const char *)a->gelu_grad_pre; d.ld_gpre = a->ld_gpre;
  d.residual = a->residual; d.ld_res = a->ld_res;
  d.row_scale = a->row_scale; d.row_scale_div = a->row_scale ? a->row_scale_div : 1;
  d.row_scale_idx = a->row_scale ? a->row_scale_idx : nullptr;
  d.act = a->act;
  d.M = a->M; d.N = a->N; d.K = a->K; d.G = a->G;
  d.group_offsets = a->group_offsets; d.tile_starts = a->tile_starts;
  d.n_tiles = (a->N + BN - 1) / BN;
  // tile order (gemm_dev.h: tile_of): bands of four row tiles when a group's weight does not fit an XCD's L2 beside the rows
  // (> 2 MB: the ViT-Base N = 2304 / 3072 launches and K = 3072); M3_GEMM_BAND=n forces n (1 = row-tile major everywhere)
  static int band_env = -1;
  if (band_env < 0) { const char *e = getenv("M3_GEMM_BAND"); band_env = e ? atoi(e) : 0; }
  d.m_band = band_env > 0 ? band_env : ((int64_t)a->N * a->K * es > ((int64_t)2 << 20) ? 4 : 1);
  const int64_t mt = (a->M + BM - 1) / BM + (a->group_offsets ? a->G : 0);
  M3_REQUIRE(mt * d.n_tiles < (int64_t)1 << 30, "m3_gemm_nt: grid too large");
  d.m_tiles_max = (int)mt;
  d.vec8 = (a->N % 8 == 0 && a->ldc % 8 == 0 && (!a->pre_out || a->ld_pre % 8 == 0) &&
            (!a->gelu_grad_pre || a->ld_gpre % 8 == 0) && (!a->residual || a->ld_res % 8 == 0)) ? 1 : 0;
  const dim3 grid((unsigned)(mt * d.n_tiles)), block(GEMM_THREADS);
  const size_t lds = 4 * BM * ROWB;  // 64 KiB
  hipStream_t s = (hipStream_t)stream;
  // 32-bit per-lane byte offsets: A rows (gathered source rows must be < M) and one B group must fit 4 GiB
  M3_REQUIRE((a->M + 1) * a->lda * es < ((int64_t)1 << 32) && (int64_t)a->N * a->ldb * es < ((int64_t)1 << 32),
             "m3_gemm_nt: operand panel exceeds the 4 GiB reach of the 32-bit lane offsets");
#ifdef M3_EXPERIMENTAL
  // weight-stationary persistent variant (gemm_ws.hip; opt-in: m3_gemm_set_variant / M3_GEMM_WS; EXPERIMENTAL builds only)
  if (g_ws_mode < 0) { const char *e = getenv("M3_GEMM_WS"); g_ws_mode = e ? atoi(e) : 0; }
  if (g_ws_mode) {
    const int rc = launch_gemm_ws(d, a, mt, g_ws_mode, s);
    if (rc <= 0) return rc;                     // 0: launched, < 0: error, 1: not a call it takes
  }
#endif
  // epilogue kinds (16-bit dtypes; anything else takes the generic epilogue)
  int epi = DMA_EPI_ANY;
  static int epi_mode = -1;                    // M3_GEMM_EPI=0: generic epilogue everywhere (diagnostics)
  if (epi_mode < 0) { const char *e = getenv("M3_GEMM_EPI"); epi_mode = e ? atoi(e) : 1; }
  if (epi_mode && es == 2) {
    const bool none = a->act == M3_ACT_NONE && !a->pre_out;
    if (none && a->gelu_grad_pre && !a->residual && !d.c_f32 && !a->bias) epi = DMA_EPI_GPRE;
    else if (none && a->residual && !a->gelu_grad_pre && d.c_f32) epi = DMA_EPI_RES;
    else if (none && !a->gelu_grad_pre && !a->residual && !d.c_f32) epi = DMA_EPI_PLAIN;
    else if (a->act == M3_ACT_GELU && a->pre_out && !a->gelu_grad_pre && !a->residual && !d.c_f32) epi = DMA_EPI_GELU;
  }
  // long contractions (the ViT-Base shapes): 256 x 256 tiles, gemm_big.hip.  m3_gemm_set_big / M3_GEMM_BIG: 0 never,
  // 1 whenever the kernel can run the shape, 2 (default) when the shape also has enough tiles to fill the chip twice
  if (g_big_mode < 0) { const char *e = getenv("M3_GEMM_BIG"); g_big_mode = e ? atoi(e) : 2; }
  if (g_big_mode && gemm_big_eligible(d, es, g_big_mode == 1)) return launch_gemm_big(d, a->dtype, epi, s);
  // variant: fp16 -> LDS-DMA kernel, fp32 (MFMA-bound, measured 2 % slower there) and odd shapes ->
  // register-staged kernel; M3_GEMM_DMA=1/0 forces one or the other (diagnostics)
  static int dma_mode = -1;
  if (dma_mode < 0) { const char *e = getenv("M3_GEMM_DMA"); dma_mode = e ? (atoi(e) ? 1 : 0) : 2; }
  const bool dma_ok = d.vec8 && (a->K * es) % DMA_RB == 0;
  if (dma_ok && (dma_mode == 1 || (dma_mode == 2 && es == 2))) {
#define M3_DMA_GO(TT)                                                                                                  \
    do {                                                                                                               \
      if (epi == DMA_EPI_GPRE) hipLaunchKernelGGL((gemm_nt_dma_kernel<TT, DMA_EPI_GPRE>), grid, block, DMA_LDS_ALL, s, d);  \
      else if (epi == DMA_EPI_RES) hipLaunchKernelGGL((gemm_nt_dma_kernel<TT, DMA_EPI_RES>), grid, block, DMA_LDS_ALL, s, d); \
      else if (epi == DMA_EPI_PLAIN) hipLaunchKernelGGL((gemm_nt_dma_kernel<TT, DMA_EPI_PLAIN>), grid, block, DMA_LDS_ALL, s, d); \
      else if (epi == DMA_EPI_GELU) hipLaunchKernelGGL((gemm_nt_dma_kernel<TT, DMA_EPI_GELU>), grid, block, DMA_LDS_ALL, s, d); \
      else hipLaunchKernelGGL((gemm_nt_dma_kernel<TT, DMA_EPI_ANY>), grid, block, DMA_LDS_ALL, s, d);                  \
    } while (0)
    if (a->dtype == M3_F16) M3_DMA_GO(half_t);
    else if (a->dtype == M3_BF16) M3_DMA_GO(bf16_t);
    else hipLaunchKernelGGL((gemm_nt_dma_kernel<float, DMA_EPI_ANY>), grid, block, DMA_LDS_ALL, s, d);
#undef M3_DMA_GO
    return check_launch("m3_gemm_nt");
  }
  const bool ktail = (a->K * es) % ROWB != 0;
  // fp32, dense: 160-row tiles where they even out the last round (gemm_nt_kernel's MI = 5).  The busiest CU's share of the
  // rows - ceil(tiles / 256 CUs) x tile rows - decides an MFMA-bound launch; M3_GEMM_F32_TALL=0 switches it off
  if (a->dtype == M3_F32 && !ktail && !a->group_offsets) {
    static int tall = -1;
    if (tall < 0) { const char *e = getenv("M3_GEMM_F32_TALL"); tall = e ? atoi(e) : 1; }
    const int64_t t128 = (a->M + 127) / 128 * d.n_tiles, t160 = (a->M + 159) / 160 * d.n_tiles;
    const int64_t cost128 = (t128 + 255) / 256 * 128, cost160 = (t160 + 255) / 256 * 160;
    if (tall && cost160 < cost128) {
      const size_t lds5 = 2 * (160 + BN) * ROWB;          // 72 KiB: two workgroups per CU
      static bool attr5 = false;
      if (!attr5) {
        (void)hipFuncSetAttribute((const void *)gemm_nt_kernel<float, false, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5);
        attr5 = true;
      }
      d.m_tiles_max = (int)((a->M + 159) / 160);
      hipLaunchKernelGGL((gemm_nt_kernel<float, false, 5>), dim3((unsigned)t160), block, lds5, s, d);
      return check_launch("m3_gemm_nt");
    }
  }
  if (a->dtype == M3_F16) {
    if (ktail) hipLaunchKernelGGL((gemm_nt_kernel<half_t, true>), grid, block, lds, s, d);
    else hipLaunchKernelGGL((gemm_nt_kernel<half_t, false>), grid, block, lds, s, d);
  } else if (a->dtype == M3_BF16) {
    if (ktail) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, true>), grid, block, lds, s, d);
    else hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, false>), grid, block, lds, s, d);
  } else {
    if (ktail) hipLaunchKernelGGL((gemm_nt_kernel<float, true>), grid, block, lds, s, d);
    else hipLaunchKernelGGL((gemm_nt_kernel<float, false>), grid, block, lds, s, d);
  }
  return check_launch("m3_gemm_nt");
}

#ifdef M3_GEMM_STAMPS
extern "C" int m3_debug_gemm_stamps(unsigned long long *dst, int wgs) {
  if (wgs > STAMP_WGS) wgs = STAMP_WGS;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gemm_stamps), (size_t)wgs * STAMP_N * sizeof(unsigned long long)) == hipSuccess ? M3_OK : M3_ERR_LAUNCH;
}
#endif
