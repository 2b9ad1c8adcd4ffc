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

template <typename T, bool KTAIL>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_kernel(const GemmDev p) {
  typedef Mma<T> MM;
  typedef typename MM::frag frag;
  constexpr int BK = ROWB / (int)sizeof(T);
  constexpr int CHUNKS = ROWB / 64;   // 64-byte fragments groups per row slice = 2

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [buf][A|B][128 rows * 128 B]

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
  const int mt = t / p.n_tiles, nt = t - mt * p.n_tiles;
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

  // ---- per-thread staging assignment: 4 x 16-byte chunks per operand per step
  // chunk q = tid + 256*i -> row = q >> 3 (0..127), c = q & 7.
  // Rows past the end of the group / of N are CLAMPED to a valid row instead of predicated: an
  // output element depends only on its own A row and B row, and those rows/columns are never
  // stored, so the loads can be unconditional (no branches -> hipcc keeps counted vmcnt waits).
  // Addresses are (wave-uniform 64-bit base that advances with k) + (32-bit per-lane byte offset):
  // the loads take the SGPR-base form and cost no per-step VALU address arithmetic.
  uint32_t a_off[4], b_off[4];
  const int c_stage = tid & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid >> 3) + 32 * i;
    int64_t m = m_begin + row;
    if (m >= m_end) m = m_end - 1;
    int64_t src = m;
    if (p.a_row_idx) src = (int64_t)div_by(p.a_row_idx[m], p.a_row_div, p.a_row_sh);
    a_off[i] = (uint32_t)(src * p.lda_b) + c_stage * 16;
    int n = n0 + row;
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
  const int rdA0 = (wr * 64 + li) * ROWB + ((lg ^ sw) << 4);                        // + i*2048
  const int rdA1 = (wr * 64 + li) * ROWB + (((4 + lg) ^ sw) << 4);
  const int rdB0 = (wc * 64 + li) * ROWB + ((lg ^ sw) << 4) + BM * ROWB;
  const int rdB1 = (wc * 64 + li) * ROWB + (((4 + lg) ^ sw) << 4) + BM * ROWB;

  // Two register sets: tile t+1 waits in one set while tile t+2 is being fetched into the other,
  // so every global load has two compute phases to land (prefetch distance 2).
  u32x4 ra0[4], rb0[4], ra1[4], rb1[4];
  auto load_global = [&](int ks, u32x4(&ra)[4], u32x4(&rb)[4]) {
    int kb = ks * ROWB;
    if (KTAIL && kb + c_stage * 16 >= kbytes) kb = -c_stage * 16;   // K tail: a valid chunk, zeroed at store
    const char *pa = a_base + kb, *pb = b_base + kb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *(const u32x4 *)(pa + a_off[i]);
      rb[i] = *(const u32x4 *)(pb + b_off[i]);
    }
  };
  auto store_lds = [&](int buf, const u32x4(&ra)[4], const u32x4(&rb)[4], int ks) {
    const bool kin = !KTAIL || (ks * ROWB + c_stage * 16) < kbytes;
    char *base = smem + buf * (2 * BM * ROWB) + st_base;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(u32x4 *)(base + i * 32 * ROWB) = kin ? ra[i] : u32x4{0u, 0u, 0u, 0u};
      *(u32x4 *)(base + i * 32 * ROWB + BM * ROWB) = kin ? rb[i] : u32x4{0u, 0u, 0u, 0u};
    }
  };

  f32x4 acc[4][4];   // [ni][mi]: rows of the MFMA tile = n, cols = m
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char *sb = smem + buf * (2 * BM * ROWB);
#pragma unroll
    for (int kc = 0; kc < CHUNKS; ++kc) {
      frag fa[4], fb[4];
      const char *pa = sb + (kc ? rdA1 : rdA0), *pb = sb + (kc ? rdB1 : rdB0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *(const frag *)(pa + i * 16 * ROWB);
        fb[i] = *(const frag *)(pb + i * 16 * ROWB);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = MM::mma(fb[ni], fa[mi], acc[ni][mi]);
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
    __syncthreads();                         // all waves are done with the operand buffers
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int row = wr * 64 + mi * 16 + li;
        const int chunk = wc * 16 + ni * 4 + lg;
        *(f32x4 *)(smem + row * 512 + ((chunk ^ (row & 31)) << 4)) = acc[ni][mi];
      }
    __syncthreads();
    const int cg = tid & 15, r16 = tid >> 4;
    const int n = n0 + cg * 8;
    if (n < p.N) {
      f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
      if (bias) { b0 = *(const f32x4 *)(bias + n); b1 = *(const f32x4 *)(bias + n + 4); }
#pragma unroll 2
      for (int ps = 0; ps < 8; ++ps) {
        const int row = ps * 16 + r16;
        const int64_t m = m_begin + row;
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
    return;
  }
  // Generic path (N or a leading dimension not a multiple of 8): lane holds for tile (ni, mi)
  // n = nb + 4*lg + r (r = 0..3), m = mb + li.
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int64_t m = m_begin + wr * 64 + mi * 16 + li;
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
  const int mt = t / p.n_tiles, nt = t - mt * p.n_tiles;
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


// ------------------------------------------------------------------------------------------------
// Weight-stationary, wave-specialised persistent variant (fp16, K = 384, N a multiple of 128: qkv, proj, fc1, the
// expert FC1 and the GELU'-fused input-gradient GEMMs).
// The tiled kernels above move both operands through LDS for every 128 x 128 tile and expose one prologue and one
// epilogue per tile.  Here a workgroup owns a run of consecutive row tiles of (mostly) ONE 128-wide column tile and
//   - waves 0-3 (one per SIMD) keep that column tile's WEIGHTS in registers as MFMA A fragments (wave w: columns
//     32 w .. 32 w + 31, 2 x 12 fragments = 96 VGPRs) and only touch LDS to read activation fragments and to hand a
//     finished fp32 accumulator tile over;
//   - waves 4-7 (again one per SIMD) are helpers with two duties per step: they stream the activation rows - the only
//     operand that moves - as 16 KiB slices (128 rows x 128 B, XOR swizzle on the source address, expert gather
//     fused) through registers, a whole tile (six slices) ahead of the MFMAs, into a two-slot ring; and they run the
//     PREVIOUS tile's epilogue (bias, GELU, pre-activation output, GELU', fp32 residual) out of the 64 KiB staging
//     image, an eighth of the tile between two step barriers, so neither load issue, nor store issue, nor the
//     epilogue's VALU work ever sits in front of an MFMA.
// The epilogue is a template parameter: a run-time `if (p.gpre)` around a load makes the compiler wait vmcnt(0) at the
// join - i.e. for the previous pass's STORE to be acknowledged (~1000 cycles a pass; measured with the barrier-arrival
// stamps below: the store role arrived last at 88 % of the barriers).  vmcnt is in-order, so the operands an
// epilogue reads from memory (GELU' pre-activations, residual rows) are requested for the whole tile in one batch at
// the head of the round, in front of that round's stores.
// Gathered row indices go through LDS: MFMA wave 0 (no stores in flight) loads the indices of tile i + 2 and leaves
// them in a 2 x 512 B table, so the helpers never wait on a load that sits behind their own stores.
// Work split: XCD x (blockIdx % 8) owns the row-tile band [x MT / 8, (x + 1) MT / 8) for all column tiles - its A
// rows stay in that XCD's L2 - and its workgroups split the band's tiles, column-major, into equal runs (a run that
// crosses a column or expert boundary reloads the weight fragments).
constexpr int WS_THREADS = 512;
constexpr int WS_NS = 3;                           // ring slots: slice q (and the head of q + 1) is read while slice q + 2 is written
constexpr int WS_SLICE = 16384;                    // 128 rows x 128 B
constexpr int WS_RING = WS_NS * WS_SLICE;          // 48 KiB
constexpr int WS_STAGE = BM * BN * 4;              // 64 KiB fp32 tile
constexpr int WS_IDX = 2 * BM * 4;                 // two tables of 128 source rows
constexpr int WS_TAB = 1024;                        // 64 tiles x {g, nt, m_begin, m_end}
constexpr int WS_LDS = WS_RING + WS_STAGE + WS_IDX + WS_TAB;
constexpr int WS_KS = 6;                           // K = 384: six 64-deep slices
enum { WS_EPI_PLAIN = 0, WS_EPI_GELU = 1, WS_EPI_GPRE = 2, WS_EPI_RES = 3 };

// Diagnostic build (-DM3_GEMM_STAMPS, tools/gemm_ws_stamps.py): lane 0 of the first wave of each role records s_memtime
// when it ARRIVES at each of its first 64 barriers; the role that arrives last at a barrier is the one the step waited for.
#ifdef M3_GEMM_STAMPS
constexpr int WS_STAMP_WGS = 256;
__device__ unsigned long long g_ws_stamps[WS_STAMP_WGS][2][64];
#define WS_ARRIVE(role)                                                                       \
  do {                                                                                        \
    if (lane == 0 && (wave & 3) == 0 && blockIdx.x < WS_STAMP_WGS && nbar < 64)               \
      g_ws_stamps[blockIdx.x][(role)][nbar] = __builtin_amdgcn_s_memtime();                   \
    ++nbar;                                                                                   \
  } while (0)
#else
#define WS_ARRIVE(role) do { } while (0)
#endif

struct WsTile { int g, nt; int64_t m_begin, m_end; };

// The tiles of a workgroup's run (band tiles are numbered column-major: all row tiles of column 0, then column 1 ...)
// are worked out once, by one thread per tile, into an LDS table {group, column tile, first row, end row}: the roles
// then look a tile up with one ds_read instead of walking tile_starts with vector loads that would sit in the vmcnt
// queue among the helpers' stores.
constexpr int WS_MAXT = 64;                        // longest run (the host checks it)
__device__ __forceinline__ WsTile ws_tile_at(const int *tab, int i) {
  const i32x4 v = *(const i32x4 *)(tab + 4 * i);
  WsTile t;
  t.g = __builtin_amdgcn_readfirstlane(v[0]); t.nt = __builtin_amdgcn_readfirstlane(v[1]);
  t.m_begin = __builtin_amdgcn_readfirstlane(v[2]); t.m_end = __builtin_amdgcn_readfirstlane(v[3]);
  return t;
}

template <int EPI>
__global__ __launch_bounds__(WS_THREADS, 2) void gemm_nt_ws_kernel(const GemmDev p) {
  typedef Mma<half_t> MM;
  typedef MM::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *const ring = smem;
  char *const stage = smem + WS_RING;
  int *const idx_tab = (int *)(smem + WS_RING + WS_STAGE);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;

  // ---- this workgroup's run of tiles
  const int MT = p.tile_starts ? p.tile_starts[p.G] : (int)((p.M + BM - 1) / BM);
  const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3, nkx = (int)gridDim.x >> 3;
  const int band0 = (int)((int64_t)xcd * MT / 8);
  const int band_len = (int)((int64_t)(xcd + 1) * MT / 8) - band0;
  if (band_len <= 0) return;                       // (uniform: no barrier has been executed)
  const int tb = band_len * p.n_tiles;
  const int idx0 = (int)((int64_t)kx * tb / nkx);
  const int ntl = (int)((int64_t)(kx + 1) * tb / nkx) - idx0;
  if (ntl <= 0) return;
  int *const tile_tab = (int *)(smem + WS_RING + WS_STAGE + WS_IDX);
  if (tid < ntl) {
    const int idx = idx0 + tid;
    const int nt = idx / band_len, mt = band0 + idx - nt * band_len;
    i32x4 e;
    if (p.tile_starts) {
      int g = 0;
      while (g + 1 < p.G && p.tile_starts[g + 1] <= mt) ++g;
      e = i32x4{g, nt, p.group_offsets[g] + (mt - p.tile_starts[g]) * BM, p.group_offsets[g + 1]};
    } else {
      e = i32x4{0, nt, mt * BM, (int)p.M};
    }
    *(i32x4 *)(tile_tab + 4 * tid) = e;
  }
  __syncthreads();
  int nbar = 0; (void)nbar;

  if (wave < 4) {
    // =============================================================== MFMA waves
    frag bf[2][2 * WS_KS];
    f32x4 bv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};      // this lane's bias columns: the accumulators start from them
    int key_cur = -1;
    const int swl = dma_swz(li);
    int rdA[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) rdA[kc] = li * 128 + (((kc * 4 + lg) ^ swl) << 4);      // + m8 * 16 * 128
    int iv0 = 0, iv1 = 0;
    frag af0[8], af1[8];                           // activation fragments: first / second 32-deep half of a slice
    bool primed = false;
#pragma nounroll
    for (int it = 0; it <= ntl; ++it) {
      const bool live = it < ntl;
      f32x4 acc[8][2];
      if (live) {
        const WsTile t = ws_tile_at(tile_tab, it);
        const int key = t.g * p.n_tiles + t.nt;
        if (key != key_cur) {                      // (re)load the weight fragments of this column tile / expert
          key_cur = key;
          const char *wb = p.B + (int64_t)t.g * p.b_group_b;
#pragma unroll
          for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int s = 0; s < 2 * WS_KS; ++s)
              bf[n2][s] = *(const frag *)(wb + (int64_t)(t.nt * BN + 32 * wave + 16 * n2 + li) * p.ldb_b + s * 64 + lg * 16);
          if (p.bias) {
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
              bv[n2] = *(const f32x4 *)(p.bias + (int64_t)t.g * p.N + t.nt * BN + 32 * wave + 16 * n2 + 4 * lg);
          }
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) { acc[a][0] = bv[0]; acc[a][1] = bv[1]; }
      }
      const bool idx_job = p.a_row_idx && wave == 0 && it + 2 < ntl;
#pragma unroll
      for (int s = 0; s < WS_KS; ++s) {
        WS_ARRIVE(0); __builtin_amdgcn_s_barrier();              // B(q): slice q is in its ring slot
        asm volatile("" ::: "memory");
        if (s == 1 && idx_job) {                   // (behind the step-0 MFMAs: their first use of the weights waits vmcnt(0))
          const WsTile t2 = ws_tile_at(tile_tab, it + 2);
          int64_t m0 = t2.m_begin + lane, m1 = m0 + 64;
          if (m0 >= t2.m_end) m0 = t2.m_end - 1;
          if (m1 >= t2.m_end) m1 = t2.m_end - 1;
          iv0 = p.a_row_idx[m0]; iv1 = p.a_row_idx[m1];
        }
        if (live) {
          // Slice q is multiplied in two 32-deep halves.  The fragments of its FIRST half were requested in the middle
          // of the previous step (the ring is written two steps ahead, so slice q was complete one barrier ago); the
          // second half is requested now and arrives under the first half's 16 MFMAs; then the next slice's first
          // half is requested under the second half's MFMAs.  No MFMA ever waits on a read issued behind its barrier.
          const char *sb = ring + (s % WS_NS) * WS_SLICE;           // slice q lives in slot q % 3 = s % 3 (six slices a tile)
          const char *sbn = ring + ((s + 1) % WS_NS) * WS_SLICE;
#if defined(WS_ABL_NOMFMA)                         // diagnostic builds: one ingredient of a step removed (results are wrong)
#define WS_MMA(a_, b_, c_) ((c_) + f32x4{(float)(b_)[0], 0.f, 0.f, 0.f})
#else
#define WS_MMA(a_, b_, c_) MM::mma(a_, b_, c_)
#endif
#ifdef WS_ABL_NOREAD
#define WS_RD(ptr_) bf[0][0]
#else
#define WS_RD(ptr_) (*(const frag *)(ptr_))
#endif
          if (!primed) {                            // first slice of the run
#pragma unroll
            for (int m8 = 0; m8 < 8; ++m8) af0[m8] = WS_RD(sb + rdA[0] + m8 * 16 * 128);
            primed = true;
          }
#pragma unroll
          for (int m8 = 0; m8 < 8; ++m8) af1[m8] = WS_RD(sb + rdA[1] + m8 * 16 * 128);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m8 = 0; m8 < 8; ++m8) {
            acc[m8][0] = WS_MMA(bf[0][2 * s], af0[m8], acc[m8][0]);
            acc[m8][1] = WS_MMA(bf[1][2 * s], af0[m8], acc[m8][1]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (s + 1 < WS_KS || it + 1 < ntl) {
#pragma unroll
            for (int m8 = 0; m8 < 8; ++m8) af0[m8] = WS_RD(sbn + rdA[0] + m8 * 16 * 128);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m8 = 0; m8 < 8; ++m8) {
            acc[m8][0] = WS_MMA(bf[0][2 * s + 1], af1[m8], acc[m8][0]);
            acc[m8][1] = WS_MMA(bf[1][2 * s + 1], af1[m8], acc[m8][1]);
            if (s + 1 == WS_KS && m8 > 0) {
              // last slice: a row block's accumulators go to the staging image as soon as they are final - row
              // m = 16 m8 + li, columns 32 wave + 16 n2 + 4 lg .. + 3 (fp32, 16 bytes) - under the MFMAs of the blocks
              // behind it.  (The helpers read the image in steps 0-4 only: it is free once B(5) has opened.)
#pragma unroll
              for (int n2 = 0; n2 < 2; ++n2) {
                const int row = (m8 - 1) * 16 + li;
                const int chunk = 8 * wave + 4 * n2 + lg;
                *(f32x4 *)(stage + row * 512 + ((chunk ^ (row & 31)) << 4)) = acc[m8 - 1][n2];
              }
            }
            if (s + 1 == WS_KS) __builtin_amdgcn_sched_barrier(0);
          }
          if (s + 1 == WS_KS) {
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
              const int row = 7 * 16 + li;
              const int chunk = 8 * wave + 4 * n2 + lg;
              *(f32x4 *)(stage + row * 512 + ((chunk ^ (row & 31)) << 4)) = acc[7][n2];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (s == 4 && idx_job) {
          int *tab = idx_tab + ((it + 2) & 1) * BM;
          tab[lane] = iv0 / p.a_row_div; tab[lane + 64] = iv1 / p.a_row_div;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // staging image (and index table) written before B(0)
    }
  } else {
    // =============================================================== helper waves: activation slices + the previous tile's epilogue
    const int hw = wave - 4, ht = tid - 4 * 64;    // pieces 4 hw .. 4 hw + 3 of a slice (8 image rows each)
    const int cg = ht & 15, r16 = ht >> 4;         // epilogue: 8 columns per thread, 16 rows per pass, 8 passes per tile
    const char *src[4];
    u32x4 lq[WS_KS][4];                            // slice ks of the tile being fetched lives in set ks
    auto set_src = [&](int it, bool from_tab) {    // per-lane source rows of tile it (clamped past the group's end)
      const WsTile t = ws_tile_at(tile_tab, it);
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        const int row = (4 * hw + pc) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ dma_swz(row);
        int64_t m = t.m_begin + row;
        if (m >= t.m_end) m = t.m_end - 1;
        int64_t sr = m;
        if (p.a_row_idx) sr = from_tab ? (int64_t)idx_tab[(it & 1) * BM + row] : (int64_t)(p.a_row_idx[m] / p.a_row_div);
        src[pc] = p.A + sr * p.lda_b + c * 16;
      }
    };
    auto fetch = [&](int ks, u32x4(&r)[4]) {
#ifdef WS_ABL_NOFETCH
      if (tid >= 0) return;
#endif
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) r[pc] = *(const u32x4 *)(src[pc] + ks * 128);
    };
    auto put = [&](int slot, const u32x4(&r)[4]) {
#ifdef WS_ABL_NOPUT
      if (tid >= 0) return;
#endif
      char *dst = ring + slot * WS_SLICE + (4 * hw) * 1024 + lane * 16;
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) *(u32x4 *)(dst + pc * 1024) = r[pc];
    };
    set_src(0, false);
#pragma unroll
    for (int ks = 0; ks < WS_KS; ++ks) fetch(ks, lq[ks]);
    if (ntl > 1) set_src(1, false);                // (the table serves tiles 2 ...)
    put(0, lq[0]);
    if (ntl * WS_KS > 1) put(1, lq[1]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // One round = the six steps of tile `it`.  LIVE: tile `it` exists (its slices go to the ring); MORE: so does tile
    // it + 1 (its slices are fetched); HAVE: tile it - 1 exists (its epilogue runs).  These are template constants and
    // rows past a group's end are clamped, not skipped, so a round is one branch-free instruction stream: the compiler
    // then counts the vector-memory operations exactly and every ring write waits for ITS slice's loads only
    // (vmcnt(N) with the right N).  With run-time `if (more)` / `if (valid)` around loads and stores it fell back to
    // vmcnt(0) in front of every ring write - a full memory round trip per step.
    // A clamped row repeats the group's last row: same operands, same result, stored to the same place.
    auto round = [&](auto live_c, auto more_c, auto have_c, const int it) {
      constexpr bool LIVE = decltype(live_c)::value, MORE = decltype(more_c)::value, HAVE = decltype(have_c)::value;
      WsTile t; t.g = 0; t.nt = 0; t.m_begin = 0; t.m_end = 0;
      int n = 0;
      u32x4 gq[8];
      f32x4 rq[8][2];
      auto row_of = [&](int pass) {                // output row of this thread in a pass, clamped to the group's last row
        const int64_t m = t.m_begin + pass * 16 + r16;
        return m < t.m_end ? m : t.m_end - 1;
      };
      if constexpr (HAVE) {
        t = ws_tile_at(tile_tab, it - 1);
        n = t.nt * BN + cg * 8;
        // what this tile's epilogue reads from memory, requested in front of this round's stores
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
          const int64_t m = row_of(pass);
          if constexpr (EPI == WS_EPI_GPRE) gq[pass] = *(const u32x4 *)(p.gpre + (m * p.ld_gpre + n) * 2);
          if constexpr (EPI == WS_EPI_RES) {
            rq[pass][0] = *(const f32x4 *)(p.residual + m * p.ld_res + n);
            rq[pass][1] = *(const f32x4 *)(p.residual + m * p.ld_res + n + 4);
          }
        }
      }
#pragma unroll
      for (int s = 0; s < WS_KS; ++s) {
        WS_ARRIVE(1); __builtin_amdgcn_s_barrier();              // B(q): slice q (written one step ago) is readable
        asm volatile("" ::: "memory");
        // epilogue passes of this step (1, 2, 2, 2, 1, none: the image is rewritten in step 5): the staging reads are
        // requested first so that their latency runs under the load / ring-write issue below
        constexpr int first[WS_KS + 1] = {0, 1, 3, 5, 7, 8, 8};
        f32x4 sv[2][2];
#ifdef WS_ABL_NOEPI
        constexpr bool EPI_ON = false;
#else
        constexpr bool EPI_ON = true;
#endif
        if constexpr (HAVE && EPI_ON) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (first[s] + j >= first[s + 1]) continue;
            const int row = (first[s] + j) * 16 + r16;
            const int sw = row & 31;
            sv[j][0] = *(const f32x4 *)(stage + row * 512 + (((2 * cg) ^ sw) << 4));
            sv[j][1] = *(const f32x4 *)(stage + row * 512 + (((2 * cg + 1) ^ sw) << 4));
          }
        }
        if constexpr (LIVE) {
          if constexpr (MORE) {
            if constexpr (HAVE) { if (s == 0) set_src(it + 1, true); }      // (tile 1's rows were set in the prologue)
            fetch(s, lq[s]);                                     // slice s of tile it + 1; set s went to LDS two steps ago
          }
          if (s + 2 < WS_KS || MORE) put((s + 2) % WS_NS, lq[(s + 2) % WS_KS]);    // slice q + 2 -> slot (q + 2) % 3
        }
        if constexpr (HAVE && EPI_ON) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (first[s] + j >= first[s + 1]) continue;
            const int pass = first[s] + j;
            const int64_t m = row_of(pass);
            f32x4 v0 = sv[j][0], v1 = sv[j][1];
            if constexpr (EPI == WS_EPI_GELU) {
              Vec8<half_t>::store((half_t *)p.pre_out + m * p.ld_pre + n, v0, v1);
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) { v0[jj] = gelu_f(v0[jj]); v1[jj] = gelu_f(v1[jj]); }
            }
            if constexpr (EPI == WS_EPI_GPRE) {
              const f16x8 h = __builtin_bit_cast(f16x8, gq[pass]);
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) { v0[jj] *= gelu_grad_f((float)h[jj]); v1[jj] *= gelu_grad_f((float)h[4 + jj]); }
            }
            if constexpr (EPI == WS_EPI_RES) {
              v0 += rq[pass][0]; v1 += rq[pass][1];
              *(f32x4 *)((float *)p.C + m * p.ldc + n) = v0;
              *(f32x4 *)((float *)p.C + m * p.ldc + n + 4) = v1;
            } else {
              Vec8<half_t>::store((half_t *)p.C + m * p.ldc + n, v0, v1);
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // ring writes / staging reads done before the next barrier
      }
    };
    const std::true_type yes; const std::false_type no;
    if (ntl == 1) {
      round(yes, no, no, 0);
    } else {
      round(yes, yes, no, 0);
#pragma nounroll
      for (int it = 1; it + 1 < ntl; ++it) round(yes, yes, yes, it);
      round(yes, no, yes, ntl - 1);
    }
    round(no, no, yes, ntl);
  }
}

}  // namespace m3

using namespace m3;

// weight-stationary variant: mask of the epilogues it may take (bit 0 plain, 1 GELU + pre-activation, 2 GELU' multiply,
// 3 fp32 residual, bit 4: grouped calls too).  Off by default (see the kernel's header); -1 = take M3_GEMM_WS from the
// environment at the first call.
static int g_ws_mode = -1;
extern "C" int m3_gemm_set_variant(int ws_mask) {
  M3_REQUIRE(ws_mask >= -1 && ws_mask < 32, "m3_gemm_set_variant: mask %d out of range", ws_mask);
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
  // weight-stationary persistent variant (opt-in: m3_gemm_set_variant / M3_GEMM_WS): fp16, K = 384, whole 128-wide
  // column tiles and one of its four epilogues (plain / GELU + pre-activation / GELU' multiply / fp32 residual)
  static int ws_grid = 0;
  if (g_ws_mode < 0) {
    const char *e = getenv("M3_GEMM_WS");
    g_ws_mode = e ? atoi(e) : 0;
  }
  const int ws_mode = g_ws_mode;
  if (ws_mode && ws_grid == 0) {
    int dev = 0; hipDeviceProp_t prop;
    const int cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    ws_grid = cus / 8 * 8;                         // one workgroup per CU, the same number on every XCD
    (void)hipFuncSetAttribute((const void *)gemm_nt_ws_kernel<WS_EPI_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
    (void)hipFuncSetAttribute((const void *)gemm_nt_ws_kernel<WS_EPI_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
    (void)hipFuncSetAttribute((const void *)gemm_nt_ws_kernel<WS_EPI_GPRE>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
    (void)hipFuncSetAttribute((const void *)gemm_nt_ws_kernel<WS_EPI_RES>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
  }
  if (ws_mode && ws_grid >= 8 && a->dtype == M3_F16 && a->K == 384 && a->N % BN == 0 && d.vec8 && a->M >= 8 * BM &&
      !a->c_row_idx && !a->row_scale && a->M < ((int64_t)1 << 31) &&
      ((mt + 7) / 8 + 1) * d.n_tiles / (ws_grid / 8) + 2 <= WS_MAXT) {               // longest run of tiles of a workgroup
    int epi = -1;
    if (a->gelu_grad_pre) { if (!a->residual && !d.c_f32 && a->act == M3_ACT_NONE && !a->pre_out) epi = WS_EPI_GPRE; }
    else if (a->residual) { if (d.c_f32 && a->act == M3_ACT_NONE && !a->pre_out) epi = WS_EPI_RES; }
    else if (a->act == M3_ACT_GELU) { if (!d.c_f32 && a->pre_out) epi = WS_EPI_GELU; }
    else if (a->act == M3_ACT_NONE && !d.c_f32 && !a->pre_out) epi = WS_EPI_PLAIN;
    if (epi >= 0 && !((ws_mode >> epi) & 1)) epi = -1;            // M3_GEMM_WS is a mask: bit e = epilogue e, bit 4 = grouped calls
    if (epi >= 0 && a->group_offsets && !((ws_mode >> 4) & 1)) epi = -1;
    const dim3 wg((unsigned)ws_grid), wb(WS_THREADS);
    if (epi == WS_EPI_PLAIN) hipLaunchKernelGGL(gemm_nt_ws_kernel<WS_EPI_PLAIN>, wg, wb, WS_LDS, s, d);
    else if (epi == WS_EPI_GELU) hipLaunchKernelGGL(gemm_nt_ws_kernel<WS_EPI_GELU>, wg, wb, WS_LDS, s, d);
    else if (epi == WS_EPI_GPRE) hipLaunchKernelGGL(gemm_nt_ws_kernel<WS_EPI_GPRE>, wg, wb, WS_LDS, s, d);
    else if (epi == WS_EPI_RES) hipLaunchKernelGGL(gemm_nt_ws_kernel<WS_EPI_RES>, wg, wb, WS_LDS, s, d);
    if (epi >= 0) return check_launch("m3_gemm_nt");
  }
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
extern "C" int m3_debug_gemm_ws_stamps(unsigned long long *dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ws_stamps), sizeof(unsigned long long) * WS_STAMP_WGS * 2 * 64) == hipSuccess ? M3_OK : M3_ERR_LAUNCH;
}
extern "C" int m3_debug_gemm_stamps(unsigned long long *dst, int wgs) {
  if (wgs > STAMP_WGS) wgs = STAMP_WGS;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gemm_stamps), (size_t)wgs * STAMP_N * sizeof(unsigned long long)) == hipSuccess ? M3_OK : M3_ERR_LAUNCH;
}
#endif
