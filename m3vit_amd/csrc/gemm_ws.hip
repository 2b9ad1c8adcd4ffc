// Weight-stationary, wave-specialised persistent NT GEMM (fp16, K = 384): EXPERIMENTAL builds only (make EXPERIMENTAL=1).
// Built and measured in round 2: per launch it beats the tiled kernel where the epilogue is heavy and
// ties elsewhere; inside the two-stream training step it loses 1-3 % - the engine never takes it.  Kept as the base of a
// persistent, cross-tile-pipelined kernel; opt-in through m3_gemm_set_variant / M3_GEMM_WS in such a build.
#include "gemm_dev.h"
#include <type_traits>

namespace m3 {

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


// 0: launched; 1: not a call this kernel takes (the caller goes on to the tiled kernels); < 0: error
int launch_gemm_ws(const GemmDev &d, const m3_gemm_args *a, int64_t mt, int ws_mode, hipStream_t s) {
  static int ws_grid = 0;
  if (ws_grid == 0) {
    int dev = 0; hipDeviceProp_t prop;
    const int cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    ws_grid = cus / 8 * 8;                         // one workgroup per CU, the same number on every XCD
    (void)hipFuncSetAttribute((const void *)gemm_nt_ws_kernel<WS_EPI_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
    (void)hipFuncSetAttribute((const void *)gemm_nt_ws_kernel<WS_EPI_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
    (void)hipFuncSetAttribute((const void *)gemm_nt_ws_kernel<WS_EPI_GPRE>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
    (void)hipFuncSetAttribute((const void *)gemm_nt_ws_kernel<WS_EPI_RES>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
  }
  if (!(ws_grid >= 8 && a->dtype == M3_F16 && a->K == 384 && a->N % BN == 0 && d.vec8 && a->M >= 8 * BM &&
        !a->c_row_idx && !a->row_scale && a->M < ((int64_t)1 << 31) &&
        ((mt + 7) / 8 + 1) * d.n_tiles / (ws_grid / 8) + 2 <= WS_MAXT))               // longest run of tiles of a workgroup
    return 1;
  int epi = -1;
  if (a->gelu_grad_pre) { if (!a->residual && !d.c_f32 && a->act == M3_ACT_NONE && !a->pre_out) epi = WS_EPI_GPRE; }
  else if (a->residual) { if (d.c_f32 && a->act == M3_ACT_NONE && !a->pre_out) epi = WS_EPI_RES; }
  else if (a->act == M3_ACT_GELU) { if (!d.c_f32 && a->pre_out) epi = WS_EPI_GELU; }
  else if (a->act == M3_ACT_NONE && !d.c_f32 && !a->pre_out) epi = WS_EPI_PLAIN;
  if (epi >= 0 && !((ws_mode >> epi) & 1)) epi = -1;            // M3_GEMM_WS is a mask: bit e = epilogue e, bit 4 = grouped calls
  if (epi >= 0 && a->group_offsets && !((ws_mode >> 4) & 1)) epi = -1;
  if (epi < 0) return 1;
  const dim3 wg((unsigned)ws_grid), wb(WS_THREADS);
  if (epi == WS_EPI_PLAIN) hipLaunchKernelGGL(gemm_nt_ws_kernel<WS_EPI_PLAIN>, wg, wb, WS_LDS, s, d);
  else if (epi == WS_EPI_GELU) hipLaunchKernelGGL(gemm_nt_ws_kernel<WS_EPI_GELU>, wg, wb, WS_LDS, s, d);
  else if (epi == WS_EPI_GPRE) hipLaunchKernelGGL(gemm_nt_ws_kernel<WS_EPI_GPRE>, wg, wb, WS_LDS, s, d);
  else hipLaunchKernelGGL(gemm_nt_ws_kernel<WS_EPI_RES>, wg, wb, WS_LDS, s, d);
  return check_launch("m3_gemm_nt");
}

}  // namespace m3

#ifdef M3_GEMM_STAMPS
extern "C" int m3_debug_gemm_ws_stamps(unsigned long long *dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(m3::g_ws_stamps), sizeof(unsigned long long) * m3::WS_STAMP_WGS * 2 * 64) == hipSuccess ? M3_OK : M3_ERR_LAUNCH;
}
#endif
