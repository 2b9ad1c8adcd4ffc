// Dispatch metadata for top-k routing on gfx950: per-expert counts, exclusive offsets,
// a STABLE slot for every (token, j) entry and the m-tile prefix the grouped GEMMs read.
//
// Replaces fastmoe's count_by_gate / assign_pos (prepare_forward) behind
// _fmoe_general_global_forward, models/moe/ckpt/custom_moe_layer.py:263-265; same
// information as compute_gating, models/moe/moe.py:19-64.  Everything stays on the
// device (the reference syncs to the host for the counts); the order inside an expert is
// increasing flat entry index, so results are run-to-run identical (fastmoe's is
// atomics-ordered).
//
// Three tiny launches: block histograms -> one-block scan -> block-local stable ranks
// (wave ballots; 64-wide).  Integer work, bit-exact against oracle/gate_route.c.
#include "common.h"

namespace m3 {

constexpr int RT_THREADS = 256;
constexpr int RT_PASSES = 4;
constexpr int RT_BLOCK = RT_THREADS * RT_PASSES;   // entries per workgroup
constexpr int RT_MAX_E = 256;

__global__ __launch_bounds__(RT_THREADS) void route_hist_kernel(const int32_t *idx, int64_t n, int E,
                                                                int32_t *blk_counts) {
  __shared__ int32_t h[RT_MAX_E];
  for (int e = threadIdx.x; e < E; e += RT_THREADS) h[e] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RT_BLOCK;
#pragma unroll
  for (int p = 0; p < RT_PASSES; ++p) {
    const int64_t i = base + p * RT_THREADS + threadIdx.x;
    if (i < n) {
      const int e = idx[i];
      if (e >= 0 && e < E) atomicAdd(&h[e], 1);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += RT_THREADS) blk_counts[(int64_t)blockIdx.x * E + e] = h[e];
}

// exclusive scan of the per-block histograms along the block axis, one WAVE per expert (64 blocks per
// step, shuffle scan), then the expert offsets / tile prefix by one thread
constexpr int RT_SCAN_THREADS = 1024;
__global__ __launch_bounds__(RT_SCAN_THREADS) void route_scan_kernel(const int32_t *blk_counts, int nblk, int E,
                                                                     int32_t *blk_base, int32_t *counts,
                                                                     int32_t *offsets, int32_t *tile_starts,
                                                                     int64_t *counts64) {
  __shared__ int32_t tot[RT_MAX_E];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = wave; e < E; e += RT_SCAN_THREADS / 64) {
    int32_t carry = 0;
    for (int b0 = 0; b0 < nblk; b0 += 64) {
      const int b = b0 + lane;
      const int32_t v = b < nblk ? blk_counts[(int64_t)b * E + e] : 0;
      int32_t inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int32_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
      }
      if (b < nblk) blk_base[(int64_t)b * E + e] = carry + inc - v;
      carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) {
      tot[e] = carry;
      counts[e] = carry;
      if (counts64) counts64[e] = carry;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t o = 0, ts = 0;
    for (int i = 0; i < E; ++i) {
      offsets[i] = o;
      tile_starts[i] = ts;
      o += tot[i];
      ts += (tot[i] + 127) / 128;
    }
    offsets[E] = o;
    tile_starts[E] = ts;
  }
}

// base_stride: prefix blocks per workgroup - 1 for m3_route_build's own histogram blocks (RT_BLOCK entries each), the
// number of 64-token gate blocks in RT_BLOCK entries for m3_route_assign (prefix of the first of them: the workgroup
// walks its entries in order, so the running counts pass through the others' prefixes by themselves)
__global__ __launch_bounds__(RT_THREADS) void route_assign_kernel(const int32_t *idx, int64_t n, int E,
                                                                  const int32_t *blk_base, const int32_t *offsets,
                                                                  int32_t *pos, int32_t *row_of_slot, int base_stride) {
  __shared__ int32_t run[RT_MAX_E];          // entries of expert e seen in earlier passes of this block
  __shared__ int32_t wcnt[4][RT_MAX_E];      // per-wave counts of the current pass
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < E; e += RT_THREADS)
    run[e] = offsets[e] + blk_base[(int64_t)blockIdx.x * base_stride * E + e];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RT_BLOCK;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int p = 0; p < RT_PASSES; ++p) {
    const int64_t i = base + p * RT_THREADS + threadIdx.x;
    int my = -1;
    if (i < n) {
      my = idx[i];
      if (my < 0 || my >= E) my = -1;
    }
    int rank = 0;
    for (int e = 0; e < E; ++e) {
      const unsigned long long m = __ballot(my == e);
      if (my == e) rank = __popcll(m & lt);
      if (lane == 0) wcnt[wave][e] = __popcll(m);
    }
    __syncthreads();
    if (my >= 0) {
      int s = run[my] + rank;
      for (int w = 0; w < wave; ++w) s += wcnt[w][my];
      pos[i] = s;
      row_of_slot[s] = (int32_t)i;
    } else if (i < n) {
      // an id outside [0, E) is a caller error (Route.check() / offsets[E] != n reports it); it gets no slot.  Keep the
      // metadata in range all the same: consumers gather through pos / row_of_slot with M = n rows.
      pos[i] = 0;
    }
    if (i < n && i >= offsets[E]) row_of_slot[i] = 0;       // slots past the last routed row (only with dropped ids)
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += RT_THREADS) run[e] += wcnt[0][e] + wcnt[1][e] + wcnt[2][e] + wcnt[3][e];
    __syncthreads();
  }
}

// ---- expert-parallel exchange plan (fastmoe expert_exchange / global_scatter bookkeeping behind
// _fmoe_general_global_forward for world_size > 1, models/moe/ckpt/custom_moe_layer.py:263-265), on the device:
// send[d * E_loc + e]: rows this rank routes to local expert e of rank d (its route_build counts, by global expert id);
// recv[s * E_loc + e]: rows rank s routes to this rank's local expert e (the all-to-all of the former).
// Rows arrive ordered (src, e); the local grouped GEMMs want (e, src): rg[i] = position in the received buffer of the
// i-th row of the expert-major order.  Also: the a2a-v split sizes (the only values the host reads), the expert-major
// group offsets and the 128-row tile prefix the grouped GEMMs take.
constexpr int EP_MAX_BLOCKS = 64 * 64;       // W * E_loc (sources x local experts)
//
// Fixed-capacity form (cap > 0; m3_ep_plan_fixed): every (source, destination) pair exchanges exactly `cap` rows (equal
// splits: nothing of the exchange is read by the host, so a step can be captured), the valid rows first.  A source's rows
// then land at s * cap + their prefix within the source; a pair that routes more than `cap` rows keeps the FIRST cap rows
// of its (expert, token) order on both sides, raises *overflow and the caller repeats the step on the exact path.
__global__ __launch_bounds__(256) void ep_plan_kernel(const int64_t *send, const int64_t *recv, int W, int E_loc,
                                                      int64_t *splits, int32_t *rg, int64_t rg_cap, int32_t *offsets,
                                                      int32_t *tile_starts, int cap, int32_t *overflow) {
  __shared__ int32_t src_start[EP_MAX_BLOCKS + 1]; // [s * E_loc + e]: first received row of block (s, e)
  __shared__ int32_t em_start[EP_MAX_BLOCKS + 1]; // [e * W + s]: first expert-major slot of block (e, s)
  __shared__ int32_t part[256];
  const int nb = W * E_loc, tid = threadIdx.x;
  // both prefix sums by the whole workgroup (a contiguous run of blocks per thread + a 256-wide scan of the run sums): with
  // W * E_loc up to 4096 one thread walking 2 x nb dependent global loads in every workgroup was the critical path in front of
  // the host's read of the split sizes
  const int per = (nb + 255) / 256;
  const int b0 = tid * per < nb ? tid * per : nb, b1 = b0 + per < nb ? b0 + per : nb;
  auto scan = [&](auto get, int32_t *out) {          // out[b] = sum_{b' < b} get(b'), out[nb] = total
    int32_t sum = 0;
    for (int q = b0; q < b1; ++q) sum += get(q);
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const int32_t v = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    int32_t o = part[tid] - sum;
    for (int q = b0; q < b1; ++q) { out[q] = o; o += get(q); }
    if (tid == 255) out[nb] = part[255];
    __syncthreads();
  };
  scan([&](int q) { return (int32_t)recv[q]; }, src_start);                                   // order (s, e); src_start[nb] unused
  // rows of block (s, e) that fit the pair's capacity (all of them on the exact path)
  auto kept = [&](int sr, int e) -> int32_t {
    const int b = sr * E_loc + e;
    if (cap <= 0) return (int32_t)recv[b];
    const int32_t before = src_start[b] - src_start[sr * E_loc];          // rows of source sr in front of expert e
    const int32_t lo_ = before < cap ? before : cap;
    const int32_t hi_ = before + (int32_t)recv[b] < cap ? before + (int32_t)recv[b] : cap;
    return hi_ - lo_;
  };
  scan([&](int q) { const int e = q / W, sr = q - e * W; return kept(sr, e); }, em_start);   // order (e, s)
  if (cap > 0) {
    // received row of block (s, e): s * cap + min(rows of s in front of e, cap)   (every thread rewrites the same values)
    __syncthreads();
    for (int sr = 0; sr < W; ++sr) {
      const int32_t base = src_start[sr * E_loc];
      __syncthreads();
      for (int e = tid; e < E_loc; e += 256) {
        const int32_t before = src_start[sr * E_loc + e] - base;
        src_start[sr * E_loc + e] = sr * cap + (before < cap ? before : cap);
      }
      __syncthreads();
    }
  }
  if (blockIdx.x == 0) {
    if (tid == 0) {
      int32_t ts = 0;
      for (int e = 0; e < E_loc; ++e) {
        offsets[e] = em_start[e * W];
        tile_starts[e] = ts;
        ts += (em_start[(e + 1) * W] - em_start[e * W] + 127) / 128;
      }
      offsets[E_loc] = em_start[nb];
      tile_starts[E_loc] = ts;
    }
    for (int d = tid; d < W; d += 256) {
      int64_t a_ = 0, b_ = 0;
      for (int e = 0; e < E_loc; ++e) { a_ += send[d * E_loc + e]; b_ += recv[d * E_loc + e]; }
      splits[d] = a_;            // rows this rank sends to rank d
      splits[W + d] = b_;        // rows this rank receives from rank d
      if (cap > 0 && overflow && (a_ > cap || b_ > cap)) *overflow = 1;
    }
  }
  const int64_t n = em_start[nb] < rg_cap ? em_start[nb] : rg_cap;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = nb - 1;                      // last block with em_start <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (em_start[mid] <= i) lo = mid; else hi = mid - 1;
    }
    const int e = lo / W, s = lo - e * W;
    rg[i] = src_start[s * E_loc + e] + (int32_t)(i - em_start[lo]);
  }
}

// Index vectors of the fixed-capacity exchange.  send: this rank's route_build counts by global expert id, so the
// expert-major slots [send_off[d], send_off[d + 1]) go to rank d.
//   pad_idx[d * cap + j]  = row_of_slot[send_off[d] + j]  (token-major entry whose row is the j-th sent to rank d; past the
//                           pair's rows: the last valid one - the receiver never looks at those rows)
//   unpad_idx[i]          = d * cap + (pos[i] - send_off[d])  (where entry i's expert output sits in the returned padded
//                           buffer; an entry that did not fit reads the pair's last row - the step is repeated anyway)
__global__ __launch_bounds__(256) void ep_pad_index_kernel(const int64_t *send, int W, int E_loc, int cap,
                                                           const int32_t *row_of_slot, const int32_t *pos, int64_t n,
                                                           int32_t *pad_idx, int32_t *unpad_idx) {
  __shared__ int32_t send_off[65];
  if (threadIdx.x == 0) {
    int32_t o = 0;
    for (int d = 0; d < W; ++d) {
      send_off[d] = o;
      for (int e = 0; e < E_loc; ++e) o += (int32_t)send[d * E_loc + e];
    }
    send_off[W] = o;
  }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t q = i0; q < (int64_t)W * cap; q += stride) {
    const int d = (int)(q / cap), j = (int)(q - (int64_t)d * cap);
    const int32_t nd = send_off[d + 1] - send_off[d];
    int32_t slot = send_off[d] + (j < nd ? j : (nd > 0 ? nd - 1 : 0));
    if (slot >= n) slot = n > 0 ? (int32_t)n - 1 : 0;
    pad_idx[q] = n > 0 ? row_of_slot[slot] : 0;
  }
  for (int64_t i = i0; i < n; i += stride) {
    const int32_t slot = pos[i];
    int d = 0;
    while (d + 1 < W && send_off[d + 1] <= slot) ++d;
    const int32_t j = slot - send_off[d];
    unpad_idx[i] = d * cap + (j < cap ? j : cap - 1);
  }
}

}  // namespace m3

using namespace m3;

static int64_t route_blocks(int64_t n) { return (n + RT_BLOCK - 1) / RT_BLOCK; }

extern "C" int64_t m3_route_ws_elems(int64_t n, int E) { return 2 * route_blocks(n > 0 ? n : 1) * (int64_t)E; }

extern "C" int m3_route_build(const int32_t *idx32, int64_t n, int E, int32_t *counts, int32_t *offsets, int32_t *pos,
                              int32_t *row_of_slot, int32_t *tile_starts, int64_t *counts64, int32_t *ws,
                              void *stream) {
  M3_REQUIRE(idx32 && counts && offsets && pos && row_of_slot && tile_starts && ws, "m3_route_build: null operand");
  M3_REQUIRE(E >= 1 && E <= RT_MAX_E, "m3_route_build: E=%d outside [1,%d]", E, RT_MAX_E);
  M3_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "m3_route_build: n out of range");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (int)route_blocks(n > 0 ? n : 1);
  int32_t *blk_counts = ws, *blk_base = ws + (int64_t)nblk * E;
  hipLaunchKernelGGL(route_hist_kernel, dim3(nblk), dim3(RT_THREADS), 0, s, idx32, n, E, blk_counts);
  int rc = check_launch("m3_route_build(hist)");
  if (rc) return rc;
  hipLaunchKernelGGL(route_scan_kernel, dim3(1), dim3(RT_SCAN_THREADS), 0, s, blk_counts, nblk, E, blk_base, counts, offsets,
                     tile_starts, counts64);
  rc = check_launch("m3_route_build(scan)");
  if (rc) return rc;
  hipLaunchKernelGGL(route_assign_kernel, dim3(nblk), dim3(RT_THREADS), 0, s, idx32, n, E, blk_base, offsets, pos,
                     row_of_slot, 1);
  return check_launch("m3_route_build(assign)");
}

extern "C" int m3_route_assign(const int32_t *idx32, int64_t n, int E, int k, const int32_t *blk_base, const int32_t *offsets,
                               int32_t *pos, int32_t *row_of_slot, void *stream) {
  M3_REQUIRE(idx32 && blk_base && offsets && pos && row_of_slot, "m3_route_assign: null operand");
  M3_REQUIRE(E >= 1 && E <= RT_MAX_E && n >= 0 && n < ((int64_t)1 << 31), "m3_route_assign: E / n out of range");
  M3_REQUIRE(k >= 1 && k <= 16 && 16 % k == 0 && n % k == 0, "m3_route_assign: k = %d must divide 16 (64-token prefix blocks)", k);
  if (n == 0) return M3_OK;
  const int nblk = (int)route_blocks(n);
  hipLaunchKernelGGL(route_assign_kernel, dim3(nblk), dim3(RT_THREADS), 0, (hipStream_t)stream, idx32, n, E, blk_base, offsets,
                     pos, row_of_slot, RT_BLOCK / (64 * k));
  return check_launch("m3_route_assign");
}

extern "C" int m3_ep_plan(const int64_t *send_counts, const int64_t *recv_counts, int W, int E_loc, int64_t *splits,
                          int32_t *regroup, int64_t regroup_cap, int32_t *offsets, int32_t *tile_starts, void *stream) {
  M3_REQUIRE(send_counts && recv_counts && splits && regroup && offsets && tile_starts, "m3_ep_plan: null operand");
  M3_REQUIRE(W >= 1 && E_loc >= 1 && W * E_loc <= EP_MAX_BLOCKS, "m3_ep_plan: W * E_loc = %d outside [1, %d]", W * E_loc, EP_MAX_BLOCKS);
  M3_REQUIRE(regroup_cap >= 0 && regroup_cap < ((int64_t)1 << 31), "m3_ep_plan: regroup capacity out of range");
  const int64_t want = (regroup_cap + 255) / 256;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 512 ? 512 : want));
  hipLaunchKernelGGL(ep_plan_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, send_counts, recv_counts, W, E_loc, splits,
                     regroup, regroup_cap, offsets, tile_starts, 0, (int32_t *)nullptr);
  return check_launch("m3_ep_plan");
}

extern "C" int m3_ep_plan_fixed(const int64_t *send_counts, const int64_t *recv_counts, int W, int E_loc, int cap,
                                const int32_t *row_of_slot, const int32_t *pos, int64_t n_rows, int64_t *splits,
                                int32_t *regroup, int32_t *offsets, int32_t *tile_starts, int32_t *pad_idx,
                                int32_t *unpad_idx, int32_t *overflow, void *stream) {
  M3_REQUIRE(send_counts && recv_counts && splits && regroup && offsets && tile_starts && pad_idx && unpad_idx && overflow &&
             row_of_slot && pos, "m3_ep_plan_fixed: null operand");
  M3_REQUIRE(W >= 1 && W <= 64 && E_loc >= 1 && W * E_loc <= EP_MAX_BLOCKS, "m3_ep_plan_fixed: W = %d, E_loc = %d out of range", W, E_loc);
  M3_REQUIRE(cap >= 1 && (int64_t)W * cap < ((int64_t)1 << 31) && n_rows >= 0 && n_rows < ((int64_t)1 << 31),
             "m3_ep_plan_fixed: capacity / row count out of range");
  const int64_t rg_cap = (int64_t)W * cap;
  const int64_t want = (rg_cap + 255) / 256;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 512 ? 512 : want));
  hipLaunchKernelGGL(ep_plan_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, send_counts, recv_counts, W, E_loc, splits,
                     regroup, rg_cap, offsets, tile_starts, cap, overflow);
  const int64_t w2 = ((n_rows > rg_cap ? n_rows : rg_cap) + 255) / 256;
  hipLaunchKernelGGL(ep_pad_index_kernel, dim3((unsigned)(w2 < 1 ? 1 : (w2 > 512 ? 512 : w2))), dim3(256), 0, (hipStream_t)stream,
                     send_counts, W, E_loc, cap, row_of_slot, pos, n_rows, pad_idx, unpad_idx);
  return check_launch("m3_ep_plan_fixed");
}
