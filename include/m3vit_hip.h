/* m3vit_hip.h - C ABI of libm3vit_hip.so: the MI355X (gfx950) implementation of the
 * M3ViT MoE-ViT forward/backward hot path.
 *
 * This is the drop-in boundary described in SURVEY.md section 8(b).  The reference
 * reaches this path through the python package `fmoe` (laekov/fastmoe, absent from
 * /root/reference) whose native half is the `fmoe_cuda` extension; the entry points
 * below are what a binding for that layer would call.  Each one cites the reference
 * call site (path:line relative to the reference root) whose work it replaces.
 *
 * Conventions
 *   - plain C, no torch types: device pointers, sizes, an opaque hipStream_t (void*).
 *   - every call is asynchronous on `stream`, allocates nothing, takes no locks and
 *     never synchronises the device; the caller owns all buffers and workspaces.
 *   - return value: 0 = ok, negative = M3_ERR_* ; m3_last_error() gives the text of
 *     the last failure on the calling thread.  Shape/alignment violations are
 *     rejected on the host before anything is launched.
 *   - dtype codes (M3_F32, M3_F16, M3_BF16) describe ACTIVATION storage; accumulation is
 *     always fp32; parameters/gradients of parameters are fp32.
 *   - indices are int32 on the device-internal path and int64 where the reference
 *     API exposes them (gate_top_k_idx from torch.topk).
 */
#ifndef M3VIT_HIP_H
#define M3VIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M3_F32 0
#define M3_F16 1
#define M3_BF16 2            /* every entry point except m3_ffn_fwd (fp16 there) */

#define M3_OK 0
#define M3_ERR_ARG (-1)      /* bad shape / alignment / null pointer */
#define M3_ERR_LAUNCH (-2)   /* hipLaunch failed (text in m3_last_error) */
#define M3_ERR_UNSUPPORTED (-3)

#define M3_ACT_NONE 0
#define M3_ACT_GELU 1        /* exact erf GELU, nn.GELU(): vision_transformer_moe.py:409-412 */

int m3_version(void);
const char *m3_last_error(void);
/* fills name[0..len) with the gcnArchName of the current device; returns CU count or <0 */
int m3_device_query(char *name, int len);
/* 1 when the library was built with `make EXPERIMENTAL=1` (the opt-in kernels the training step never takes are compiled in:
 * m3_ffn_fwd, the weight-stationary variant behind m3_gemm_set_variant, the wide tiles behind m3_wgrad_set_wide), else 0:
 * those three entry points then exist but refuse (M3_ERR_ARG).  No reference counterpart. */
int m3_experimental(void);

/* ---------------------------------------------------------------- gate (a1-a3)
 * NoisyGate_VMoE.forward, models/moe/ckpt/noisy_gate_vmoe.py:91-93,168,197-207:
 *   logit = x @ w_gate (+ logit_bias) ; noisy = logit + noise*noise_std ;
 *   p = softmax(noisy) ; top-(k+1) ; score/idx = first k (not renormalised).
 * x [T,D] (x_dtype, row stride ldx elements), w_gate [D,E] fp32 row-major,
 * logit_bias [E] fp32 or NULL (task-conditioned term tsf @ w_gate[D:], the
 * cat at custom_moe_layer.py:176-179 folded into a bias), noise [T,E] fp32 or NULL.
 * Outputs: idx i64 [T,k], score f32 [T,k], top_logits f32 [T,min(k+1,E)];
 * optional dense clean/noisy/gates f32 [T,E] (NULL to skip), optional
 * idx32 i32 [T,k] copy for the routing kernels.
 * part_importance f32 [nblk,E], part_load i32 [nblk,E] with nblk =
 * m3_gate_num_blocks(T): per-block partial sums of gates.sum(0) and (gates>0).sum(0)
 * (vision_transformer_moe.py:453-459), reduced in fixed order by m3_gate_reduce /
 * m3_balance_loss.
 * E in {2..64}, k < = 8, k <= E. */
int m3_gate_num_blocks(int64_t T);
int m3_gate_dw_blocks(int64_t T);
typedef struct m3_gate_fwd_args {
  const void *x; int32_t x_dtype; int64_t T; int32_t D; int64_t ldx;
  const float *w_gate; int32_t E;
  const float *logit_bias;          /* [E] or NULL */
  const float *noise;               /* [T,E] N(0,1) draws or NULL (randn_like at noisy_gate_vmoe.py:168) */
  float noise_std;                  /* noise_std / E_tot * training, :92-93 */
  int32_t k;
  int64_t *idx; int32_t *idx32;     /* [T,k]; idx32 optional */
  int32_t *idx_next;                /* [T] index of the (k+1)-th expert (top_idx[:, k], :198-200) or NULL */
  float *score; float *top_logits;  /* [T,k], [T,min(k+1,E)] */
  float *clean; float *noisy; float *gates;   /* dense [T,E] or NULL */
  float *part_importance; int32_t *part_load; /* [nblk,E] */
  float *part_load_prob;            /* [nblk,E] or NULL: partial sums of _prob_in_top_k
                                       (vision_transformer_moe.py:33-71), the load of noisy training
                                       (:456-457; needs noise, noise_std != 0, k < E, clean and noisy) */
  int32_t *part_count;              /* [nblk,E] or NULL: tokens of each 64-token block whose top-k holds expert e - the
                                       routing histogram m3_route_build would count from idx (m3_balance_route scans it) */
} m3_gate_fwd_args;
int m3_gate_fwd(const m3_gate_fwd_args *args, void *stream);
/* importance f32 [E], load i64 [E] from the per-block partials */
int m3_gate_reduce(const float *part_importance, const int32_t *part_load, int nblk, int E,
                   float *importance, int64_t *load, void *stream);
/* Block-level balance loss, vision_transformer_moe.py:453-459,540 with cv_squared :73-87 and
 * _gates_to_load :23-31: reduces the partials (fixed order) to importance f32 [E], load i64 [E]
 * (count form) and, when part_load_prob is given, load_prob f32 [E] (Normal-CDF form, which then
 * replaces the count in the loss); loss = cv^2(importance) + cv^2(load) is written to loss_out and
 * added to loss_acc (either may be NULL); d_importance / d_load_prob [E] receive d loss / d(.)
 * (NULL to skip; the count form of load carries no gradient). */
int m3_balance_loss(const float *part_importance, const int32_t *part_load, const float *part_load_prob,
                    int nblk, int E, float *importance, int64_t *load, float *load_prob,
                    float *loss_out, float *loss_acc, float *d_importance, float *d_load_prob, void *stream);
/* Backward of the gate -> d_logits f32 [T,E] through the scatter (:206-207), the softmax (:197)
 * and, for noisy training, the Normal-CDF load term.  Upstream gradients (each may be NULL):
 * d_score [T,k] (from the combine), d_top [T,min(k+1,E)] (w.r.t. top_logits), d_importance [E] and
 * d_load_prob [E] (from m3_balance_loss; both multiplied by balance_scale = the loss weight).
 * probs are recomputed from `noisy` [T,E]; clean / top_logits / idx_next / noise_std are only read
 * for d_load_prob (and idx_next for d_top).  balance_scale_dev (may be NULL): one device-resident float that multiplies
 * balance_scale - the upstream gradient of cv_loss when the caller is torch.autograd (the reference forms
 * cv_loss * moe_noisy_gate_loss_weight, train/train_utils.py:277, and autograd hands the weight back as a tensor); read
 * by the kernel, so a captured launch follows its value. */
typedef struct m3_gate_bwd_args {
  const float *noisy; const float *clean; const float *top_logits;
  const int64_t *idx; const int32_t *idx_next;
  const float *d_score; const float *d_top; const float *d_importance; const float *d_load_prob;
  float balance_scale; float noise_std;
  int64_t T; int32_t E; int32_t k;
  float *d_logits;
  const float *balance_scale_dev;
  void *d_logits_act; int32_t act_dtype;   /* optional second copy of d_logits in M3_F16 / M3_BF16 / M3_F32 (NULL: none) */
} m3_gate_bwd_args;
int m3_gate_bwd_logits(const m3_gate_bwd_args *args, void *stream);
/* m3_balance_loss and the scan pass of m3_route_build in ONE launch (two workgroups): the dispatch metadata of
 * custom_moe_layer.py:263-265 straight from the gate kernel's per-block counts, two launches less on every MoE layer's
 * critical path.  part_count i32 [nblk,E] from m3_gate_fwd; outputs blk_base i32 [nblk,E] (exclusive prefix along the
 * blocks, for m3_route_assign), counts i32 [E], offsets / tile_starts i32 [E+1], counts64 i64 [E] or NULL - as
 * m3_route_build writes them. */
int m3_balance_route(const float *part_importance, const int32_t *part_load, const float *part_load_prob,
                     int nblk, int E, float *importance, int64_t *load, float *load_prob,
                     float *loss_out, float *loss_acc, float *d_importance, float *d_load_prob,
                     const int32_t *part_count, int32_t *blk_base, int32_t *counts, int32_t *offsets,
                     int32_t *tile_starts, int64_t *counts64, void *stream);
/* d_w_gate[D,E] (+)= x^T d_logits ; dx[T,D] (+)= d_logits w_gate^T  (noisy_gate_vmoe.py:91).
 * part_dw f32 [m3_gate_dw_blocks(T), D, E] workspace; dx fp32 accumulate (beta_dx 0/1). */
int m3_gate_bwd_params(const void *x, int x_dtype, int64_t T, int D, int64_t ldx,
                       const float *w_gate, int E, const float *d_logits,
                       float *part_dw, float *d_w_gate, int beta_dw,
                       float *dx, int64_t lddx, int beta_dx, void *stream);

/* ------------------------------------------------------- dispatch metadata (a5)
 * What fastmoe's prepare_forward (count_by_gate + assign_pos) computes behind
 * _fmoe_general_global_forward, models/moe/ckpt/custom_moe_layer.py:263-265, with a
 * stable slot order.  idx32 [n] flat (n = T*k) expert ids in [0,E).
 * Outputs (all device): counts i32 [E], offsets i32 [E+1], pos i32 [n] (slot of flat
 * entry i), row_of_slot i32 [n] (inverse), tile_starts i32 [E+1] (prefix of
 * ceil(count/128) m-tiles per expert, read by the grouped GEMMs),
 * counts64 i64 [E] (optional, fwd_expert_count as the reference API exposes it).
 * ws i32 [m3_route_ws_elems(n,E)] scratch. */
int64_t m3_route_ws_elems(int64_t n, int E);
int m3_route_build(const int32_t *idx32, int64_t n, int E, int32_t *counts, int32_t *offsets,
                   int32_t *pos, int32_t *row_of_slot, int32_t *tile_starts, int64_t *counts64,
                   int32_t *ws, void *stream);

/* The slot pass of m3_route_build on its own: stable slots pos / row_of_slot i32 [n] of the n = T*k entries of idx32 from
 * the per-64-token-block prefixes blk_base and the expert offsets that m3_balance_route wrote.  k must divide 16. */
int m3_route_assign(const int32_t *idx32, int64_t n, int E, int k, const int32_t *blk_base, const int32_t *offsets,
                    int32_t *pos, int32_t *row_of_slot, void *stream);

/* Expert-parallel exchange plan, on the device (what fastmoe's expert_exchange / global_scatter bookkeeping computes
 * on the host behind _fmoe_general_global_forward with world_size > 1, models/moe/ckpt/custom_moe_layer.py:263-265;
 * experts sharded E_loc per rank, utils/common_config.py:179-185).
 * send_counts i64 [W*E_loc]: this rank's route_build counts by GLOBAL expert id (entry d*E_loc+e goes to local expert e
 * of rank d); recv_counts i64 [W*E_loc]: their all-to-all (entry s*E_loc+e = rows rank s sends to my expert e).
 * Outputs: splits i64 [2W] = rows sent to each rank, then rows received from each rank (the a2a-v sizes: the ONLY
 * values the host reads); regroup i32 [regroup_cap >= rows received]: received rows arrive ordered (src, e), the
 * grouped GEMMs take slot i of the (e, src) order from row regroup[i] (as a_row_idx / c_row_idx); offsets,
 * tile_starts i32 [E_loc+1] as m3_route_build writes them.  W * E_loc <= 4096. */
int m3_ep_plan(const int64_t *send_counts, const int64_t *recv_counts, int W, int E_loc, int64_t *splits,
               int32_t *regroup, int64_t regroup_cap, int32_t *offsets, int32_t *tile_starts, void *stream);
/* The same plan for an exchange with a FIXED row capacity per (source, destination) pair (same reference call site): every
 * pair exchanges exactly `cap` rows, the valid ones first, so the all-to-all has equal splits, the host reads nothing and a
 * step with sharded experts can be captured into a hipGraph.  row_of_slot / pos i32 [n_rows]: this rank's m3_route_build
 * output.  Outputs: regroup i32 [W*cap] (indices into the padded received buffer [W*cap, D]; the first offsets[E_loc]
 * entries are valid), offsets / tile_starts i32 [E_loc+1], pad_idx i32 [W*cap] (token-major entry whose row goes to
 * position q of the padded send buffer: gather with m3_gather_rows(div = k)), unpad_idx i32 [n_rows] (position of entry
 * i's expert output in the returned padded buffer), splits i64 [2W] (true row counts, informational), *overflow i32 set to
 * 1 (never cleared) when some pair routes more than cap rows: such a pair keeps its first cap rows on both sides, the
 * results are incomplete and the caller repeats the step with m3_ep_plan.  W <= 64, W * E_loc <= 4096. */
int m3_ep_plan_fixed(const int64_t *send_counts, const int64_t *recv_counts, int W, int E_loc, int cap,
                     const int32_t *row_of_slot, const int32_t *pos, int64_t n_rows, int64_t *splits,
                     int32_t *regroup, int32_t *offsets, int32_t *tile_starts, int32_t *pad_idx,
                     int32_t *unpad_idx, int32_t *overflow, void *stream);

/* The exchange itself over RCCL (xGMI), for a caller that does not go through torch.distributed: what fastmoe's
 * expert_exchange / global_scatter / global_gather do behind _fmoe_general_global_forward (custom_moe_layer.py:263-265,
 * world_size > 1; experts sharded per utils/common_config.py:179-185).  The only state the library keeps: communicators.
 *   m3_ep_unique_id        rank 0: 128 bytes (ncclUniqueId) to hand to every rank out of band (e.g. a broadcast over gloo)
 *   m3_ep_init             collective over the `world` ranks -> *handle; one communicator per expert-parallel group
 *   m3_ep_exchange_counts  all-to-all of e_loc int64 counts per peer (device buffers [world * e_loc], layout as m3_ep_plan takes)
 *   m3_ep_dispatch         all-to-all-v of rows (row_bytes each): peer p gets in_splits[p] consecutive rows of send_rows and
 *                          delivers out_splits[p] rows into recv_rows, in rank order; the split arrays are HOST arrays of
 *                          `world` row counts - what m3_ep_plan returned in `splits`.  One grouped set of ncclSend / ncclRecv
 *                          pairs on the caller's stream: a distinct peer per xGMI link, nothing chunked into ring steps
 *   m3_ep_return           the way home: the same exchange with the two split vectors swapped
 *   m3_ep_destroy          frees the communicator
 * librccl is opened at the first m3_ep_unique_id / m3_ep_init call (dlopen): the library has no link-time dependency on it.
 * Asynchronous on `stream` like every other entry point; errors of the collective library come back as M3_ERR_LAUNCH with
 * its text in m3_last_error().  (The engine's default exchange is torch.distributed, backend nccl = RCCL, which the two-rank
 * gloo rehearsals can test; BackboneEngine(ep_native=True) routes the row exchanges through these entry points instead.) */
int m3_ep_unique_id(void *out128);
int m3_ep_init(const void *unique_id128, int rank, int world, int *handle);
int m3_ep_destroy(int handle);
int m3_ep_exchange_counts(int handle, const int64_t *send_counts, int64_t *recv_counts, int e_loc, void *stream);
int m3_ep_dispatch(int handle, const void *send_rows, const int64_t *in_splits, void *recv_rows, const int64_t *out_splits,
                   int64_t row_bytes, void *stream);
int m3_ep_return(int handle, const void *send_rows, const int64_t *out_splits, void *recv_rows, const int64_t *in_splits,
                 int64_t row_bytes, void *stream);

/* -------------------------------------------------- GEMM family (a6, a8, a10)
 * C[m, n] = epilogue( sum_k A[arow(m), k] * B[g(m)][n, k] )      ("NT": both K-contiguous)
 *   FMoELinear fwd/dgrad: custom_moe_layer.py:32-33,41,43; qkv/proj Linear:
 *   vision_transformer_moe.py:295,297,303,311; Mlp fc1/fc2 :255-261; PatchEmbed :330-341.
 * Dense call: G = 1, group_offsets = tile_starts = NULL, rows 0..M-1.
 * Grouped call: rows are expert-major slots, group g owns rows
 *   [group_offsets[g], group_offsets[g+1]); tile_starts as produced by m3_route_build;
 *   M = upper bound on rows (T*k); B group stride = N*ldb elements.
 * a_row_idx (i32, optional): source row of A for slot m is a_row_idx[m] / a_row_div
 *   (gather fused into the operand load: MOEScatter).  c_row_idx (optional): destination
 *   row of C for slot m (MOEGather back to token-major).
 * Epilogue, in this order: + bias[g][n] (fp32) ; store pre-activation to pre_out (act
 * dtype) if non-NULL ; act (GELU) ; * gelu'(gelu_grad_pre[m,n]) if non-NULL ;
 * * row_scale[srow(m) / row_scale_div] (fp32) if non-NULL, srow(m) = row_scale_idx[m] when that is given, else crow(m)
 * - the per-sample DropPath factor of the residual branch the GEMM closes, vision_transformer_moe.py:167-185,441,450,
 * or (row_scale_idx = the slot -> routed-entry map, div 1) the gate score of a routed row: the backward of the combine
 * bmm(gate_score, moe_outp), custom_moe_layer.py:298-305, is d moe_outp[t*k+j] = score[t,j] * d out[t], a row scaling
 * that commutes with the expert GEMM behind it, so the scaled copy [T*k, D] is never materialised ;
 * + residual[m,n] (fp32) if non-NULL ; store as c_dtype (M3_F32 or the act dtype).
 * Requirements: K*sizeof(elem) % 16 == 0, lda/ldb rows 16-byte aligned, N % 4 == 0. */
typedef struct {
  const void *A; int64_t lda;
  const int32_t *a_row_idx; int32_t a_row_div;
  const void *B; int64_t ldb;
  void *C; int64_t ldc; int32_t c_dtype;
  const int32_t *c_row_idx;
  const float *bias;               /* [G][N] or NULL */
  void *pre_out; int64_t ld_pre;   /* act dtype, or NULL */
  const void *gelu_grad_pre; int64_t ld_gpre;
  const float *residual; int64_t ld_res;
  int32_t act;
  int64_t M; int32_t N; int32_t K;
  int32_t G;
  const int32_t *group_offsets;    /* [G+1] device, or NULL for dense */
  const int32_t *tile_starts;      /* [G+1] device, or NULL for dense */
  int32_t dtype;                   /* M3_F32 / M3_F16 / M3_BF16: element type of A, B, pre */
  const float *row_scale;          /* fp32 [ceil(rows / row_scale_div)] or NULL */
  int32_t row_scale_div;
  const int32_t *row_scale_idx;    /* i32 [M] device or NULL: which row_scale entry slot m takes (before the division) */
} m3_gemm_args;
int m3_gemm_nt(const m3_gemm_args *args, void *stream);
/* Tuning knob, no reference counterpart: which calls of m3_gemm_nt may take the weight-stationary persistent kernel
 * (fp16, K = 384, N % 128 == 0, M >= 1024; csrc/gemm.hip).  ws_mask bits: 0 plain epilogue, 1 GELU + pre-activation
 * output, 2 GELU'(pre) multiply, 3 fp32 residual, 4 grouped calls as well.  Default 0 (the tiled kernels take every
 * call: measured faster inside the training step); -1 re-reads M3_GEMM_WS from the environment.  Results are the same
 * up to fp32 summation order either way. */
int m3_gemm_set_variant(int ws_mask);
/* Tuning knob, no reference counterpart: which calls of m3_gemm_nt take the 256 x 256-tile kernel for long contractions
 * (16-bit operands, K * 2 bytes a multiple of 128 and >= 1024, at most 64 groups; csrc/gemm_big.hip - the ViT-Base shapes
 * of BASELINE configs[3] / configs[4]).  0 never, 1 every call the kernel can run, 2 (default) those where it measured faster
 * with streamed operands: K >= 2048 and at least 96 tiles; -1 re-reads M3_GEMM_BIG from the environment.  Results are the same up to fp32
 * summation order either way. */
int m3_gemm_set_big(int mode);

/* Fused FFN forward (fp16 activations):
 *   Y[crow(m), :] = (residual[crow(m), :] +) GELU(X[arow(m), :] W1[g]^T + b1[g]) W2[g]^T + b2[g]
 * One launch for `_Expert.forward` (models/moe/ckpt/custom_moe_layer.py:36-44: htoh4 -> GELU -> h4toh through
 * FMoELinear :32-33) with the MOEScatter / MOEGather row movement of `_fmoe_general_global_forward` (:263-265)
 * fused into the operand load / the store, and for the dense `Mlp.forward`
 * (models/moe/ckpt/vision_transformer_moe.py:255-261) with Block's residual add (:450).  The hidden activations
 * [rows, H] are never written: this is the forward of the reference's default activation-checkpointing mode
 * (vision_transformer_moe.py:495-524); a backward that recomputes them is not built (the engine's checkpoint mode re-runs
 * the unfused block forward instead).
 * X [*, D] (row stride ldx elements), W1 [G][H][D], W2p [G][D][H] = W2 with the h index permuted inside every
 * aligned group of 32 (position 8a + 4b + c holds h = 16b + 4a + c, a < 4, b < 2, c < 4: M3_CAST_PERM32 of
 * m3_cast_batch), b1 [G][H], b2 [G][D] fp32 (NULL = 0).  Y [*, D] f16 or fp32 (row stride ldy elements);
 * residual fp32 only with fp32 Y.  pre_out / act_out (optional, f16 [rows in slot order][H]): x W1^T + b1 and its
 * GELU, for a backward that keeps the hidden activations instead of recomputing them.  Grouped call: rows are expert-major slots, group g owns
 * [group_offsets[g], group_offsets[g+1]); x_row_idx / y_row_idx as a_row_idx / c_row_idx of m3_gemm_nt.
 * D in {384, 768}; H a multiple of 64; G <= 64. */
typedef struct {
  const void *X; int64_t ldx; const int32_t *x_row_idx; int32_t x_row_div;
  const void *W1; const void *W2p;
  const float *b1; const float *b2;
  void *Y; int64_t ldy; int32_t y_dtype; const int32_t *y_row_idx;
  const float *residual; int64_t ld_res;
  void *pre_out; void *act_out;    /* f16 [M][H] (row m = slot m), or NULL */
  int64_t M; int32_t D; int32_t H; int32_t G;
  const int32_t *group_offsets;    /* [G+1] device, or NULL for dense */
  int32_t dtype;                   /* M3_F16 */
} m3_ffn_args;
int m3_ffn_fwd(const m3_ffn_args *args, void *stream);

/* Weight gradient ("TN", contraction over rows):
 *   dW[g][n, k] (+)= sum_{m in group g} dC[crow(m), n] * A[arow(m), k]
 *   FMoELinear backward (fastmoe linear_backward behind custom_moe_layer.py:32-33) and
 *   nn.Linear weight grads.  dC [*, N], A [*, K] in the act dtype; dW fp32 [G][N][K].
 * Deterministic split over rows: `splits` partial slabs in ws (fp32
 * [splits][G][N][K]) then m3_wgrad_reduce sums them in order (beta = 1 accumulates
 * into dW, as the joint multi-task backward does: train/train_utils.py:449-457). */
/* A slab reduction that has not been launched yet (the arguments of m3_wgrad_reduce, or with chunk_rows > 0 of
 * m3_wgrad_reduce_grouped): the next m3_wgrad_tn call of the stream can carry it out in front of its own work
 * (m3_wgrad_args.prev), which saves one launch per weight-gradient GEMM. */
typedef struct m3_wgrad_reduce_desc {
  const float *ws; int32_t splits; int64_t elems;
  const int32_t *group_offsets; int32_t G; int32_t chunk_rows;
  float *dW; int32_t beta;
  const float *bias_ws; int64_t bias_elems; float *db; int32_t beta_db;
} m3_wgrad_reduce_desc;
typedef struct {
  const void *dC; int64_t lddc; const int32_t *c_row_idx;
  const void *A; int64_t lda; const int32_t *a_row_idx; int32_t a_row_div;
  int64_t M; int32_t N; int32_t K; int32_t G;
  const int32_t *group_offsets;    /* [G+1] device or NULL (dense: rows 0..M-1) */
  int32_t splits;
  float *ws;                       /* [splits][G][N][K] */
  int32_t dtype;
  float *bias_ws;                  /* optional [splits][G][N]: bias-grad column sums of dC, fused into the
                                      same pass (one extra MFMA row); reduced by m3_wgrad_reduce (same launch as
                                      the weight slabs) or m3_wgrad_bias_reduce */
  int32_t chunk_rows;              /* 0: every group is cut into `splits` equal parts.  > 0 (grouped calls, a
                                      multiple of 64): balanced mode - a work unit is chunk_rows rows of one group,
                                      group g gets ceil(rows_g / chunk_rows) consecutive units of equal size (taken from the
                                      device-resident offsets: a hot expert gets proportionally more workgroups), ws is
                                      [units][N][K] (bias_ws [units][N]) and m3_wgrad_reduce_grouped sums each
                                      group's slabs */
  int32_t units;                   /* balanced mode: slab slots >= sum_g ceil(rows_g / chunk_rows)
                                      (M / chunk_rows + G always suffices) */
  int32_t c_row_div;               /* with c_row_idx: the dC row of slot m is c_row_idx[m] / c_row_div (0 or 1: no division) */
  const float *c_row_scale;        /* with c_row_idx, fp32 or NULL: that row is multiplied by c_row_scale[c_row_idx[m]] on
                                      its way in.  Together: dC = the [T, N] gradient of the combine's OUTPUT, read through
                                      the slot -> routed-entry map with div = k and scaled by the entry's gate score - the
                                      combine backward d moe_outp[t*k+j] = score[t,j] * d out[t]
                                      (custom_moe_layer.py:298-305) without materialising the [T*k, N] copy */
  const m3_wgrad_reduce_desc *prev; /* optional (host pointer, read during the call): the PREVIOUS call's slab reduction, whose
                                      slabs must live in another buffer than `ws`; its blocks run first in this launch.  The
                                      caller reduces the last call of a sequence itself (m3_wgrad_reduce*). */
  float *direct_dW;                /* optional: DIRECT mode (needs splits == 1, chunk_rows == 0, bias_ws NULL): every (group, tile)
                                      then belongs to exactly one workgroup, which adds its result into dW [G][N][K] itself
                                      (direct_beta 1: dW += result, 0: dW = result) - no slabs (ws may be NULL), no reduction
                                      call.  For weights whose gradient is large against the rows contracted (the ViT-Base
                                      experts: 151 MB per layer), where slabs + reduction move four times the result. */
  float *direct_db;                /* direct mode, optional: the bias gradient [G][N] (column sums of dC) likewise */
  int32_t direct_beta, direct_beta_db;
} m3_wgrad_args;
int m3_wgrad_tn(const m3_wgrad_args *args, void *stream);
/* The output tile (n x k) m3_wgrad_tn uses for a shape - 128 x 128, or, with the wide tiles switched on, for fp16
 * 128 x 384 (K = 384, N >= 768 a multiple of 128) / 384 x 128 (N = 384, K >= 768 a multiple of 128) with one
 * 512-thread workgroup per CU.  Callers that pick `splits` / `units` themselves size them (and the slab workspace
 * splits * G * N * K) for ceil(N / tn) * ceil(K / tk) tiles per group.
 * m3_wgrad_set_wide: tuning knob, no reference counterpart: 1 = wide tiles where they apply, 0 = 128 x 128 everywhere
 * (default: measured no faster inside the training step), -1 = re-read M3_WGRAD_WIDE from the environment.  Switch it
 * before sizing any workspace. */
int m3_wgrad_set_wide(int on);
/* Tuning knob, no reference counterpart: which weight-gradient launches take the LDS-DMA kernel (wgrad_dma_kernel,
 * csrc/wgrad.hip: four workgroups per CU, operands global -> LDS directly; fp16 / bf16 / fp32, power-of-two gather divisors, a
 * per-row factor with fp16 / fp32 only).  0 = none (the register-staged kernel everywhere), 2 = every launch the kernel can
 * run, 1 (default) = those where it measured faster with operands streamed from HBM: fp32 always; 16-bit when tiles x groups
 * >= 1024 (one part per group: the ViT-Base experts) or N * K >= 1.5 M elements.  -1 = re-read M3_WGRAD_DMA.  Same results up to
 * fp32 summation order (64 instead of 32 contraction rows per accumulation step in 16 bit). */
int m3_wgrad_set_dma(int on);
/* Tuning knob, no reference counterpart: 256 x 256 output tiles (wgrad_big_kernel, csrc/wgrad.hip: eight waves, one workgroup
 * per CU, two LDS stages filled by LDS-DMA) for 16-bit weights whose N and K are multiples of 256 - the ViT-Base shapes (768,
 * 2304, 3072).  1 = on (default), 0 = 128 x 128 everywhere, -1 = re-read M3_WGRAD_BIG.  m3_wgrad_tile reports (256, 256) for the
 * shapes it takes; switch before sizing workspaces.  Same results up to fp32 summation order. */
int m3_wgrad_set_big(int on);
int m3_wgrad_tile(int N, int K, int dtype, int *tn, int *tk);
/* 1 when m3_wgrad_tn runs a plain call of this shape (one group, no gathers / factor / bias / balanced units / direct mode) with
 * the streaming kernel for K = 16 / 32 - the router's weight, dW_gate = h^T d_logits (custom_moe_layer.py:213-217) - instead of
 * a 128 x 128 MFMA tile padded eightfold: the caller then sizes `splits` for a stream over dC (ops.default_wgrad_splits:
 * at least 64 rows per part, at most 256 parts).  Slab layout and reduction are unchanged.  M3_WGRAD_SKINNY=0 switches it off. */
int m3_wgrad_skinny(int N, int K, int G);
/* balanced mode: dW[g] (+)= sum over group g's units of ws[u] (elems = N*K per group), unit order; optionally the
 * same for the bias slabs (bias_elems = N per group) */
int m3_wgrad_reduce_grouped(const float *ws, const int32_t *group_offsets, int G, int chunk_rows, int64_t elems,
                            float *dW, int beta, const float *bias_ws, int64_t bias_elems, float *db, int beta_db,
                            void *stream);
/* dW (+)= sum over the splits of the weight slabs, fixed order; optionally db (+)= the same over
 * the bias slabs (bias_ws NULL to skip). */
int m3_wgrad_reduce(const float *ws, int splits, int64_t elems, float *dW, int beta,
                    const float *bias_ws, int64_t bias_elems, float *db, int beta_db, void *stream);
/* db[g][n] (+)= sum_s bias_ws[s][g][n], elems = G*N */
int m3_wgrad_bias_reduce(const float *bias_ws, int splits, int64_t elems, float *db, int beta, void *stream);
/* Column sums for bias grads: db[g][n] (+)= sum_{m in group g} dC[crow(m), n].
 * ws fp32 [m3_colsum_ws_elems(M, N, G)]. */
int64_t m3_colsum_ws_elems(int64_t M, int N, int G);
int m3_colsum(const void *dC, int dtype, int64_t lddc, const int32_t *c_row_idx, int64_t M, int N,
              int G, const int32_t *group_offsets, float *ws, float *db, int beta, void *stream);

/* ------------------------------------------------------------- combine (a7)
 * out[t,:] = residual[t,:] + sum_j score[t,j] * y[t*k+j,:]
 *   bmm(gate_score[T,1,k], moe_outp[T,k,D]), custom_moe_layer.py:298-305, fused with
 *   the residual add at vision_transformer_moe.py:450.  y [T*k, D] act dtype (token
 *   major), residual/out fp32 [T, D] (residual may be NULL). */
int m3_combine_fwd(const void *y, int dtype, const float *score, const float *residual,
                   int64_t T, int k, int D, float *out, void *stream);
/* dy[t*k+j,:] = score[t,j] * dout[t,:] (act dtype) ; dscore[t,j] = <dout[t,:], y[t*k+j,:]>.
 * dy may be NULL (d score only): the expert backward can take score * dout straight from dout through
 * m3_gemm_args.row_scale_idx / m3_wgrad_args.c_row_scale instead of reading a materialised dy. */
int m3_combine_bwd(const float *dout, const void *y, int dtype, const float *score,
                   int64_t T, int k, int D, void *dy, float *dscore, void *stream);
/* Input gradient at an MoE layer's branch point, in one pass:
 *   dh[t,:] = sum_j dxe[t*k+j,:]  +  d_logits[t,:] @ w_gate[:D,:]^T
 * the gather-sum of the k routed copies (MOEScatter.backward behind custom_moe_layer.py:254-259) and the gate's share
 * (backward of `inp @ w_gate`, noisy_gate_vmoe.py:91).  dxe [T*k, D] act dtype (token major), d_logits fp32 [T, E],
 * w_gate fp32 [D, E] (row-major, the parameter's own layout; E*(D+4)*4 bytes must fit 64 KB of LDS), dh [T, D] stored as
 * dh_dtype: M3_F32 or the activation dtype (the sum is formed in fp32 either way). */
int m3_combine_gate_bwd(const void *dxe, int dtype, int64_t T, int k, int D, const float *d_logits,
                        const float *w_gate, int E, void *dh, int dh_dtype, void *stream);

/* Row movement of fastmoe's MOEScatter / MOEGather (custom_moe_layer.py:14,263-265) for
 * callers that run an arbitrary expert_fn between them (the fused path does not need it):
 *   dst[i,:] = sum_{j<k} src[idx[i*k+j] / div, :]     (k = 1: plain gather; k > 1: the
 *   backward of the scatter, summing the k routed copies of a token). */
int m3_gather_rows(const void *src, int dtype, const int32_t *idx, int div, int64_t nout, int k,
                   int D, void *dst, void *stream);

/* ----------------------------------------------------------- LayerNorm (a9)
 * nn.LayerNorm(D, eps) on the fp32 residual stream, output in the act dtype
 * (vision_transformer_moe.py:441-442, eps 1e-6 :567).  Saves mean/rstd fp32 [T]. */
int m3_layernorm_fwd(const float *x, int64_t T, int D, const float *gamma, const float *beta,
                     float eps, void *y, int y_dtype, float *mean, float *rstd, void *stream);
/* dx[t,:] = dx_res[t,:] + LN'(dy[t,:]) ; dgamma/dbeta via per-block partials in ws
 * (fp32 [2][m3_ln_bwd_blocks(T, D)][D]) reduced in fixed order (beta = accumulate).
 * dx_act (optional): a copy of dx in the activation dtype for the GEMMs that consume it next. */
int m3_ln_bwd_blocks(int64_t T, int D);
/* dgamma == dbeta == NULL: the per-block partials are left in ws for m3_layernorm_bwd_reduce. */
int m3_layernorm_bwd(const void *dy, int dy_dtype, const float *x, const float *mean,
                     const float *rstd, const float *gamma, const float *dx_res,
                     int64_t T, int D, float *dx, float *ws, float *dgamma, float *dbeta,
                     int beta, void *dx_act, int dx_act_dtype, void *stream);

/* dgamma / dbeta of `count` LayerNorms in ONE launch: layer j (first <= j < first + count) has its partials
 * (written by m3_layernorm_bwd with dgamma = dbeta = NULL) at ws + j * layer_stride floats and its outputs in
 * grads_dev[j] (a DEVICE array); fixed summation order; beta = accumulate.  The reference's autograd produces these one
 * LayerNorm at a time (vision_transformer_moe.py:441-442); here the 24-workgroup reduction behind every LayerNorm
 * backward becomes one launch per group of blocks. */
typedef struct m3_ln_param_grads { float *dgamma; float *dbeta; } m3_ln_param_grads;
int m3_layernorm_bwd_reduce(const float *ws, int64_t layer_stride, int nblk, int D,
                            const m3_ln_param_grads *grads_dev, int first, int count, int beta, void *stream);

/* ----------------------------------------------------------- attention (a8)
 * softmax(q k^T * dh^-0.5) v over the packed qkv activations written by the qkv
 * Linear, Attention.forward vision_transformer_moe.py:299-313.
 * qkv [B*N, 3*C] act dtype laid out [token][3][heads][dh]; o [B*N, C]; lse f32 [B,heads,N].
 * dh in {32, 64}. */
int m3_attention_fwd(const void *qkv, int dtype, int B, int N, int heads, int dh,
                     void *o, float *lse, void *stream);
/* dqkv [B*N, 3*C] from do [B*N, C].  One workgroup per (image, head, 256-key block); for N > 256 the
 * key blocks' dQ contributions go to fp32 slabs in dq_ws ([ceil(N/256)][B*heads][N][dh] =
 * m3_attention_bwd_ws_elems(B,N,heads,dh) floats, contents need no initialisation; NULL when N <= 256)
 * and are summed in key-block order by a second launch. */
int64_t m3_attention_bwd_ws_elems(int B, int N, int heads, int dh);
int m3_attention_bwd(const void *qkv, const void *o, const void *d_o, const float *lse,
                     int dtype, int B, int N, int heads, int dh, void *dqkv, float *dq_ws,
                     void *stream);

/* ------------------------------------------------------------- elementwise
 * cast / transpose helpers for parameters (fp32 master -> act dtype operand copies):
 *   dst[g][c][r] = (T) src[g][r][c]  when transpose, else dst = (T) src. */
int m3_cast_matrix(const float *src, int G, int rows, int cols, int transpose,
                   void *dst, int dst_dtype, void *stream);
/* The same for many matrices in ONE launch (all operand copies of a model per optimizer step):
 * descs_dev is a DEVICE array of n_desc descriptors; a job writes dst (same layout as src) and/or dst_t
 * (transposed, [g][cols][rows]) - either may be NULL - from one read of src; tile_start = running sum of
 * G * ceil(rows/32) * ceil(cols/32) over the preceding descriptors, total_tiles = the full sum. */
#define M3_CAST_PERM32 1    /* dst: last index permuted inside aligned groups of 32 (cols % 32 == 0): position
                               8a + 4b + c holds source column 16b + 4a + c - the k order in which an MFMA 16x16x32
                               accumulator pair becomes the next product's operand (m3_ffn_fwd) */
#define M3_CAST_PERM32_T 2  /* the same for the last index (= source row, rows % 32 == 0) of dst_t */
typedef struct m3_cast_desc {
  const float *src; void *dst; void *dst_t;
  int32_t G, rows, cols;
  int32_t tile_start;
  int32_t flags;                   /* M3_CAST_PERM32 / M3_CAST_PERM32_T */
  int32_t pad1;
} m3_cast_desc;
int m3_cast_batch(const m3_cast_desc *descs_dev, int n_desc, int total_tiles, int dst_dtype, void *stream);
/* dst[i] += src[i] (fp32, n elements): sums the flat gradient buffers of task passes that ran
 * concurrently (the reference accumulates them through autograd, train/train_utils.py:437-457). */
int m3_add_f32(float *dst, const float *src, int64_t n, void *stream);
/* dst(T)[i] = src(f32)[i] ; n elements */
int m3_cast_f32(const float *src, int64_t n, void *dst, int dst_dtype, void *stream);
/* dst(T)[r, :] = row_scale[r / div] * src(f32)[r, :]  (rows x cols, cols % 4 == 0): the gradient that enters a residual
 * branch behind a DropPath (vision_transformer_moe.py:167-185: d branch = scale[sample] * d x), as the activation-dtype
 * operand of the branch's backward GEMMs. */
int m3_scale_rows_cast(const float *src, int64_t rows, int cols, const float *row_scale, int div, void *dst,
                       int dst_dtype, void *stream);
/* patchify: images [B,3,H,W] fp32 NCHW -> rows [B*(H/P)*(W/P), 3*P*P] act dtype in
 * conv-weight order (c, py, px): PatchEmbed conv16/16, vision_transformer_moe.py:330-341 */
int m3_im2row(const float *img, int B, int Cin, int H, int W, int P, void *rows, int dtype,
              void *stream);
/* tokens[b,0,:] = cls + pos[0] ; tokens[b,1+i,:] = patch[b*np+i,:] + pos[1+i]  (fp32 out)
 * forward_features :782-791.  patch fp32 [B*np, D]. */
int m3_assemble_tokens(const float *patch, const float *cls, const float *pos, int B, int np_,
                       int D, float *tokens, void *stream);

/* backward of m3_assemble_tokens: dpatch (act dtype, may be NULL) [B*np, D] = dtok[:,1:,:];
 * dpos f32 [np+1, D] (+)= sum_b dtok[b] ; dcls f32 [D] (+)= sum_b dtok[b,0]  (beta 0/1). */
int m3_tokens_bwd(const float *dtok, int B, int np_, int D, void *dpatch, int dtype, float *dpos,
                  float *dcls, int beta, void *stream);

/* -------------------------------------------------- decoder-head element-wise stage ("custom later")
 * ReLU (optional) + bilinear x2 up-sampling with align_corners = False in one pass, channels-last: the element-wise half of a
 * stage of models/heads/vit_up_head.py:181-214 (F.relu(syncbn(conv(x))) then F.interpolate(.., size = 2x, mode='bilinear')).
 * x [N, H, W, C] (x_dtype; C a multiple of 8 for 16-bit, 4 for fp32) -> y [N, 2H, 2W, C] in x's dtype or M3_F32.
 * Backward: dx [N, H, W, C] (x's dtype) = relu'(x) * transpose-resize(dy), dy in x's dtype or M3_F32; a gather over the 4 x 4
 * window of output gradients each input pixel fed (deterministic). */
int m3_relu_up2x_fwd(const void *x, int x_dtype, int64_t N, int H, int W, int C, int relu, void *y, int y_dtype,
                     void *stream);
int m3_relu_up2x_bwd(const void *dy, int dy_dtype, const void *x, int x_dtype, int64_t N, int H, int W, int C,
                     int relu, void *dx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* M3VIT_HIP_H */
