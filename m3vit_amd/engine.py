"""Fused forward/backward executor for the MoE-ViT backbone hot path on one MI355X.

Host-side mirror of VisionTransformerMoE.forward_features / Block / Attention /
FMoETransformerMLP (models/moe/ckpt/vision_transformer_moe.py:283-313,438-487,780-880,
models/moe/ckpt/custom_moe_layer.py:161-322) as a straight-line sequence of HIP kernel
launches on one stream, with no host synchronisation anywhere (the reference has ~9
`.item()` syncs per MoE block plus fastmoe's count `.cpu()`):

  * fp32 master parameters (reference state_dict names/shapes), fp32 residual stream,
    activations feeding MFMA stored in the activation dtype (fp16 or fp32);
  * every Linear is the hand-written NT GEMM; dgrads use transposed operand copies of
    the weights that are refreshed once per step by `prepare_weights()`;
  * the MoE MLP is gate -> route_build -> grouped FC1 (+bias+GELU, row gather fused into
    the A load) -> grouped FC2 (+bias, scatter to token-major fused into the store) ->
    combine (+residual);
  * activations needed by the backward are kept resident (288 GB HBM: a whole ViT-S pass
    at batch 128 is ~5 GB), nothing is recomputed (the reference's
    --use_checkpointing False mode, run_exps.sh:17).

torch is used for buffer ownership and for the O(E) cv-loss arithmetic only.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

import os

from . import ops
from .ops import M3_ACT_GELU

_ROUTE_FOLD = os.environ.get("M3_ROUTE_FOLD", "1") != "0"       # A/B knob: 0 = the three-launch m3_route_build behind the gate


class _Cfg:
    """Duck-typed config: any object with the attribute names of oracle.BackboneCfg /
    the VisionTransformerMoE ctor works (img_size, patch_size, in_chans, embed_dim, depth,
    num_heads, mlp_ratio, moe_mlp_ratio, moe_experts, moe_top_k, gate_dim, multi_gate,
    gate_task_specific_dim, vmoe_noisy_std)."""


class BackboneEngine:
    def __init__(self, cfg, params: Dict[str, torch.Tensor], batch: int, dtype=torch.float16,
                 device="cuda:0", ep_group=None, ep_world: int = 1, ep_rank: int = 0, share: "BackboneEngine" = None,
                 wgrad_stream: bool = False, checkpoint: bool = False, ep_capacity: float = 0.0,
                 experts_are_local: bool = False, ep_chunks: int = 1, ep_native: bool = False):
        """params: GLOBAL parameters (all E experts).  With ep_world > 1 this rank keeps experts
        [ep_rank*E/W, (ep_rank+1)*E/W) (utils/common_config.py:179-185) and exchanges routed rows with
        the other ranks over torch.distributed (RCCL) - see _experts_fwd_ep.
        share: another engine of the same configuration whose parameters and operand copies this one
        uses (params is ignored); it gets its own activations, scratch and gradient buffer, so the two
        can run different task passes concurrently on different HIP streams.
        wgrad_stream: launch the weight-gradient GEMMs of backward() on a second HIP stream - they hang off
        the dgrad chain (nothing downstream reads them), so they can fill the chain's memory-bound phases.
        checkpoint: the reference's default memory mode (`use_checkpointing`, train_fastmoe.py:178 ->
        torch.utils.checkpoint around every block, vision_transformer_moe.py:495-524): forward keeps only each block's
        input; backward re-runs a block's forward (same kernels, same inputs: bit-identical activations) right before
        its backward.  All blocks then share ONE set of activation buffers.  With sharded experts (ep_world > 1, the
        reference's default combination) the recompute repeats NO collective: what crossed the wire in the forward - the
        exchange plan, the rows received for the local experts and the expert outputs that came back - is kept per MoE
        block, and only the local work (gate, FC1 of the local experts) is re-run.
        ep_capacity (with ep_world > 1; 0 = off): exchange the routed rows with a FIXED capacity of
        ceil(ep_capacity * R / W) rows per (source, destination) pair (R = T * k rows routed by a rank; 1.25 leaves a
        quarter of head room over a uniform routing) instead of the exact a2a-v: equal splits, the plan stays on the device
        (m3_ep_plan_fixed), NOTHING of the exchange is read by the host - one host read per STEP (ep_overflowed()) instead
        of one per MoE layer and pass, and a step that a collective library can capture.  A pair that routes more rows than
        the capacity raises the flag; the step's results are then incomplete and the caller repeats it on the exact path
        (MultiTaskStep does).
        ep_chunks (with ep_world > 1, exact exchange; 1 = one all-to-all-v each way): cut every exchange into this many
        chunks of E_loc / ep_chunks local experts (on every destination) and overlap them with the experts' GEMMs inside ONE
        pass - see _experts_fwd_ep_chunked.  Same results bit for bit.
        ep_native (with ep_world > 1; opt-in, never run on more than one rank on this build box): the count and row exchanges
        go through the library's own RCCL entry points (m3vit_amd/ep_native.py: m3_ep_exchange_counts / m3_ep_dispatch /
        m3_ep_return) instead of torch.distributed's all_to_all_single.
        experts_are_local (with ep_world > 1): the expert tensors in `params` are already this rank's slice [E / W, ..] (a
        module built the way utils/common_config.py:179-185 builds it: moe_experts // world_size experts per rank) and are
        taken as they are - in place - instead of being cut out of global tensors."""
        assert not (checkpoint and wgrad_stream), "checkpoint mode re-uses the activation buffers a wgrad stream may still read"
        self.checkpoint = bool(checkpoint)
        assert dtype in (torch.float16, torch.bfloat16, torch.float32), "activation dtype: float16, bfloat16 or float32"
        self.cfg = cfg
        self.dev = torch.device(device)
        self.dt = dtype
        self.B = batch
        self.ep_group, self.ep_world, self.ep_rank = ep_group, int(ep_world), int(ep_rank)
        self.ep_capacity = float(ep_capacity) if int(ep_world) > 1 else 0.0
        assert not (self.ep_capacity and checkpoint), "fixed-capacity exchange: not with activation checkpointing"
        self.ep_fixed = self.ep_capacity > 0.0                 # switched off by the step runner for the exact repeat of a step
        self.D = cfg.embed_dim
        self.heads = cfg.num_heads
        self.dh = self.D // self.heads
        self.P = cfg.patch_size
        self.hp, self.wp = cfg.img_size[0] // self.P, cfg.img_size[1] // self.P
        self.np_ = self.hp * self.wp
        self.N = self.np_ + 1
        self.T = self.B * self.N
        self.E = cfg.moe_experts
        self.k = cfg.moe_top_k
        self.Hd = int(self.D * cfg.mlp_ratio)
        self.Hm = int(self.D * cfg.moe_mlp_ratio)
        self.R = self.T * self.k
        self.depth = cfg.depth
        assert self.E % self.ep_world == 0, "experts must divide evenly over the EP ranks"
        self.E_loc = self.E // self.ep_world
        self.ep_chunks = int(ep_chunks) if (self.ep_world > 1 and int(ep_chunks) > 1) else 1
        self.ep_native = None
        if ep_native and self.ep_world > 1:
            assert not ep_capacity, "ep_native carries the exact exchange only (the fixed-capacity form is one equal-split all-to-all)"
            if share is not None and getattr(share, "ep_native", None) is not None:
                self.ep_native = share.ep_native                   # one communicator per expert-parallel group and process
            else:
                from .ep_native import NativeExchange
                self.ep_native = NativeExchange(self.ep_rank, self.ep_world, group=ep_group, device=device)
        assert self.E_loc % self.ep_chunks == 0, "ep_chunks must divide the experts per rank"
        is_exp = lambda n: ".mlp.experts." in n                                  # noqa: E731
        lo, hi = self.ep_rank * self.E_loc, (self.ep_rank + 1) * self.E_loc
        if share is not None:
            self.params = share.params
        else:
            if self.ep_world > 1:
                names = [n for n in params if not is_exp(n)] + [n for n in params if is_exp(n)]     # experts last
            else:
                # replicated experts: everything is all-reduced, so order the flat buffer by WHEN a gradient is
                # final in backward() - blocks from the top down, then the embeddings: after
                # backward_blocks(.., b) the prefix flat_grads[:grad_prefix(b)] is final and can be all-reduced
                # while the blocks below still run
                names = sorted(params, key=lambda n: -self._block_of(n))
            cut = self.ep_world > 1 and not experts_are_local
            self.params = {n: (params[n][lo:hi] if (is_exp(n) and cut) else params[n])
                           .to(self.dev, torch.float32).contiguous() for n in names}
            for n, p in self.params.items():
                if is_exp(n):
                    assert p.shape[0] == self.E_loc, f"{n}: {p.shape[0]} experts on this rank, expected {self.E_loc}"
        self.n_dense = sum(p.numel() for n, p in self.params.items() if not is_exp(n))
        self.n_upper = self.grad_prefix(self.split_block)
        # one flat fp32 gradient buffer (views per parameter): zeroing is one memset and the data-parallel
        # sync is one RCCL all-reduce (xGMI is point-to-point: few large collectives)
        total = sum(p.numel() for p in self.params.values())
        self.flat_grads = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.grads, o = {}, 0
        for n, p in self.params.items():
            self.grads[n] = self.flat_grads[o:o + p.numel()].view_as(p)
            o += p.numel()
        dense_only = bool(getattr(cfg, "dense_only", False))
        self.is_moe = [(i % 2 == 1) and not dense_only for i in range(self.depth)]
        self.wg_stream = torch.cuda.Stream(device=self.dev) if (wgrad_stream and self.ep_world == 1) else None
        self._readers, self._ev_pool, self._ev_i = {}, [], 0
        self._alloc()
        if share is not None:
            self.wc, self.wt, self.wgate_c, self.cast_plan = share.wc, share.wt, share.wgate_c, None
        else:
            self.prepare_weights()

    @property
    def split_block(self):
        """first block of the 'upper half' (its gradients occupy flat_grads[:n_upper])"""
        return self.cfg.depth // 2

    @property
    def stem_blocks(self):
        """number of leading blocks before the first MoE block: together with the patch embedding they see neither the
        task id nor the gate, so every task pass of a step computes the same values there (MultiTaskStep share_stem)"""
        return next((i for i in range(self.depth) if self.is_moe[i]), self.depth)

    @staticmethod
    def _block_of(name: str) -> int:
        return int(name.split(".")[1]) if name.startswith("blocks.") else -1

    def grad_prefix(self, block: int) -> int:
        """number of leading elements of flat_grads that belong to blocks >= `block` (0 under expert parallelism,
        where the buffer is ordered dense-first instead)"""
        if self.ep_world > 1:
            return 0
        return sum(p.numel() for n, p in self.params.items() if self._block_of(n) >= block)

    # ------------------------------------------------------------------ buffers
    def _e(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dt, device=self.dev)

    def _alloc(self):
        T, D, R = self.T, self.D, self.R
        f32 = torch.float32
        self.act = []
        shared = {}                                            # checkpoint mode: one buffer per (block kind, name)

        keep = ("x2", "y") if self.ep_world > 1 else ("x2",)     # expert parallel: the returned expert outputs stay per block

        def buf(a, key, *shape, dtype=None):
            if self.checkpoint and key not in keep:
                k_ = (a["_moe"], key)
                if k_ not in shared:
                    shared[k_] = self._e(*shape, dtype=dtype)
                a[key] = shared[k_]
            else:
                a[key] = self._e(*shape, dtype=dtype)

        for i in range(self.depth):
            a = {"_moe": bool(self.is_moe[i])}
            a["x_in"] = None                                   # alias of previous block's output
            buf(a, "mean1", T, dtype=f32); buf(a, "rstd1", T, dtype=f32)
            buf(a, "h1", T, D)
            buf(a, "qkv", T, 3 * D)
            buf(a, "o", T, D)
            buf(a, "lse", self.B, self.heads, self.N, dtype=f32)
            buf(a, "x1", T, D, dtype=f32)
            buf(a, "mean2", T, dtype=f32); buf(a, "rstd2", T, dtype=f32)
            buf(a, "h2", T, D)
            if self.is_moe[i]:
                buf(a, "hid_pre", R, self.Hm)
                buf(a, "hid", R, self.Hm)
                buf(a, "y", R, D)
            else:
                buf(a, "pre", T, self.Hd)
                buf(a, "u", T, self.Hd)
            buf(a, "x2", T, D, dtype=f32)
            self.act.append(a)
        Kp = self.cfg.in_chans * self.P * self.P
        self.rows = self._e(self.B * self.np_, Kp)
        self.patch = self._e(self.B * self.np_, D, dtype=f32)
        self.x0 = self._e(T, D, dtype=f32)
        # backward scratch (reused by every block)
        Hmax = max(self.Hd, self.Hm)
        self.s_dxa = self._e(T, D, dtype=f32)
        self.s_dxb = self._e(T, D, dtype=f32)
        self.s_dx_t = self._e(T, D)
        self.s_dpre = self._e(max(T * self.Hd, R * self.Hm))
        self.s_dh = self._e(T, D)
        self.s_dh32 = self._e(T, D, dtype=f32)
        self.s_do = self._e(T, D)
        self.s_dqkv = self._e(T, 3 * D)
        self.s_dy = self._e(R, D) if self.ep_world > 1 else None       # local experts read score * d x straight from d x
        self.s_dxe = self._e(R, D)
        self.s_dscore = self._e(T, self.k, dtype=f32)
        self.s_dpatch = self._e(self.B * self.np_, D)
        self.ones_k = torch.ones(T, self.k, dtype=f32, device=self.dev)
        # shared workspaces
        wg = 0
        shapes = [(T, self.Hd, D, 1), (T, D, self.Hd, 1), (T, 3 * D, D, 1), (T, D, D, 1),
                  (R, self.Hm, D, self.E), (R, D, self.Hm, self.E), (self.B * self.np_, D, Kp, 1), (T, D, self.E, 1)]
        if self.ep_capacity:                                   # the padded exchange contracts over the capacity bound
            ncap = self.ep_world * ((-(-int(self.ep_capacity * R + self.ep_world - 1) // self.ep_world) + 7) // 8 * 8)
            shapes += [(ncap, self.Hm, D, self.E_loc), (ncap, D, self.Hm, self.E_loc)]
        for (M, N, K, G) in shapes:
            wg = max(wg, ops.wgrad_ws_elems(M, N, K, G, grouped=G > 1, dtype=self.dt))
        # weight-gradient slab reductions ride in front of the NEXT weight-gradient launch of the pass (ops.WgradQueue: two
        # slab workspaces used in turn); with a wgrad side stream every call reduces for itself (its launches are not in
        # one stream order with flush())
        self.wq = ops.WgradQueue(wg, self.dev) if self.wg_stream is None else None
        self.ws_wgrad = self._e(wg, dtype=f32) if self.wq is None else None
        cs = max(int(ops.lib().m3_colsum_ws_elems(T, 3 * D, 1)), int(ops.lib().m3_colsum_ws_elems(T, self.Hd, 1)),
                 int(ops.lib().m3_colsum_ws_elems(R, max(self.Hm, D), self.E)))
        self.ws_colsum = self._e(cs, dtype=f32)
        # LayerNorm parameter-gradient partials: one slot per LayerNorm (2 * block + {0: norm1, 1: norm2}), reduced by ONE
        # launch per backward_blocks() call instead of one 24-workgroup launch behind every LayerNorm backward
        self.ln_nblk = int(ops.lib().m3_ln_bwd_blocks(T, D))
        self.ws_ln = self._e(2 * self.depth, 2, self.ln_nblk, D, dtype=f32)
        self.ln_table = ops.LnGradTable([(self.grads[f"blocks.{i}.{n}.weight"], self.grads[f"blocks.{i}.{n}.bias"])
                                         for i in range(self.depth) for n in ("norm1", "norm2")], self.dev)
        need_dq = int(ops.lib().m3_attention_bwd_ws_elems(self.B, self.N, self.heads, self.dh))
        self.ws_dq = self._e(need_dq, dtype=f32) if need_dq else None
        self.ws_gate_dw = self._e(ops.lib().m3_gate_dw_blocks(T) * self.cfg_d_gate() * self.E, dtype=f32)
        # gate backward through the MFMA GEMMs when E rows are 16-byte multiples
        es = 2 if self.dt in (torch.float16, torch.bfloat16) else 4
        self.gate_via_gemm = (self.E * es) % 16 == 0
        # ... and the gate's share of d h2 folded into the gather-sum of the routed rows' gradients (m3_combine_gate_bwd)
        # when w_gate^T fits its LDS image
        self.fused_gate_dx = self.E * (D + 4) * 4 <= 64 * 1024
        # task-conditioned gate (custom_moe_layer.py:161-181): one shared w_gate [D + gtsd, E] per MoE block
        self.task_cond = self.cfg.gate_task_specific_dim >= 0 and not self.cfg.multi_gate
        self.s_dl_t = self._e(T, self.E)
        self.s_dl = self._e(T, self.E, dtype=f32)
        self.cv_acc = torch.zeros(1, dtype=f32, device=self.dev)
        if self.ep_world > 1:
            # expert-parallel exchange plans: one regroup index per MoE block (kept for the backward; a rank can
            # receive at most what all ranks route) and the pinned landing buffer of the split sizes
            self.ep_regroup = {i: torch.empty(self.ep_world * R, dtype=torch.int32, device=self.dev)
                               for i in range(self.depth) if self.is_moe[i]}
            self.ep_splits_host = torch.empty(2 * self.ep_world, dtype=torch.int64, pin_memory=True)
            if self.ep_chunks > 1:
                # chunked exchange: rows are routed by a CHUNK-MAJOR key - (chunk of the local expert, destination rank,
                # expert inside the chunk) - so that what goes to every rank for one chunk of its experts is one
                # contiguous run of the send buffer, in destination order (one all_to_all_single per chunk)
                C, W, Ec = self.ep_chunks, self.ep_world, self.E_loc // self.ep_chunks
                e = torch.arange(self.E)
                d_, rest = e // self.E_loc, e % self.E_loc
                self.ep_key = ((rest // Ec) * (W * Ec) + d_ * Ec + rest % Ec).to(torch.int32).to(self.dev)
                self.ep_regroup_c = {i: torch.empty(C, W * R, dtype=torch.int32, device=self.dev)
                                     for i in range(self.depth) if self.is_moe[i]}
                self.ep_splits_host_c = torch.empty(C, 2 * W, dtype=torch.int64, pin_memory=True)
            if self.ep_capacity:
                W = self.ep_world
                self.ep_cap = -(-int(self.ep_capacity * R + W - 1) // W)                # rows per (source, destination) pair
                self.ep_cap = (self.ep_cap + 7) // 8 * 8
                self.ep_overflow = torch.zeros(1, dtype=torch.int32, device=self.dev)
                n = W * self.ep_cap
                # static buffers of the padded exchange, per MoE block (a captured step replays on fixed addresses)
                # (zeroed once: the rows of a pair's share that no routed row fills travel over the wire as they are - never read
                # back, unpad_idx does not select them, but they should not be whatever the allocator left there)
                z = lambda *shape: torch.zeros(*shape, dtype=self.dt, device=self.dev)                   # noqa: E731
                self.ep_fx = {i: dict(x_send=z(n, D), x_recv=z(n, D), hid_pre=z(n, self.Hm),
                                      hid=z(n, self.Hm), y_recv=z(n, D), y_back=z(n, D),
                                      recv_counts=torch.zeros(self.E, dtype=torch.int64, device=self.dev), plan=None)
                              for i in range(self.depth) if self.is_moe[i]}
                self.ep_fx_bwd = dict(dy_send=z(n, D), dy_recv=z(n, D), dhp=z(n, self.Hm),
                                      dx_recv=z(n, D), dx_back=z(n, D))

    def cfg_d_gate(self):
        g = self.cfg.gate_task_specific_dim
        return self.D if g < 0 else self.D + g

    # ------------------------------------------------------------- weight copies
    def _linear_names(self):
        names = ["patch_embed.proj"]
        for i in range(self.depth):
            b = f"blocks.{i}."
            names += [b + "attn.qkv", b + "attn.proj"]
            if self.is_moe[i]:
                names += [b + "mlp.experts.htoh4", b + "mlp.experts.h4toh"]
            else:
                names += [b + "mlp.fc1", b + "mlp.fc2"]
        return names

    def prepare_weights(self):
        """Refresh the activation-dtype operand copies W [.., N, K] and W^T [.., K, N] of every
        Linear / FMoELinear weight (and the w_gate copies) from the fp32 masters: one batched launch
        per optimizer step."""
        if not hasattr(self, "wc"):
            self.wc, self.wt, self.wgate_c = {}, {}, {}
            jobs = []
            for n in self._linear_names():
                w = self.params[n + ".weight"]
                w2 = w.reshape(w.shape[0], -1) if n == "patch_embed.proj" else w
                self.wc[n] = w2 if self.dt == torch.float32 else torch.empty_like(w2, dtype=self.dt)
                if n != "patch_embed.proj":
                    self.wt[n] = torch.empty(*w2.shape[:-2], w2.shape[-1], w2.shape[-2], dtype=self.dt, device=self.dev)
                plain = None if self.dt == torch.float32 else self.wc[n]
                if plain is not None or n in self.wt:
                    jobs.append((w2, plain, self.wt.get(n)))          # both copies from one read of the master
            if self.dt != torch.float32:
                for n, p in self.params.items():
                    if n.endswith("w_gate"):
                        self.wgate_c[n] = torch.empty_like(p, dtype=self.dt)
                        jobs.append((p, self.wgate_c[n], None))
            self.cast_plan = ops.CastPlan(jobs, self.dt) if jobs else None
        if self.cast_plan is not None:
            self.cast_plan.run()

    def zero_grad(self):
        self.flat_grads.zero_()
        if self.wq is not None:
            # a step that aborted between a queued weight-gradient call and flush() leaves a slab-reduction descriptor behind
            # whose slabs were never (fully) written; it must not ride in front of THIS step's first launch
            self.wq.reset()

    # ------------------------------------------------------------- checkpoints
    def state_dict(self):
        """This rank's parameters as CPU tensors under the reference's key names (experts: the LOCAL slice under
        expert parallelism, as train_fastmoe's rank shards hold them - m3vit_amd/checkpoint.py)."""
        from collections import OrderedDict
        return OrderedDict((n, p.detach().cpu().clone()) for n, p in self.params.items())

    def load_state(self, state, strict: bool = True):
        """Copy a state_dict (local expert slices, e.g. from checkpoint.to_backbone_state) into the fp32 masters
        and refresh the operand copies.  Returns the keys of `state` that were not used."""
        missing = [n for n in self.params if n not in state]
        if strict and missing:
            raise KeyError(f"state_dict lacks {missing[:5]}{'...' if len(missing) > 5 else ''}")
        for n, p in self.params.items():
            if n in state:
                src = state[n]
                if tuple(src.shape) != tuple(p.shape):
                    raise ValueError(f"{n}: checkpoint shape {tuple(src.shape)} != parameter shape {tuple(p.shape)}")
                p.copy_(src.to(p.device, torch.float32))
        self.prepare_weights()
        return [k for k in state if k not in self.params]

    # ------------------------------------------------------------------ forward
    def _gate_weight(self, i, task_id):
        b = f"blocks.{i}.mlp.gate."
        if self.cfg.multi_gate:
            return b + f"{task_id}.w_gate"           # custom_moe_layer.py:213-214
        return b + "w_gate"

    def _task_feature(self, task_id):
        """tsf = gate_task_represent(one_hot(task_id)) (vision_transformer_moe.py:793-797; new_Mlp :263-281:
        fc1 -> GELU -> fc2 -> LayerNorm) and the per-block logit bias tsf @ w_gate[D:], which is what
        cat(inp, tsf.repeat(T, 1)) @ w_gate (custom_moe_layer.py:176-179) adds to every token's logits.
        A [num_tasks] -> [gtsd] vector MLP: kept as a torch autograd graph over leaf copies of its six
        parameters; backward() finishes it from the d(logit bias) the gate kernels produce."""
        import torch.nn.functional as F
        p, D = self.params, self.D
        names = [f"gate_task_represent.{n}" for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias",
                                                      "norm.weight", "norm.bias")]
        leaves = [p[n].detach().requires_grad_() for n in names]
        with torch.enable_grad():
            h = F.gelu(leaves[0][:, task_id] + leaves[1])          # fc1(one_hot(task)) = column task of fc1.weight
            h = F.linear(h, leaves[2], leaves[3])
            tsf = F.layer_norm(h, (h.shape[-1],), leaves[4], leaves[5], 1e-6)
        self._tsf = dict(names=names, leaves=leaves, tsf=tsf)
        tv = tsf.detach()
        return {i: tv @ p[self._gate_weight(i, task_id)][D:] for i in range(self.depth) if self.is_moe[i]}

    def _task_feature_block_bwd(self, a, dl, st):
        """One MoE block's share of the task-conditioned gate backward, written straight after the block's d logits
        exist: w_gate[D:] += outer(tsf, colsum(d logits)) and d tsf += w_gate[D:] @ colsum(d logits).  It must be part of
        backward_blocks(): a data-parallel step all-reduces a block's gradient slice as soon as backward_blocks() has
        passed the block (step.py), so nothing may be added to it later."""
        p, gr, D = self.params, self.grads, self.D
        tv = self._tsf["tsf"].detach()
        dlb = ops.colsum(dl, torch.empty(self.E, device=self.dev), ws=self.ws_colsum)
        gr[a["wname"]][D:].addr_(tv, dlb)
        if st.get("d_tsf") is None:
            st["d_tsf"] = torch.zeros_like(tv)
        st["d_tsf"] += p[a["wname"]][D:] @ dlb

    def _task_feature_bwd(self, d_tsf):
        """d tsf -> the gate_task_represent parameters (block -1: the last gradient slice)."""
        t = self._tsf
        if d_tsf is not None:
            for n, g in zip(t["names"], torch.autograd.grad(t["tsf"], t["leaves"], d_tsf)):
                self.grads[n] += g
        self._tsf = None

    def _block_forward(self, i, x, loss_acc):
        """block i on residual stream x -> its output buffer (vision_transformer_moe.py:438-562: pre-LN attention +
        MLP / MoE branch).  Called by forward() and, in checkpoint mode, again by backward_blocks() right before the
        block's backward (loss_acc None there: the balance loss was already counted)."""
        task_id, tsf_bias, noises, path_scales = self._fwd_ctx
        P_, p = self.P, self.params
        B, T, D = self.B, self.T, self.D
        N = self.N
        a = self.act[i]
        b = f"blocks.{i}."
        a["x_in"] = x
        ops.layernorm_fwd(x, p[b + "norm1.weight"], p[b + "norm1.bias"], a["h1"], a["mean1"], a["rstd1"])
        ops.gemm_nt(a["h1"], self.wc[b + "attn.qkv"], a["qkv"], bias=p[b + "attn.qkv.bias"])
        ops.attention_fwd(a["qkv"], B, self.N, self.heads, self.dh, a["o"], a["lse"])
        ps = None if path_scales is None else path_scales.get(i)
        a["ps"] = ps
        sa, sm = (None, None) if ps is None else ps
        ops.gemm_nt(a["o"], self.wc[b + "attn.proj"], a["x1"], bias=p[b + "attn.proj.bias"], residual=x,
                    row_scale=sa, row_scale_div=N)
        ops.layernorm_fwd(a["x1"], p[b + "norm2.weight"], p[b + "norm2.bias"], a["h2"], a["mean2"], a["rstd2"])
        if not self.is_moe[i]:
            ops.gemm_nt(a["h2"], self.wc[b + "mlp.fc1"], a["u"], bias=p[b + "mlp.fc1.bias"], act=M3_ACT_GELU,
                        pre_out=a["pre"])
            ops.gemm_nt(a["u"], self.wc[b + "mlp.fc2"], a["x2"], bias=p[b + "mlp.fc2.bias"], residual=a["x1"],
                        row_scale=sm, row_scale_div=N)
        else:
            wname = self._gate_weight(i, task_id)
            a["wname"] = wname
            wg = p[wname]
            wg_tok = wg if wg.shape[0] == D else wg[:D]
            noise = None if noises is None else noises.get(i)
            std = (self.cfg.vmoe_noisy_std / self.E) if noise is not None else 0.0
            # gate + balance loss (importance, load, cv^2 and its gradient) + dispatch metadata: three launches (the routing
            # histogram comes out of the gate kernel, its scan rides in the balance launch)
            g = ops.gate_fwd(a["h2"], wg_tok, self.k, logit_bias=None if tsf_bias is None else tsf_bias[i],
                             noise=noise, noise_std=std, dense=True, loss_acc=loss_acc, route=self.ep_world == 1 and _ROUTE_FOLD)
            a["gate"] = g
            if self.ep_world > 1:
                self._experts_fwd_ep(i, a, g, recompute=loss_acc is None)
            else:
                r = g["route"] if g["route"] is not None else ops.route_build(g["idx32"], self.E)
                a["route"] = r
                ops.gemm_nt(a["h2"], self.wc[b + "mlp.experts.htoh4"], a["hid"], M=self.R,
                            bias=p[b + "mlp.experts.htoh4.bias"], act=M3_ACT_GELU, pre_out=a["hid_pre"],
                            a_row_idx=r.row_of_slot, a_row_div=self.k, group_offsets=r.offsets,
                            tile_starts=r.tile_starts)
                ops.gemm_nt(a["hid"], self.wc[b + "mlp.experts.h4toh"], a["y"], M=self.R,
                            bias=p[b + "mlp.experts.h4toh.bias"], c_row_idx=r.row_of_slot,
                            group_offsets=r.offsets, tile_starts=r.tile_starts)
            if sm is not None:                          # out = x1 + scale[sample] * sum_j score_j y_j
                a["sm_tok"] = sm.view(B, 1).expand(B, N).reshape(T, 1).contiguous()
                a["score_s"] = g["score"] * a["sm_tok"]
            ops.combine_fwd(a["y"], a["score_s"] if sm is not None else g["score"], a["x1"], a["x2"])
        return a["x2"]

    def forward(self, images: torch.Tensor, task_id: Optional[int], tsf_bias=None, noises=None, path_scales=None,
                stem_of: "BackboneEngine" = None):
        """Returns (tokens fp32 [B,N,D], total_cv_loss).  noises: {block: [T,E]} caller-supplied N(0,1).
        Task-conditioned configs compute the per-block logit bias tsf @ w_gate[D:] here (tsf_bias overrides
        it with a caller-supplied {block: [E]} and then no gradient flows to the task embedding).
        path_scales: {block: (attn [B], mlp [B])} fp32 per-sample DropPath factors of the block's two residual
        branches (stochastic depth, vision_transformer_moe.py:167-185: floor(keep + U) / keep, drawn by the caller -
        pretrain/configs/deit_moe_small.yaml:51 trains with drop_path > 0); the factor rides on the epilogue of the
        GEMM that closes the branch (proj, fc2) or on the combine scores (MoE), and on the gradient entering the
        branch in backward.
        stem_of: an engine (this one or another of the same step) whose forward_stem() already holds the patch embedding
        and the blocks below the first MoE block for THESE images; this pass starts at the first MoE block from that
        output and its backward stops there (backward_blocks(depth - 1, stem_blocks); the caller adds the passes'
        d x and runs the stem's backward once - MultiTaskStep)."""
        x = self.forward_begin(images, task_id, tsf_bias=tsf_bias, noises=noises, path_scales=path_scales, stem_of=stem_of)
        for i in range(self.stem_blocks if stem_of is not None else 0, self.depth):
            x = self._block_forward(i, x, self.cv_acc)
        return self.forward_end(x)

    def forward_stem(self, images: torch.Tensor):
        """patch embedding, cls / pos and the blocks below the first MoE block (task-independent: no gate, no task id;
        the reference's DropPath schedule gives block 0 rate 0, vision_transformer_moe.py:761).  Returns their output."""
        self._fwd_ctx = (None, None, None, None)
        x = self._embed(images)
        for i in range(self.stem_blocks):
            x = self._block_forward(i, x, None)
        return x

    def _embed(self, images):
        ops.im2row(images, self.P, self.rows)
        ops.gemm_nt(self.rows, self.wc["patch_embed.proj"], self.patch, bias=self.params["patch_embed.proj.bias"])
        ops.assemble_tokens(self.patch, self.params["cls_token"], self.params["pos_embed"], self.B, self.np_, self.D, self.x0)
        return self.x0

    # The forward in resumable pieces (patch embedding / one block at a time / result), so that a step runner can
    # interleave the blocks of several task passes on the host (expert parallelism: while one pass waits for its
    # exchange's split sizes, the other pass's queued kernels keep the GPU busy).
    def forward_begin(self, images: torch.Tensor, task_id: Optional[int], tsf_bias=None, noises=None, path_scales=None,
                      stem_of: "BackboneEngine" = None):
        self._tsf = None
        if self.task_cond and tsf_bias is None and task_id is not None:
            tsf_bias = self._task_feature(task_id)
        self.cv_acc.zero_()
        self.task_id = task_id
        self._fwd_ctx = (task_id, tsf_bias, noises, path_scales)
        if stem_of is None:
            return self._embed(images)
        s = self.stem_blocks
        assert path_scales is None or all(path_scales.get(i) is None for i in range(s)), "a shared stem takes no DropPath draw"
        return stem_of.act[s - 1]["x2"] if s > 0 else stem_of.x0

    def forward_end(self, x):
        # total cv_loss = sum over MoE blocks of cv^2(importance) + cv^2(load)  (vision_transformer_moe.py:453-459,540)
        return x.view(self.B, self.N, self.D), self.cv_acc[0].clone()

    # ------------------------------------------------------------- expert parallel
    def _a2a(self, x, in_splits, out_splits):
        import torch.distributed as dist
        out = torch.empty((sum(out_splits),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        if self.ep_native is not None:
            self.ep_native.dispatch_async(out, x.contiguous(), out_splits, in_splits).wait()
            return out
        dist.all_to_all_single(out, x.contiguous(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                               group=self.ep_group)
        return out

    def _experts_fwd_ep(self, i, a, g, recompute=False):
        """EP forward of one MoE layer: ONE count exchange + ONE row exchange each way (m3vit_amd/ep.py has
        the same logic for the module API).  Rows are routed by GLOBAL expert id, so the expert-major send
        buffer is already grouped by destination rank; received rows are regrouped from (src, expert) to
        (expert, src) order for the local grouped GEMMs.
        recompute (checkpoint mode, called from backward_blocks): the gate has just been re-run on the same input, so the
        routing is the forward's; the plan, the received rows and the returned outputs a["y"] were kept - only FC1 of the
        local experts is run again (its hidden activations are what the backward needs), no collective."""
        import torch.distributed as dist
        b = f"blocks.{i}."
        p, k, D, dev = self.params, self.k, self.D, self.dev
        if self.ep_fixed:
            return self._experts_fwd_ep_fixed(i, a, g)
        if self.ep_chunks > 1 and not recompute:
            return self._experts_fwd_ep_chunked(i, a, g)
        if recompute:
            ep = a["ep"]
            n = ep["n"]
            if n > 0:
                ep["hid_pre"], ep["hid"] = self._e(n, self.Hm), self._e(n, self.Hm)
                ops.gemm_nt(ep["x_recv"], self.wc[b + "mlp.experts.htoh4"], ep["hid"], M=n,
                            bias=p[b + "mlp.experts.htoh4.bias"], act=M3_ACT_GELU, pre_out=ep["hid_pre"],
                            a_row_idx=ep["rg"], a_row_div=1, group_offsets=ep["offsets"], tile_starts=ep["tile_starts"])
            return
        r = ops.route_build(g["idx32"], self.E, want_counts64=True)
        a["route"] = r
        x_send = self._e(self.R, D)
        ops.gather_rows(a["h2"], r.row_of_slot, x_send, div=k)
        if self.ep_native is not None:
            recv = self.ep_native.exchange_counts(r.counts64)
        else:
            recv = torch.empty_like(r.counts64)
            dist.all_to_all_single(recv, r.counts64, group=self.ep_group)
        # the plan (regroup index, expert-major offsets, tile prefix) is built on the device; the host reads the
        # 2 W split sizes the a2a-v API needs and nothing else
        plan = ops.ep_plan(r.counts64, recv, self.ep_world, self.E_loc, self.ep_regroup[i], splits_host=self.ep_splits_host)
        n = plan.n_recv
        x_recv = self._a2a(x_send, plan.in_splits, plan.out_splits)
        ep = dict(plan=plan, n=n, rg=plan.regroup, offsets=plan.offsets, tile_starts=plan.tile_starts)
        y_recv = self._e(n, D)
        ep["x_recv"] = x_recv
        if n > 0:
            # the (src, expert) -> (expert, src) regroup is the A-row gather of FC1 and the C-row scatter of FC2:
            # expert-major slot i reads x_recv[rg[i]] and writes y_recv[rg[i]] - no regrouped copies
            ep["hid_pre"], ep["hid"] = self._e(n, self.Hm), self._e(n, self.Hm)
            ops.gemm_nt(x_recv, self.wc[b + "mlp.experts.htoh4"], ep["hid"], M=n,
                        bias=p[b + "mlp.experts.htoh4.bias"], act=M3_ACT_GELU, pre_out=ep["hid_pre"],
                        a_row_idx=ep["rg"], a_row_div=1, group_offsets=ep["offsets"], tile_starts=ep["tile_starts"])
            ops.gemm_nt(ep["hid"], self.wc[b + "mlp.experts.h4toh"], y_recv, M=n, bias=p[b + "mlp.experts.h4toh.bias"],
                        c_row_idx=ep["rg"], group_offsets=ep["offsets"], tile_starts=ep["tile_starts"])
        y_send = self._a2a(y_recv, plan.out_splits, plan.in_splits)
        ops.gather_rows(y_send, r.pos, a["y"])                       # back to token-major [T*k, D]
        if self.checkpoint:                                          # local hidden activations: recomputed in backward
            ep["hid_pre"] = ep["hid"] = None
        a["ep"] = ep

    # ------------------------------------------------------------------ expert parallel, exchange overlapped inside ONE pass
    def _a2a_async(self, out, x, out_splits, in_splits):
        import torch.distributed as dist
        if self.ep_native is not None:
            return self.ep_native.dispatch_async(out, x, out_splits, in_splits)
        return dist.all_to_all_single(out, x, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.ep_group,
                                      async_op=True)

    def _experts_fwd_ep_chunked(self, i, a, g):
        """_experts_fwd_ep with every exchange cut into ep_chunks all-to-all-v's, chunk c = the rows for local experts
        [c E_loc / C, (c + 1) E_loc / C) of EVERY rank (custom_moe_layer.py:263-265): all chunks' row
        exchanges are queued on the collective library's stream at once; the grouped FC1 / FC2 of chunk c start when ITS rows
        have arrived - the later chunks are still in flight - and its outputs start their way home under the next chunk's GEMMs.
        Exposed per direction: one chunk's exchange instead of the whole.  The count exchange and the plans (C m3_ep_plan
        launches, ONE host read of the C * 2 W split sizes) are as before.  What the backward and a checkpoint recompute need
        is kept in the unchunked form - one received buffer (the chunks side by side), one regroup index, offsets and tile
        prefix over all local experts - so every row keeps its expert-major position and the results are bit-identical to the
        one-exchange path."""
        import torch.distributed as dist
        b = f"blocks.{i}."
        p, k, D, dev = self.params, self.k, self.D, self.dev
        C, W = self.ep_chunks, self.ep_world
        Ec = self.E_loc // C
        key = self.ep_key[g["idx32"].reshape(-1).long()].view(-1, k).contiguous()          # chunk-major routing keys
        r = ops.route_build(key, self.E, want_counts64=True)
        a["route"] = r
        x_send = self._e(self.R, D)
        ops.gather_rows(a["h2"], r.row_of_slot, x_send, div=k)
        send = r.counts64.view(C, W, Ec)
        snd_dm = send.permute(1, 0, 2).contiguous()                                         # destination-major for the count exchange
        if self.ep_native is not None:
            rcv_dm = self.ep_native.exchange_counts(snd_dm.view(-1)).view_as(snd_dm)
        else:
            rcv_dm = torch.empty_like(snd_dm)
            dist.all_to_all_single(rcv_dm.view(-1), snd_dm.view(-1), group=self.ep_group)
        recv = rcv_dm.permute(1, 0, 2).contiguous()                                         # [C][source][expert of the chunk]
        plans = ops.ep_plan_chunks(send.contiguous(), recv, W, Ec, self.ep_regroup_c[i], self.ep_splits_host_c)
        ns = [sum(pl.in_splits) for pl in plans]
        nr = [pl.n_recv for pl in plans]
        sb = [sum(ns[:c]) for c in range(C + 1)]
        rb = [sum(nr[:c]) for c in range(C + 1)]
        n = rb[C]
        x_recv, y_recv = self._e(n, D), self._e(n, D)
        hid_pre, hid = self._e(n, self.Hm), self._e(n, self.Hm)
        y_send = self._e(self.R, D)
        works = [self._a2a_async(x_recv[rb[c]:rb[c + 1]], x_send[sb[c]:sb[c + 1]], plans[c].out_splits, plans[c].in_splits)
                 for c in range(C)]
        back = []
        w1, b1 = self.wc[b + "mlp.experts.htoh4"], p[b + "mlp.experts.htoh4.bias"]
        w2, b2 = self.wc[b + "mlp.experts.h4toh"], p[b + "mlp.experts.h4toh.bias"]
        for c in range(C):
            works[c].wait()
            pl, es = plans[c], slice(c * Ec, (c + 1) * Ec)
            if nr[c] > 0:
                ops.gemm_nt(x_recv[rb[c]:rb[c + 1]], w1[es], hid[rb[c]:rb[c + 1]], M=nr[c], bias=b1[es], act=M3_ACT_GELU,
                            pre_out=hid_pre[rb[c]:rb[c + 1]], a_row_idx=pl.regroup, a_row_div=1, group_offsets=pl.offsets,
                            tile_starts=pl.tile_starts)
                ops.gemm_nt(hid[rb[c]:rb[c + 1]], w2[es], y_recv[rb[c]:rb[c + 1]], M=nr[c], bias=b2[es], c_row_idx=pl.regroup,
                            group_offsets=pl.offsets, tile_starts=pl.tile_starts)
            back.append(self._a2a_async(y_send[sb[c]:sb[c + 1]], y_recv[rb[c]:rb[c + 1]], pl.in_splits, pl.out_splits))
        # the unchunked view of the plan for the backward / a checkpoint recompute: chunk c's rows sit at rb[c] of the received
        # buffer and its experts at c * Ec of the local experts
        rg = torch.cat([plans[c].regroup + rb[c] for c in range(C)]) if n else self.ep_regroup_c[i][0, :0]
        offs = torch.cat([plans[c].offsets[:-1] + rb[c] for c in range(C)] +
                         [torch.full((1,), n, dtype=torch.int32, device=dev)])
        tb = [0]
        tsc = [plans[c].tile_starts for c in range(C)]
        ts_all, base_t = [], torch.zeros((), dtype=torch.int32, device=dev)
        for c in range(C):
            ts_all.append(tsc[c][:-1] + base_t)
            base_t = base_t + tsc[c][-1]
        ts_all.append(base_t.reshape(1))
        ep = dict(plan=None, chunked=True, plans=plans, sb=sb, rb=rb, n=n, rg=rg.contiguous(), offsets=offs.contiguous(),
                  tile_starts=torch.cat(ts_all).contiguous(), x_recv=x_recv, hid_pre=hid_pre, hid=hid)
        for w_ in back:
            w_.wait()
        ops.gather_rows(y_send, r.pos, a["y"])                       # back to token-major [T*k, D]
        if self.checkpoint:                                          # local hidden activations: recomputed in backward
            ep["hid_pre"] = ep["hid"] = None
        a["ep"] = ep

    def _experts_bwd_ep_chunked(self, i, a):
        """mirror of _experts_fwd_ep_chunked: the d y rows travel in the same chunks; chunk c's input-gradient GEMMs run while the
        later chunks are in flight and its d x rows go home under the next chunk's GEMMs; the weight gradients (all local
        experts at once, on the side-by-side buffers - the same launches as the one-exchange path) run last, under the
        returning exchanges."""
        b = f"blocks.{i}."
        D, r, ep = self.D, a["route"], a["ep"]
        C, Ec = self.ep_chunks, self.E_loc // self.ep_chunks
        plans, sb, rb, n = ep["plans"], ep["sb"], ep["rb"], ep["n"]
        dy_send = ops.gather_rows(self.s_dy, r.row_of_slot, self._e(self.R, D))
        dy_recv, dx_recv, dhp = self._e(n, D), self._e(n, D), self._e(n, self.Hm)
        dx_send = self._e(self.R, D)
        works = [self._a2a_async(dy_recv[rb[c]:rb[c + 1]], dy_send[sb[c]:sb[c + 1]], plans[c].out_splits, plans[c].in_splits)
                 for c in range(C)]
        back = []
        wt2, wt1 = self.wt[b + "mlp.experts.h4toh"], self.wt[b + "mlp.experts.htoh4"]
        for c in range(C):
            works[c].wait()
            pl, es = plans[c], slice(c * Ec, (c + 1) * Ec)
            rows = slice(rb[c], rb[c + 1])
            if rb[c + 1] > rb[c]:
                ops.gemm_nt(dy_recv[rows], wt2[es], dhp[rows], M=rb[c + 1] - rb[c], gelu_grad_pre=ep["hid_pre"][rows],
                            a_row_idx=pl.regroup, a_row_div=1, group_offsets=pl.offsets, tile_starts=pl.tile_starts)
                ops.gemm_nt(dhp[rows], wt1[es], dx_recv[rows], M=rb[c + 1] - rb[c], c_row_idx=pl.regroup,
                            group_offsets=pl.offsets, tile_starts=pl.tile_starts)
            back.append(self._a2a_async(dx_send[sb[c]:sb[c + 1]], dx_recv[rows], pl.in_splits, pl.out_splits))
        if n > 0:
            rg = ep["rg"]
            self._wgrad(dy_recv, ep["hid"], b + "mlp.experts.h4toh.weight", M=n, c_row_idx=rg,
                        group_offsets=ep["offsets"], bias=b + "mlp.experts.h4toh.bias")
            self._wgrad(dhp, ep["x_recv"], b + "mlp.experts.htoh4.weight", M=n, a_row_idx=rg, a_row_div=1,
                        group_offsets=ep["offsets"], bias=b + "mlp.experts.htoh4.bias")
        for w_ in back:
            w_.wait()
        ops.gather_rows(dx_send, r.pos, self.s_dxe)

    def _a2a_equal(self, x, out):
        """all-to-all with equal splits (the fixed-capacity exchange): no sizes, nothing for the host to read"""
        import torch.distributed as dist
        dist.all_to_all_single(out, x, group=self.ep_group)
        return out

    def _experts_fwd_ep_fixed(self, i, a, g):
        """_experts_fwd_ep with ep_cap rows per (source, destination) pair: the send buffer is the padded [W * cap, D]
        image gathered straight from h2 through pad_idx, rows arrive at source * cap + ..., the grouped GEMMs take the
        device-resident regroup / offsets / tile prefix with M = the capacity bound (surplus workgroups retire on the
        device-side tile prefix), and the outputs come home through unpad_idx."""
        import torch.distributed as dist
        b = f"blocks.{i}."
        p, k = self.params, self.k
        fx = self.ep_fx[i]
        W, cap = self.ep_world, self.ep_cap
        n = W * cap
        r = ops.route_build(g["idx32"], self.E, want_counts64=True)
        a["route"] = r
        dist.all_to_all_single(fx["recv_counts"], r.counts64, group=self.ep_group)
        plan = fx["plan"] = ops.ep_plan_fixed(r.counts64, fx["recv_counts"], W, self.E_loc, cap, r, self.ep_overflow,
                                              bufs=fx["plan"])
        ops.gather_rows(a["h2"], plan.pad_idx, fx["x_send"], div=k)
        self._a2a_equal(fx["x_send"], fx["x_recv"])
        ops.gemm_nt(fx["x_recv"], self.wc[b + "mlp.experts.htoh4"], fx["hid"], M=n, bias=p[b + "mlp.experts.htoh4.bias"],
                    act=M3_ACT_GELU, pre_out=fx["hid_pre"], a_row_idx=plan.regroup, a_row_div=1,
                    group_offsets=plan.offsets, tile_starts=plan.tile_starts)
        ops.gemm_nt(fx["hid"], self.wc[b + "mlp.experts.h4toh"], fx["y_recv"], M=n, bias=p[b + "mlp.experts.h4toh.bias"],
                    c_row_idx=plan.regroup, group_offsets=plan.offsets, tile_starts=plan.tile_starts)
        self._a2a_equal(fx["y_recv"], fx["y_back"])
        ops.gather_rows(fx["y_back"], plan.unpad_idx, a["y"])            # back to token-major [T*k, D]
        a["ep"] = dict(fixed=True)

    def _experts_bwd_ep_fixed(self, i, a):
        b = f"blocks.{i}."
        fx, fb = self.ep_fx[i], self.ep_fx_bwd
        plan = fx["plan"]
        n = self.ep_world * self.ep_cap
        rg = plan.regroup
        ops.gather_rows(self.s_dy, plan.pad_idx, fb["dy_send"])
        self._a2a_equal(fb["dy_send"], fb["dy_recv"])
        self._wgrad(fb["dy_recv"], fx["hid"], b + "mlp.experts.h4toh.weight", M=n, c_row_idx=rg,
                    group_offsets=plan.offsets, bias=b + "mlp.experts.h4toh.bias")
        ops.gemm_nt(fb["dy_recv"], self.wt[b + "mlp.experts.h4toh"], fb["dhp"], M=n, gelu_grad_pre=fx["hid_pre"],
                    a_row_idx=rg, a_row_div=1, group_offsets=plan.offsets, tile_starts=plan.tile_starts)
        self._wgrad(fb["dhp"], fx["x_recv"], b + "mlp.experts.htoh4.weight", M=n, a_row_idx=rg, a_row_div=1,
                    group_offsets=plan.offsets, bias=b + "mlp.experts.htoh4.bias")
        ops.gemm_nt(fb["dhp"], self.wt[b + "mlp.experts.htoh4"], fb["dx_recv"], M=n, c_row_idx=rg,
                    group_offsets=plan.offsets, tile_starts=plan.tile_starts)
        self._a2a_equal(fb["dx_recv"], fb["dx_back"])
        ops.gather_rows(fb["dx_back"], plan.unpad_idx, self.s_dxe)

    def ep_overflowed(self) -> bool:
        """fixed-capacity exchange: did any (source, destination) pair of any layer since the last call route more rows
        than the capacity?  ONE host read; clears the flag."""
        if not self.ep_capacity:
            return False
        over = bool(int(self.ep_overflow.item()))
        if over:
            self.ep_overflow.zero_()
        return over

    def _experts_bwd_ep(self, i, a):
        """mirror of _experts_fwd_ep: self.s_dy (token-major d y) -> expert grads (local experts only) and
        self.s_dxe (token-major d of the routed input copies)."""
        if a["ep"].get("fixed"):
            return self._experts_bwd_ep_fixed(i, a)
        if a["ep"].get("chunked"):
            return self._experts_bwd_ep_chunked(i, a)
        b = f"blocks.{i}."
        k, D, r, ep = self.k, self.D, a["route"], a["ep"]
        plan, n = ep["plan"], ep["n"]
        dy_send = ops.gather_rows(self.s_dy, r.row_of_slot, self._e(self.R, D))
        dy_recv = self._a2a(dy_send, plan.in_splits, plan.out_splits)
        dx_recv = self._e(n, D)
        if n > 0:
            rg = ep["rg"]
            self._wgrad(dy_recv, ep["hid"], b + "mlp.experts.h4toh.weight", M=n, c_row_idx=rg,
                        group_offsets=ep["offsets"], bias=b + "mlp.experts.h4toh.bias")
            dhp = self._e(n, self.Hm)
            ops.gemm_nt(dy_recv, self.wt[b + "mlp.experts.h4toh"], dhp, M=n, gelu_grad_pre=ep["hid_pre"],
                        a_row_idx=rg, a_row_div=1, group_offsets=ep["offsets"], tile_starts=ep["tile_starts"])
            self._wgrad(dhp, ep["x_recv"], b + "mlp.experts.htoh4.weight", M=n, a_row_idx=rg, a_row_div=1,
                        group_offsets=ep["offsets"], bias=b + "mlp.experts.htoh4.bias")
            ops.gemm_nt(dhp, self.wt[b + "mlp.experts.htoh4"], dx_recv, M=n, c_row_idx=rg,
                        group_offsets=ep["offsets"], tile_starts=ep["tile_starts"])
        dx_send = self._a2a(dx_recv, plan.out_splits, plan.in_splits)
        ops.gather_rows(dx_send, r.pos, self.s_dxe)

    def sync_grads(self, group=None, world: int = 1):
        """Data-parallel gradient sync (mean): everything when experts are replicated, only the non-expert
        slice of the flat buffer under expert parallelism (expert params have dp_comm == "none",
        custom_moe_layer.py:159)."""
        if world <= 1:
            return
        import torch.distributed as dist
        buf = self.flat_grads[: self.n_dense] if self.ep_world > 1 else self.flat_grads
        dist.all_reduce(buf, group=group)
        buf.div_(world)

    # ----------------------------------------------------------------- backward
    def _fork(self, reads, fn):
        """Run fn() - weight-gradient launches - on the wgrad stream, ordered after everything queued so far
        on the current stream; `reads` names the scratch buffers it reads, which the main chain may only
        overwrite after _before_write(name)."""
        if self.wg_stream is None:
            fn()
            return
        ready = self._event()
        ready.record(torch.cuda.current_stream())
        self.wg_stream.wait_event(ready)
        with torch.cuda.stream(self.wg_stream):
            fn()
            done = self._event()
            done.record(self.wg_stream)
        for r in reads:
            self._readers[r] = done

    def _event(self):
        # events come from a pool that lives as long as the engine (none is destroyed while a hipGraph
        # capture that recorded it is open) and is walked in the same order every backward()
        if self._ev_i == len(self._ev_pool):
            self._ev_pool.append(torch.cuda.Event())
        self._ev_i += 1
        return self._ev_pool[self._ev_i - 1]

    def _before_write(self, *names):
        for n in names:
            ev = self._readers.pop(n, None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

    def _join_wgrad(self):
        if self.wg_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wg_stream)
            self._readers.clear()

    def _wgrad(self, dC, A, name, M=None, bias=None, reads=(), **kw):
        """weight grad (+ fused bias grad) accumulated into self.grads"""
        self._fork(reads, lambda: ops.wgrad_tn(dC, A, self.grads[name], M=M, beta=1, ws=self.ws_wgrad, queue=self.wq,
                                               db=self.grads[bias] if bias is not None else None, **kw))

    def backward(self, d_tokens: torch.Tensor, cv_weight=0.0):
        """Accumulates parameter gradients of  <tokens, d_tokens> + cv_weight * total_cv_loss
        into self.grads (beta = 1: the joint multi-task backward, train/train_utils.py:437-457).
        cv_weight: a python float, or a 1-element fp32 device tensor (the upstream gradient of cv_loss handed over by
        torch.autograd - read by the gate's backward kernel, so a captured backward follows its value)."""
        self.backward_begin(d_tokens, cv_weight)
        self.backward_blocks(self.depth - 1, 0)
        return self.backward_end()

    # The backward in three resumable pieces, so that a data-parallel trainer can all-reduce the gradients
    # of the upper blocks (complete after backward_blocks(depth-1, s)) while the lower blocks still run.
    def backward_begin(self, d_tokens: torch.Tensor, cv_weight=0.0):
        self._ev_i = 0
        dx = self.s_dxa
        dx.copy_(d_tokens.reshape(self.T, self.D))
        self._bw = dict(dx=dx, other=self.s_dxb, have_dx_t=False, cv_weight=cv_weight, d_tsf=None)

    def backward_blocks(self, hi: int, lo: int):
        """blocks hi, hi-1, ..., lo (inclusive); after it the gradients of every parameter of those blocks
        are final on the wgrad stream (backward_sync_wgrad) / current stream."""
        p, gr = self.params, self.grads
        B, T, D, R, k = self.B, self.T, self.D, self.R, self.k
        st = self._bw
        dx, other, have_dx_t, cv_weight = st["dx"], st["other"], st["have_dx_t"], st["cv_weight"]
        for i in range(hi, lo - 1, -1):
            a = self.act[i]
            b = f"blocks.{i}."
            if self.checkpoint:                             # the shared activation buffers hold another block's values
                self._block_forward(i, a["x_in"], None)
            sa, sm = (None, None) if a.get("ps") is None else a["ps"]
            if not self.is_moe[i]:
                if sm is not None:                          # DropPath: the gradient entering the branch is scale * d x
                    self._before_write("dx_t")
                    ops.scale_rows_cast(dx, sm, self.N, self.s_dx_t)
                elif not have_dx_t:
                    self._before_write("dx_t")
                    ops.cast_f32(dx, self.s_dx_t)
                dpre = self.s_dpre[: T * self.Hd].view(T, self.Hd)
                self._wgrad(self.s_dx_t, a["u"], b + "mlp.fc2.weight", bias=b + "mlp.fc2.bias", reads=("dx_t",))
                self._before_write("dpre")
                ops.gemm_nt(self.s_dx_t, self.wt[b + "mlp.fc2"], dpre, gelu_grad_pre=a["pre"])
                self._wgrad(dpre, a["h2"], b + "mlp.fc1.weight", bias=b + "mlp.fc1.bias", reads=("dpre",))
                ops.gemm_nt(dpre, self.wt[b + "mlp.fc1"], self.s_dh)
                dh2 = self.s_dh
            else:
                g, r = a["gate"], a["route"]
                score = a["score_s"] if sm is not None else g["score"]
                if self.ep_world > 1:
                    self._before_write("dy")
                    ops.combine_bwd(dx, a["y"], score, self.s_dy, self.s_dscore)
                    if sm is not None:
                        self.s_dscore.mul_(a["sm_tok"])         # d score = scale * d(scale * score)
                    self._experts_bwd_ep(i, a)
                else:
                    # The combine's backward d y[t*k+j] = score[t,j] * d x[t] is a row scaling, and a row scaling commutes
                    # with the expert GEMMs behind it: FC2's input-gradient GEMM gathers its rows from d x (activation
                    # dtype copy, [T, D]) through row_of_slot / k and applies the score in its epilogue; FC2's weight
                    # gradient takes the same rows and multiplies by the score where they enter LDS.  The [T*k, D] copy
                    # d y is never written (one 2*R*D-byte store and two such reads less per MoE block); the combine
                    # backward only produces d score.
                    if not have_dx_t:
                        self._before_write("dx_t")
                        ops.cast_f32(dx, self.s_dx_t)
                    ops.combine_bwd(dx, a["y"], score, None, self.s_dscore)
                    if sm is not None:
                        self.s_dscore.mul_(a["sm_tok"])         # d score = scale * d(scale * score)
                    dhp = self.s_dpre[: R * self.Hm].view(R, self.Hm)
                    self._wgrad(self.s_dx_t, a["hid"], b + "mlp.experts.h4toh.weight", M=R, c_row_idx=r.row_of_slot,
                                c_row_div=k, c_row_scale=score, group_offsets=r.offsets,
                                bias=b + "mlp.experts.h4toh.bias", reads=("dx_t",))
                    self._before_write("dpre")
                    ops.gemm_nt(self.s_dx_t, self.wt[b + "mlp.experts.h4toh"], dhp, M=R, gelu_grad_pre=a["hid_pre"],
                                a_row_idx=r.row_of_slot, a_row_div=k, row_scale=score, row_scale_idx=r.row_of_slot,
                                group_offsets=r.offsets, tile_starts=r.tile_starts)
                    self._wgrad(dhp, a["h2"], b + "mlp.experts.htoh4.weight", M=R, a_row_idx=r.row_of_slot,
                                a_row_div=k, group_offsets=r.offsets, bias=b + "mlp.experts.htoh4.bias", reads=("dpre",))
                    ops.gemm_nt(dhp, self.wt[b + "mlp.experts.htoh4"], self.s_dxe, M=R, c_row_idx=r.row_of_slot,
                                group_offsets=r.offsets, tile_starts=r.tile_starts)
                # gate: d score from the combine, d importance / d load from the cv loss
                cvw_dev = cv_weight if isinstance(cv_weight, torch.Tensor) else None
                bal = cvw_dev is not None or cv_weight != 0.0
                self._before_write("dl")
                dl = ops.gate_bwd_logits(g["noisy"], g["idx"], self.s_dscore, g["d_importance"] if bal else None, k,
                                         balance_scale=1.0 if cvw_dev is not None else cv_weight, balance_scale_dev=cvw_dev,
                                         idx_next=g["idx_next"],
                                         d_load_prob=g["d_load_prob"] if bal else None, clean=g["clean"],
                                         top_logits=g["top_logits"], noise_std=g["noise_std"], out=self.s_dl,
                                         out_act=self.s_dl_t if (self.gate_via_gemm and self.dt != torch.float32) else None)
                # token rows of w_gate ([:D]; the task-conditioned rows [D:]: _task_feature_block_bwd below)
                wg, dwg = p[a["wname"]][:D], gr[a["wname"]][:D]
                dh2_moe = self.s_dh32
                if self.gate_via_gemm:
                    # (a dedicated VALU kernel for d w_gate - m3_gate_bwd_params, 4 columns x E accumulators per thread - was
                    # measured in round 3: 24 us + 6 us partial reduce against 17.5 us for the padded TN GEMM whose reduce rides
                    # along: the GEMM stays)
                    # d w_gate += h2^T dl (TN GEMM) ; dh2 = sum_j dxe[t,j] + dl w_gate^T
                    dl_t = dl if self.dt == torch.float32 else self.s_dl_t      # (written by the gate's backward kernel itself)
                    self._fork(("dl",), lambda: ops.wgrad_tn(a["h2"], dl_t, dwg, beta=1, ws=self.ws_wgrad, queue=self.wq))
                    if self.fused_gate_dx:
                        # one pass over the [T, D] result, stored in the activation dtype like a dense block's d h2 (the sum
                        # itself is formed in fp32): the LayerNorm backward below reads half the bytes
                        ops.combine_gate_bwd(self.s_dxe, k, dl, wg, self.s_dh)
                        dh2_moe = self.s_dh
                    else:
                        ops.combine_fwd(self.s_dxe, self.ones_k, None, self.s_dh32)
                        wg_t = wg if self.dt == torch.float32 else self.wgate_c[a["wname"]][:D]
                        ops.gemm_nt(dl_t, wg_t, self.s_dh32, residual=self.s_dh32)
                else:
                    ops.combine_fwd(self.s_dxe, self.ones_k, None, self.s_dh32)       # dh2 = sum_j dxe[t,j]
                    ops.gate_bwd_params(a["h2"], wg, dl, d_w_gate=dwg, beta_dw=1, dx=self.s_dh32,
                                        beta_dx=1, part_dw=self.ws_gate_dw)
                if self._tsf is not None:
                    self._task_feature_block_bwd(a, dl, st)
                dh2 = dh2_moe
            self._before_write("dx_t")
            ops.layernorm_bwd(dh2, a["x1"], a["mean2"], a["rstd2"], p[b + "norm2.weight"], dx, other, None, None,
                              ws=self.ws_ln[2 * i + 1],
                              dx_act=self.s_dx_t if sa is None else None)     # also emits the activation-dtype copy
            dx, other = other, dx                                        # dx = d x1
            if sa is not None:
                ops.scale_rows_cast(dx, sa, self.N, self.s_dx_t)         # DropPath of the attention branch
            self._wgrad(self.s_dx_t, a["o"], b + "attn.proj.weight", bias=b + "attn.proj.bias", reads=("dx_t",))
            ops.gemm_nt(self.s_dx_t, self.wt[b + "attn.proj"], self.s_do)
            self._before_write("dqkv")
            ops.attention_bwd(a["qkv"], a["o"], self.s_do, a["lse"], B, self.N, self.heads, self.dh, self.s_dqkv,
                              dq_ws=self.ws_dq)
            self._wgrad(self.s_dqkv, a["h1"], b + "attn.qkv.weight", bias=b + "attn.qkv.bias", reads=("dqkv",))
            ops.gemm_nt(self.s_dqkv, self.wt[b + "attn.qkv"], self.s_dh)
            # does the block below consume the activation-dtype copy of d x directly?  (a dense block without DropPath: fc2's
            # backward GEMMs; an MoE block with local experts: FC2's backward GEMMs, which apply the gate score themselves)
            nxt_dx_t = i > 0 and ((self.is_moe[i - 1] and self.ep_world == 1) or
                                  (not self.is_moe[i - 1] and self.act[i - 1].get("ps") is None))
            if nxt_dx_t:
                self._before_write("dx_t")
            ops.layernorm_bwd(self.s_dh, a["x_in"], a["mean1"], a["rstd1"], p[b + "norm1.weight"], dx, other, None, None,
                              ws=self.ws_ln[2 * i],
                              dx_act=self.s_dx_t if nxt_dx_t else None)
            dx, other = other, dx
            have_dx_t = nxt_dx_t
        # the last weight-gradient call's slabs (a data-parallel step all-reduces these blocks' slice next); on the stream
        # the weight-gradient launches went to
        if self.wq is not None:
            self._fork((), self.wq.flush)
        # norm weight / bias gradients of the blocks just done, all in one launch (their partial slots are contiguous)
        if hi >= lo:
            ops.layernorm_bwd_reduce(self.ws_ln, self.ln_nblk, self.D, self.ln_table, 2 * lo, 2 * (hi - lo + 1), beta=1)
        st.update(dx=dx, other=other, have_dx_t=have_dx_t)

    def backward_sync_wgrad(self):
        """make the current stream wait for the weight-gradient launches issued so far (no-op without a wgrad stream)"""
        self._join_wgrad()

    def accept_dx(self, others):
        """shared stem: add the other task passes' gradient at the stem's output to this pass's; the stem's backward
        then runs once on the sum (it is linear in d x)."""
        st = self._bw
        for o in others:
            ops.add_f32(st["dx"].view(-1), o._bw["dx"].view(-1))
        st["have_dx_t"] = False                       # the activation-dtype copy was this pass's d x alone

    def backward_end(self, stem: bool = True):
        """stem=False: a pass that started from another engine's stem - nothing below its first block is its own"""
        p, gr = self.params, self.grads
        B, D = self.B, self.D
        dx = self._bw["dx"]
        if stem:
            # patch embedding / cls / pos
            ops.tokens_bwd(dx, B, self.np_, D, self.s_dpatch, gr["pos_embed"].view(self.N, D), gr["cls_token"].view(D), beta=1)
            gw = gr["patch_embed.proj.weight"].view(D, -1)
            self._fork((), lambda: ops.wgrad_tn(self.s_dpatch, self.rows, gw, beta=1, ws=self.ws_wgrad, queue=self.wq,
                                                db=gr["patch_embed.proj.bias"]))
        if self.wq is not None:
            self._fork((), self.wq.flush)
        self._join_wgrad()
        if self._tsf is not None:
            self._task_feature_bwd(self._bw.get("d_tsf"))
        return dx
