"""VisionTransformerMoE / Block / Attention / Mlp / PatchEmbed mirrors
(models/moe/ckpt/vision_transformer_moe.py:244-261,283-341,379-562,564-886) with the reference's
module names, state_dict keys and forward signatures, running on the HIP kernels.

Scope notes: drop / attn_drop must be 0 (as in every BASELINE config), drop_path is supported (torch-level per-sample
scaling of the residual branches, as pretrain/configs/deit_moe_small.yaml uses it); pretrained
weight loading, hybrid backbones, distilled tokens, wandb statistics and the sem regularisers are
out of scope.  `forward` returns (tokens [B,N,D], total_cv_loss) like :882-886.

convention="origin" (ctor flag of VisionTransformerMoE / Block, handed down to the layer and the gate) selects the API
of models/moe/origin/* instead - what train_fastmoe.py builds with --use_checkpointing False (:425-435): the gate
returns (idx, score) and keeps the block's balance loss in `self.loss` (origin/noisy_gate_vmoe.py:267-297), the layer
returns a tensor (origin/custom_moe_layer.py:180-181), Block and backbone return tokens only
(origin/vision_transformer_moe.py:275-283,552-563) and the trainer collects the loss with
m3vit_amd.moe_utils.collect_noisy_gating_loss (utils/moe_utils.py:201-207).  Same parameters, same kernels."""
from functools import partial

import torch
import torch.nn as nn

from .functional import AttentionCoreFn, LayerNormFn, MlpFn, PlainLinearFn
from .gate import NoisyGate_VMoE
from .moe_layer import FMoETransformerMLP


from .balance import block_balance_loss, cv_squared, gates_to_load as _gates_to_load, prob_in_top_k as _prob_in_top_k  # noqa: E402,F401


class HipLayerNorm(nn.LayerNorm):
    """nn.LayerNorm whose forward/backward are the HIP row kernels; output in `act_dtype`."""
    act_dtype = torch.float32

    def forward(self, x):
        return LayerNormFn.apply(x, self.weight, self.bias, self.eps, self.act_dtype)


class Dropout(nn.Dropout):
    """nn.Dropout that knows its call site (element-wise dropout, drop_rate > 0: configs/nyud/vit_moe/*drop0.1*.yml; the six
    sites are listed in oracle/ref_torch.py).  `mask_fn(site, shape, p) -> factors` (class attribute, tests only) pins the
    masks; otherwise torch draws them.  Dropout runs on the per-op module path: the fused executor has no dropout."""
    mask_fn = None

    def __init__(self, p=0.0, site=""):
        super().__init__(p)
        self.site = site

    def forward(self, x, site=None):
        if self.p == 0.0 or not self.training:
            return x
        if Dropout.mask_fn is not None:
            return x * Dropout.mask_fn(site or self.site, x.shape, self.p).to(device=x.device, dtype=x.dtype)
        return nn.functional.dropout(x, self.p, True)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0., site=""):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = Dropout(drop, site)
        self.site = site

    def forward(self, x):
        if self.drop.p == 0.0 or not self.training:
            return MlpFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
        # vision_transformer_moe.py:255-261 with drop > 0: the ONE Dropout module is applied after the activation and after fc2
        h = self.act(PlainLinearFn.apply(x, self.fc1.weight, self.fc1.bias))
        h = self.drop(h, self.site + "mlp.act")
        return self.drop(PlainLinearFn.apply(h, self.fc2.weight, self.fc2.bias), self.site + "mlp.out")


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., site=""):
        super().__init__()
        assert attn_drop == 0.0, "dropout on the attention probabilities (inside the attention kernel) is not supported"
        self.num_heads = num_heads
        head_dim = dim // num_heads
        assert qk_scale is None or abs(qk_scale - head_dim ** -0.5) < 1e-12, "custom qk_scale not supported"
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=bool(qkv_bias))
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = Dropout(proj_drop, site + "attn.proj")

    def forward(self, x):
        qkv = PlainLinearFn.apply(x, self.qkv.weight, self.qkv.bias)
        o = AttentionCoreFn.apply(qkv, self.num_heads)
        return self.proj_drop(PlainLinearFn.apply(o, self.proj.weight, self.proj.bias))


class DropPath(nn.Module):
    """Stochastic depth per sample on a residual branch (vision_transformer_moe.py:167-185): in training the branch of
    a sample is dropped with probability drop_prob and the kept ones are scaled by 1 / (1 - drop_prob).  The factor
    drawn last is kept in `last_scale` ([B], for tests)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob or 0.0)
        self.last_scale = None

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        draw = torch.rand((x.shape[0],) + (1,) * (x.dim() - 1), dtype=x.dtype, device=x.device)
        scale = torch.floor(keep + draw) / keep
        self.last_scale = scale.flatten()
        return x * scale


class _Im2Row(torch.autograd.Function):
    """patchify (m3_im2row): images [B, C, H, W] -> rows [B * patches, C * P * P].  The patches do not overlap, so the
    gradient with respect to the images (the reference's Conv2d provides one; no trainer of the reference asks for it) is the
    inverse permutation of d rows - only computed when the images require a gradient."""

    @staticmethod
    def forward(ctx, x, P, num_patches, act_dtype):
        from . import ops
        B, C, H, W = x.shape
        rows = torch.empty(B * num_patches, C * P * P, dtype=act_dtype, device=x.device)
        ops.im2row(x.detach().contiguous().float(), P, rows)
        ctx.geom = (B, C, H, W, P, x.dtype)
        return rows

    @staticmethod
    def backward(ctx, d_rows):
        B, C, H, W, P, dt = ctx.geom
        if not ctx.needs_input_grad[0]:
            return None, None, None, None
        d = d_rows.float().view(B, H // P, W // P, C, P, P).permute(0, 3, 1, 4, 2, 5).reshape(B, C, H, W)
        return d.to(dt), None, None, None


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        img_size = tuple(img_size) if isinstance(img_size, (tuple, list)) else (img_size, img_size)
        self.img_size = img_size
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size[1] // patch_size) * (img_size[0] // patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x, act_dtype=torch.float32):
        """-> [B, num_patches, D] (the reference returns the conv map; flatten(2).transpose(1,2) is done
        by the caller at :783-784 - folded in here)."""
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1]
        P = self.patch_size[0]
        rows = _Im2Row.apply(x, P, self.num_patches, act_dtype)
        w2 = self.proj.weight.reshape(self.proj.weight.shape[0], -1)
        y = PlainLinearFn.apply(rows, w2, self.proj.bias)
        return y.view(B, self.num_patches, -1)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=HipLayerNorm, moe=False, moe_mlp_ratio=-1,
                 moe_experts=64, moe_top_k=2, moe_gate_dim=-1, world_size=1, gate_return_decoupled_activation=False,
                 moe_gate_type="noisy_vmoe", vmoe_noisy_std=1, gate_task_specific_dim=-1, multi_gate=False,
                 regu_experts_fromtask=False, num_experts_pertask=-1, num_tasks=-1, gate_input_ahead=False,
                 regu_sem=False, sem_force=False, regu_subimage=False, expert_prune=False, use_checkpointing=False,
                 convention="ckpt", site=""):
        super().__init__()
        assert attn_drop == 0.0, "dropout on the attention probabilities is not supported"
        assert convention in ("ckpt", "origin")
        self.convention = convention
        self.moe = moe
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, proj_drop=drop, site=site)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()      # :397-398 (one module, two draws per block)
        self.norm2 = norm_layer(dim)
        self.gate_input_ahead = gate_input_ahead
        if moe:
            self.tot_expert = moe_experts * world_size
            self.moe_top_k = moe_top_k
            activation = nn.Sequential(act_layer(), Dropout(drop, site + "experts.act"))
            if moe_gate_dim < 0:
                moe_gate_dim = dim
            if moe_mlp_ratio < 0:
                moe_mlp_ratio = mlp_ratio
            if moe_gate_type != "noisy_vmoe":
                raise ValueError("only moe_gate_type='noisy_vmoe' works with this layer (SURVEY App. A.8)")
            self.mlp = FMoETransformerMLP(num_expert=moe_experts, d_model=dim, d_gate=moe_gate_dim,
                                          d_hidden=int(dim * moe_mlp_ratio), world_size=world_size, top_k=moe_top_k,
                                          activation=activation, gate=NoisyGate_VMoE, vmoe_noisy_std=vmoe_noisy_std,
                                          gate_task_specific_dim=gate_task_specific_dim, multi_gate=multi_gate,
                                          regu_experts_fromtask=regu_experts_fromtask,
                                          num_experts_pertask=num_experts_pertask, num_tasks=num_tasks,
                                          expert_prune=expert_prune, sem_force=sem_force, convention=convention)
            self.mlp_drop = Dropout(drop, site + "mlp_drop")
        else:
            self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop, site=site)

    def forward(self, x, gate_inp=None, task_id=None, task_specific_feature=None, sem=None):
        if self.gate_input_ahead:
            gate_inp = x
        x = x + self.drop_path(self.attn(self.norm1(x)).to(x.dtype))
        normed = self.norm2(x)
        if self.convention == "origin":                                   # origin/vision_transformer_moe.py:275-283
            if not self.moe:
                return x + self.drop_path(self.mlp(normed).to(x.dtype))
            return x + self.drop_path(self.mlp_drop(self.mlp(normed, gate_inp, task_id, task_specific_feature, sem)).to(x.dtype))
        if not self.moe:
            return x + self.drop_path(self.mlp(normed).to(x.dtype)), None
        out, clean, noisy, std, top_logits, gates = self.mlp(normed, gate_inp, task_id, task_specific_feature, sem)
        x = x + self.drop_path(self.mlp_drop(out).to(x.dtype))
        cv_loss = block_balance_loss(gates, clean, noisy, std, top_logits, self.mlp.top_k) if self.training else 0   # :453-459,540
        return x, cv_loss


class new_Mlp(nn.Module):
    """Task embedding net (:263-281): fc1 -> GELU -> fc2 -> LayerNorm, on torch ops (O(gtsd^2) work)."""

    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.norm = nn.LayerNorm(out_features, eps=1e-6)

    def forward(self, x):
        return self.norm(self.fc2(self.act(self.fc1(x))))


class VisionTransformerMoE(nn.Module):
    def __init__(self, model_name="vit_small_patch16_224", img_size=224, patch_size=16, in_chans=3, embed_dim=384,
                 depth=12, num_heads=12, num_classes=19, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., moe_mlp_ratio=-1, moe_experts=64, moe_top_k=2, world_size=1,
                 gate_dim=-1, moe_gate_type="noisy_vmoe", vmoe_noisy_std=1, gate_task_specific_dim=-1,
                 multi_gate=False, regu_experts_fromtask=False, num_experts_pertask=-1, num_tasks=-1,
                 gate_input_ahead=False, expert_prune=False, use_checkpointing=False, act_dtype=torch.float32,
                 random_init=True, sem_force=False, convention="ckpt", fused="auto", **kwargs):
        """fused: "auto" (default) / True - forward(x, task_id) runs as ONE autograd node on the straight-line executor
        (m3vit_amd/fused.py: hipGraph replay, the parameters' .grad are views of its flat gradient buffer) whenever the call
        is one it covers, and through the per-op autograd Functions below otherwise (`fused_fallback_reason` says why);
        False - always per-op.  The environment variable M3VIT_FUSED=0 forces False."""
        super().__init__()
        assert convention in ("ckpt", "origin")
        self.convention = convention
        import os
        self.fused = False if os.environ.get("M3VIT_FUSED", "1") == "0" else fused
        self.use_checkpointing = bool(use_checkpointing)
        self.world_size = int(world_size)
        self._fused = None
        self.fused_fallback_reason = None
        self._cfg_kwargs = dict(img_size=tuple(img_size) if isinstance(img_size, (tuple, list)) else (img_size, img_size),
                                patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                                mlp_ratio=mlp_ratio, moe_mlp_ratio=moe_mlp_ratio if moe_mlp_ratio >= 0 else mlp_ratio,
                                moe_experts=moe_experts * world_size, moe_top_k=moe_top_k,     # the gate scores all E_loc * W experts
                                gate_dim=gate_dim if gate_dim >= 0 else embed_dim,
                                multi_gate=bool(multi_gate), gate_task_specific_dim=gate_task_specific_dim,
                                vmoe_noisy_std=float(vmoe_noisy_std))
        why = None
        if drop_rate > 0.0:
            why = "element-wise dropout (drop_rate > 0)"
        elif not qkv_bias:
            why = "qkv_bias=False"
        elif world_size > 1 and use_checkpointing:
            why = "expert parallel layer with activation checkpointing"
        elif regu_experts_fromtask or expert_prune or gate_input_ahead:
            why = "regu_experts_fromtask / expert_prune / gate_input_ahead routing edits"
        elif embed_dim // num_heads not in (32, 64) or moe_experts * world_size > 64 or moe_experts * world_size < 2 or \
                moe_top_k > moe_experts * world_size:
            why = "head dim not 32 / 64 or expert count outside [2, 64]"
        elif multi_gate and self._cfg_kwargs["gate_dim"] <= embed_dim:
            why = "multi_gate without tasks"
        elif (not multi_gate) and gate_task_specific_dim >= 0 and self._cfg_kwargs["gate_dim"] <= embed_dim:
            why = "task-conditioned gate without tasks"
        self._fused_static_ok, self._fused_static_why = why is None, why
        assert attn_drop_rate == 0.0, "dropout on the attention probabilities is not supported"
        dpr = [drop_path_rate * i / max(depth - 1, 1) for i in range(depth)]          # linspace(0, rate, depth), :632
        self.img_size = tuple(img_size) if isinstance(img_size, (tuple, list)) else (img_size, img_size)
        self.patch_size = patch_size
        self.embed_dim = self.num_features = embed_dim
        self.depth = depth
        self.act_dtype = act_dtype
        norm_layer = partial(_make_norm, act_dtype=act_dtype)
        self.patch_embed = PatchEmbed(self.img_size, patch_size, in_chans, embed_dim)
        self.num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = None
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, embed_dim))
        self.pos_drop = Dropout(drop_rate, "pos")
        self.num_tasks = gate_dim - embed_dim
        self.gate_task_specific_dim = gate_task_specific_dim
        self.multi_gate = multi_gate
        if gate_task_specific_dim < 0 or multi_gate:
            self.gate_task_represent = None
        else:
            self.gate_task_represent = new_Mlp(self.num_tasks, int(gate_task_specific_dim), gate_task_specific_dim)
        blocks = []
        for i in range(depth):
            if i % 2 == 0:
                blocks.append(Block(embed_dim, num_heads, mlp_ratio, qkv_bias, qk_scale, drop=drop_rate, drop_path=dpr[i],
                                    norm_layer=norm_layer, convention=convention, site=f"blocks.{i}."))
            else:
                blocks.append(Block(embed_dim, num_heads, mlp_ratio, qkv_bias, qk_scale, drop=drop_rate, drop_path=dpr[i],
                                    norm_layer=norm_layer, moe=True, site=f"blocks.{i}.",
                                    moe_mlp_ratio=moe_mlp_ratio, moe_experts=moe_experts, moe_top_k=moe_top_k,
                                    moe_gate_dim=gate_dim, world_size=world_size, moe_gate_type=moe_gate_type,
                                    vmoe_noisy_std=vmoe_noisy_std, gate_task_specific_dim=gate_task_specific_dim,
                                    multi_gate=multi_gate, regu_experts_fromtask=regu_experts_fromtask,
                                    num_experts_pertask=num_experts_pertask, num_tasks=num_tasks,
                                    gate_input_ahead=gate_input_ahead, expert_prune=expert_prune,
                                    sem_force=sem_force, convention=convention))
        self.blocks = nn.Sequential(*blocks)
        self.pre_logits = nn.Identity()
        self.init_weights()

    def init_weights(self):
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        for _, m in self.named_modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

    def _forward_fused(self, x, task_id):
        from .fused import FusedBackbone
        if self._fused is None:
            self._fused = FusedBackbone(self)
        tok, cv = self._fused.forward(x, task_id)
        if self.convention != "origin":
            return tok, cv
        # origin convention (origin/vision_transformer_moe.py:552-563): tokens only; the trainer sums the gates' stored
        # losses (utils/moe_utils.py:201-207) - the summed balance loss of the pass is handed to ONE gate, the others hold none
        last = None
        for blk in self.blocks:
            if blk.moe:
                gates = blk.mlp.gate if isinstance(blk.mlp.gate, nn.ModuleList) else [blk.mlp.gate]
                for g in gates:
                    g.set_loss(None)
                last = blk.mlp.gate[task_id] if isinstance(blk.mlp.gate, nn.ModuleList) else blk.mlp.gate
        if last is not None:
            last.set_loss(cv if self.training else 0)
        return tok

    def forward_features(self, x, gate_inp, task_id, sem):
        if self.fused:
            from .fused import FusedBackbone
            self.fused_fallback_reason = FusedBackbone.unsupported(self, x, gate_inp, task_id, sem)
            if self.fused_fallback_reason is None:
                return self._forward_fused(x, task_id)
            if self.fused is True:
                raise RuntimeError(f"VisionTransformerMoE(fused=True): {self.fused_fallback_reason}")
        B = x.shape[0]
        x = self.patch_embed(x, self.act_dtype).float()
        x = self.pos_drop(torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1) + self.pos_embed)      # :791
        tsf = None
        if (task_id is not None) and (self.gate_task_represent is not None):
            one_hot = torch.zeros(self.num_tasks, device=x.device)
            one_hot[task_id] = 1.0
            tsf = self.gate_task_represent(one_hot)
        if self.convention == "origin":                                   # origin/vision_transformer_moe.py:536-563
            for blk in self.blocks:
                x = blk(x, gate_inp, task_id, tsf, sem=sem) if blk.moe else blk(x)
            return x
        total_cv = torch.tensor(0.0, device=x.device, dtype=x.dtype, requires_grad=True)
        for blk in self.blocks:
            if blk.moe:
                x, cv = blk(x, gate_inp, task_id, tsf, sem=sem)
                if cv is not None:
                    total_cv = total_cv + cv
            else:
                x, _ = blk(x)
        return x, total_cv

    def forward(self, x, gate_inp=None, task_id=None, sem=None):
        return self.forward_features(x, gate_inp, task_id=task_id, sem=sem)


def _make_norm(dim, act_dtype=torch.float32):
    n = HipLayerNorm(dim, eps=1e-6)
    n.act_dtype = act_dtype
    return n
