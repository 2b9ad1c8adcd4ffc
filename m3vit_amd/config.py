"""Shape/config holder and random initialisation for the MoE-ViT backbone, mirroring the
VisionTransformerMoE constructor arguments that change the hot-path maths
(models/moe/ckpt/vision_transformer_moe.py:565-572) and its init
(:697-705 trunc_normal(.02) Linear weights, zero biases, LayerNorm (1, 0);
noisy_gate_vmoe.py:69 kaiming_uniform(a=sqrt 5) w_gate; expert tensor layout
utils/helpers.py:645-662)."""
from __future__ import annotations

import math
from typing import Dict

import torch


class BackboneConfig:
    def __init__(self, img_size=(224, 224), patch_size=16, in_chans=3, embed_dim=384, depth=12, num_heads=12,
                 mlp_ratio=4.0, moe_mlp_ratio=1.0, moe_experts=16, moe_top_k=4, gate_dim=386, multi_gate=True,
                 gate_task_specific_dim=-1, vmoe_noisy_std=0.0, dense_only=False):
        self.dense_only = dense_only
        self.img_size = tuple(img_size)
        self.patch_size = patch_size
        self.in_chans = in_chans
        self.embed_dim = embed_dim
        self.depth = depth
        self.num_heads = num_heads
        self.mlp_ratio = mlp_ratio
        self.moe_mlp_ratio = moe_mlp_ratio
        self.moe_experts = moe_experts
        self.moe_top_k = moe_top_k
        self.gate_dim = gate_dim
        self.multi_gate = multi_gate
        self.gate_task_specific_dim = gate_task_specific_dim
        self.vmoe_noisy_std = vmoe_noisy_std

    @property
    def num_tasks(self):
        return self.gate_dim - self.embed_dim

    @property
    def num_tokens(self):
        return (self.img_size[0] // self.patch_size) * (self.img_size[1] // self.patch_size) + 1

    @property
    def d_gate(self):
        return self.embed_dim if self.gate_task_specific_dim < 0 else self.embed_dim + self.gate_task_specific_dim

    def is_moe(self, i):
        return (i % 2 == 1) and not self.dense_only

    # FLOPs of one forward backbone pass per image
    def fwd_flops_per_image(self):
        N, D = self.num_tokens, self.embed_dim
        Hd, Hm = int(D * self.mlp_ratio), int(D * self.moe_mlp_ratio)
        n_moe = sum(1 for i in range(self.depth) if self.is_moe(i))
        f_attn = self.depth * (2 * N * D * 3 * D + 4 * N * N * D + 2 * N * D * D)
        f_dense = (self.depth - n_moe) * (4 * N * D * Hd)
        f_moe = n_moe * (4 * N * self.moe_top_k * D * Hm + 2 * N * self.d_gate * self.moe_experts)
        f_patch = 2 * (N - 1) * (self.in_chans * self.patch_size ** 2) * D
        return f_attn + f_dense + f_moe + f_patch

    def stem_flops_per_image(self):
        """forward FLOPs of the task-independent stem: patch embedding + the (dense) blocks below the first MoE block"""
        N, D = self.num_tokens, self.embed_dim
        Hd = int(D * self.mlp_ratio)
        s = next((i for i in range(self.depth) if self.is_moe(i)), self.depth)
        f_block = 2 * N * D * 3 * D + 4 * N * N * D + 2 * N * D * D + 4 * N * D * Hd
        return s * f_block + 2 * (N - 1) * (self.in_chans * self.patch_size ** 2) * D


VIT_SMALL_MOE = dict(img_size=(224, 224), embed_dim=384, depth=12, num_heads=12, mlp_ratio=4.0, moe_mlp_ratio=1.0,
                     moe_experts=16, moe_top_k=4, gate_dim=386, multi_gate=True)


def _tn(shape, std, gen):
    t = torch.empty(shape)
    torch.nn.init.trunc_normal_(t, std=std, generator=gen)
    return t


def init_params(cfg: BackboneConfig, seed: int = 1, zero_bias: bool = True) -> Dict[str, torch.Tensor]:
    """fp32 CPU tensors keyed like the reference state_dict."""
    g = torch.Generator().manual_seed(seed)
    D, E = cfg.embed_dim, cfg.moe_experts
    Hd, Hm = int(D * cfg.mlp_ratio), int(D * cfg.moe_mlp_ratio)

    def bias(*s):
        return torch.zeros(*s) if zero_bias else _tn(s, .02, g)

    P = {"patch_embed.proj.weight": _tn((D, cfg.in_chans, cfg.patch_size, cfg.patch_size), .02, g),
         "patch_embed.proj.bias": bias(D),
         "cls_token": _tn((1, 1, D), .02, g), "pos_embed": _tn((1, cfg.num_tokens, D), .02, g)}
    if cfg.gate_task_specific_dim >= 0 and not cfg.multi_gate:
        # gate_task_represent = new_Mlp(num_tasks, gtsd, gtsd) (vision_transformer_moe.py:638-641, :263-281)
        gt = cfg.gate_task_specific_dim
        P["gate_task_represent.fc1.weight"] = _tn((gt, cfg.num_tasks), .02, g)
        P["gate_task_represent.fc1.bias"] = bias(gt)
        P["gate_task_represent.fc2.weight"] = _tn((gt, gt), .02, g)
        P["gate_task_represent.fc2.bias"] = bias(gt)
        P["gate_task_represent.norm.weight"] = torch.ones(gt) if zero_bias else 1 + _tn((gt,), .02, g)
        P["gate_task_represent.norm.bias"] = bias(gt)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        for n in ("norm1", "norm2"):
            P[b + n + ".weight"] = torch.ones(D) if zero_bias else 1 + _tn((D,), .02, g)
            P[b + n + ".bias"] = bias(D)
        P[b + "attn.qkv.weight"] = _tn((3 * D, D), .02, g); P[b + "attn.qkv.bias"] = bias(3 * D)
        P[b + "attn.proj.weight"] = _tn((D, D), .02, g); P[b + "attn.proj.bias"] = bias(D)
        if cfg.is_moe(i):
            P[b + "mlp.experts.htoh4.weight"] = _tn((E, Hm, D), .02, g); P[b + "mlp.experts.htoh4.bias"] = bias(E, Hm)
            P[b + "mlp.experts.h4toh.weight"] = _tn((E, D, Hm), .02, g); P[b + "mlp.experts.h4toh.bias"] = bias(E, D)
            for t in range(cfg.num_tasks if cfg.multi_gate else 1):
                w = torch.empty(cfg.d_gate, E)
                torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5), generator=g)
                P[b + (f"mlp.gate.{t}.w_gate" if cfg.multi_gate else "mlp.gate.w_gate")] = w
        else:
            P[b + "mlp.fc1.weight"] = _tn((Hd, D), .02, g); P[b + "mlp.fc1.bias"] = bias(Hd)
            P[b + "mlp.fc2.weight"] = _tn((D, Hd), .02, g); P[b + "mlp.fc2.bias"] = bias(D)
    return P
