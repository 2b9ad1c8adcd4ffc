"""Plain-torch CPU restatement of the M3ViT MoE-ViT forward path (TEST INFRASTRUCTURE).

Every function cites the reference lines (relative to /root/reference) it follows.
Gradients come from torch autograd over these functions; the HIP backward kernels
are checked against them.  Works in fp32 (the reference's train_fastmoe.py dtype)
or fp64 (tighter checker).  No nn.Module magic: parameters are plain tensors in a
dict keyed like the reference's state_dict.

This module must never be imported from m3vit_amd/ (see oracle/__init__.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

__all__ = [
    "gate_vmoe", "route_build", "experts_ffn", "moe_dispatch_ffn", "moe_layer",
    "layernorm", "attention", "mlp_dense", "block_forward", "patch_embed",
    "backbone_forward", "cv_squared", "gates_to_load", "prob_in_top_k",
    "task_embedding", "init_backbone_params", "BackboneCfg",
]


# --------------------------------------------------------------------------- gate
def gate_vmoe(x: torch.Tensor, w_gate: torch.Tensor, top_k: int,
              noise: Optional[torch.Tensor] = None, noise_std: float = 0.0,
              training: bool = True, idx_override: Optional[torch.Tensor] = None):
    """NoisyGate_VMoE.forward, models/moe/ckpt/noisy_gate_vmoe.py:80-264
    (torch-only twin models/moe/gates.py:405-466).

    clean = x @ w_gate (:91); std = noise_std / E_tot * training (:92-93);
    noisy = clean + noise * std (:168, noise = randn_like supplied by the caller);
    p = softmax(noisy, 1) (:197); top_logits, top_idx = p.topk(min(k+1, E)) (:198-200);
    score = top_logits[:, :k], idx = top_idx[:, :k] (:202-204), NOT renormalised;
    gates = zeros.scatter(1, idx, score) (:206-207).
    Returns ((idx, score), clean, noisy, std, top_logits, gates) (:257-264).

    idx_override (test aid, not in the reference): use these expert indices instead of the top-k and
    take their scores from the same softmax - lets a reduced-precision run be checked value-for-value
    on ITS OWN routing once that routing has been verified separately on its own gate input.
    """
    x2 = x.reshape(-1, x.shape[-1])
    E = w_gate.shape[1]
    clean = x2 @ w_gate
    std = (noise_std / E) * (1.0 if training else 0.0)
    if noise is not None and std != 0.0:
        noisy = clean + noise * std
    else:
        noisy = clean + 0.0
    p = torch.softmax(noisy, dim=1)
    top_logits, top_idx = p.topk(min(top_k + 1, E), dim=1)
    score = top_logits[:, :top_k]
    idx = top_idx[:, :top_k]
    if idx_override is not None:
        idx = idx_override.to(device=p.device, dtype=torch.int64)
        score = p.gather(1, idx)
    gates = torch.zeros_like(p).scatter(1, idx, score)
    return (idx, score), clean, noisy, std, top_logits, gates


def gates_to_load(gates: torch.Tensor) -> torch.Tensor:
    """_gates_to_load, models/moe/ckpt/vision_transformer_moe.py:23-31."""
    return (gates > 0).sum(0)


def prob_in_top_k(clean, noisy, noise_stddev, noisy_top_values, top_k):
    """_prob_in_top_k, models/moe/ckpt/vision_transformer_moe.py:33-71 (mixed
    logit/probability domain reproduced as in the reference)."""
    from torch.distributions.normal import Normal
    batch = clean.size(0)
    m = noisy_top_values.size(1)
    flat = noisy_top_values.flatten()
    pos_in = torch.arange(batch, device=clean.device) * m + top_k
    thr_in = flat.gather(0, pos_in).unsqueeze(1)
    is_in = noisy > thr_in
    thr_out = flat.gather(0, pos_in - 1).unsqueeze(1)
    normal = Normal(torch.tensor([0.0], dtype=clean.dtype, device=clean.device), torch.tensor([1.0], dtype=clean.dtype, device=clean.device))
    p_in = normal.cdf((clean - thr_in) / noise_stddev)
    p_out = normal.cdf((clean - thr_out) / noise_stddev)
    return torch.where(is_in, p_in, p_out)


def cv_squared(x: torch.Tensor) -> torch.Tensor:
    """cv_squared, models/moe/ckpt/vision_transformer_moe.py:73-87:
    unbiased var / (mean^2 + 1e-10); zero for a single expert."""
    if x.shape[0] == 1:
        return torch.zeros((), dtype=torch.float32, device=x.device)
    xf = x.float() if x.dtype not in (torch.float32, torch.float64) else x
    return xf.var() / (xf.mean() ** 2 + 1e-10)


# ------------------------------------------------------------------------ routing
def route_build(idx: torch.Tensor, num_expert: int):
    """What fastmoe's prepare_forward (count_by_gate + assign_pos) produces for
    models/moe/ckpt/custom_moe_layer.py:263, restated with a STABLE slot order
    (fastmoe's order inside an expert is atomics-based).

    idx [T,k] -> counts[E], offsets[E+1], pos[T*k] (slot of flat entry t*k+j in the
    expert-major buffer), row_of_slot[T*k] (inverse).  Same information as
    compute_gating (models/moe/moe.py:19-64): expert_size == counts,
    index_sorted_experts == row_of_slot, batch_index == row_of_slot // k.
    """
    flat = idx.reshape(-1).to(torch.int64)
    counts = torch.bincount(flat, minlength=num_expert)
    offsets = torch.zeros(num_expert + 1, dtype=torch.int64, device=flat.device)
    offsets[1:] = counts.cumsum(0)
    row_of_slot = torch.sort(flat, stable=True).indices
    pos = torch.empty_like(row_of_slot)
    pos[row_of_slot] = torch.arange(flat.numel(), dtype=torch.int64, device=flat.device)
    return counts, offsets, pos, row_of_slot


# ------------------------------------------------------------------------ dropout sites
# Element-wise dropout (drop_rate > 0: configs/nyud/vit_moe/*drop0.1*.yml) sits at six call sites of the backbone:
#   "pos"                      pos_drop on the embedded tokens               vision_transformer_moe.py:791
#   "blocks.i.attn.proj"       Attention.proj_drop                           :297,312
#   "blocks.i.mlp.act/.out"    Mlp.drop after the activation and after fc2   :258,260
#   "blocks.i.experts.act"     Dropout inside the experts' activation        :409-412
#   "blocks.i.mlp_drop"        mlp_drop on the MoE output                    :434,450
# backbone_forward(dropout=fn) calls fn(site, tensor) -> tensor there (tests pin the masks by site name); None = p = 0.
_DROP = {"fn": None, "prefix": ""}


def _drop(site, x):
    fn = _DROP["fn"]
    return x if fn is None else fn(_DROP["prefix"] + site, x)


# ------------------------------------------------------------------------ experts
def gelu_erf(x):
    """nn.GELU() (exact erf), activation at models/moe/ckpt/vision_transformer_moe.py:409-412."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def experts_ffn(rows: torch.Tensor, counts, w1, b1, w2, b2):
    """_Expert.forward, models/moe/ckpt/custom_moe_layer.py:36-44 over FMoELinear
    semantics (:32-33): rows are grouped by expert (counts[e] consecutive rows each);
    y_e = GELU(x_e W1_e^T + b1_e) W2_e^T + b2_e with W1 [E,H,D], W2 [E,D,H]
    (layout: utils/helpers.py:645-662)."""
    hs = []
    start = 0
    ns = [int(c) for c in counts]
    for e, n in enumerate(ns):
        hs.append(gelu_erf(F.linear(rows[start:start + n], w1[e], b1[e])))
        start += n
    if not hs:
        return rows.new_zeros((0, w2.shape[1]))
    h_all = _drop("experts.act", torch.cat(hs, 0))             # Sequential(GELU(), Dropout(drop)) on the expert-major rows
    outs, start = [], 0
    for e, n in enumerate(ns):
        outs.append(F.linear(h_all[start:start + n], w2[e], b2[e]))
        start += n
    return torch.cat(outs, 0)


def moe_dispatch_ffn(x: torch.Tensor, idx: torch.Tensor, w1, b1, w2, b2):
    """_fmoe_general_global_forward(moe_inp, gate_top_k_idx, expert_fn, E, 1),
    call site models/moe/ckpt/custom_moe_layer.py:263-265: returns token-major
    [T*k, D] with row t*k+j = expert idx[t,j] applied to token t (evidenced by the
    view(-1, top_k, dim) + bmm at :291-305)."""
    T, k = idx.shape
    E = w1.shape[0]
    counts, offsets, pos, row_of_slot = route_build(idx, E)
    rows = x[row_of_slot // k]                       # MOEScatter (local gather)
    y = experts_ffn(rows, counts, w1, b1, w2, b2)    # expert-major
    return y[pos]                                    # MOEGather back to token-major


def moe_layer(x: torch.Tensor, gate_x: torch.Tensor, w_gate, w1, b1, w2, b2, top_k: int,
              noise=None, noise_std: float = 0.0, training: bool = True, idx_override=None):
    """FMoETransformerMLP.forward / forward_moe, models/moe/ckpt/custom_moe_layer.py:161-322
    (gate :213-219, dispatch :263-265, combine bmm(score[T,1,k], out[T,k,D]) :291-305).
    gate_x is the gate input ([T,D] or [T,D+gtsd] after the task-conditioning cat :176-179).
    Returns (out[T,D], clean, noisy, std, top_logits, gates, idx, score)."""
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    (idx, score), clean, noisy, std, top_logits, gates = gate_vmoe(
        gate_x, w_gate, top_k, noise, noise_std, training, idx_override)
    y = moe_dispatch_ffn(x2, idx, w1, b1, w2, b2)            # [T*k, D]
    y = y.view(-1, top_k, y.shape[-1])
    out = torch.bmm(score.view(-1, 1, top_k), y).reshape(-1, y.shape[-1])
    return out.reshape(shp), clean, noisy, std, top_logits, gates, idx, score


# ---------------------------------------------------------------- attention / block
def layernorm(x, w, b, eps: float = 1e-6):
    """norm_layer = partial(nn.LayerNorm, eps=1e-6), vision_transformer_moe.py:567."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def attention(x, wqkv, bqkv, wproj, bproj, num_heads: int):
    """Attention.forward, models/moe/ckpt/vision_transformer_moe.py:299-313
    (dense twin models/backbones/vit.py:177-207): qkv Linear -> [3,B,h,N,dh];
    softmax(q k^T * dh^-0.5) v; proj.  attn_drop = proj_drop = 0 in all configs."""
    B, N, C = x.shape
    dh = C // num_heads
    qkv = F.linear(x, wqkv, bqkv).reshape(B, N, 3, num_heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (dh ** -0.5)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return _drop("attn.proj", F.linear(o, wproj, bproj))


def mlp_dense(x, w1, b1, w2, b2):
    """Mlp.forward, models/moe/ckpt/vision_transformer_moe.py:255-261 (drop = 0)."""
    return _drop("mlp.out", F.linear(_drop("mlp.act", gelu_erf(F.linear(x, w1, b1))), w2, b2))


def task_embedding(params: Dict[str, torch.Tensor], num_tasks: int, task_id: int):
    """gate_task_represent(one_hot(task_id)), vision_transformer_moe.py:793-797 with
    new_Mlp :263-281 (fc1 -> GELU -> fc2 -> LayerNorm eps 1e-6)."""
    p = params
    one_hot = torch.zeros(num_tasks, dtype=p["gate_task_represent.fc1.weight"].dtype, device=p["gate_task_represent.fc1.weight"].device)
    one_hot[task_id] = 1.0
    h = gelu_erf(F.linear(one_hot, p["gate_task_represent.fc1.weight"], p["gate_task_represent.fc1.bias"]))
    h = F.linear(h, p["gate_task_represent.fc2.weight"], p["gate_task_represent.fc2.bias"])
    return layernorm(h, p["gate_task_represent.norm.weight"], p["gate_task_represent.norm.bias"])


class BackboneCfg:
    """Shape/config holder mirroring the VisionTransformerMoE ctor arguments that
    change the hot-path maths (vision_transformer_moe.py:565-572)."""

    def __init__(self, img_size=(224, 224), patch_size=16, in_chans=3, embed_dim=384, depth=12,
                 num_heads=12, mlp_ratio=4.0, moe_mlp_ratio=1.0, moe_experts=16, moe_top_k=4,
                 gate_dim=386, multi_gate=True, gate_task_specific_dim=-1, vmoe_noisy_std=0.0, dense_only=False):
        self.dense_only = dense_only        # BASELINE configs[0]: models/backbones/vit.py (no MoE blocks)
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
        return self.gate_dim - self.embed_dim          # vision_transformer_moe.py:634

    @property
    def num_tokens(self):
        return (self.img_size[0] // self.patch_size) * (self.img_size[1] // self.patch_size) + 1

    @property
    def d_gate(self):
        # custom_moe_layer.py:127-130
        return self.embed_dim if self.gate_task_specific_dim < 0 else self.embed_dim + self.gate_task_specific_dim

    def is_moe(self, i):
        return (i % 2 == 1) and not self.dense_only    # vision_transformer_moe.py:643-657


def _trunc_normal(shape, std, gen, dtype):
    t = torch.empty(shape, dtype=dtype)
    torch.nn.init.trunc_normal_(t, std=std, generator=gen)
    return t


def init_backbone_params(cfg: BackboneCfg, seed: int = 1, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Random-init weights with the reference's shapes/names (random_init: True).
    Linear trunc_normal(std .02)/bias 0, LN (1, 0): vision_transformer_moe.py:697-705;
    w_gate kaiming_uniform(a=sqrt 5): noisy_gate_vmoe.py:69; expert tensors
    [E,H,D]/[E,H]/[E,D,H]/[E,D]: utils/helpers.py:645-662.  Biases get small random
    values instead of zeros so that bias paths are exercised by parity tests."""
    g = torch.Generator().manual_seed(seed)
    D = cfg.embed_dim
    P = {}
    P["patch_embed.proj.weight"] = _trunc_normal((D, cfg.in_chans, cfg.patch_size, cfg.patch_size), .02, g, dtype)
    P["patch_embed.proj.bias"] = _trunc_normal((D,), .02, g, dtype)
    P["cls_token"] = _trunc_normal((1, 1, D), .02, g, dtype)
    P["pos_embed"] = _trunc_normal((1, cfg.num_tokens, D), .02, g, dtype)
    Hd = int(D * cfg.mlp_ratio)
    Hm = int(D * cfg.moe_mlp_ratio)
    E = cfg.moe_experts
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        for n in ("norm1", "norm2"):
            P[b + n + ".weight"] = 1.0 + _trunc_normal((D,), .02, g, dtype)
            P[b + n + ".bias"] = _trunc_normal((D,), .02, g, dtype)
        P[b + "attn.qkv.weight"] = _trunc_normal((3 * D, D), .02, g, dtype)
        P[b + "attn.qkv.bias"] = _trunc_normal((3 * D,), .02, g, dtype)
        P[b + "attn.proj.weight"] = _trunc_normal((D, D), .02, g, dtype)
        P[b + "attn.proj.bias"] = _trunc_normal((D,), .02, g, dtype)
        if cfg.is_moe(i):
            P[b + "mlp.experts.htoh4.weight"] = _trunc_normal((E, Hm, D), .02, g, dtype)
            P[b + "mlp.experts.htoh4.bias"] = _trunc_normal((E, Hm), .02, g, dtype)
            P[b + "mlp.experts.h4toh.weight"] = _trunc_normal((E, D, Hm), .02, g, dtype)
            P[b + "mlp.experts.h4toh.bias"] = _trunc_normal((E, D), .02, g, dtype)
            n_gates = cfg.num_tasks if cfg.multi_gate else 1
            for t in range(n_gates):
                w = torch.empty(cfg.d_gate, E, dtype=dtype)
                torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5), generator=g)
                key = b + (f"mlp.gate.{t}.w_gate" if cfg.multi_gate else "mlp.gate.w_gate")
                P[key] = w
        else:
            P[b + "mlp.fc1.weight"] = _trunc_normal((Hd, D), .02, g, dtype)
            P[b + "mlp.fc1.bias"] = _trunc_normal((Hd,), .02, g, dtype)
            P[b + "mlp.fc2.weight"] = _trunc_normal((D, Hd), .02, g, dtype)
            P[b + "mlp.fc2.bias"] = _trunc_normal((D,), .02, g, dtype)
    if cfg.gate_task_specific_dim >= 0 and not cfg.multi_gate:
        gt = cfg.gate_task_specific_dim
        P["gate_task_represent.fc1.weight"] = _trunc_normal((gt, cfg.num_tasks), .02, g, dtype)
        P["gate_task_represent.fc1.bias"] = _trunc_normal((gt,), .02, g, dtype)
        P["gate_task_represent.fc2.weight"] = _trunc_normal((gt, gt), .02, g, dtype)
        P["gate_task_represent.fc2.bias"] = _trunc_normal((gt,), .02, g, dtype)
        P["gate_task_represent.norm.weight"] = 1.0 + _trunc_normal((gt,), .02, g, dtype)
        P["gate_task_represent.norm.bias"] = _trunc_normal((gt,), .02, g, dtype)
    return P


def patch_embed(images, w, b, cls_token, pos_embed):
    """PatchEmbed.forward (conv 16x16 / stride 16) :330-341 then flatten/transpose,
    cls concat and + pos_embed, VisionTransformerMoE.forward_features :782-791."""
    x = F.conv2d(images, w, b, stride=w.shape[-1])
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((cls_token.expand(x.shape[0], -1, -1), x), dim=1)
    return x + pos_embed


def block_forward(params, cfg: BackboneCfg, i: int, x, task_id: Optional[int], tsf=None,
                  training: bool = True, noise=None, idx_override=None, path_scale=None):
    """Block._ckpt_main_moe / _ckpt_non_moe, vision_transformer_moe.py:438-487, and the
    cv-loss part of Block.forward :539-543 (mlp_drop = 0).  path_scale: None (drop_path = 0) or the two per-sample
    factors DropPath (:167-185) multiplies the attention and the MLP / MoE branch with: mask / keep_prob, each [B].
    Returns (x_out, cv_loss or None, aux dict)."""
    p = params
    b = f"blocks.{i}."
    _DROP["prefix"] = b
    s_attn = s_mlp = 1.0
    if path_scale is not None:
        s_attn, s_mlp = (v.to(x.dtype).view(-1, 1, 1) for v in path_scale)
    x = x + s_attn * attention(layernorm(x, p[b + "norm1.weight"], p[b + "norm1.bias"]),
                               p[b + "attn.qkv.weight"], p[b + "attn.qkv.bias"],
                               p[b + "attn.proj.weight"], p[b + "attn.proj.bias"], cfg.num_heads)
    normed = layernorm(x, p[b + "norm2.weight"], p[b + "norm2.bias"])
    if not cfg.is_moe(i):
        x = x + s_mlp * mlp_dense(normed, p[b + "mlp.fc1.weight"], p[b + "mlp.fc1.bias"],
                                  p[b + "mlp.fc2.weight"], p[b + "mlp.fc2.bias"])
        return x, None, {}
    T = normed.shape[0] * normed.shape[1]
    flat = normed.reshape(T, -1)
    if cfg.multi_gate:
        w_gate = p[b + f"mlp.gate.{task_id}.w_gate"]          # custom_moe_layer.py:213-214
        gate_x = flat
    else:
        w_gate = p[b + "mlp.gate.w_gate"]
        gate_x = flat
        if task_id is not None and tsf is not None:           # custom_moe_layer.py:176-179
            gate_x = torch.cat((flat, tsf.repeat(T, 1)), dim=-1)
    out, clean, noisy, std, top_logits, gates, idx, score = moe_layer(
        normed, gate_x, w_gate,
        p[b + "mlp.experts.htoh4.weight"], p[b + "mlp.experts.htoh4.bias"],
        p[b + "mlp.experts.h4toh.weight"], p[b + "mlp.experts.h4toh.bias"],
        cfg.moe_top_k, noise=noise, noise_std=cfg.vmoe_noisy_std, training=training,
        idx_override=idx_override)
    x = x + s_mlp * _drop("mlp_drop", out)
    importance = gates.sum(0)                                  # :453
    E = gates.shape[1]
    if cfg.moe_top_k < E and abs(std) > 1e-6:                  # :456-459
        load = prob_in_top_k(clean, noisy, std, top_logits, cfg.moe_top_k).sum(0)
    else:
        load = gates_to_load(gates)
    cv = (cv_squared(importance) + cv_squared(load)) if training else None   # :539-543
    return x, cv, {"importance": importance, "load": load, "idx": idx, "score": score,
                   "gates": gates, "clean": clean, "gate_x": gate_x, "w_gate": w_gate}


def backbone_forward(params, cfg: BackboneCfg, images, task_id: Optional[int], training: bool = True,
                     noises=None, route_override=None, path_scales=None, dropout=None):
    """VisionTransformerMoE.forward_features, vision_transformer_moe.py:780-880:
    returns (tokens[B,N,D] of the last block, total_cv_loss).  dropout: fn(site, tensor) -> tensor applied at the
    element-wise dropout sites (see _DROP above)."""
    p = params
    _DROP["fn"], _DROP["prefix"] = dropout, ""
    try:
        return _backbone_forward(p, cfg, images, task_id, training, noises, route_override, path_scales)
    finally:
        _DROP["fn"], _DROP["prefix"] = None, ""


def _backbone_forward(p, cfg, images, task_id, training, noises, route_override, path_scales):
    x = patch_embed(images, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"],
                    p["cls_token"], p["pos_embed"])
    x = _drop("pos", x)
    tsf = None
    if task_id is not None and "gate_task_represent.fc1.weight" in p:
        tsf = task_embedding(p, cfg.num_tasks, task_id)
    total_cv = torch.zeros((), dtype=x.dtype, device=x.device)
    aux_all = []
    for i in range(cfg.depth):
        noise = None if noises is None else noises.get(i)
        ovr = None if route_override is None else route_override.get(i)
        ps = None if path_scales is None else path_scales.get(i)
        x, cv, aux = block_forward(p, cfg, i, x, task_id, tsf, training, noise, ovr, ps)
        if cv is not None:
            total_cv = total_cv + cv
        aux_all.append(aux)
    return x, total_cv, aux_all
