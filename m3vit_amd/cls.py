"""224x224 classification wrapper + AMP training step around the hot path.

Mirrors `MoEViTForImageNet` / `MoEViTConfig` (pretrain/models/moe_vit_cls.py:14-212: encoder ->
final LayerNorm -> cls-token head, returns {"logits", "cv_loss"}) and the body of the AMP iteration
(pretrain/engine/train_one_epoch.py:28-61: autocast forward, loss + moe_cv_weight * cv_loss,
GradScaler scale / unscale / clip / step / update).  The encoder is `m3vit_amd.vit.VisionTransformerMoE`
(HIP kernels); the final norm and the head run on the same LayerNorm / GEMM kernels.  fp16 autocast is
expressed as `act_dtype=torch.float16` (fp16 activations into the MFMA GEMMs, fp32 residual stream and
accumulation, fp32 master weights) - the custom autograd Functions pick their dtypes themselves, so a
surrounding `torch.autocast` context is harmless and not needed.

Out of scope here as in the reference's own hot path: distillation token / head_dist, DeiT warm start
(network fetch), mixup, EMA."""
from dataclasses import dataclass
import math

import torch
import torch.nn as nn

from .functional import LayerNormFn, PlainLinearFn
from .vit import VisionTransformerMoE


@dataclass
class MoEViTConfig:                      # pretrain/models/moe_vit_cls.py:14-42 (same field names)
    model_name: str = "vit_small_patch16_224"
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    embed_dim: int = 384
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    qkv_bias: bool = True
    distilled: bool = False
    drop_rate: float = 0.0
    attn_drop_rate: float = 0.0
    drop_path_rate: float = 0.0
    num_classes: int = 1000
    random_init: bool = True
    pos_embed_interp: bool = False
    align_corners: bool = False
    moe_mlp_ratio: float = 1.0
    moe_experts: int = 16
    moe_top_k: int = 4
    moe_gate_type: str = "noisy_vmoe"
    vmoe_noisy_std: float = 1.0
    gate_dim: int = 384
    gate_task_specific_dim: int = -1
    multi_gate: bool = False
    world_size: int = 1
    use_checkpointing: bool = False


class HipLinear(nn.Linear):
    """nn.Linear whose forward / backward are the NT / TN GEMM kernels; activations in `act_dtype`."""
    act_dtype = torch.float32

    def forward(self, x):
        return PlainLinearFn.apply(x.to(self.act_dtype), self.weight, self.bias)


class MoEViTForImageNet(nn.Module):
    def __init__(self, cfg: MoEViTConfig, act_dtype=torch.float32):
        super().__init__()
        if cfg.distilled:
            raise NotImplementedError("distillation token / head_dist are outside the hot path")
        if not cfg.random_init:
            raise NotImplementedError("DeiT warm start needs a network fetch; load a state_dict instead")
        self.cfg = cfg
        self.act_dtype = act_dtype
        self.encoder = VisionTransformerMoE(
            model_name=cfg.model_name, img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans,
            embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, num_classes=cfg.num_classes,
            mlp_ratio=cfg.mlp_ratio, qkv_bias=cfg.qkv_bias, qk_scale=None, drop_rate=cfg.drop_rate,
            attn_drop_rate=cfg.attn_drop_rate, drop_path_rate=cfg.drop_path_rate, moe_mlp_ratio=cfg.moe_mlp_ratio,
            moe_experts=cfg.moe_experts, moe_top_k=cfg.moe_top_k, world_size=cfg.world_size, gate_dim=cfg.gate_dim,
            moe_gate_type=cfg.moe_gate_type, vmoe_noisy_std=cfg.vmoe_noisy_std,
            gate_task_specific_dim=cfg.gate_task_specific_dim, multi_gate=cfg.multi_gate, num_tasks=-1,
            use_checkpointing=cfg.use_checkpointing, act_dtype=act_dtype, random_init=cfg.random_init)
        self.norm = nn.LayerNorm(cfg.embed_dim)                       # eps = 1e-5 (nn.LayerNorm default, :97)
        self.head = HipLinear(cfg.embed_dim, cfg.num_classes)
        self.head.act_dtype = act_dtype
        self.head_dist = None
        nn.init.trunc_normal_(self.head.weight, std=0.02)
        nn.init.zeros_(self.head.bias)

    def no_weight_decay(self):
        return {"encoder.pos_embed", "encoder.cls_token"}

    def forward(self, x):
        tokens, cv_loss = self.encoder(x)
        if tokens.ndim != 3:
            raise RuntimeError(f"Expected token output [B, N, C], got shape {tuple(tokens.shape)}")
        # LayerNorm is per row and only the cls row feeds the head: normalise that row alone (same value and
        # gradient as norm(tokens)[:, 0], :189-190)
        cls = LayerNormFn.apply(tokens[:, 0].contiguous(), self.norm.weight, self.norm.bias, self.norm.eps,
                                self.act_dtype)
        logits_cls = self.head(cls).float()
        return {"logits": logits_cls, "cv_loss": cv_loss}


def amp_train_step(model, criterion, optimizer, scaler, samples, targets, moe_cv_weight=0.01, clip_grad=None):
    """One iteration of pretrain/engine/train_one_epoch.py:35-61.  criterion(samples, logits, targets) as there.
    Returns (loss value, cv_loss value or None)."""
    out = model(samples)
    logits = out["logits"] if isinstance(out, dict) else out
    cv_loss = out.get("cv_loss", None) if isinstance(out, dict) else None
    loss = criterion(samples, logits, targets)
    cv = None
    if cv_loss is not None:
        cv = cv_loss.mean() if torch.is_tensor(cv_loss) else torch.tensor(float(cv_loss), device=logits.device)
        loss = loss + moe_cv_weight * cv
    loss_value = loss.item()
    if not math.isfinite(loss_value):
        raise RuntimeError(f"Loss is {loss_value}, stopping training")
    optimizer.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    if clip_grad is not None:
        scaler.unscale_(optimizer)
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad)
    scaler.step(optimizer)
    scaler.update()
    return loss_value, (float(cv.detach().item()) if cv is not None else None)
