"""Decoder head + multi-task wrapper around the hot path - the step AFTER the path.

`VisionTransformerUpHead` mirrors models/heads/vit_up_head.py:73-224 (module / state_dict names `norm`,
`conv_0..conv_4`, `syncbn_fc_0..3`; the "drop the cls token unless the token count is a multiple of 48" quirk of
:152-155; LayerNorm eps 1e-6; 3x3 conv + BN + ReLU + bilinear x2 stages, 1x1 classifier).  The token LayerNorm runs
on the HIP row kernel; the convolutions, batch norms and resizes are plain torch modules (MIOpen on ROCm) - they
are outside the MoE / attention path this repository hand-writes, and SURVEY plans "MIOpen first".
`MultiTaskModel` mirrors the backbone/decoder plumbing of models/models.py:215-342 for the two gate layouts the
path supports: one backbone pass shared by all heads (single / task-conditioned gate called without a task), or
one pass per task with that task's gate (multi-gate, :299-320), every head output resized to the input size.
Also mirrored, as plain torch modules behind the path (convolutions = MIOpen): the heads' multi-level outputs
(`output_level_0..2`, vit_up_head.py:128-131,184-214), their three TAM feature taps (:190-206) and `TamModule`
(models/models.py:11-134: a sigmoid gate B from the stacked task features, a 2x down / 2x up modulation M, one
3x3 + 1x1 head per task on cat_t(feature_t * (1 + M))), wired into `MultiTaskModel` as `tam_level0/1/2` (:246-279).
The sem regulariser is out of scope."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .functional import LayerNormFn


class ReluUp2xFn(torch.autograd.Function):
    """F.interpolate(F.relu(x), scale 2, mode='bilinear', align_corners=False) on a channels-last CUDA tensor as ONE kernel each
    way (m3_relu_up2x_fwd / _bwd; relu=False: the resize alone).  The output keeps x's dtype (out_fp32: fp32) - under fp16
    autocast torch's own resize returns fp32, 4x the input, and a cast kernel follows each way."""

    @staticmethod
    def forward(ctx, x, relu, out_fp32):
        from . import ops
        ctx.save_for_backward(x)
        ctx.relu = bool(relu)
        return ops.relu_up2x_fwd(x, relu=ctx.relu, out_dtype=torch.float32 if out_fp32 else None)

    @staticmethod
    def backward(ctx, g):
        from . import ops
        x, = ctx.saved_tensors
        if g.dtype not in (x.dtype, torch.float32):
            g = g.float()
        g = g.contiguous(memory_format=torch.channels_last)
        return ops.relu_up2x_bwd(g, x, relu=ctx.relu), None, None


def _fusable(x, align_corners):
    """the fused ReLU + x2 resize takes channels-last CUDA activations (what MIOpen's convolutions return under autocast) whose
    channel count fills 16-byte vectors"""
    return (x.is_cuda and not align_corners and x.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16, torch.float32)
            and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] % (4 if x.dtype == torch.float32 else 8) == 0)


class VisionTransformerUpHead(nn.Module):
    def __init__(self, img_size=(480, 640), patch_size=16, embed_dim=384, num_classes=40, num_conv=4,
                 num_upsampe_layer=4, conv3x3_conv1x1=True, align_corners=False, sync_bn=False,
                 act_dtype=torch.float32, channels=256, multi_level=False, tam=False, amp=False, fused_resize=True):
        """amp: run the conv / BN / resize stages under fp16 autocast on the GPU, as the reference does under its AMP step
        (pretrain/engine/train_one_epoch.py:35-61): at 480 x 640 (8 images, 40 classes) forward + backward 33.4 -> 13.9 ms,
        MIOpen's fp16 3x3 convolutions running at 406-475 TFLOP/s (tools/head_bench.py)."""
        super().__init__()
        self.amp = bool(amp)
        # fused_resize: ReLU + bilinear x2 of a stage as one hand-written kernel each way (ReluUp2xFn) whenever the stage's
        # tensor is channels-last on the GPU; forward + backward of the head at 480 x 640 under autocast: 13.7 -> see
        # tools/head_bench.py.  False: torch's relu + interpolate (what rounds 1-3 ran)
        self.fused_resize = bool(fused_resize)
        self.multi_level, self.tam = bool(multi_level), bool(tam)       # p['multi_level'], p['model_kwargs']['tam'] (:96-105)
        if (num_conv, num_upsampe_layer) not in ((4, 4), (2, 2), (2, 1)):
            raise NotImplementedError("supported stacks: num_conv / num_upsampe_layer = 4/4, 2/2, 2/1")
        self.img_size = tuple(img_size) if isinstance(img_size, (tuple, list)) else (img_size, img_size)
        self.h, self.w = self.img_size[0] // patch_size, self.img_size[1] // patch_size
        self.num_conv, self.num_upsampe_layer = num_conv, num_upsampe_layer
        self.align_corners = align_corners
        self.act_dtype = act_dtype
        self.num_classes = num_classes
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        bn = nn.SyncBatchNorm if sync_bn else nn.BatchNorm2d
        if num_conv == 2:
            self.conv_0 = nn.Conv2d(embed_dim, channels, 3, 1, 1) if conv3x3_conv1x1 else nn.Conv2d(embed_dim, channels, 1, 1)
            self.conv_1 = nn.Conv2d(channels, num_classes, 1, 1)
            self.syncbn_fc_0 = bn(channels)
        else:
            self.conv_0 = nn.Conv2d(embed_dim, channels, 3, 1, 1)
            for i in (1, 2, 3):
                setattr(self, f"conv_{i}", nn.Conv2d(channels, channels, 3, 1, 1))
            self.conv_4 = nn.Conv2d(channels, num_classes, 1, 1)
            for i in range(4):
                setattr(self, f"syncbn_fc_{i}", bn(channels))
        if self.multi_level:                                           # :128-131
            for i in range(3):
                setattr(self, f"output_level_{i}", nn.Conv2d(channels, num_classes, 1, 1))

    def _up(self, x, factor=None, size=None):
        if size is None:
            size = (x.shape[-2] * factor, x.shape[-1] * factor)
        return F.interpolate(x, size=size, mode="bilinear", align_corners=self.align_corners)

    def forward(self, x):
        if x.dim() == 3:
            if x.shape[1] % 48 != 0:                 # :153-154 (cls token present)
                x = x[:, 1:]
            n, hw, c = x.shape
            x = LayerNormFn.apply(x.contiguous().float(), self.norm.weight, self.norm.bias, self.norm.eps,
                                  self.act_dtype).float()
            x = x.transpose(1, 2).reshape(n, c, self.h, self.w)
        if self.amp and x.is_cuda:
            with torch.autocast("cuda", dtype=torch.float16):
                return self._stages(x)
        return self._stages(x)

    def _stages(self, x):
        if self.num_conv == 2:
            x = F.relu(self.syncbn_fc_0(self.conv_0(x)))
            if self.num_upsampe_layer == 2:
                x = self._up(x, size=x.shape[-1] * 4)          # :171 (square size, as the reference writes it)
            return self._up(self.conv_1(x), size=self.img_size)
        out, taps = {}, []
        want_taps = self.tam and self.training
        for i in range(4):
            x = getattr(self, f"syncbn_fc_{i}")(getattr(self, f"conv_{i}")(x))
            if self.fused_resize and i < 3 and not (want_taps and i >= 1) and _fusable(x, self.align_corners):
                x = ReluUp2xFn.apply(x, True, False)                     # relu + resize in one pass, dtype kept
            else:
                x = F.relu(x)
                if i >= 1:
                    taps.append(x)                                       # tam_feature0..2: after conv_1..3 (:190-206)
                if i < 3:
                    x = self._up(x, 2)
            if i < 3 and self.multi_level:
                out[f"level{i + 1}"] = getattr(self, f"output_level_{i}")(x)     # :184-200
        x = self.conv_4(x)
        x = ReluUp2xFn.apply(x, False, True) if (self.fused_resize and _fusable(x, self.align_corners)) else self._up(x, 2)
        if self.multi_level:
            out["final"] = x
            return out
        if self.tam and self.training:
            return x, taps[0], taps[1], taps[2]
        return x


class TamModule(nn.Module):
    """models/models.py:11-134 with the reference's module names (layers0..2, encoder0/1, decoder0/1, layers3/4[task]).
    norm: the layer build_norm_layer(norm_cfg) would give (BatchNorm2d / SyncBatchNorm).  The reference's activation
    checkpointing around the blocks changes memory only."""

    def __init__(self, tasks, input_channels, num_output, norm=nn.BatchNorm2d):
        super().__init__()
        self.tasks = list(tasks)
        if not 2 <= len(self.tasks) <= 5:
            raise ValueError("TamModule is defined for 2..5 tasks (models/models.py:87-95)")
        nt, c = len(self.tasks), input_channels
        self.layers0 = nn.Sequential(nn.Conv2d(nt * c, c, 3, 1, 1), norm(c))
        self.layers1 = nn.Sequential(nn.Conv2d(c, c, 3, 1, 1), norm(c))
        self.layers2 = nn.Sequential(nn.Conv2d(nt * c, c, 3, 1, 1), norm(c))
        self.encoder0 = nn.Sequential(nn.Conv2d(c, c, 3, 2, 1), norm(c))
        self.encoder1 = nn.Sequential(nn.Conv2d(c, c, 3, 2, 1), norm(c))
        self.decoder0 = nn.Sequential(nn.ConvTranspose2d(c, c, 3, 2, 1, output_padding=1), norm(c))
        self.decoder1 = nn.Sequential(nn.ConvTranspose2d(c, c, 3, 2, 1, output_padding=1), norm(c))
        self.layers3 = nn.ModuleDict({t: nn.Sequential(nn.Conv2d(nt * c, 256, 3, 1, 1), norm(256)) for t in self.tasks})
        self.layers4 = nn.ModuleDict({t: nn.Sequential(nn.Conv2d(256, num_output[t], 1, 1)) for t in self.tasks})

    @staticmethod
    def gate_weights(n_tasks, B):
        """the per-task factors of the gated concat (:87-95): B and 1 - B shared out over the first / last tasks"""
        return {2: [B, 1 - B], 3: [B, (1 - B) / 2, (1 - B) / 2], 4: [B / 2, B / 2, (1 - B) / 2, (1 - B) / 2],
                5: [B / 2, B / 2, (1 - B) / 3, (1 - B) / 3, (1 - B) / 3]}[n_tasks]

    def forward(self, deepfeature):
        feats = [deepfeature[t] for t in self.tasks]
        batch, c, H, W = feats[0].shape
        stacked = torch.stack(feats, dim=1).reshape(batch, len(feats) * c, H, W)
        B = torch.sigmoid(self.layers1(F.relu(self.layers0(stacked))))                       # _block0
        Fb = torch.cat([f * w for f, w in zip(feats, self.gate_weights(len(feats), B))], dim=1)
        Fb = F.relu(self.layers2(Fb))                                                        # _block2
        Fb = F.relu(self.encoder1(F.relu(self.encoder0(Fb))))                                # _encoder_block
        M = torch.sigmoid(self.decoder1(F.relu(self.decoder0(Fb))))                          # _decoder_block
        Ftam = torch.cat([f * (1 + M) for f in feats], dim=1)
        return {t: self.layers4[t](F.relu(self.layers3[t](Ftam))) for t in self.tasks}


class MultiTaskModel(nn.Module):
    """backbone(x[, task_id]) -> (tokens, cv_loss); decoders: {task: head}; tasks_id: {task: gate index}.
    tam_models: {0 / 1 / 2: TamModule} for the heads' feature taps (the reference's tam_level0/1/2, training only):
    their per-task outputs are added to the result as 'tam_level{l}_{task}' (models/models.py:246-279,313-327)."""

    def __init__(self, backbone: nn.Module, decoders: nn.ModuleDict, tasks, multi_gate: bool = False, tam_models=None):
        super().__init__()
        assert set(decoders.keys()) == set(tasks)
        self.backbone, self.decoders = backbone, decoders
        self.tasks = list(tasks)
        self.tasks_id = {t: i for i, t in enumerate(self.tasks)}          # models/models.py ctor: enumerate(tasks)
        self.multi_gate = multi_gate
        self.tam = bool(tam_models)
        self.tam_models = nn.ModuleDict({str(k): m for k, m in (tam_models or {}).items()})

    def _head(self, task, tokens, out_size, feats=None):
        y = self.decoders[task](tokens)
        if self.tam and self.training:                                    # head returns (out, tap0, tap1, tap2)
            y, *taps = y
            for lvl in self.tam_models:
                feats[lvl][task] = taps[int(lvl)]
        return F.interpolate(y, out_size, mode="bilinear")

    def _tam_outputs(self, out, feats, out_size):
        for lvl, model in self.tam_models.items():
            y = model(feats[lvl])
            for t in self.tasks:
                out[f"tam_level{lvl}_{t}"] = F.interpolate(y[t], out_size, mode="bilinear", align_corners=False)

    def forward(self, x, single_task=None, task_id=None):
        if task_id is not None:
            assert self.tasks_id[single_task] == task_id                  # :216-217
        out_size = x.shape[2:]
        use_tam = self.tam and self.training
        feats = {lvl: {} for lvl in self.tam_models} if use_tam else None
        if not self.multi_gate:
            tokens, cv = self.backbone(x) if task_id is None else self.backbone(x, task_id=task_id)
            if single_task is not None:                                   # :251-256 (no TAM on the single-task call)
                y = self.decoders[single_task](tokens)
                y = y[0] if isinstance(y, tuple) else y
                return {single_task: F.interpolate(y, out_size, mode="bilinear")}, cv
            out = {t: self._head(t, tokens, out_size, feats) for t in self.tasks}
            if use_tam:
                self._tam_outputs(out, feats, out_size)
            return out, cv
        out, total = {}, None
        for t in self.tasks:                                              # :299-320: one pass per task's gate
            tokens, cv = self.backbone(x, task_id=self.tasks_id[t])
            total = cv if total is None else total + cv
            out[t] = self._head(t, tokens, out_size, feats)
        if use_tam:
            self._tam_outputs(out, feats, out_size)
        return out, total
