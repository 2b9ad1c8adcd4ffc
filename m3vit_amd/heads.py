"""Decoder head + multi-task wrapper around the hot path - the step AFTER the path.

`VisionTransformerUpHead` mirrors models/heads/vit_up_head.py:73-224 (module / state_dict names `norm`,
`conv_0..conv_4`, `syncbn_fc_0..3`; the "drop the cls token unless the token count is a multiple of 48" quirk of
:152-155; LayerNorm eps 1e-6; 3x3 conv + BN + ReLU + bilinear x2 stages, 1x1 classifier).  The token LayerNorm runs
on the HIP row kernel; the convolutions, batch norms and resizes are plain torch modules (MIOpen on ROCm) - they
are outside the MoE / attention path this repository hand-writes, and SURVEY plans "MIOpen first".
`MultiTaskModel` mirrors the backbone/decoder plumbing of models/models.py:215-342 for the two gate layouts the
path supports: one backbone pass shared by all heads (single / task-conditioned gate called without a task), or
one pass per task with that task's gate (multi-gate, :299-320), every head output resized to the input size.
TAM feature aggregation, multi-level outputs and the sem regulariser are out of scope."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .functional import LayerNormFn


class VisionTransformerUpHead(nn.Module):
    def __init__(self, img_size=(480, 640), patch_size=16, embed_dim=384, num_classes=40, num_conv=4,
                 num_upsampe_layer=4, conv3x3_conv1x1=True, align_corners=False, sync_bn=False,
                 act_dtype=torch.float32, channels=256):
        super().__init__()
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
        if self.num_conv == 2:
            x = F.relu(self.syncbn_fc_0(self.conv_0(x)))
            if self.num_upsampe_layer == 2:
                x = self._up(x, size=x.shape[-1] * 4)          # :171 (square size, as the reference writes it)
            return self._up(self.conv_1(x), size=self.img_size)
        for i in range(4):
            x = F.relu(getattr(self, f"syncbn_fc_{i}")(getattr(self, f"conv_{i}")(x)))
            if i < 3:
                x = self._up(x, 2)
        return self._up(self.conv_4(x), 2)


class MultiTaskModel(nn.Module):
    """backbone(x[, task_id]) -> (tokens, cv_loss); decoders: {task: head}; tasks_id: {task: gate index}."""

    def __init__(self, backbone: nn.Module, decoders: nn.ModuleDict, tasks, multi_gate: bool = False):
        super().__init__()
        assert set(decoders.keys()) == set(tasks)
        self.backbone, self.decoders = backbone, decoders
        self.tasks = list(tasks)
        self.tasks_id = {t: i for i, t in enumerate(self.tasks)}          # models/models.py ctor: enumerate(tasks)
        self.multi_gate = multi_gate

    def _head(self, task, tokens, out_size):
        return F.interpolate(self.decoders[task](tokens), out_size, mode="bilinear")

    def forward(self, x, single_task=None, task_id=None):
        if task_id is not None:
            assert self.tasks_id[single_task] == task_id                  # :216-217
        out_size = x.shape[2:]
        if not self.multi_gate:
            tokens, cv = self.backbone(x) if task_id is None else self.backbone(x, task_id=task_id)
            names = [single_task] if single_task is not None else self.tasks
            return {t: self._head(t, tokens, out_size) for t in names}, cv
        out, total = {}, None
        for t in self.tasks:                                              # :299-320: one pass per task's gate
            tokens, cv = self.backbone(x, task_id=self.tasks_id[t])
            total = cv if total is None else total + cv
            out[t] = self._head(t, tokens, out_size)
        return out, total
