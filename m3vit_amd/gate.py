"""NoisyGate_VMoE mirror (models/moe/ckpt/noisy_gate_vmoe.py:16-275) on the fused HIP gate kernel.

Same constructor arguments, parameter name (`w_gate [d_model, tot_expert]`, kaiming_uniform a=sqrt 5),
`select_idx`, `get_activation/has_activation`, BaseGate loss bookkeeping, and the 6-tuple return
((idx, score), clean_logits, noisy_logits, noise_stddev, top_logits, gates) of :257-264.
noise_stddev is a Python float (:92-93).  Noise is drawn with torch.randn (same distribution as
randn_like at :168) and handed to the kernel; pass `noise=` to forward to pin it.
The regu_sem / regu_subimage regularisers (:95-162) are OUT OF SCOPE and raise.

convention="origin": the API of models/moe/origin/noisy_gate_vmoe.py:168-297 - forward returns (idx, score) only,
stores the block's balance loss cv^2(importance) + cv^2(load) (0 in eval) with set_loss() for
utils/moe_utils.py::collect_noisy_gating_loss, and keeps the softmax probabilities as `activation`."""
import math

import torch
import torch.nn as nn

from .fmoe.gates.base_gate import BaseGate
from .functional import GateFn


class NoisyGate_VMoE(BaseGate):
    def __init__(self, d_model, num_expert, world_size, top_k=2, noise_std=1, no_noise=False,
                 return_decoupled_activation=False, regu_experts_fromtask=False, num_experts_pertask=-1, num_tasks=-1,
                 regu_sem=False, sem_force=False, regu_subimage=False, group_size=4, convention="ckpt"):
        super().__init__(num_expert, world_size)
        assert convention in ("ckpt", "origin")
        self.convention = convention
        if regu_sem or regu_subimage or return_decoupled_activation:
            raise NotImplementedError("regu_sem / regu_subimage / decoupled activation are outside the hot path")
        self.w_gate = nn.Parameter(torch.zeros(d_model, self.tot_expert), requires_grad=True)
        self.top_k = top_k
        self.no_noise = no_noise
        self.noise_std = noise_std
        self.group_size = group_size
        self.activation = None
        self.select_idx = None
        self.regu_experts_fromtask = regu_experts_fromtask
        self.num_experts_pertask = num_experts_pertask
        self.num_tasks = num_tasks
        if self.regu_experts_fromtask:
            self.start_experts_id = []
            start_id = 0
            for i in range(self.num_tasks):
                start_id = start_id + int(i * (self.tot_expert - self.num_experts_pertask) / (self.num_tasks - 1))
                self.start_experts_id.append(start_id)
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.kaiming_uniform_(self.w_gate, a=math.sqrt(5))

    def forward(self, inp, task_id=None, sem=None, noise=None, task_specific_feature=None):
        """inp [..., D] (token features; the task-conditioned tail of the gate input is passed as
        `task_specific_feature` [gtsd] and folded into a logit bias instead of materialising the
        cat of custom_moe_layer.py:176-179)."""
        shp = list(inp.shape)
        D = shp[-1]
        other = shp[:-1]
        x = inp.reshape(-1, D)
        w = self.w_gate
        if self.regu_experts_fromtask and task_id is not None:          # :87-89
            s0 = self.start_experts_id[task_id]
            w = w[:, s0:s0 + self.num_experts_pertask].contiguous()
            raw_std = self.noise_std / self.num_experts_pertask
        else:
            raw_std = self.noise_std / self.tot_expert
        noise_stddev = raw_std * self.training                          # python float, :93
        if self.no_noise:
            noise_stddev *= 0
        if self.select_idx is not None:
            raise NotImplementedError("expert pruning via select_idx is outside the hot path")
        bias = None
        if task_specific_feature is not None:
            bias = (task_specific_feature.float().reshape(1, -1) @ w[D:]).reshape(-1).contiguous()
        if noise is None and abs(noise_stddev) > 0:
            noise = torch.randn(x.shape[0], w.shape[1], device=x.device)
        E = w.shape[1]
        idx, score, clean, noisy, top_logits, gates, idx32, imp, load = GateFn.apply(
            x, w, min(self.top_k, E), noise if abs(noise_stddev) > 0 else None, float(noise_stddev), bias)
        self._last = dict(idx32=idx32, importance=imp, load=load)
        k = idx.shape[1]
        if self.convention == "origin":                                   # origin/noisy_gate_vmoe.py:277-297
            if self.training:
                from .balance import block_balance_loss
                loss = block_balance_loss(gates, clean, noisy, noise_stddev, top_logits, k)
            else:
                loss = 0
            self.set_loss(loss)
            self.activation = torch.softmax(noisy.detach(), dim=1).reshape(other + [-1]).contiguous()
            return idx.reshape(other + [k]).contiguous(), score.reshape(other + [k]).contiguous()
        return ((idx.reshape(other + [k]), score.reshape(other + [k])), clean, noisy, noise_stddev, top_logits, gates)

    def get_activation(self, clear=True):
        a = self.activation
        if clear:
            self.activation = None
        return a

    @property
    def has_activation(self):
        return self.activation is not None
