"""fmoe.layers.FMoE and _fmoe_general_global_forward with the attributes / call contract the
reference relies on (custom_moe_layer.py:99-159,197-265)."""
import torch
import torch.nn as nn

from .functions import MOEGather, MOEScatter, prepare_forward
from .gates import NaiveGate
from .linear import FMoELinear


def mark_module_parallel_comm(module, comm):
    """Tag every parameter of `module` with dp_comm = comm ("none" for expert weights, so that
    DistributedGroupedDataParallel.allreduce_params skips them)."""
    for p in module.parameters():
        setattr(p, "dp_comm", comm)


def _fmoe_general_global_forward(inp, gate, expert_fn, num_expert, world_size, **kwargs):
    """(moe_inp [T,D], gate_top_k_idx i64 [T,k], expert_fn, E_local, world_size) -> [T*k, D_out]
    token-major (row t*k+j = expert gate[t,j] applied to token t); differentiable w.r.t. moe_inp and
    whatever expert_fn closes over.  Call site custom_moe_layer.py:263-265."""
    k = gate.shape[1] if gate.dim() > 1 else 1
    if world_size > 1:
        from ..ep import general_global_forward_ep
        return general_global_forward_ep(inp, gate, expert_fn, num_expert, world_size)
    route, _, _, fwd_expert_count, _ = prepare_forward(gate, num_expert, world_size)
    x = MOEScatter.apply(inp, route, k)
    x = expert_fn(x, fwd_expert_count)
    return MOEGather.apply(x, route)


class FMoE(nn.Module):
    def __init__(self, num_expert=32, d_model=1024, world_size=1, mp_group=None, slice_group=None,
                 moe_group=None, top_k=2, gate=NaiveGate, expert=None, gate_hook=None, mask=None, mask_dict=None):
        super().__init__()
        self.num_expert = num_expert
        self.d_model = d_model
        self.world_size = world_size
        self.slice_group = slice_group if slice_group is not None else mp_group
        if self.slice_group is None:
            self.slice_size, self.slice_rank = 1, 0
        else:
            self.slice_size = self.slice_group.size()
            self.slice_rank = self.slice_group.rank()
        self.top_k = top_k
        if isinstance(expert, list):
            self.experts = nn.ModuleList([e(d_model) for e in expert])
            self.experts_fused = False
            self.num_expert = num_expert = len(expert)
        elif expert is not None:
            self.experts = nn.ModuleList([expert(d_model) for _ in range(num_expert)])
            self.experts_fused = False
        else:
            self.experts_fused = True
        try:
            self.gate = gate(d_model, num_expert, world_size, top_k)
        except TypeError:
            self.gate = None          # subclasses (FMoETransformerMLP) build their own gate(s)
        self.gate_hook = gate_hook
        self.mask = mask
        self.mask_dict = mask_dict
        self.moe_group = moe_group

    def expert_fn(self, inp, fwd_expert_count):
        if self.experts_fused:
            return self.experts(inp, fwd_expert_count)
        outputs, base = [], 0
        counts = [int(c) for c in torch.as_tensor(fwd_expert_count).tolist()]
        for i, n in enumerate(counts):
            outputs.append(self.experts[i](inp[base:base + n]))
            base += n
        return torch.cat(outputs, dim=0)

    def mark_parallel_comm(self, expert_dp_comm="none"):
        if self.experts is not None:
            comm = expert_dp_comm
            if isinstance(self.experts, list):
                for e in self.experts:
                    mark_module_parallel_comm(e, comm)
            else:
                mark_module_parallel_comm(self.experts, comm)
        if self.gate is not None:
            mark_module_parallel_comm(self.gate, "gate")

    def forward(self, moe_inp):
        gate_top_k_idx, gate_score = self.gate(moe_inp)
        fwd = _fmoe_general_global_forward(moe_inp, gate_top_k_idx, self.expert_fn, self.num_expert, self.world_size)
        out = fwd.view(-1, self.top_k, self.d_model)
        return torch.bmm(gate_score.view(-1, 1, self.top_k), out).reshape(-1, self.d_model)
