"""fmoe.linear.FMoELinear on the grouped MFMA GEMM (custom_moe_layer.py:32-33):
weight [E, out, in], bias [E, out]; forward(inp[R, in], fwd_expert_count) -> [R, out] with rows
grouped by expert."""
import math

import torch
import torch.nn as nn

from ..functional import GroupedLinearFn
from .functions import route_meta_from_counts


class FMoELinear(nn.Module):
    def __init__(self, num_expert: int, in_feat: int, out_feat: int, bias: bool = True, rank: int = 0):
        super().__init__()
        self.num_expert = num_expert
        self.in_feat = in_feat
        self.out_feat = out_feat
        self.rank = rank
        self.weight = nn.Parameter(torch.Tensor(num_expert, out_feat, in_feat))
        if bias:
            self.bias = nn.Parameter(torch.zeros(num_expert, out_feat))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        # fastmoe: kaiming_uniform(a=sqrt 5) per expert
        torch.nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward(self, inp, fwd_expert_count):
        offsets, tile_starts = route_meta_from_counts(fwd_expert_count, inp.device)
        return GroupedLinearFn.apply(inp, self.weight, self.bias, offsets, tile_starts)

    def extra_repr(self):
        return f"num_expert={self.num_expert}, in_features={self.in_feat}, out_features={self.out_feat}, " \
               f"bias={self.bias is not None}, rank={self.rank}"
