"""fmoe.gates.NaiveGate: linear scores -> top-k -> softmax over the k (only the default value of
FMoETransformerMLP's `gate` argument in the reference, custom_moe_layer.py:82)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .base_gate import BaseGate


class NaiveGate(BaseGate):
    def __init__(self, d_model, num_expert, world_size, top_k=2):
        super().__init__(num_expert, world_size)
        self.gate = nn.Linear(d_model, self.tot_expert)
        self.top_k = top_k

    def forward(self, inp, return_all_scores=False):
        gate = self.gate(inp)
        val, idx = torch.topk(gate, k=self.top_k, dim=-1, largest=True, sorted=False)
        val = val.view(-1, self.top_k)
        score = F.softmax(val, dim=-1)
        if return_all_scores:
            return idx, score, gate
        return idx, score
