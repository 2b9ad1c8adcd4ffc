"""FMoETransformerMLP / _Expert mirror (models/moe/ckpt/custom_moe_layer.py:24-44,66-322).

Same constructor signature, attribute names and state_dict keys
(`experts.htoh4.weight [E,H,D]`, `experts.htoh4.bias [E,H]`, `experts.h4toh.weight [E,D,H]`,
`experts.h4toh.bias [E,D]`, `gate.w_gate` or `gate.{t}.w_gate`), same forward signature and
6-tuple return (:180-181).  forward_moe runs the fused HIP path
(gate -> route_build -> grouped FC1+GELU -> grouped FC2 -> combine) when the expert activation
is the reference's Sequential(GELU(), Dropout(0)); any other activation goes through the
composable _fmoe_general_global_forward + bmm exactly as :263-305."""
import torch
import torch.nn as nn

from .fmoe.gates import NaiveGate
from .fmoe.layers import FMoE, _fmoe_general_global_forward
from .fmoe.linear import FMoELinear
from .functional import CombineFn, GroupedFFNFn
from .gate import NoisyGate_VMoE


class _Expert(nn.Module):
    def __init__(self, num_expert, d_model, d_hidden, activation, rank=0):
        super().__init__()
        self.htoh4 = FMoELinear(num_expert, d_model, d_hidden, bias=True, rank=rank)
        self.h4toh = FMoELinear(num_expert, d_hidden, d_model, bias=True, rank=rank)
        self.activation = activation

    def forward(self, inp, fwd_expert_count):
        x = self.htoh4(inp, fwd_expert_count)
        x = self.activation(x)
        x = self.h4toh(x, fwd_expert_count)
        return x


def _is_plain_gelu(act):
    if isinstance(act, nn.GELU):
        return getattr(act, "approximate", "none") == "none"
    if isinstance(act, nn.Sequential):
        mods = list(act)
        return (len(mods) >= 1 and isinstance(mods[0], nn.GELU) and getattr(mods[0], "approximate", "none") == "none"
                and all(isinstance(m, nn.Dropout) and m.p == 0.0 for m in mods[1:]))
    return False


class FMoETransformerMLP(FMoE):
    def __init__(self, num_expert=32, d_model=1024, d_gate=1024, d_hidden=4096, activation=torch.nn.GELU(),
                 expert_dp_comm="none", expert_rank=0, gate=NaiveGate, world_size=1, top_k=2, vmoe_noisy_std=1,
                 gate_return_decoupled_activation=False, gate_task_specific_dim=-1, multi_gate=False,
                 regu_experts_fromtask=False, num_experts_pertask=-1, num_tasks=-1, regu_sem=False, sem_force=False,
                 regu_subimage=False, expert_prune=False, prune_threshold=0.1, convention="ckpt", **kwargs):
        super().__init__(num_expert=num_expert, d_model=d_model, gate=gate, world_size=world_size, top_k=top_k, **kwargs)
        assert convention in ("ckpt", "origin")
        self.convention = convention       # "origin": forward returns the tensor only (origin/custom_moe_layer.py:161-181)
        if regu_sem or regu_subimage:
            raise NotImplementedError("regu_sem / regu_subimage are outside the hot path (SURVEY 8a a4)")
        self.sem_force = sem_force
        if sem_force:
            # NYUD-40 classes -> 8 expert pairs (custom_moe_layer.py:112-113; configuration data of the reference)
            self.force_id = [[0], [1, 17, 18, 19, 20], [2, 12, 13, 14, 15, 16], [3, 9, 10, 11], [4, 5], [6, 7, 8, 38],
                             [21, 22, 23, 24, 25, 26, 39], [27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37]]
        self.our_d_gate = d_gate
        self.our_d_model = d_model
        self.num_expert = num_expert
        self.regu_experts_fromtask = regu_experts_fromtask
        self.num_experts_pertask = num_experts_pertask
        self.num_tasks = num_tasks
        self.expert_prune = expert_prune
        self.prune_threshold = prune_threshold
        if self.regu_experts_fromtask:
            self.start_experts_id = []
            start_id = 0
            for i in range(self.num_tasks):
                start_id = start_id + int(i * (self.num_expert - self.num_experts_pertask) / (self.num_tasks - 1))
                self.start_experts_id.append(start_id)
        self.experts = _Expert(num_expert, d_model, d_hidden, activation, rank=expert_rank)
        self.experts_fused = True
        self.gate_task_specific_dim = gate_task_specific_dim
        self.multi_gate = multi_gate
        d_gate_in = d_model if gate_task_specific_dim < 0 else d_model + gate_task_specific_dim
        if gate is not NoisyGate_VMoE:
            raise ValueError("No such gating type")       # only noisy_vmoe works with this layer
        mk = lambda: gate(d_gate_in, num_expert, world_size, top_k,                      # noqa: E731
                          return_decoupled_activation=gate_return_decoupled_activation, noise_std=vmoe_noisy_std,
                          regu_experts_fromtask=regu_experts_fromtask, num_experts_pertask=num_experts_pertask,
                          num_tasks=num_tasks, regu_sem=False, sem_force=False, regu_subimage=False, convention=convention)
        if self.multi_gate:
            self.gate = nn.ModuleList([mk() for _ in range(self.our_d_gate - self.our_d_model)])
        else:
            self.gate = mk()
        self.mark_parallel_comm(expert_dp_comm)

    def mark_parallel_comm(self, expert_dp_comm="none"):
        from .fmoe.layers import mark_module_parallel_comm
        mark_module_parallel_comm(self.experts, expert_dp_comm)
        mark_module_parallel_comm(self.gate, "gate")

    def forward(self, inp, gate_inp=None, task_id=None, task_specific_feature=None, sem=None):
        if gate_inp is None:
            gate_inp = inp
        original_shape = inp.shape
        inp = inp.reshape(-1, self.d_model)
        gate_inp = gate_inp.reshape(-1, gate_inp.shape[-1])
        tsf = None
        if (task_id is not None) and (task_specific_feature is not None):
            assert self.multi_gate is False
            tsf = task_specific_feature
        out, clean, noisy, std, top_logits, gates = self.forward_moe(gate_inp, inp, task_id=task_id, sem=sem, tsf=tsf)
        if self.convention == "origin":
            return out.reshape(original_shape)
        return out.reshape(original_shape), clean, noisy, std, top_logits, gates

    def _force_by_semantics(self, idx, score, sem):
        """sem_force (custom_moe_layer.py:225-243), vectorised: a patch whose semantic class is in force_id[j] is sent
        to the expert pair (2j, 2j+1) (pattern repeated to fill top_k); token 0 of every image (cls) keeps its
        routing; EVERY score becomes 0.5 (constant: no gradient reaches the gate through the combine)."""
        k = self.top_k
        B = sem.shape[0]
        idx3 = idx.reshape(B, -1, k).clone()
        semf = sem.reshape(B, -1).to(idx.device)
        patches = idx3[:, 1:1 + semf.shape[1]]
        for j, classes in enumerate(self.force_id):                # later groups override earlier ones, as the loop does
            hit = torch.isin(semf, torch.tensor(classes, device=semf.device, dtype=semf.dtype))
            pattern = torch.tensor(([2 * j, 2 * j + 1] * ((k + 1) // 2))[:k], device=idx.device, dtype=idx.dtype)
            patches[hit] = pattern
        return idx3.reshape(-1, k), torch.full_like(score, 0.5)

    def forward_moe(self, gate_inp, moe_inp, task_id=None, sem=None, tsf=None):
        clean = noisy = std = top_logits = gates = None
        if (task_id is not None) and self.multi_gate:
            g = self.gate[task_id]
            if self.convention == "origin":                 # origin/custom_moe_layer.py:213-215: a 2-tuple, no keep-alive
                idx, score = g(gate_inp)
            else:
                (idx, score), clean, noisy, std, top_logits, gates = g(gate_inp, sem=sem)
                unused = sum(p.sum() for i, gg in enumerate(self.gate) if i != task_id for p in gg.parameters())
                clean = clean + 0.0 * unused                 # DDP keep-alive, :216-217
        else:
            g = self.gate
            if self.convention == "origin":
                idx, score = g(gate_inp, task_id=task_id, sem=sem, task_specific_feature=tsf)
            else:
                (idx, score), clean, noisy, std, top_logits, gates = g(gate_inp, task_id=task_id, sem=sem,
                                                                       task_specific_feature=tsf)
        if self.expert_prune:
            score = torch.where(score > self.prune_threshold, score, torch.zeros_like(score))
        idx32 = g._last["idx32"]
        if self.sem_force and (sem is not None):
            idx, score = self._force_by_semantics(idx, score, sem)          # :225-243
            idx32 = idx.to(torch.int32)
        if self.regu_experts_fromtask and (task_id is not None):
            idx = idx + self.start_experts_id[task_id]
            idx32 = idx.to(torch.int32)
        if self.gate_hook is not None:
            self.gate_hook(idx, score, None)
        e = self.experts
        if self.world_size == 1 and _is_plain_gelu(e.activation) and self.mask is None:
            out = GroupedFFNFn.apply(moe_inp, idx32.reshape(-1, idx.shape[-1]), score.reshape(-1, idx.shape[-1]),
                                     e.htoh4.weight, e.htoh4.bias, e.h4toh.weight, e.h4toh.bias)
            out = out.to(moe_inp.dtype)
        else:
            fwd = _fmoe_general_global_forward(moe_inp, idx.reshape(-1, idx.shape[-1]), self.expert_fn, self.num_expert,
                                               self.world_size)
            out = CombineFn.apply(fwd.view(-1, self.top_k, fwd.shape[-1]), score.reshape(-1, self.top_k)).to(moe_inp.dtype)
        return out, clean, noisy, std, top_logits, gates


class TokenFMoETransformerMLP(FMoE):
    """Pre-routed MoE MLP (models/moe/token/custom_moe_layer.py:55-156): the Block owns the gate and hands the
    layer `gate_top_k_idx [T,k]` and `gate_score [T,k]`; the layer only dispatches, runs the experts and
    combines - on any subset of tokens (the token-MoE branch gathers the tokens selected by its compute mask
    before calling it, token/vision_transformer_moe.py:730-812).  Same constructor, attributes and state_dict
    keys as the reference; same kernels as FMoETransformerMLP behind it."""

    def __init__(self, num_expert=32, d_model=1024, d_gate=1024, d_hidden=4096, activation=torch.nn.GELU(),
                 expert_dp_comm="none", expert_rank=0, world_size=1, top_k=2, **kwargs):
        # a dummy gate for the parent, as in the reference (:76-77)
        super().__init__(num_expert=num_expert, d_model=d_model, gate=NaiveGate, world_size=world_size, top_k=top_k,
                         **kwargs)
        self.our_d_model = d_model
        self.num_expert = num_expert
        self.experts = _Expert(num_expert, d_model, d_hidden, activation, rank=expert_rank)
        self.mark_parallel_comm(expert_dp_comm)

    def forward(self, inp, gate_top_k_idx, gate_score):
        original_shape = inp.shape
        moe_inp = inp.reshape(-1, self.d_model)
        return self.forward_moe(moe_inp, gate_top_k_idx, gate_score).reshape(original_shape)

    def forward_moe(self, moe_inp, gate_top_k_idx, gate_score):
        T = moe_inp.shape[0]
        idx = gate_top_k_idx.reshape(T, self.top_k)
        score = gate_score.reshape(T, self.top_k)
        e = self.experts
        if self.world_size == 1 and _is_plain_gelu(e.activation) and self.mask is None:
            out = GroupedFFNFn.apply(moe_inp, idx.to(torch.int32), score, e.htoh4.weight, e.htoh4.bias,
                                     e.h4toh.weight, e.h4toh.bias)
            return out.to(moe_inp.dtype)
        fwd = _fmoe_general_global_forward(moe_inp, idx, self.expert_fn, self.num_expert, self.world_size)
        return CombineFn.apply(fwd.view(-1, self.top_k, fwd.shape[-1]), score).to(moe_inp.dtype)
