"""The hooks of the reference's utils/moe_utils.py that a trainer calls around the hot path, on the HIP modules
(north_star: "keeps the models/moe FMoE layer API and utils/moe_utils hooks").  Checkpoint formats live in
m3vit_amd/checkpoint.py (filter_state / save_moe_model_to_dir / read_specific_group_experts: utils/moe_utils.py:128-198)."""
import torch

from .checkpoint import expert_only as filter_state, shard_experts  # noqa: F401
from .gate import NoisyGate_VMoE
from .moe_layer import FMoETransformerMLP


def collect_noisy_gating_loss(model, weight):
    """utils/moe_utils.py:201-207: sum of the balance losses the (origin-convention) gates stored during the forward,
    times `weight`; get_loss() clears them."""
    loss = 0
    for module in model.modules():
        if isinstance(module, NoisyGate_VMoE) and module.has_loss:
            loss += module.get_loss()
    return loss * weight


def collect_moe_activation(model, batch_size, activation_suppress="pool", return_name=False):
    """utils/moe_utils.py:226-248: the gates' stored activations [B, N, E] (softmax probabilities), pooled over the
    tokens ("pool") or flattened per image ("concat")."""
    acts, names = [], []
    for name, module in model.named_modules():
        if isinstance(module, NoisyGate_VMoE) and module.has_activation:
            a = module.get_activation()
            a = a.reshape(batch_size, -1, a.shape[-1])
            if activation_suppress == "pool":
                a = a.mean(1)
            elif activation_suppress == "concat":
                a = a.reshape(batch_size, -1)
            else:
                raise ValueError("No activation_suppress of {}".format(activation_suppress))
            acts.append(a)
            names.append(name)
    return (acts, names) if return_name else acts


def set_moe_layer_train_mode(model):
    """utils/moe_utils.py:303-306"""
    for module in model.modules():
        if isinstance(module, FMoETransformerMLP):
            module.train()


def read_specific_group_experts(moe_state_dict, rank, num_experts):
    """utils/moe_utils.py:191-198: keep experts [rank * num_experts, (rank + 1) * num_experts) of a global state"""
    return shard_experts(moe_state_dict, rank, num_experts)


def sync_weights(model, except_key_words):
    """utils/moe_utils.py:310-324: broadcast every state_dict entry whose key contains none of `except_key_words`
    (the expert tensors, which differ per rank under expert parallelism) from rank 0, then reload."""
    state_dict = model.state_dict()
    for key, item in state_dict.items():
        if not any(w in key for w in except_key_words):
            torch.distributed.broadcast(item, 0)
    model.load_state_dict(state_dict)
