"""Balance-loss arithmetic on the gate kernel's outputs (models/moe/ckpt/vision_transformer_moe.py:23-87; the origin
variant has the same functions as gate methods: models/moe/origin/noisy_gate_vmoe.py:70-130).  [T, E] / [E] elementwise
torch ops for the module API; the fused executor (m3vit_amd.engine) has them inside the gate kernels."""
import torch


def gates_to_load(gates):
    return (gates > 0).sum(0)


def prob_in_top_k(clean_values, noisy_values, noise_stddev, noisy_top_values, k):
    """Probability that each expert stays in the top k under fresh noise (vision_transformer_moe.py:33-71; the
    thresholds are the (k+1)-th / k-th entries of top_logits, i.e. probabilities, compared with logits, exactly as the
    reference does)."""
    thr_in = noisy_top_values[:, k:k + 1]
    thr_out = noisy_top_values[:, k - 1:k]
    is_in = noisy_values > thr_in
    normal = torch.distributions.normal.Normal(torch.zeros((), device=clean_values.device),
                                               torch.ones((), device=clean_values.device))
    prob_if_in = normal.cdf((clean_values - thr_in) / noise_stddev)
    prob_if_out = normal.cdf((clean_values - thr_out) / noise_stddev)
    return torch.where(is_in, prob_if_in, prob_if_out)


def cv_squared(x):
    eps = 1e-10
    if x.shape[0] == 1:
        return torch.Tensor([0])
    return x.float().var() / (x.float().mean() ** 2 + eps)


def block_balance_loss(gates, clean, noisy, std, top_logits, top_k):
    """cv^2(importance) + cv^2(load) of one MoE block (:453-459,540; origin gate :277-288): load is the Normal-CDF
    form when noise is on and k < E, else the count."""
    importance = gates.sum(0)
    if top_k < gates.shape[1] and abs(std) > 1e-6:
        load = prob_in_top_k(clean, noisy, std, top_logits, top_k).sum(0)
    else:
        load = gates_to_load(gates)
    return cv_squared(importance) + cv_squared(load)
