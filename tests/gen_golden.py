"""Generate golden input/output vectors from the reference's importable pure-torch
twins of the hot path.  Run in the build container only (needs /root/reference):

    python tests/gen_golden.py

The reference modules used (never copied; only their OUTPUTS are stored):
  models/moe/gates.py            NoisyGate_VMoE       (gate arithmetic, a1 twin)
  models/moe/moe.py              compute_gating, MoE.forward composition (a5/a7 twin)
  models/moe/parallel_experts.py ParallelLinear       (a6 grouped GEMM fwd/dgrad/wgrad)

Fixtures written to tests/golden/*.npz (small, a few hundred KB total):
  g1_gate_e16 / g1_gate_e64  x, w_gate -> idx, score, probs           (std = 0, eval)
  g2_gate_taskcond           x, tsf, w_gate[D+gtsd,E] -> idx, score   (gate_inp = cat(x, tsf))
  g2b_gate_noise             x, w_gate, noise, std -> idx, score      (training, caller noise)
  g3_route                   idx -> expert_size, batch_index, index_sorted_experts
  g4_grouped_linear          ragged counts incl. an empty expert: fwd + dX/dW/db
  g5_moe_layer               whole layer out + dX/dW1/db1/dW2/db2/dw_gate
  g6_balance_noisy           x, w_gate, noise, std -> the gate's own balance loss in training mode
                             (cv^2(importance) + cv^2(_prob_in_top_k load)) and d loss / d(x, w_gate)
  g7_checkpoint_formats      pretrain/utils/moe_checkpoint.py (f.4): to_mtl_backbone_state_dict, get_first_expert_dim0,
                             infer_expert_format, build_mtl_meta over a table of cases, merge_moe_sharded_directory on a
                             4-rank shard directory written the way train_fastmoe's ranks write it
Inputs are resampled until the top-(k+2) probabilities of every token are
separated by > 1e-4 relative, so torch.topk's unspecified tie order cannot matter.
"""
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
REF = "/root/reference"
sys.path.insert(0, REF)
from models.moe.gates import NoisyGate_VMoE          # noqa: E402
from models.moe.moe import compute_gating            # noqa: E402
from models.moe.parallel_experts import ParallelLinear  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
os.makedirs(OUT, exist_ok=True)


def tie_free(probs: torch.Tensor, k: int, rel=1e-4) -> bool:
    top = probs.topk(min(k + 2, probs.shape[1]), dim=1).values
    gap = (top[:, :-1] - top[:, 1:]) / top[:, :-1]
    return bool((gap > rel).all())


def make_gate(Dg, E, k, seed, noise_std=0.0):
    torch.manual_seed(seed)
    g = NoisyGate_VMoE(Dg, num_expert=E, world_size=1, top_k=k, noise_std=noise_std)
    return g


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.asarray(v.detach() if torch.is_tensor(v) else v).shape) for k, v in arrs.items()})


def g1(E, name, T=64, D=384, k=4):
    for seed in range(100, 200):
        gate = make_gate(D, E, k, seed).eval()
        x = torch.randn(T, D)
        with torch.no_grad():
            idx, score = gate(x)
            probs = gate.get_activation()
        if tie_free(probs, k):
            save(name, x=x, w_gate=gate.w_gate, idx=idx, score=score, probs=probs, k=k)
            return
    raise RuntimeError("no tie-free seed")


def g2(T=48, D=384, gtsd=64, E=16, k=4):
    for seed in range(200, 300):
        gate = make_gate(D + gtsd, E, k, seed).eval()
        x = torch.randn(T, D)
        tsf = torch.randn(gtsd)
        gate_inp = torch.cat((x, tsf.repeat(T, 1)), dim=-1)   # custom_moe_layer.py:176-179
        with torch.no_grad():
            idx, score = gate(gate_inp)
            probs = gate.get_activation()
        if tie_free(probs, k):
            save("g2_gate_taskcond", x=x, tsf=tsf, w_gate=gate.w_gate, idx=idx, score=score, probs=probs, k=k)
            return
    raise RuntimeError("no tie-free seed")


def g2b(T=48, D=384, E=16, k=4, std=1.0):
    for seed in range(300, 400):
        gate = make_gate(D, E, k, seed, noise_std=std).train()
        x = torch.randn(T, D)
        # reproduce the gate's randn_like draw: same generator state before the call
        torch.manual_seed(seed + 1000)
        noise = torch.randn(T, E)
        torch.manual_seed(seed + 1000)
        with torch.no_grad():
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                idx, score = gate(x)
            probs = gate.get_activation()
            gate.get_loss()
        if tie_free(probs, k):
            save("g2b_gate_noise", x=x, w_gate=gate.w_gate, noise=noise, std=np.float32(std), idx=idx, score=score,
                 probs=probs, k=k)
            return
    raise RuntimeError("no tie-free seed")


def g6(T=96, D=64, E=8, k=2, std=1.0):
    """the reference gate's set_loss() value with noise (Normal-CDF load) and its gradients"""
    import contextlib, io
    for seed in range(500, 600):
        gate = make_gate(D, E, k, seed, noise_std=std).train()
        x = torch.randn(T, D, requires_grad=True)
        torch.manual_seed(seed + 1000)
        noise = torch.randn(T, E)
        torch.manual_seed(seed + 1000)
        with contextlib.redirect_stdout(io.StringIO()):
            idx, score = gate(x)
        probs = gate.get_activation()
        if not tie_free(probs.detach(), k):
            continue
        loss = gate.get_loss()
        dx, dw = torch.autograd.grad(loss, [x, gate.w_gate])
        save("g6_balance_noisy", x=x, w_gate=gate.w_gate, noise=noise, std=np.float32(std), k=k, idx=idx, score=score,
             loss=loss, dx=dx, dw_gate=dw)
        return
    raise RuntimeError("no tie-free seed")


def g3():
    d = np.load(os.path.join(OUT, "g1_gate_e16.npz"))
    idx = torch.tensor(d["idx"])
    score = torch.tensor(d["score"])
    k = int(d["k"])
    probs = torch.zeros(idx.shape[0], 16).scatter_(1, idx, score)
    batch_gates, batch_index, expert_size, gates, index_sorted = compute_gating(k, probs, score, idx)
    save("g3_route", idx=idx, k=k, E=16, expert_size=expert_size, batch_index=batch_index,
         index_sorted_experts=index_sorted, batch_gates=batch_gates)


def g4(E=4, Din=96, Dout=160):
    torch.manual_seed(7)
    counts = torch.tensor([37, 0, 70, 21])          # ragged, one empty expert
    R = int(counts.sum())
    x = torch.randn(R, Din, requires_grad=True)
    w = (torch.randn(E, Din, Dout) * 0.05).requires_grad_()    # ParallelLinear layout [E,in,out]
    b = (torch.randn(E, Dout) * 0.05).requires_grad_()
    y = ParallelLinear.apply(x, counts, w, b)
    gy = torch.randn_like(y)
    y.backward(gy)
    save("g4_grouped_linear", x=x, w_in_out=w, b=b, counts=counts, y=y, gy=gy, dx=x.grad, dw_in_out=w.grad, db=b.grad)


def g5(T=80, D=64, H=96, E=8, k=2):
    for seed in range(500, 600):
        gate = make_gate(D, E, k, seed).eval()
        torch.manual_seed(seed)
        x = torch.randn(T, D, requires_grad=True)
        w1 = (torch.randn(E, D, H) * 0.08).requires_grad_()
        b1 = (torch.randn(E, H) * 0.05).requires_grad_()
        w2 = (torch.randn(E, H, D) * 0.08).requires_grad_()
        b2 = (torch.randn(E, D) * 0.05).requires_grad_()
        idx, score = gate(x)                                   # MoE.forward, moe.py:121-155
        probs_full = gate.get_activation()
        if not tie_free(probs_full.detach(), k):
            continue
        probs = torch.zeros(T, E).scatter(1, idx, score)
        batch_gates, batch_index, expert_size, gates, iso = compute_gating(k, probs, score, idx)
        xin = x[batch_index]
        h = ParallelLinear.apply(xin, expert_size, w1, b1)
        h = torch.nn.functional.gelu(h)
        yo = ParallelLinear.apply(h, expert_size, w2, b2)
        yo = yo * batch_gates[:, None]
        out = yo.new_zeros(T, D).index_add(0, batch_index, yo)
        gout = torch.randn_like(out)
        out.backward(gout)
        save("g5_moe_layer", x=x, w_gate=gate.w_gate, w1_in_out=w1, b1=b1, w2_in_out=w2, b2=b2, k=k,
             idx=idx, score=score, out=out, gout=gout, dx=x.grad, dw1_in_out=w1.grad, db1=b1.grad,
             dw2_in_out=w2.grad, db2=b2.grad, dw_gate=gate.w_gate.grad)
        return
    raise RuntimeError("no tie-free seed")


def g7():
    """EP checkpoint formats: run the reference's pretrain/utils/moe_checkpoint.py on small synthetic state dicts; the
    case table (inputs) travels with the outputs so that the test replays exactly these calls."""
    import json
    import tempfile
    from collections import OrderedDict
    from pretrain.utils import moe_checkpoint as M
    rng = np.random.RandomState(77)
    E, W = 8, 4

    def t(*shape):
        return torch.from_numpy(rng.randn(*shape).astype(np.float32))
    wrapped = OrderedDict([
        ("module.encoder.pos_embed", t(1, 5, 4)),
        ("module.encoder.blocks.0.attn.qkv.weight", t(12, 4)),
        ("module.encoder.blocks.1.mlp.gate.w_gate", t(4, E)),
        ("module.encoder.blocks.1.mlp.experts.htoh4.weight", t(E, 6, 4)),
        ("module.encoder.blocks.1.mlp.experts.htoh4.bias", t(E, 6)),
        ("module.encoder.blocks.1.mlp.experts.h4toh.weight", t(E, 4, 6)),
        ("module.encoder.blocks.1.mlp.experts.h4toh.bias", t(E, 4)),
        ("module.norm.weight", t(4)),
        ("module.head.weight", t(5, 4)),
        ("module.head.bias", t(5)),
        ("blocks.0.norm1.weight", t(4)),                      # an already-backbone key passes through
    ])
    out, dropped = M.to_mtl_backbone_state_dict(wrapped)
    glob = out                                                # backbone key space, E experts along dim 0
    loc = OrderedDict((k, (v[2:4] if M.is_expert_key(k) else v)) for k, v in glob.items())
    dense = OrderedDict((k, v) for k, v in glob.items() if not M.is_expert_key(k))
    states = {"global": glob, "local": loc, "dense": dense}
    infer_cases = []
    for ck in ({}, {"meta": {"expert_format": "local"}}, {"meta": {"expert_format": "global"}}, {"meta": {"expert_format": "other"}},
               {"args": {"moe_experts": 8, "world_size": 4}}, {"args": {"moe_experts": 8}}, {"args": {"moe_experts": 16, "world_size": 2}},
               {"meta": "not a dict", "args": {"moe_experts": 2, "world_size": 4}}):
        for sname in ("global", "local", "dense"):
            for egx, ews in ((None, None), (8, None), (8, 4), (2, None), (16, 2), (None, 4), (8, 1)):
                got = M.infer_expert_format(ck, states[sname], expected_global_experts=egx, expected_world_size=ews)
                infer_cases.append(dict(checkpoint=ck, state=sname, expected_global_experts=egx, expected_world_size=ews, out=got))
    meta_cases = []
    for sname in ("global", "local", "dense"):
        for kw in (dict(world_size=1), dict(world_size=4), dict(world_size=3), dict(world_size=4, moe_experts_global=16),
                   dict(world_size=2, moe_experts_local=1), dict(world_size=4, moe_experts_global=32, moe_experts_local=8)):
            meta_cases.append(dict(state=sname, kwargs=kw, out=M.build_mtl_meta(states[sname], "gen_golden", **kw)))
    # a train_fastmoe-style shard directory (utils/moe_utils.py:164-175): rank 0 the whole state with ITS experts, the
    # other ranks their expert tensors only
    with tempfile.TemporaryDirectory() as d:
        for r in range(W):
            st = OrderedDict()
            for k, v in glob.items():
                if M.is_expert_key(k):
                    st[k] = v[r * (E // W):(r + 1) * (E // W)].clone()
                elif r == 0:
                    st[k] = v
            torch.save({"state_dict": st, "epoch": 7, "rank": r}, os.path.join(d, f"{r}.pth"))
        base, merged, n = M.merge_moe_sharded_directory(d)
    assert n == W and all(torch.equal(merged[k], glob[k]) for k in glob)
    arrs = {"in/" + k: v for k, v in wrapped.items()}
    arrs.update({"merged/" + k: v for k, v in merged.items()})
    save("g7_checkpoint_formats", table=np.array(json.dumps(dict(
        E=E, W=W, backbone_keys=list(out.keys()), dropped=dropped, first_dim0={k: M.get_first_expert_dim0(v) for k, v in states.items()},
        infer=infer_cases, meta=meta_cases, merged_keys=list(merged.keys()), n_shards=n, base_epoch=base["epoch"]))), **arrs)


if __name__ == "__main__":
    g1(16, "g1_gate_e16")
    g1(64, "g1_gate_e64")
    g2()
    g2b()
    g6()
    g3()
    g4()
    g5()
    g7()
