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
  g8_attention_*             the reference's OWN Attention modules (models/moe/ckpt/vision_transformer_moe.py:283-313 and its
                             dense twin models/backbones/vit.py:177-207): out, d x, d qkv / proj parameters; N = 197 with 2 heads
                             of 32 and N = 1025 with one head of 64
  g9_dense_block             the reference's dense Block, both classes (ckpt Block(moe=False) :438-487 - returns (x, None) - and
                             models/backbones/vit.py:216-246) on the same weights: out + every gradient
  g9b_dense_vit              the reference's whole dense backbone, models/backbones/vit.py VisionTransformer (BASELINE
                             configs[0]'s class; PatchEmbed, cls token, pos_embed, blocks): tokens + every gradient
  g10_balance_helpers        module-level _prob_in_top_k / cv_squared / _gates_to_load of the ckpt backbone
                             (vision_transformer_moe.py:23-87) with noise_stddev > 0, incl. their gradients
  g11_upcycling              utils/helpers.py:481-713 _inject_moe_expert_from_deit_mlp (split upcycling at ratio 1 with and without
                             GELU weight scaling, replicate at ratio 4, truncate mode, deit_warm_start) and :714-867
                             _auto_virtual_group_size / _inject_virtual_group_init_for_gates on tiny synthetic dense state dicts
The backbone files import cv2 / timm / tree (absent here) and fmoe (this repository's shim): those imports are satisfied by
IMPORT-LINE-ONLY placeholder modules (reference_backbone_modules below) whose every attribute is a tripwire that raises when
used; the generators assert that no placeholder attribute is read while the reference code runs.
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


# ------------------------------------------------------------------ the reference's backbone files, g8 - g11
class _Tripwire:
    """what a placeholder module hands out for any attribute: importing the name works, USING it raises"""

    def __init__(self, name):
        object.__setattr__(self, "_name", name)

    def _boom(self, *a, **k):
        raise RuntimeError(f"placeholder attribute {self._name} was used: the fixture would not be the reference's output")
    __call__ = __getattr__ = __getitem__ = __iter__ = __mul__ = __add__ = __bool__ = _boom


class _Placeholder(__import__("types").ModuleType):
    """import-line-only stand-in for a module this image lacks; every attribute read is logged"""
    reads = []

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        _Placeholder.reads.append((self.__name__, k))
        return _Tripwire(self.__name__ + "." + k)


def reference_backbone_modules():
    """(models.backbones.vit, models.moe.ckpt.vision_transformer_moe, utils.helpers) of the reference, imported with cv2,
    timm, timm.layers and tree as placeholders and fmoe = this repository's shim (classes only: nothing of it runs on CPU)."""
    import importlib.machinery
    for n in ("cv2", "timm", "timm.layers", "tree"):
        if n not in sys.modules:
            m = _Placeholder(n)
            m.__path__ = []
            m.__spec__ = importlib.machinery.ModuleSpec(n, None)
            sys.modules[n] = m
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import m3vit_amd
    m3vit_amd.install_fmoe_shim()
    import models.backbones.vit as RV
    import models.moe.ckpt.vision_transformer_moe as RM
    import utils.helpers as RH
    # the only reads so far are the `from timm.layers import lecun_normal_` lines themselves
    assert set(_Placeholder.reads) <= {("timm.layers", "lecun_normal_")}, _Placeholder.reads
    _Placeholder.reads.clear()
    return RV, RM, RH


def _no_placeholder_was_used():
    assert not _Placeholder.reads, f"a placeholder module was read while reference code ran: {_Placeholder.reads}"


def _grads(mod, prefix=""):
    return {prefix + "d_" + n: p.grad for n, p in mod.named_parameters()}


def g8(RV, RM):
    for name, B, N, D, h, seed in (("g8_attention_n197_dh32", 2, 197, 64, 2, 801), ("g8_attention_n1025_dh64", 1, 1025, 64, 1, 802)):
        torch.manual_seed(seed)
        att = RM.Attention(D, num_heads=h, qkv_bias=True)
        for p in att.parameters():
            torch.nn.init.normal_(p, std=0.2)
        twin = RV.Attention(D, num_heads=h, qkv_bias=True)
        twin.load_state_dict(att.state_dict())
        x = torch.randn(B, N, D, requires_grad=True)
        gout = torch.randn(B, N, D)
        out = att(x)
        out.backward(gout)
        x2 = x.detach().clone().requires_grad_()
        out2 = twin(x2)
        out2.backward(gout)
        assert torch.equal(out, out2) and torch.equal(x.grad, x2.grad)       # the two reference classes are the same arithmetic
        save(name, x=x, gout=gout, out=out, dx=x.grad, heads=h, **{"p_" + n: v for n, v in att.state_dict().items()}, **_grads(att))
    _no_placeholder_was_used()


def g9(RV, RM):
    torch.manual_seed(901)
    B, N, D, h = 2, 50, 64, 2
    norm = __import__("functools").partial(torch.nn.LayerNorm, eps=1e-6)             # vision_transformer_moe.py:567
    blk = RM.Block(D, h, mlp_ratio=4., qkv_bias=True, norm_layer=norm, moe=False, use_checkpointing=False)
    for p in blk.parameters():
        torch.nn.init.normal_(p, std=0.15)
    twin = RV.Block(D, h, mlp_ratio=4., qkv_bias=True, norm_layer=norm)
    twin.load_state_dict(blk.state_dict())
    x = torch.randn(B, N, D, requires_grad=True)
    gout = torch.randn(B, N, D)
    out, cv = blk(x)                                                                   # Block.forward :489-562 returns (x, cv_loss)
    assert cv is None
    out.backward(gout)
    x2 = x.detach().clone().requires_grad_()
    out2 = twin(x2)
    out2.backward(gout)
    assert torch.allclose(out, out2, atol=1e-6) and torch.allclose(x.grad, x2.grad, atol=1e-6)
    ck = RM.Block(D, h, mlp_ratio=4., qkv_bias=True, norm_layer=norm, moe=False, use_checkpointing=True)    # :495-524
    ck.load_state_dict(blk.state_dict())
    x3 = x.detach().clone().requires_grad_()
    out3, _ = ck(x3)
    out3.backward(gout)
    assert torch.allclose(out, out3, atol=1e-6) and torch.allclose(x.grad, x3.grad, atol=1e-6)
    save("g9_dense_block", x=x, gout=gout, out=out, dx=x.grad, heads=h, **{"p_" + n: v for n, v in blk.state_dict().items()},
         **_grads(blk))
    # the whole dense backbone (BASELINE configs[0]'s class), tiny: 32 x 48 image, 2 x 3 patches + cls
    torch.manual_seed(902)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        vit = RV.VisionTransformer(model_name="vit_tiny_patch16_224", img_size=(32, 48), patch_size=16, embed_dim=64, depth=3,
                                   num_heads=2, num_classes=5, drop_rate=0., random_init=True)
    for n, p in vit.named_parameters():
        if p.dim() == 1:
            torch.nn.init.normal_(p, mean=1.0 if n.endswith("weight") else 0.0, std=0.1)
    img = torch.randn(2, 3, 32, 48, requires_grad=True)
    tok = vit(img)
    gtok = torch.randn_like(tok)
    tok.backward(gtok)
    save("g9b_dense_vit", images=img, gtok=gtok, tokens=tok, dimages=img.grad, heads=2, depth=3,
         **{"p_" + n: v for n, v in vit.state_dict().items()}, **_grads(vit))
    _no_placeholder_was_used()


def g10(RM):
    torch.manual_seed(1001)
    T, E, k, std = 96, 8, 2, 0.7
    clean = torch.randn(T, E, requires_grad=True)
    noise = torch.randn(T, E)
    noisy = clean + noise * std
    probs = noisy.softmax(dim=1)
    top = probs.topk(k + 1, dim=1).values                           # noisy_gate_vmoe.py:197-200: top_logits are post-softmax
    load_rows = RM._prob_in_top_k(clean, noisy, std, top, k)       # vision_transformer_moe.py:33-71
    load = load_rows.sum(0)
    gates = torch.zeros(T, E).scatter(1, probs.topk(k, dim=1).indices, probs.topk(k, dim=1).values)
    importance = gates.sum(0)
    loss = RM.cv_squared(importance) + RM.cv_squared(load)          # :73-87, :540
    loss.backward()
    save("g10_balance_helpers", clean=clean, noise=noise, std=std, k=k, top=top, prob_rows=load_rows, load=load,
         count_load=RM._gates_to_load(gates), gates=gates, importance=importance, cv_importance=RM.cv_squared(importance),
         cv_load=RM.cv_squared(load), loss=loss, dclean=clean.grad,
         cv_single=RM.cv_squared(torch.tensor([3.0])))               # E == 1: Tensor([0]) (:83-84)
    _no_placeholder_was_used()


def g11(RH):
    """DeiT dense MLP -> experts upcycling and the virtual-group gate init, run on stand-in models that carry exactly the
    attributes the reference functions read (utils/helpers.py:498-580, :756-867)."""
    import contextlib
    import io
    import json
    from collections import OrderedDict
    from types import SimpleNamespace as NS
    D, Hd = 8, 32
    rng = np.random.RandomState(1101)

    def t(*shape):
        return torch.from_numpy(rng.randn(*shape).astype(np.float32))

    def dense_state(depth):
        sd = OrderedDict()
        for i in range(depth):
            sd[f"blocks.{i}.mlp.fc1.weight"] = t(Hd, D); sd[f"blocks.{i}.mlp.fc1.bias"] = t(Hd)
            sd[f"blocks.{i}.mlp.fc2.weight"] = t(D, Hd); sd[f"blocks.{i}.mlp.fc2.bias"] = t(D)
        sd["pos_embed"] = t(1, 3, D)
        return sd

    class Model:
        def __init__(self, depth, moe_blocks, e_local, world, expert_hidden, ratio, top_k, multi_gate=0):
            self.moe_mlp_ratio, self.mlp_ratio, self.moe_top_k, self.moe_experts = ratio, 4.0, top_k, e_local
            self.blocks, self._sd = [], OrderedDict()
            for i in range(depth):
                moe = i in moe_blocks
                b = NS(moe=moe, mlp=NS(num_expert=e_local, world_size=world), moe_top_k=top_k, world_size=world,
                       tot_expert=e_local * world)
                self.blocks.append(b if moe else NS(moe=False, mlp=NS()))
                if moe:
                    self._sd[f"blocks.{i}.mlp.experts.htoh4.weight"] = torch.zeros(e_local, expert_hidden, D)
                    self._sd[f"blocks.{i}.mlp.experts.htoh4.bias"] = torch.zeros(e_local, expert_hidden)
                    self._sd[f"blocks.{i}.mlp.experts.h4toh.weight"] = torch.zeros(e_local, D, expert_hidden)
                    self._sd[f"blocks.{i}.mlp.experts.h4toh.bias"] = torch.zeros(e_local, D)
                    for g in (range(multi_gate) if multi_gate else [None]):
                        self._sd[f"blocks.{i}.mlp.gate." + (f"{g}.w_gate" if g is not None else "w_gate")] = torch.zeros(D, e_local * world)

        def state_dict(self):
            return self._sd

    cases = [
        dict(name="split_ratio1", depth=4, moe=[1, 3], e_local=8, world=1, eh=8, ratio=1.0, top_k=4, cfg={}, mode="deit_upcycling"),
        dict(name="split_scaled_ep2", depth=2, moe=[1], e_local=8, world=2, eh=8, ratio=1.0, top_k=4,
             cfg={"use_weight_scaling": True}, mode="deit_upcycling"),
        dict(name="replicate_ratio4", depth=2, moe=[1], e_local=4, world=1, eh=32, ratio=4.0, top_k=2, cfg={}, mode="deit_upcycling"),
        dict(name="truncate", depth=2, moe=[1], e_local=2, world=2, eh=8, ratio=1.0, top_k=2, cfg={}, mode="deit_upcycling"),
        dict(name="warm_start_forced_split", depth=2, moe=[0, 1], e_local=4, world=1, eh=8, ratio=4.0, top_k=1,
             cfg={"use_weight_scaling": True}, mode="deit_warm_start"),
        dict(name="granularity2", depth=2, moe=[1], e_local=4, world=1, eh=16, ratio=1.0, top_k=2, cfg={"use_weight_scaling": True},
             mode="deit_upcycling"),
    ]
    arrs, table = {}, []
    for c in cases:
        sd = dense_state(c["depth"])
        m = Model(c["depth"], c["moe"], c["e_local"], c["world"], c["eh"], c["ratio"], c["top_k"])
        for k_, v in sd.items():
            arrs[f"{c['name']}/in/{k_}"] = v.clone()
        with contextlib.redirect_stdout(io.StringIO()):
            out = RH._inject_moe_expert_from_deit_mlp(sd, m, c["cfg"], deit_init_mode=c["mode"])
        for k_, v in out.items():
            if "experts" in k_:
                arrs[f"{c['name']}/out/{k_}"] = v
        table.append({k_: v for k_, v in c.items()})
    # virtual-group size: a table of argument combinations
    sizes = []
    for tot in (0, 1, 8, 16, 64, 12):
        for loc in (None, 2, 4, 8):
            for ws in (None, 1, 2, 3):
                for dh, eh in ((None, None), (32, 8), (32, 32), (32, 12), (1536, 384)):
                    sizes.append(dict(tot=tot, local_experts=loc, world_size=ws, dense_hidden=dh, expert_hidden=eh,
                                      out=RH._auto_virtual_group_size(tot, local_experts=loc, world_size=ws, dense_hidden=dh,
                                                                      expert_hidden=eh)))
    # gate init: same RNG stream (torch.manual_seed) -> the w_gate tensors themselves are the fixture
    vg = []
    for name, kw, mg in (("vg_single", dict(depth=3, moe_blocks=[1, 2], e_local=8, world=1, expert_hidden=8, ratio=1.0, top_k=4), 0),
                         ("vg_multi_ep2", dict(depth=2, moe_blocks=[1], e_local=4, world=2, expert_hidden=16, ratio=1.0, top_k=2), 2)):
        sd = dense_state(kw["depth"])
        m = Model(multi_gate=mg, **kw)
        torch.manual_seed(1102)
        with contextlib.redirect_stdout(io.StringIO()):
            out = RH._inject_virtual_group_init_for_gates(sd, m, cfg={}, init="normal", std=0.02)
        for k_, v in out.items():
            if k_.endswith("w_gate"):
                arrs[f"{name}/out/{k_}"] = v
        vg.append(dict(name=name, seed=1102, multi_gate=mg, D=D, Hd=Hd, **kw))
    save("g11_upcycling", table=np.array(json.dumps(dict(D=D, Hd=Hd, cases=table, group_sizes=sizes, gate_init=vg))), **arrs)
    _no_placeholder_was_used()


if __name__ == "__main__":
    if "--backbone-only" not in sys.argv:
        _main_moe = True
    else:
        _main_moe = False
    RV, RM, RH = reference_backbone_modules()
    g8(RV, RM)
    g9(RV, RM)
    g10(RM)
    g11(RH)
    if not _main_moe:
        sys.exit(0)
    g1(16, "g1_gate_e16")
    g1(64, "g1_gate_e64")
    g2()
    g2b()
    g6()
    g3()
    g4()
    g5()
    g7()
