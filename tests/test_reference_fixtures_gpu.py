"""GPU: the HIP path run DIRECTLY on the fixtures the reference's own code produced (tests/gen_golden.py) - no oracle in
between:
  g3  models/moe/moe.py compute_gating                               -> m3_route_build
  g5  whole MoE layer, reference composition gates.py + compute_gating + ParallelLinear (out + every gradient)
                                                                      -> m3vit_amd.moe_layer.FMoETransformerMLP
  g8  the reference's Attention classes (ckpt vision_transformer_moe.py:283-313, backbones/vit.py:177-207)
                                                                      -> m3vit_amd.vit.Attention (qkv / proj GEMMs + attention kernels)
  g9  the reference's dense Block classes                            -> m3vit_amd.vit.Block(moe=False)
  g9b the reference's dense backbone models/backbones/vit.py VisionTransformer (BASELINE configs[0]'s class)
                                                                      -> BackboneEngine(dense_only), tokens + every gradient
fp32 bound 2e-5 relative L2 (tokens) / 1e-4 (gradients); fp16 activations 1e-3 / 3e-3 (the kernels' documented bounds)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.float16]
TOK = {torch.float32: 2e-5, torch.float16: 1e-3}
GRAD = {torch.float32: 1e-4, torch.float16: 3e-3}


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu().flatten(); b = torch.as_tensor(b).detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _load_params(mod, g, prefix="p_"):
    sd = {k[len(prefix):]: torch.tensor(v) for k, v in g.items() if k.startswith(prefix)}
    missing, unexpected = mod.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def _check_grads(mod, g, tol, prefix="d_"):
    bad = [(n, rel(p.grad, g[prefix + n])) for n, p in mod.named_parameters() if rel(p.grad, g[prefix + n]) > tol]
    assert not bad, bad


def test_route_build_on_the_reference_routing_fixture(golden_dir):
    _need_gpu()
    from m3vit_amd import ops
    g = load(golden_dir, "g3_route")
    k, E = int(g["k"]), int(g["E"])
    r = ops.route_build(torch.tensor(g["idx"]).to(torch.int32).cuda(), E)
    assert np.array_equal(r.counts.cpu().numpy(), g["expert_size"])
    off, ros = r.offsets.cpu().tolist(), r.row_of_slot.cpu()
    for e in range(E):        # inside an expert the reference's (unstable) sort may order the slots differently: compare sets
        a, b = off[e], off[e + 1]
        assert set(ros[a:b].tolist()) == set(g["index_sorted_experts"][a:b].tolist())
        assert sorted((ros[a:b] // k).tolist()) == sorted(g["batch_index"][a:b].tolist())
    assert torch.equal(ros[r.pos.cpu().long()], torch.arange(ros.numel(), dtype=ros.dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_moe_layer_mirror_on_the_reference_layer_fixture(golden_dir, dtype):
    """g5: out, d x, d W1 / b1 / W2 / b2, d w_gate of the reference composition; FMoELinear's layout [E, out, in] is the
    transpose of ParallelLinear's [E, in, out]"""
    _need_gpu()
    from m3vit_amd.gate import NoisyGate_VMoE
    from m3vit_amd.moe_layer import FMoETransformerMLP
    from m3vit_amd.vit import HipLayerNorm                                    # noqa: F401  (act dtype is the input's)
    g = load(golden_dir, "g5_moe_layer")
    k = int(g["k"])
    T, D = g["x"].shape
    E, _, H = g["w1_in_out"].shape
    layer = FMoETransformerMLP(num_expert=E, d_model=D, d_gate=D, d_hidden=H, gate=NoisyGate_VMoE, top_k=k, vmoe_noisy_std=0,
                               activation=torch.nn.Sequential(torch.nn.GELU(), torch.nn.Dropout(0.))).cuda().eval()
    with torch.no_grad():
        layer.gate.w_gate.copy_(torch.tensor(g["w_gate"]))
        layer.experts.htoh4.weight.copy_(torch.tensor(g["w1_in_out"]).transpose(1, 2))
        layer.experts.htoh4.bias.copy_(torch.tensor(g["b1"]))
        layer.experts.h4toh.weight.copy_(torch.tensor(g["w2_in_out"]).transpose(1, 2))
        layer.experts.h4toh.bias.copy_(torch.tensor(g["b2"]))
    x = torch.tensor(g["x"]).to(dtype).cuda().requires_grad_()
    out, clean, noisy, std, top_logits, gates = layer(x.view(1, T, D))
    if dtype == torch.float32:
        assert np.array_equal(gates.gt(0).sum(1).cpu().numpy(), np.full(T, k))
        idx = gates.topk(k, dim=1).indices.sort(dim=1).values.cpu().numpy()
        assert np.array_equal(idx, np.sort(g["idx"], axis=1))                  # bit-exact expert selection
    assert rel(out.view(T, D), g["out"]) < TOK[dtype] * 2
    out.view(T, D).backward(torch.tensor(g["gout"]).to(out.dtype).cuda())
    tol = GRAD[dtype]
    assert rel(x.grad, g["dx"]) < tol
    assert rel(layer.experts.htoh4.weight.grad.transpose(1, 2), g["dw1_in_out"]) < tol
    assert rel(layer.experts.htoh4.bias.grad, g["db1"]) < tol
    assert rel(layer.experts.h4toh.weight.grad.transpose(1, 2), g["dw2_in_out"]) < tol
    assert rel(layer.experts.h4toh.bias.grad, g["db2"]) < tol
    assert rel(layer.gate.w_gate.grad, g["dw_gate"]) < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", ["g8_attention_n197_dh32", "g8_attention_n1025_dh64"])
def test_attention_mirror_on_the_reference_attention_fixture(golden_dir, name, dtype):
    _need_gpu()
    from m3vit_amd.vit import Attention
    g = load(golden_dir, name)
    B, N, D = g["x"].shape
    att = Attention(D, num_heads=int(g["heads"]), qkv_bias=True).cuda()
    _load_params(att, g)
    x = torch.tensor(g["x"]).to(dtype).cuda().requires_grad_()
    out = att(x)
    assert rel(out, g["out"]) < TOK[dtype] * (1 if dtype == torch.float32 else 2)
    out.backward(torch.tensor(g["gout"]).to(out.dtype).cuda())
    assert rel(x.grad, g["dx"]) < GRAD[dtype]
    _check_grads(att, g, GRAD[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
def test_dense_block_mirror_on_the_reference_block_fixture(golden_dir, dtype):
    _need_gpu()
    from m3vit_amd.vit import Block, _make_norm
    g = load(golden_dir, "g9_dense_block")
    B, N, D = g["x"].shape
    blk = Block(D, int(g["heads"]), mlp_ratio=4., qkv_bias=True, norm_layer=lambda d: _make_norm(d, dtype), moe=False).cuda()
    _load_params(blk, g)
    x = torch.tensor(g["x"]).cuda().requires_grad_()                          # the residual stream stays fp32
    out, cv = blk(x)
    assert cv is None and rel(out, g["out"]) < TOK[dtype]
    out.backward(torch.tensor(g["gout"]).cuda())
    assert rel(x.grad, g["dx"]) < GRAD[dtype]
    _check_grads(blk, g, GRAD[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
def test_engine_on_the_reference_dense_backbone_fixture(golden_dir, dtype):
    """the executor the benchmark drives, dense_only (BASELINE configs[0]'s backbone class), on the reference
    VisionTransformer's own weights, images and upstream gradient"""
    _need_gpu()
    from m3vit_amd.config import BackboneConfig
    from m3vit_amd.engine import BackboneEngine
    g = load(golden_dir, "g9b_dense_vit")
    P = {k[2:]: torch.tensor(v) for k, v in g.items() if k.startswith("p_")}
    B, _, Hh, Ww = g["images"].shape
    cfg = BackboneConfig(img_size=(Hh, Ww), embed_dim=64, depth=int(g["depth"]), num_heads=int(g["heads"]), mlp_ratio=4.0,
                         dense_only=True, gate_dim=64, multi_gate=False)
    eng = BackboneEngine(cfg, P, batch=B, dtype=dtype)
    eng.zero_grad()
    tok, cv = eng.forward(torch.tensor(g["images"]).cuda(), None)
    assert rel(tok, g["tokens"]) < TOK[dtype]
    eng.backward(torch.tensor(g["gtok"]).cuda())
    bad = [(n, rel(gr, g["d_" + n])) for n, gr in eng.grads.items() if rel(gr, g["d_" + n]) > GRAD[dtype]]
    assert not bad, bad
    assert set(eng.grads) == {k[2:] for k in g if k.startswith("d_") and k != "dimages" and k != "dx"}
