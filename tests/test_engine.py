"""GPU: the fused backbone executor (m3vit_amd.engine) against the oracle's
backbone_forward + torch autograd on the same seeded inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().double().cpu().flatten(); b = b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float16, 4e-3)])
def test_backbone_fwd_bwd_matches_oracle(dtype, tol):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from m3vit_amd.engine import BackboneEngine
    from oracle import ref_torch as R
    cfg = R.BackboneCfg(img_size=(32, 48), embed_dim=64, depth=4, num_heads=2, mlp_ratio=4.0, moe_mlp_ratio=1.0,
                        moe_experts=4, moe_top_k=2, gate_dim=66, multi_gate=True)
    B = 3
    P = R.init_backbone_params(cfg, seed=5)
    torch.manual_seed(0)
    img = torch.randn(B, 3, 32, 48)
    dtok = torch.randn(B, cfg.num_tokens, 64) * 0.1
    cvw = 0.01
    eng = BackboneEngine(cfg, P, batch=B, dtype=dtype)
    eng.zero_grad()
    Pr = {k: v.clone().double().requires_grad_() for k, v in P.items()}
    tot = 0.0
    for task in (0, 1):
        tok, cv = eng.forward(img.cuda(), task)
        tok_ref, cv_ref, aux = R.backbone_forward(Pr, cfg, img.double(), task)
        # identical routing in every MoE block, then values
        for i in (1, 3):
            assert torch.equal(eng.act[i]["gate"]["idx"].cpu(), aux[i]["idx"]), f"routing differs in block {i}"
        assert rel(tok, tok_ref) < tol
        assert abs(float(cv) - float(cv_ref)) < 1e-3 * max(1.0, abs(float(cv_ref)))
        eng.backward(dtok.cuda(), cv_weight=cvw)
        tot = tot + (tok_ref * dtok.double()).sum() + cvw * cv_ref
    tot.backward()
    bad = []
    for name, g in eng.grads.items():
        ref = Pr[name].grad
        if ref is None:
            assert float(g.abs().max()) == 0.0, name
            continue
        e = rel(g, ref)
        if e > tol * 5:
            bad.append((name, e))
    assert not bad, bad


def test_config0_dense_vit_tiny_plumbing():
    """BASELINE configs[0]: dense ViT-Tiny/16 (no MoE), 8 images of 480x640 -> tokens [8,1201,192]
    (configs/nyud/vit/pup_vit_tiny_multi_task_baseline.yml; the reference path is train_vit.py ->
    models/backbones/vit.py).  Forward + backward of the engine's dense path against the oracle."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from m3vit_amd.engine import BackboneEngine
    from oracle import ref_torch as R
    cfg = R.BackboneCfg(img_size=(480, 640), embed_dim=192, depth=2, num_heads=3, mlp_ratio=4.0, dense_only=True,
                        gate_dim=192)
    B = 2
    P = R.init_backbone_params(cfg, seed=11)
    torch.manual_seed(1)
    img = torch.randn(B, 3, 480, 640)
    dtok = torch.randn(B, cfg.num_tokens, 192) * 0.1
    assert cfg.num_tokens == 1201
    eng = BackboneEngine(cfg, P, batch=B, dtype=torch.float32)
    tok, cv = eng.forward(img.cuda(), None)
    Pr = {k: v.clone().requires_grad_() for k, v in P.items()}
    tr, _, _ = R.backbone_forward(Pr, cfg, img, None)
    assert tok.shape == (B, 1201, 192) and rel(tok, tr) < 2e-4
    eng.backward(dtok.cuda())
    (tr * dtok).sum().backward()
    bad = [(n, rel(g, Pr[n].grad)) for n, g in eng.grads.items() if rel(g, Pr[n].grad) > 2e-3]
    assert not bad, bad
