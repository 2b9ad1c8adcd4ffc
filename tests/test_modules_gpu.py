"""GPU: the reference-shaped module API (m3vit_amd.vit / moe_layer / fmoe shim) against the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().double().cpu().flatten(); b = b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("fused", [False, "auto"])
@pytest.mark.parametrize("std", [0.0, 1.0])
def test_vit_mirror_matches_oracle_fwd_bwd(std, fused):
    """fused = False: every op its own autograd Function (m3vit_amd/functional.py); "auto": the whole backbone call as one
    autograd node on the fused executor (m3vit_amd/fused.py; more in tests/test_fused_module_gpu.py).
    std = 1: the module draws the gate noise itself (torch.randn on the GPU, noisy_gate_vmoe.py:168);
    the test re-draws the same tensors from the same seed for the oracle.  The load term is then the
    Normal-CDF form computed from the gate outputs (vision_transformer_moe.py:456-457)."""
    _need_gpu()
    from m3vit_amd.vit import VisionTransformerMoE
    from oracle import ref_torch as R
    kw = dict(img_size=(32, 48), embed_dim=64, depth=4, num_heads=2, moe_experts=8 if std else 4, moe_top_k=2,
              gate_dim=66, multi_gate=True)
    cfg = R.BackboneCfg(mlp_ratio=4.0, moe_mlp_ratio=1.0, vmoe_noisy_std=std, **kw)
    P = R.init_backbone_params(cfg, seed=9)
    m = VisionTransformerMoE(mlp_ratio=4.0, moe_mlp_ratio=1, vmoe_noisy_std=std, fused=fused, **kw).cuda()
    m.load_state_dict(P)                                        # reference key names and shapes
    m.train()
    img = torch.randn(3, 3, 32, 48)
    dtok = torch.randn(3, cfg.num_tokens, 64) * 0.1
    Pr = {k: v.clone().double().requires_grad_() for k, v in P.items()}
    loss_ref = 0.0
    for task in (0, 1):
        torch.manual_seed(50 + task)
        tok, cv = m(img.cuda(), task_id=task)
        assert (m._fused is not None) == (fused == "auto")
        noises = None
        if std:
            torch.manual_seed(50 + task)
            noises = {i: torch.randn(3 * cfg.num_tokens, cfg.moe_experts, device="cuda").double().cpu() for i in (1, 3)}
        tr, cr, _ = R.backbone_forward(Pr, cfg, img.double(), task, noises=noises)
        assert rel(tok, tr) < 2e-4
        assert abs(float(cv) - float(cr)) < 1e-3 * max(1.0, float(cr))
        ((tok * dtok.cuda()).sum() + 0.01 * cv).backward()
        loss_ref = loss_ref + (tr * dtok.double()).sum() + 0.01 * cr
    loss_ref.backward()
    bad = [(n, rel(p.grad, Pr[n].grad)) for n, p in m.named_parameters() if Pr[n].grad is not None and rel(p.grad, Pr[n].grad) > 1e-3]
    assert not bad, bad


def test_vit_mirror_drop_path_matches_oracle():
    """Stochastic depth (drop_path_rate > 0, pretrain/configs/deit_moe_small.yaml:51): every block draws two per-sample
    masks in training (attention branch, MLP / MoE branch), rate rising linearly over the blocks; the factors the
    modules drew are fed to the oracle.  Eval mode is the identity."""
    _need_gpu()
    from m3vit_amd.vit import DropPath, VisionTransformerMoE
    from oracle import ref_torch as R
    kw = dict(img_size=(32, 32), embed_dim=64, depth=4, num_heads=2, moe_experts=4, moe_top_k=2, gate_dim=66,
              multi_gate=True)
    cfg = R.BackboneCfg(mlp_ratio=4.0, moe_mlp_ratio=1.0, vmoe_noisy_std=0.0, **kw)
    P = R.init_backbone_params(cfg, seed=12)
    m = VisionTransformerMoE(mlp_ratio=4.0, moe_mlp_ratio=1, vmoe_noisy_std=0.0, drop_path_rate=0.5, fused=False, **kw).cuda()
    m.load_state_dict(P)                 # (per-op path: the test reads the DropPath modules' draws; fused: tests/test_fused_module_gpu.py)
    rates = [blk.drop_path.drop_prob if isinstance(blk.drop_path, DropPath) else 0.0 for blk in m.blocks]
    assert rates == pytest.approx([0.0, 0.5 / 3, 1.0 / 3, 0.5])
    drawn = {}
    for i, blk in enumerate(m.blocks):
        if isinstance(blk.drop_path, DropPath):
            blk.drop_path.register_forward_hook(lambda mod, a, out, i=i: drawn.setdefault(i, []).append(mod.last_scale.cpu()))
    img = torch.randn(6, 3, 32, 32)
    dtok = torch.randn(6, cfg.num_tokens, 64) * 0.1
    m.train()
    torch.manual_seed(3)
    tok, cv = m(img.cuda(), task_id=1)
    assert all(len(v) == 2 for v in drawn.values()) and sorted(drawn) == [1, 2, 3]
    allv = torch.cat([torch.cat(v) for v in drawn.values()])
    assert bool((allv == 0).any()) and bool((allv > 1).any())           # some branches dropped, the kept ones scaled up
    scales = {i: (torch.ones(6), torch.ones(6)) if i not in drawn else tuple(drawn[i]) for i in range(4)}
    Pr = {k: v.clone().double().requires_grad_() for k, v in P.items()}
    tr, cr, _ = R.backbone_forward(Pr, cfg, img.double(), 1, path_scales=scales)
    assert rel(tok, tr) < 2e-4
    ((tok * dtok.cuda()).sum() + 0.01 * cv).backward()
    ((tr * dtok.double()).sum() + 0.01 * cr).backward()
    bad = [(n, rel(p.grad, Pr[n].grad)) for n, p in m.named_parameters() if Pr[n].grad is not None and rel(p.grad, Pr[n].grad) > 1e-3]
    assert not bad, bad
    m.eval()
    with torch.no_grad():
        tok_e, _ = m(img.cuda(), task_id=1)
        te, _, _ = R.backbone_forward(Pr, cfg, img.double(), 1, training=False)
    assert rel(tok_e, te) < 2e-4


def test_vit_mirror_elementwise_dropout_matches_oracle():
    """drop_rate > 0 (configs/nyud/vit_moe/*drop0.1*.yml): element-wise dropout at the reference's six call sites - pos_drop
    (vision_transformer_moe.py:791), Attention.proj_drop (:297,312), Mlp.drop after the activation and after fc2 (:258,260),
    the Dropout inside the experts' activation (:409-412), mlp_drop on the MoE output (:434,450) - with the masks pinned by
    site name on both sides; forward and every gradient against the float64 oracle; eval mode is the drop_rate = 0 model.
    Runs on the per-op path (the fused executor has no dropout and says so)."""
    _need_gpu()
    import zlib
    from m3vit_amd.vit import Dropout, VisionTransformerMoE
    from oracle import ref_torch as R
    kw = dict(img_size=(32, 32), embed_dim=64, depth=4, num_heads=2, moe_experts=4, moe_top_k=2, gate_dim=66, multi_gate=True)
    cfg = R.BackboneCfg(mlp_ratio=4.0, moe_mlp_ratio=1.0, vmoe_noisy_std=0.0, **kw)
    P = R.init_backbone_params(cfg, seed=13)
    p_drop = 0.25
    m = VisionTransformerMoE(mlp_ratio=4.0, moe_mlp_ratio=1, vmoe_noisy_std=0.0, drop_rate=p_drop, **kw).cuda()
    m.load_state_dict(P)
    seen = []

    def mask(site, shape, p):
        n = 1
        for d in shape:
            n *= d
        g = torch.Generator().manual_seed(zlib.crc32(site.encode()))
        seen.append(site)
        return ((torch.rand(n, generator=g) >= p).float() / (1.0 - p)).view(shape)

    img = torch.randn(3, 3, 32, 32)
    dtok = torch.randn(3, cfg.num_tokens, 64) * 0.1
    Dropout.mask_fn = mask
    try:
        m.train()
        tok, cv = m(img.cuda(), task_id=1)
        assert m.fused_fallback_reason == "element-wise dropout (drop_rate > 0)" and m._fused is None
        ours = list(seen)
        del seen[:]
        Pr = {k: v.clone().double().requires_grad_() for k, v in P.items()}
        tr, cr, _ = R.backbone_forward(Pr, cfg, img.double(), 1, dropout=lambda site, x: x * mask(site, x.shape, p_drop).double())
        assert sorted(ours) == sorted(seen) and len(ours) == 1 + 4 + 2 * 2 + 2 * 2, (ours, seen)
        assert rel(tok, tr) < 2e-4, rel(tok, tr)
        ((tok * dtok.cuda()).sum() + 0.01 * cv).backward()
        ((tr * dtok.double()).sum() + 0.01 * cr).backward()
        bad = [(n, rel(p.grad, Pr[n].grad)) for n, p in m.named_parameters()
               if Pr[n].grad is not None and rel(p.grad, Pr[n].grad) > 1e-3]
        assert not bad, bad
    finally:
        Dropout.mask_fn = None
    ref = VisionTransformerMoE(mlp_ratio=4.0, moe_mlp_ratio=1, vmoe_noisy_std=0.0, drop_rate=0.0, fused=False, **kw).cuda()
    ref.load_state_dict(P)
    m.eval(); ref.eval()
    with torch.no_grad():
        a, _ = m(img.cuda(), task_id=0)
        b, _ = ref(img.cuda(), task_id=0)
    assert rel(a, b) < 1e-5                                  # eval: dropout is the identity
    m.train()                                                # torch's own draws: a different mask every call, expectation kept
    with torch.no_grad():
        x1, _ = m(img.cuda(), task_id=0)
        x2, _ = m(img.cuda(), task_id=0)
    assert not torch.equal(x1, x2)


def test_composable_fmoe_path_with_custom_activation():
    """_fmoe_general_global_forward + FMoELinear with an activation the fused path does not cover."""
    _need_gpu()
    from m3vit_amd.fmoe.layers import _fmoe_general_global_forward
    from m3vit_amd.fmoe.linear import FMoELinear
    from oracle import ref_torch as R
    torch.manual_seed(3)
    T, D, H, E, k = 300, 64, 96, 4, 2
    l1, l2 = FMoELinear(E, D, H).cuda(), FMoELinear(E, H, D).cuda()
    x = torch.randn(T, D, device="cuda", requires_grad=True)
    idx = torch.stack([torch.randperm(E)[:k] for _ in range(T)]).cuda()

    def expert_fn(rows, cnt):
        return l2(torch.tanh(l1(rows, cnt)), cnt)
    y = _fmoe_general_global_forward(x, idx, expert_fn, E, 1)
    gy = torch.randn_like(y)
    y.backward(gy)
    # oracle: same routing, tanh experts
    xr = x.detach().double().cpu().requires_grad_()
    w1 = l1.weight.detach().double().cpu().requires_grad_(); w2 = l2.weight.detach().double().cpu().requires_grad_()
    counts, offsets, pos, ros = R.route_build(idx.cpu(), E)
    rows = xr[ros // k]
    outs, s = [], 0
    for e, n in enumerate(counts.tolist()):
        h = torch.tanh(torch.nn.functional.linear(rows[s:s + n], w1[e], l1.bias[e].detach().double().cpu()))
        outs.append(torch.nn.functional.linear(h, w2[e], l2.bias[e].detach().double().cpu())); s += n
    ref = torch.cat(outs)[pos]
    assert rel(y, ref) < 2e-5
    ref.backward(gy.double().cpu())
    assert rel(x.grad, xr.grad) < 2e-5 and rel(l1.weight.grad, w1.grad) < 2e-5 and rel(l2.weight.grad, w2.grad) < 2e-5


def test_task_conditioned_layer_matches_oracle():
    _need_gpu()
    from m3vit_amd.gate import NoisyGate_VMoE
    from m3vit_amd.moe_layer import FMoETransformerMLP
    from oracle import ref_torch as R
    torch.manual_seed(5)
    D, H, E, k, gtsd, T = 64, 64, 8, 2, 16, 200
    layer = FMoETransformerMLP(num_expert=E, d_model=D, d_gate=D + 3, d_hidden=H, gate=NoisyGate_VMoE, top_k=k,
                               vmoe_noisy_std=0, gate_task_specific_dim=gtsd,
                               activation=torch.nn.Sequential(torch.nn.GELU(), torch.nn.Dropout(0.))).cuda()
    with torch.no_grad():
        for p in layer.experts.parameters():
            p.normal_(0, 0.1)
    x = torch.randn(2, T // 2, D, device="cuda", requires_grad=True)
    tsf = torch.randn(gtsd, device="cuda", requires_grad=True)
    out, clean, noisy, std, top_logits, gates = layer(x, None, 1, tsf)
    assert out.shape == x.shape and std == 0.0 and top_logits.shape == (T, k + 1) and gates.shape == (T, E)
    gout = torch.randn_like(out)
    (out * gout).sum().backward()
    xr = x.detach().double().cpu().requires_grad_(); tr = tsf.detach().double().cpu().requires_grad_()
    wg = layer.gate.w_gate.detach().double().cpu().requires_grad_()
    gx = torch.cat((xr.reshape(T, D), tr.repeat(T, 1)), -1)                  # the reference's cat (:176-179)
    e = layer.experts
    o_ref, *_ = R.moe_layer(xr.reshape(T, D), gx, wg, e.htoh4.weight.detach().double().cpu(), e.htoh4.bias.detach().double().cpu(),
                            e.h4toh.weight.detach().double().cpu(), e.h4toh.bias.detach().double().cpu(), k)
    assert rel(out.reshape(T, D), o_ref) < 2e-5
    (o_ref * gout.double().cpu().reshape(T, D)).sum().backward()
    assert rel(x.grad, xr.grad) < 1e-4 and rel(tsf.grad, tr.grad) < 1e-4 and rel(layer.gate.w_gate.grad, wg.grad) < 1e-4


@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_pre_routed_layer_on_a_token_subset(act):
    """TokenFMoETransformerMLP.forward(inp, idx, score) (models/moe/token/custom_moe_layer.py:88-156) on the tokens a
    compute mask kept: the caller's routing is used as given (indices need not be a top-k of anything, scores need
    not be normalised), gradients reach the tokens, the scores and the expert parameters.  GELU experts run the
    fused grouped-FFN path, any other activation the composable scatter / FMoELinear / gather path."""
    _need_gpu()
    from m3vit_amd.moe_layer import TokenFMoETransformerMLP
    from oracle import ref_torch as R
    torch.manual_seed(4)
    E, D, H, k, T = 6, 64, 48, 3, 150
    fn = torch.nn.GELU() if act == "gelu" else torch.nn.ReLU()
    layer = TokenFMoETransformerMLP(num_expert=E, d_model=D, d_hidden=H, activation=fn, top_k=k).cuda()
    for p_ in layer.experts.parameters():
        torch.nn.init.normal_(p_, std=0.1)
    full = torch.randn(2, 100, D, device="cuda")
    keep = torch.randperm(200, device="cuda")[:T]                    # the compute-mask gather
    x = full.reshape(-1, D)[keep].clone().requires_grad_()
    idx = torch.stack([torch.randperm(E, device="cuda")[:k] for _ in range(T)])
    idx[:, 0] = 2                                                    # a crowded expert; expert 5 may stay empty
    idx[idx == 5] = 1
    score = torch.rand(T, k, device="cuda").requires_grad_()
    out = layer(x.view(1, T, D), idx, score).view(T, D)
    w1, b1, w2, b2 = [p_.detach().double().cpu().requires_grad_() for p_ in
                      (layer.experts.htoh4.weight, layer.experts.htoh4.bias, layer.experts.h4toh.weight, layer.experts.h4toh.bias)]
    xr = x.detach().double().cpu().requires_grad_(); sr = score.detach().double().cpu().requires_grad_()
    ref = torch.zeros(T, D, dtype=torch.float64)
    f = R.gelu_erf if act == "gelu" else torch.relu
    for j in range(k):
        e = idx[:, j].cpu()
        h = f(torch.einsum("td,thd->th", xr, w1[e]) + b1[e])
        ref = ref + sr[:, j:j + 1] * (torch.einsum("th,tdh->td", h, w2[e]) + b2[e])
    assert rel(out, ref) < 2e-5
    g = torch.randn(T, D, device="cuda")
    out.backward(g)
    ref.backward(g.double().cpu())
    assert rel(x.grad, xr.grad) < 1e-4 and rel(score.grad, sr.grad) < 1e-4
    for p_, r_ in ((layer.experts.htoh4.weight, w1), (layer.experts.htoh4.bias, b1), (layer.experts.h4toh.weight, w2),
                   (layer.experts.h4toh.bias, b2)):
        assert rel(p_.grad, r_.grad) < 1e-4


def _cls_cfg(**kw):
    from m3vit_amd.cls import MoEViTConfig
    base = dict(img_size=32, embed_dim=64, depth=2, num_heads=2, num_classes=16, moe_experts=4, moe_top_k=2,
                gate_dim=64, vmoe_noisy_std=0.0)
    base.update(kw)
    return MoEViTConfig(**base)


def test_classification_wrapper_matches_oracle():
    """MoEViTForImageNet (pretrain/models/moe_vit_cls.py:45-212): encoder -> LayerNorm(eps 1e-5) -> cls head;
    logits, cv_loss and every gradient of CE + 0.01 cv against the oracle backbone + torch head."""
    _need_gpu()
    import torch.nn.functional as F
    from m3vit_amd.cls import MoEViTForImageNet
    from oracle import ref_torch as R
    torch.manual_seed(6)
    m = MoEViTForImageNet(_cls_cfg()).cuda().train()
    assert set(k.split(".")[0] for k in m.state_dict()) == {"encoder", "norm", "head"}      # :48,95-96
    cfg = R.BackboneCfg(img_size=(32, 32), embed_dim=64, depth=2, num_heads=2, mlp_ratio=4.0, moe_mlp_ratio=1.0,
                        moe_experts=4, moe_top_k=2, gate_dim=64, multi_gate=False)
    Pr = {k[len("encoder."):]: v.detach().double().cpu().requires_grad_() for k, v in m.state_dict().items()
          if k.startswith("encoder.")}
    hw = {k: v.detach().double().cpu().requires_grad_() for k, v in m.state_dict().items() if not k.startswith("encoder.")}
    x = torch.randn(5, 3, 32, 32)
    y = torch.randint(0, 16, (5,))
    out = m(x.cuda())
    tok, cv, _ = R.backbone_forward(Pr, cfg, x.double(), None)
    cls = F.layer_norm(tok[:, 0], (64,), hw["norm.weight"], hw["norm.bias"], 1e-5)
    logits = F.linear(cls, hw["head.weight"], hw["head.bias"])
    assert out["logits"].shape == (5, 16) and rel(out["logits"], logits) < 2e-4
    assert abs(float(out["cv_loss"].detach()) - float(cv.detach())) < 1e-3 * max(1.0, float(cv.detach()))
    (F.cross_entropy(out["logits"], y.cuda()) + 0.01 * out["cv_loss"]).backward()
    (F.cross_entropy(logits, y) + 0.01 * cv).backward()
    ref = {**{"encoder." + k: v for k, v in Pr.items()}, **hw}
    bad = [(n, rel(p.grad, ref[n].grad)) for n, p in m.named_parameters() if ref[n].grad is not None and rel(p.grad, ref[n].grad) > 2e-3]
    assert not bad, bad


def test_amp_training_step_fp16():
    """The AMP iteration of pretrain/engine/train_one_epoch.py:35-61 (GradScaler: scale, unscale, clip, step, update)
    on fp16 activations: the loss goes down, no step is skipped at a sane scale, an absurd scale is detected
    (fp16 gradient overflow -> inf -> step skipped, scale halved) exactly as autocast training behaves."""
    _need_gpu()
    import torch.nn.functional as F
    from m3vit_amd.cls import MoEViTForImageNet, amp_train_step
    torch.manual_seed(7)
    m = MoEViTForImageNet(_cls_cfg(vmoe_noisy_std=1.0), act_dtype=torch.float16).cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=2e-3, weight_decay=0.05)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    x = torch.randn(16, 3, 32, 32, device="cuda")
    y = torch.randint(0, 16, (16,), device="cuda")
    crit = lambda samples, logits, targets: F.cross_entropy(logits, targets)          # noqa: E731
    losses = [amp_train_step(m, crit, opt, scaler, x, y, moe_cv_weight=0.01, clip_grad=1.0)[0] for _ in range(12)]
    assert all(l == l for l in losses) and losses[-1] < 0.7 * losses[0], losses
    assert scaler.get_scale() == 1024.0                                               # nothing overflowed
    before = [p.detach().clone() for p in m.parameters()]
    big = torch.amp.GradScaler("cuda", init_scale=2.0 ** 40)
    amp_train_step(m, crit, opt, big, x, y)
    assert big.get_scale() == 2.0 ** 39                                               # inf found: halved ...
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, m.parameters()))    # ... and the step skipped


def test_decoder_head_and_multitask_wrapper():
    """VisionTransformerUpHead / MultiTaskModel (models/heads/vit_up_head.py:73-224, models/models.py:215-342): key
    names, the cls-token quirk, output sizes, and the values against the same stack written with plain torch ops on
    the oracle backbone's tokens; multi-gate = one backbone pass per task, cv losses added."""
    _need_gpu()
    import torch.nn.functional as F
    from m3vit_amd.heads import MultiTaskModel, VisionTransformerUpHead
    from m3vit_amd.vit import VisionTransformerMoE
    from oracle import ref_torch as R
    torch.manual_seed(8)
    kw = dict(img_size=(32, 48), embed_dim=64, depth=2, num_heads=2, moe_experts=4, moe_top_k=2, gate_dim=66, multi_gate=True)
    bb = VisionTransformerMoE(mlp_ratio=4.0, moe_mlp_ratio=1, vmoe_noisy_std=0, **kw)
    tasks = ["semseg", "depth"]
    heads = torch.nn.ModuleDict({"semseg": VisionTransformerUpHead((32, 48), 16, 64, num_classes=5),
                                 "depth": VisionTransformerUpHead((32, 48), 16, 64, num_classes=1, num_conv=2, num_upsampe_layer=2)})
    m = MultiTaskModel(bb, heads, tasks, multi_gate=True).cuda().eval()
    assert {"norm.weight", "conv_0.weight", "conv_4.bias", "syncbn_fc_3.running_var"} <= set(heads["semseg"].state_dict())
    x = torch.randn(2, 3, 32, 48)
    out, cv = m(x.cuda())
    assert out["semseg"].shape == (2, 5, 32, 48) and out["depth"].shape == (2, 1, 32, 48)
    # reference computation: oracle backbone tokens (float64) -> the same head stack in float64 torch
    cfg = R.BackboneCfg(mlp_ratio=4.0, moe_mlp_ratio=1.0, **kw)
    P = {k: v.detach().double().cpu() for k, v in bb.state_dict().items()}
    cv_ref = 0.0
    for t_i, t in enumerate(tasks):
        tok, c, _ = R.backbone_forward(P, cfg, x.double(), t_i, training=False)
        h = heads[t].double().cpu()
        z = F.layer_norm(tok[:, 1:], (64,), h.norm.weight, h.norm.bias, 1e-6).transpose(1, 2).reshape(2, 64, 2, 3)
        if t == "semseg":
            for i in range(4):
                z = F.relu(getattr(h, f"syncbn_fc_{i}")(getattr(h, f"conv_{i}")(z)))
                if i < 3:
                    z = F.interpolate(z, scale_factor=2, mode="bilinear", align_corners=False)
            z = F.interpolate(h.conv_4(z), scale_factor=2, mode="bilinear", align_corners=False)
        else:
            z = F.relu(h.syncbn_fc_0(h.conv_0(z)))
            z = F.interpolate(z, size=z.shape[-1] * 4, mode="bilinear", align_corners=False)
            z = F.interpolate(h.conv_1(z), size=(32, 48), mode="bilinear", align_corners=False)
        z = F.interpolate(z, (32, 48), mode="bilinear")
        assert rel(out[t], z) < 2e-4, t
        heads[t].float().cuda()
    # single-task call on a shared (non multi-gate) wrapper returns only that head
    one = MultiTaskModel(bb, heads, tasks, multi_gate=False).cuda().eval()
    o1, _ = one(x.cuda(), single_task="depth", task_id=1)
    assert list(o1) == ["depth"] and torch.allclose(o1["depth"], out["depth"], atol=1e-5)


def test_sem_force_routing_override():
    """sem_force (custom_moe_layer.py:225-243): patches are routed by their semantic class to fixed expert pairs, the
    cls token keeps the gate's choice, all scores become 0.5.  Indices against a literal transcription of the
    reference's triple loop; output against the oracle's dispatch + bmm on those indices / scores."""
    _need_gpu()
    from m3vit_amd.gate import NoisyGate_VMoE
    from m3vit_amd.moe_layer import FMoETransformerMLP
    from oracle import ref_torch as R
    torch.manual_seed(12)
    E, D, k, B, N = 16, 64, 4, 2, 7                                  # 6 patches + cls per image
    layer = FMoETransformerMLP(num_expert=E, d_model=D, d_gate=D, d_hidden=D, gate=NoisyGate_VMoE, top_k=k,
                               vmoe_noisy_std=0, sem_force=True,
                               activation=torch.nn.Sequential(torch.nn.GELU(), torch.nn.Dropout(0.))).cuda()
    x = torch.randn(B, N, D, device="cuda")
    sem = torch.tensor([[0, 17, 255, 38, 5, 30], [12, 12, 3, 255, 21, 1]])     # 255: not in any group -> gate's routing
    seen = {}
    layer.gate_hook = lambda idx, score, _: seen.update(idx=idx.clone(), score=score.clone())
    out, clean, *_ = layer(x, None, None, None, sem)
    # the gate's own routing, then the reference's loop
    (gidx, _), *_ = R.gate_vmoe(x.reshape(-1, D).double().cpu(), layer.gate.w_gate.detach().double().cpu(), k)
    want = gidx.reshape(B, N, k).clone()
    for b in range(B):
        for i in range(sem.shape[1]):
            for j, grp in enumerate(layer.force_id):
                if int(sem[b, i]) in grp:
                    want[b, i + 1, :] = torch.tensor(([2 * j, 2 * j + 1] * ((k + 1) // 2))[:k])
    assert torch.equal(seen["idx"].cpu(), want.reshape(-1, k))
    assert torch.equal(seen["score"].cpu(), torch.full((B * N, k), 0.5))
    e = layer.experts
    y = R.moe_dispatch_ffn(x.reshape(-1, D).double().cpu(), want.reshape(-1, k), e.htoh4.weight.detach().double().cpu(),
                           e.htoh4.bias.detach().double().cpu(), e.h4toh.weight.detach().double().cpu(),
                           e.h4toh.bias.detach().double().cpu())
    ref = 0.5 * y.view(-1, k, D).sum(1)
    assert rel(out.reshape(-1, D), ref) < 2e-5


def _layer_and_refs(**kw):
    from m3vit_amd.gate import NoisyGate_VMoE
    from m3vit_amd.moe_layer import FMoETransformerMLP
    layer = FMoETransformerMLP(gate=NoisyGate_VMoE, vmoe_noisy_std=0,
                               activation=torch.nn.Sequential(torch.nn.GELU(), torch.nn.Dropout(0.)), **kw).cuda()
    e = layer.experts
    refs = {n: p.detach().double().cpu().requires_grad_() for n, p in
            (("w1", e.htoh4.weight), ("b1", e.htoh4.bias), ("w2", e.h4toh.weight), ("b2", e.h4toh.bias))}
    return layer, refs


def test_expert_prune_matches_reference_lines():
    """expert_prune (custom_moe_layer.py:221-224): gate_score = where(gate_score > prune_threshold, gate_score, 0) in
    front of the dispatch.  Forward and every gradient against a literal transcription on the oracle's gate / dispatch
    (float64 autograd): pruned slots contribute nothing and pass no gradient to the gate."""
    _need_gpu()
    from oracle import ref_torch as R
    torch.manual_seed(21)
    E, D, k, T = 8, 64, 4, 96
    thr = 0.16                                     # between the typical 2nd and 3rd softmax scores at E = 8
    layer, refs = _layer_and_refs(num_expert=E, d_model=D, d_gate=D, d_hidden=D, top_k=k, expert_prune=True, prune_threshold=thr)
    with torch.no_grad():
        layer.gate.w_gate.mul_(4.0)                # spread the softmax so that some of the k scores fall under the threshold
    x = torch.randn(T, D, device="cuda", requires_grad=True)
    dout = torch.randn(T, D, device="cuda") * 0.1
    seen = {}
    layer.gate_hook = lambda idx, score, _: seen.update(idx=idx.clone(), score=score.detach().clone())
    out, *_ = layer(x)
    out.backward(dout)
    # --- reference lines
    xr = x.detach().double().cpu().requires_grad_()
    wg = layer.gate.w_gate.detach().double().cpu().requires_grad_()
    (idx, score), *_ = R.gate_vmoe(xr, wg, k)
    score = torch.where(score > thr, score, torch.zeros_like(score))                       # :222
    frac = float((score == 0).float().mean())
    assert 0.05 < frac < 0.95, f"the test must prune some, not all, slots (pruned {frac:.0%})"
    y = R.moe_dispatch_ffn(xr, idx, refs["w1"], refs["b1"], refs["w2"], refs["b2"])         # :263-265
    ref = torch.bmm(score.view(T, 1, k), y.view(T, k, D)).reshape(T, D)                    # :291-305
    ref.backward(dout.double().cpu())
    assert torch.equal(seen["idx"].cpu(), idx) and torch.equal(seen["score"].cpu() == 0, score == 0)
    assert rel(out, ref) < 2e-5
    e = layer.experts
    for name, got, want in (("x", x.grad, xr.grad), ("w_gate", layer.gate.w_gate.grad, wg.grad),
                            ("w1", e.htoh4.weight.grad, refs["w1"].grad), ("b1", e.htoh4.bias.grad, refs["b1"].grad),
                            ("w2", e.h4toh.weight.grad, refs["w2"].grad), ("b2", e.h4toh.bias.grad, refs["b2"].grad)):
        assert rel(got, want) < 1e-4, name


@pytest.mark.parametrize("task", [0, 1, 2])
def test_regu_experts_fromtask_matches_reference_lines(task):
    """regu_experts_fromtask (noisy_gate_vmoe.py:87-89, custom_moe_layer.py:244-246): the gate scores only the
    num_experts_pertask columns of w_gate that start at start_experts_id[task] (ctor :106-112 / gate :40-46: a running
    sum), and the layer shifts the chosen indices by that start.  Forward and every gradient against a literal
    transcription; w_gate columns outside the task's slice get exactly zero gradient."""
    _need_gpu()
    from oracle import ref_torch as R
    torch.manual_seed(22)
    E, D, k, T, npt, ntask = 16, 64, 2, 80, 6, 3
    layer, refs = _layer_and_refs(num_expert=E, d_model=D, d_gate=D, d_hidden=D, top_k=k, regu_experts_fromtask=True,
                                  num_experts_pertask=npt, num_tasks=ntask)
    # the reference's start ids (running sum, both ctors)
    starts, s = [], 0
    for i in range(ntask):
        s = s + int(i * (E - npt) / (ntask - 1))
        starts.append(s)
    assert layer.start_experts_id == starts and layer.gate.start_experts_id == starts
    if starts[task] + npt > E:
        pytest.skip("the reference's running-sum start ids leave the expert range for this task")
    x = torch.randn(T, D, device="cuda", requires_grad=True)
    dout = torch.randn(T, D, device="cuda") * 0.1
    seen = {}
    layer.gate_hook = lambda idx, score, _: seen.update(idx=idx.clone())
    out, *_ = layer(x, task_id=task)
    out.backward(dout)
    xr = x.detach().double().cpu().requires_grad_()
    wg = layer.gate.w_gate.detach().double().cpu().requires_grad_()
    (idx, score), *_ = R.gate_vmoe(xr, wg[:, starts[task]:starts[task] + npt], k)          # noisy_gate_vmoe.py:87-88
    idx = idx + starts[task]                                                               # custom_moe_layer.py:246
    y = R.moe_dispatch_ffn(xr, idx, refs["w1"], refs["b1"], refs["w2"], refs["b2"])
    ref = torch.bmm(score.view(T, 1, k), y.view(T, k, D)).reshape(T, D)
    ref.backward(dout.double().cpu())
    assert torch.equal(seen["idx"].cpu(), idx)
    assert int(idx.min()) >= starts[task] and int(idx.max()) < starts[task] + npt
    assert rel(out, ref) < 2e-5
    e = layer.experts
    gw = layer.gate.w_gate.grad
    outside = torch.ones(E, dtype=torch.bool)
    outside[starts[task]:starts[task] + npt] = False
    assert float(gw[:, outside.cuda()].abs().max()) == 0.0
    for name, got, want in (("x", x.grad, xr.grad), ("w_gate", gw, wg.grad),
                            ("w1", e.htoh4.weight.grad, refs["w1"].grad), ("b1", e.htoh4.bias.grad, refs["b1"].grad),
                            ("w2", e.h4toh.weight.grad, refs["w2"].grad), ("b2", e.h4toh.bias.grad, refs["b2"].grad)):
        assert rel(got, want) < 1e-4, name


@pytest.mark.parametrize("std", [0.0, 1.0])
def test_origin_convention_matches_ckpt_convention(std):
    """convention="origin" (models/moe/origin/*: gate -> (idx, score) + self.loss, layer -> tensor, backbone -> tokens,
    loss gathered by utils/moe_utils.py::collect_noisy_gating_loss) against the ckpt-convention model with the same
    parameters (itself pinned to the oracle by test_vit_mirror_matches_oracle_fwd_bwd): identical tokens, the collected
    loss = weight * the ckpt backbone's cv loss, identical gradients of tokens . d + loss; eval mode stores loss 0."""
    _need_gpu()
    from m3vit_amd.moe_utils import collect_noisy_gating_loss
    from m3vit_amd.vit import VisionTransformerMoE
    kw = dict(img_size=(32, 48), embed_dim=64, depth=4, num_heads=2, moe_experts=8, moe_top_k=2, gate_dim=66,
              multi_gate=True, mlp_ratio=4.0, moe_mlp_ratio=1, vmoe_noisy_std=std)
    torch.manual_seed(3)
    a = VisionTransformerMoE(**kw).cuda()
    b = VisionTransformerMoE(convention="origin", **kw).cuda()
    b.load_state_dict(a.state_dict())
    a.train(); b.train()
    img = torch.randn(3, 3, 32, 48).cuda()
    dtok = (torch.randn(3, a.num_patches + 1, 64) * 0.1).cuda()
    w = 0.01
    for task in (0, 1):
        torch.manual_seed(70 + task)                       # the gates draw their noise with torch.randn
        tok_a, cv_a = a(img, task_id=task)
        torch.manual_seed(70 + task)
        out_b = b(img, task_id=task)
        assert torch.is_tensor(out_b) and out_b.shape == tok_a.shape          # tokens only
        assert torch.equal(out_b, tok_a)
        loss_b = collect_noisy_gating_loss(b, w)
        assert abs(float(loss_b) - w * float(cv_a)) <= 1e-6 * max(1.0, abs(w * float(cv_a)))
        assert not any(m.has_loss for m in b.modules() if hasattr(m, "has_loss"))      # get_loss() cleared them
        ((tok_a * dtok).sum() + w * cv_a).backward()
        ((out_b * dtok).sum() + loss_b).backward()
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None:
            assert pb.grad is None or float(pb.grad.abs().max()) == 0.0, n
            continue
        if pb.grad is None:                                # ckpt: 0.0 * sum(other gates) keep-alive gives zeros, origin: None
            assert float(pa.grad.abs().max()) == 0.0, n
            continue
        assert rel(pb.grad, pa.grad) < 1e-5, n
    # the layer / gate level returns
    blk = b.blocks[1]
    x = torch.randn(2, 7, 64, device="cuda")
    y = blk.mlp(x, None, 0, None, None)
    assert torch.is_tensor(y) and y.shape == x.shape
    g_out = blk.mlp.gate[0](x)
    assert isinstance(g_out, tuple) and len(g_out) == 2 and g_out[0].shape == (2, 7, 2) and g_out[0].dtype == torch.int64
    assert blk.mlp.gate[0].has_activation and blk.mlp.gate[0].get_activation().shape == (2, 7, 8)
    b.eval()
    b(img, task_id=0)
    assert float(collect_noisy_gating_loss(b, 1.0)) == 0.0


@pytest.mark.parametrize("dtype,gdtype", [(torch.float32, torch.float32), (torch.float16, torch.float16),
                                          (torch.float16, torch.float32), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("shape", [(2, 8, 1, 1), (1, 16, 1, 5), (3, 8, 4, 1), (2, 40, 7, 9), (2, 256, 30, 40)])
@pytest.mark.parametrize("relu", [True, False])
def test_relu_up2x_kernels_match_torch(dtype, gdtype, shape, relu):
    """m3_relu_up2x_fwd / _bwd (ReLU + bilinear x2, align_corners False, channels-last) against F.interpolate(F.relu(x)) and
    its autograd in float64, incl. one-pixel planes (every neighbour clamped) and the fp32-output / fp32-gradient variants the
    classifier stage of the decoder head uses (models/heads/vit_up_head.py:181-214)."""
    _need_gpu()
    import torch.nn.functional as F
    from m3vit_amd.heads import ReluUp2xFn
    if dtype == torch.float32 and shape[1] % 4 or dtype != torch.float32 and shape[1] % 8:
        pytest.skip("channel count does not fill a vector")
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    out_fp32 = gdtype == torch.float32 and dtype != torch.float32
    y = ReluUp2xFn.apply(x, relu, out_fp32)
    assert y.dtype == (torch.float32 if out_fp32 else dtype) and y.is_contiguous(memory_format=torch.channels_last)
    xr = x.detach().double().cpu().requires_grad_()
    yr = F.interpolate(F.relu(xr) if relu else xr, scale_factor=2, mode="bilinear", align_corners=False)
    tol = {torch.float32: 1e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
    assert tuple(y.shape) == tuple(yr.shape) and rel(y, yr) < tol, rel(y, yr)
    dy = torch.randn(yr.shape, generator=g)
    y.backward(dy.to(y.dtype).cuda().contiguous(memory_format=torch.channels_last))
    yr.backward(dy.to(y.dtype).double())
    assert x.grad.dtype == dtype and rel(x.grad, xr.grad) < tol, rel(x.grad, xr.grad)


def test_head_fused_resize_matches_torch_stages():
    """VisionTransformerUpHead(fused_resize=True) - ReLU + x2 resize of every stage as one kernel each way, the classifier
    stage's resize with fp32 output - against the same head on torch's relu + interpolate, under fp16 autocast (the
    convolutions return channels-last tensors there): logits and every parameter gradient."""
    _need_gpu()
    from m3vit_amd.heads import ReluUp2xFn, VisionTransformerUpHead
    torch.manual_seed(4)
    a = VisionTransformerUpHead(img_size=(64, 96), embed_dim=64, num_classes=40, amp=True, fused_resize=True).cuda().train()
    b = VisionTransformerUpHead(img_size=(64, 96), embed_dim=64, num_classes=40, amp=True, fused_resize=False).cuda().train()
    b.load_state_dict(a.state_dict())
    tok = torch.randn(3, 4 * 6 + 1, 64, device="cuda")
    calls = []
    orig = ReluUp2xFn.forward
    ReluUp2xFn.forward = staticmethod(lambda ctx, x, relu, out_fp32: (calls.append((tuple(x.shape), relu, out_fp32)), orig(ctx, x, relu, out_fp32))[1])
    try:
        ya = a(tok)
    finally:
        ReluUp2xFn.forward = orig
    assert len(calls) == 4 and calls[-1][1:] == (False, True) and all(c[1:] == (True, False) for c in calls[:3]), calls
    yb = b(tok)
    assert ya.shape == (3, 40, 64, 96) and ya.dtype == yb.dtype == torch.float32
    # both are fp16-autocast approximations of the fp32 head (batch-norm statistics over few pixels amplify fp16 rounding
    # of the stage inputs): the fused stages must sit as close to the fp32 head as torch's own fp16 stages do
    ref = VisionTransformerUpHead(img_size=(64, 96), embed_dim=64, num_classes=40, amp=False, fused_resize=False).cuda().train()
    ref.load_state_dict(a.state_dict())
    yr = ref(tok)
    ea, eb = rel(ya, yr), rel(yb, yr)
    assert ea < 1.5 * eb + 1e-3 and ea < 1e-2, (ea, eb)
    w = torch.randn_like(ya)
    for m, y in ((a, ya), (b, yb), (ref, yr)):
        (y * w).sum().backward()
    worst = []
    for (n, p), (_, q), (_, r) in zip(a.named_parameters(), b.named_parameters(), ref.named_parameters()):
        if n.startswith("conv_") and n.endswith(".bias") and n != "conv_4.bias":
            continue                      # (a bias in front of a batch norm has a zero gradient: rounding noise only)
        ga, gb = rel(p.grad, r.grad), rel(q.grad, r.grad)
        worst.append((n, ga, gb))
        assert ga < 1.5 * gb + 5e-3, (n, ga, gb)
    print("head gradients vs the fp32 head (fused, torch fp16 stages):", [(n, f"{x:.1e}", f"{y_:.1e}") for n, x, y_ in worst[:4]])


def test_head_amp_option_matches_fp32_head():
    """VisionTransformerUpHead(amp=True): the conv / BN / resize stages under fp16 autocast (the reference's AMP step) - same
    parameters and state_dict keys as the fp32 head, outputs and parameter gradients within fp16 rounding of it."""
    _need_gpu()
    from m3vit_amd.heads import VisionTransformerUpHead
    torch.manual_seed(5)
    kw = dict(img_size=(64, 96), embed_dim=64, num_classes=7, channels=32)
    a = VisionTransformerUpHead(**kw).cuda().train()
    b = VisionTransformerUpHead(amp=True, **kw).cuda().train()
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    b.load_state_dict(a.state_dict())
    tok = torch.randn(3, 4 * 6 + 1, 64, device="cuda")
    ta, tb = tok.clone().requires_grad_(), tok.clone().requires_grad_()
    ya, yb = a(ta), b(tb)
    assert ya.shape == yb.shape == (3, 7, 64, 96) and yb.dtype in (torch.float16, torch.float32)
    assert rel(yb, ya) < 2e-2
    d = torch.randn_like(ya)
    (ya * d).sum().backward()
    (yb.float() * d).sum().backward()
    assert rel(tb.grad, ta.grad) < 1e-1          # four BN + ReLU stages on a 3-image batch: ReLU flips near zero dominate
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if n in ("conv_0.bias", "conv_1.bias", "conv_2.bias", "conv_3.bias"):
            continue                                   # a bias in front of a batch norm: its gradient is rounding noise in both heads
        if pa.grad is not None and float(pa.grad.norm()) > 0:
            assert pb.grad.dtype == torch.float32 and rel(pb.grad, pa.grad) < 1e-1, n


def test_head_multi_level_outputs_and_tam_wiring():
    """vit_up_head.py:128-131,184-214 (multi_level: a 1x1 classifier after every upsampling) and :190-206 + models/models.py:
    246-279,313-327 (tam: the heads hand their three intermediate features to one TamModule per level, whose per-task
    outputs are added as 'tam_level{l}_{task}', training only) against a functional transcription of those lines."""
    _need_gpu()
    import torch.nn.functional as F
    from m3vit_amd.heads import MultiTaskModel, TamModule, VisionTransformerUpHead
    from m3vit_amd.vit import VisionTransformerMoE
    torch.manual_seed(11)
    tok = torch.randn(2, 1 + 6, 64, device="cuda")                       # cls + 2 x 3 patches

    def stack(h, x, stop=4):
        """conv / BN / ReLU / x2 stages of the 4-conv head (:181-214) in plain functional ops; returns per-stage tensors"""
        z = F.layer_norm(x[:, 1:], (64,), h.norm.weight, h.norm.bias, 1e-6).transpose(1, 2).reshape(2, 64, 2, 3)
        post_relu, post_up = [], []
        for i in range(4):
            conv, bn = getattr(h, f"conv_{i}"), getattr(h, f"syncbn_fc_{i}")
            z = F.conv2d(z, conv.weight, conv.bias, padding=1)
            z = F.relu(F.batch_norm(z, bn.running_mean, bn.running_var, bn.weight, bn.bias, training=False, eps=bn.eps))
            post_relu.append(z)
            if i < 3:
                z = F.interpolate(z, scale_factor=2, mode="bilinear", align_corners=False)
                post_up.append(z)
        final = F.interpolate(F.conv2d(z, h.conv_4.weight, h.conv_4.bias), scale_factor=2, mode="bilinear", align_corners=False)
        return post_relu, post_up, final

    ml = VisionTransformerUpHead((32, 48), 16, 64, num_classes=5, multi_level=True).cuda().eval()
    assert {"output_level_0.weight", "output_level_2.bias"} <= set(ml.state_dict())
    out = ml(tok)
    assert list(out) == ["level1", "level2", "level3", "final"]
    _, post_up, final = stack(ml, tok)
    for i in range(3):
        lv = getattr(ml, f"output_level_{i}")
        assert rel(out[f"level{i + 1}"], F.conv2d(post_up[i], lv.weight, lv.bias)) < 1e-5
    assert rel(out["final"], final) < 1e-5 and out["final"].shape == (2, 5, 32, 48)

    # TAM: eval BN statistics inside the heads (so the transcription can use running stats), wrapper in training mode
    kw = dict(img_size=(32, 48), embed_dim=64, depth=2, num_heads=2, moe_experts=4, moe_top_k=2, gate_dim=66, multi_gate=True)
    bb = VisionTransformerMoE(mlp_ratio=4.0, moe_mlp_ratio=1, vmoe_noisy_std=0, **kw)
    tasks = ["semseg", "depth"]
    nout = {"semseg": 5, "depth": 1}
    heads = torch.nn.ModuleDict({t: VisionTransformerUpHead((32, 48), 16, 64, num_classes=nout[t], tam=True) for t in tasks})
    # (the 2x-down / 2x-up modulation path needs feature maps divisible by 4: levels 1 and 2 are 8 x 12 and 16 x 24 here)
    tams = {1: TamModule(tasks, 256, nout), 2: TamModule(tasks, 256, nout)}
    m = MultiTaskModel(bb, heads, tasks, multi_gate=True, tam_models=tams).cuda()
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eval()
    x = torch.randn(2, 3, 32, 48, device="cuda")
    out, cv = m(x)
    assert set(out) == {"semseg", "depth", "tam_level1_semseg", "tam_level1_depth", "tam_level2_semseg", "tam_level2_depth"}
    feats = {1: {}, 2: {}}
    for i, t in enumerate(tasks):
        tokens, _ = bb(x, task_id=i)
        post_relu, _, final = stack(heads[t], tokens.float())
        feats[1][t], feats[2][t] = post_relu[2], post_relu[3]            # tam_feature1 / 2: after conv_2 / conv_3
        assert rel(out[t], F.interpolate(final, (32, 48), mode="bilinear")) < 1e-4
    for lvl in (1, 2):
        y = tams[lvl](feats[lvl])
        for t in tasks:
            assert rel(out[f"tam_level{lvl}_{t}"], F.interpolate(y[t], (32, 48), mode="bilinear", align_corners=False)) < 1e-4
    m.eval()                                                             # inference: no TAM outputs, heads return tensors
    out_e, _ = m(x)
    assert set(out_e) == set(tasks)


def test_patch_embed_mirror_matches_conv2d_including_the_image_gradient():
    """PatchEmbed (vision_transformer_moe.py:315-341: Conv2d(kernel = stride = patch)) on m3_im2row + the GEMM: tokens, d weight,
    d bias and - when the images require it - d images, against torch's convolution in float64"""
    _need_gpu()
    from m3vit_amd.vit import PatchEmbed
    torch.manual_seed(3)
    pe = PatchEmbed(img_size=(32, 48), patch_size=16, in_chans=3, embed_dim=64).cuda()
    x = torch.randn(2, 3, 32, 48, device="cuda", requires_grad=True)
    y = pe(x)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    x64 = x.detach().double().cpu().requires_grad_()
    w64, b64 = pe.proj.weight.detach().double().cpu().requires_grad_(), pe.proj.bias.detach().double().cpu().requires_grad_()
    y64 = torch.nn.functional.conv2d(x64, w64, b64, stride=16).flatten(2).transpose(1, 2)
    (y64 * g.double().cpu()).sum().backward()
    assert rel(y, y64) < 2e-5 and rel(x.grad, x64.grad) < 2e-5
    assert rel(pe.proj.weight.grad, w64.grad) < 1e-4 and rel(pe.proj.bias.grad, b64.grad) < 1e-4
