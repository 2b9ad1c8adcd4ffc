"""GPU: the bf16 activation dtype (M3_BF16, SURVEY section 8 b's dtype enum) through every entry point that takes it -
GEMM (dense epilogues, grouped gather / scatter), weight gradients (dense, grouped, fused bias), column sums, LayerNorm
forward / backward, combine forward / backward, gate (indices bit-exact against the C oracle on the same bf16-rounded
rows), operand casts, row gather - against torch fp64 on the same bf16-rounded operands (bf16 has 8 significant bits:
outputs stored in bf16 are compared at 6e-3, fp32 outputs of bf16 products at 2e-3), and the MoE layer mirror
(FMoETransformerMLP: gate -> dispatch -> grouped FFN -> combine) forward + backward on bfloat16 input rows; attention
forward / backward (LDS-resident and streamed kernels); and the fused executor end to end in bf16 against the float64
oracle.  Only the fused FFN kernel (m3_ffn_fwd) stays fp16."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
TOL_BF, TOL_F32 = 6e-3, 2e-3


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from m3vit_amd import ops as _ops
    return _ops


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, scale=1.0, seed=0, dtype=BF):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def gelu64(x):
    return 0.5 * x * (1 + torch.erf(x / 2 ** 0.5))


def gelu_grad64(x):
    return 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * np.pi) ** 0.5


@pytest.mark.parametrize("M,N,K", [(1500, 1152, 384), (700, 384, 1536), (333, 200, 72)])
def test_gemm_bf16_epilogues(ops, M, N, K):
    A, B, bias = rnd(M, K, seed=1), rnd(N, K, scale=0.05, seed=2), rnd(N, scale=0.1, seed=3, dtype=torch.float32)
    ref = A.double() @ B.double().t() + bias.double()
    C = torch.empty(M, N, dtype=BF, device=dev()); pre = torch.empty(M, N, dtype=BF, device=dev())
    ops.gemm_nt(A.to(dev()), B.to(dev()), C, bias=bias.to(dev()), act=ops.M3_ACT_GELU, pre_out=pre)
    assert rel(pre, ref) < TOL_BF and rel(C, gelu64(ref)) < TOL_BF
    res = rnd(M, N, seed=4, dtype=torch.float32)
    C32 = torch.empty(M, N, dtype=torch.float32, device=dev())
    ops.gemm_nt(A.to(dev()), B.to(dev()), C32, bias=bias.to(dev()), residual=res.to(dev()))
    assert rel(C32, ref + res.double()) < TOL_F32
    gp = rnd(M, N, seed=5)
    Cg = torch.empty(M, N, dtype=BF, device=dev())
    ops.gemm_nt(A.to(dev()), B.to(dev()), Cg, gelu_grad_pre=gp.to(dev()))
    assert rel(Cg, (A.double() @ B.double().t()) * gelu_grad64(gp.double())) < TOL_BF


def test_grouped_gemm_and_wgrad_bf16(ops):
    """expert FC1 (gathered rows, GELU, pre) + FC2 (token-major scatter) and both weight gradients with fused bias
    gradients; ragged groups and an empty expert"""
    E, k, T, D, H = 8, 2, 900, 384, 768
    g = torch.Generator().manual_seed(6)
    choices = torch.tensor([e for e in range(E) if e != 3])
    idx = torch.stack([choices[torch.randperm(E - 1, generator=g)[:k]] for _ in range(T)])
    r = ops.route_build(idx.to(torch.int32).to(dev()), E)
    R = T * k
    x = rnd(T, D, seed=7)
    w1, b1 = rnd(E, H, D, scale=0.05, seed=8), rnd(E, H, scale=0.1, seed=9, dtype=torch.float32)
    w2, b2 = rnd(E, D, H, scale=0.05, seed=10), rnd(E, D, scale=0.1, seed=11, dtype=torch.float32)
    hid = torch.empty(R, H, dtype=BF, device=dev()); pre = torch.empty(R, H, dtype=BF, device=dev())
    ops.gemm_nt(x.to(dev()), w1.to(dev()), hid, M=R, bias=b1.to(dev()), act=ops.M3_ACT_GELU, pre_out=pre,
                a_row_idx=r.row_of_slot, a_row_div=k, group_offsets=r.offsets, tile_starts=r.tile_starts)
    y = torch.empty(R, D, dtype=BF, device=dev())
    ops.gemm_nt(hid, w2.to(dev()), y, M=R, bias=b2.to(dev()), c_row_idx=r.row_of_slot, group_offsets=r.offsets,
                tile_starts=r.tile_starts)
    ros = r.row_of_slot.cpu().long(); off = r.offsets.cpu().tolist()
    e_of = torch.zeros(R, dtype=torch.long)
    for e in range(E):
        e_of[off[e]:off[e + 1]] = e
    p64 = torch.einsum("rd,rhd->rh", x.double()[ros // k], w1.double()[e_of]) + b1.double()[e_of]
    assert rel(pre, p64) < TOL_BF and rel(hid, gelu64(p64)) < TOL_BF
    y64 = torch.einsum("rh,rdh->rd", hid.double().cpu(), w2.double()[e_of]) + b2.double()[e_of]
    want = torch.empty(R, D, dtype=torch.float64); want[ros] = y64
    assert rel(y, want) < TOL_BF
    dpre, dy = rnd(R, H, scale=0.5, seed=12), rnd(R, D, scale=0.5, seed=13)
    dw1, db1 = torch.zeros(E, H, D, device=dev()), torch.zeros(E, H, device=dev())
    ops.wgrad_tn(dpre.to(dev()), x.to(dev()), dw1, M=R, a_row_idx=r.row_of_slot, a_row_div=k, group_offsets=r.offsets, db=db1)
    dw2, db2 = torch.zeros(E, D, H, device=dev()), torch.zeros(E, D, device=dev())
    ops.wgrad_tn(dy.to(dev()), hid, dw2, M=R, c_row_idx=r.row_of_slot, group_offsets=r.offsets, db=db2)
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        if off[e] == off[e + 1]:
            assert float(dw1[e].abs().max()) == 0.0
            continue
        assert rel(dw1[e], dpre.double()[sl].t() @ x.double()[ros[sl] // k]) < TOL_F32
        assert rel(db1[e], dpre.double()[sl].sum(0)) < TOL_F32
        assert rel(dw2[e], dy.double()[ros[sl]].t() @ hid.double().cpu()[sl]) < TOL_F32
        assert rel(db2[e], dy.double()[ros[sl]].sum(0)) < TOL_F32


def test_dense_wgrad_and_colsum_bf16(ops):
    M, N, K = 3001, 384, 256
    dC, A = rnd(M, N, seed=14), rnd(M, K, seed=15)
    dW0 = rnd(N, K, seed=16, dtype=torch.float32)
    dW, db = dW0.to(dev()), torch.zeros(N, device=dev())
    ops.wgrad_tn(dC.to(dev()), A.to(dev()), dW, beta=1, db=db)
    assert rel(dW, dW0.double() + dC.double().t() @ A.double()) < TOL_F32
    assert rel(db, dC.double().sum(0)) < TOL_F32
    cs = torch.zeros(N, device=dev())
    ops.colsum(dC.to(dev()), cs)
    assert rel(cs, dC.double().sum(0)) < TOL_F32


def test_layernorm_and_combine_bf16(ops):
    T, D, k = 777, 384, 4
    x = rnd(T, D, seed=17, dtype=torch.float32) * 2 + 0.3
    gam, bet = rnd(D, seed=18, dtype=torch.float32) * 0.2 + 1, rnd(D, seed=19, dtype=torch.float32) * 0.1
    y = torch.empty(T, D, dtype=BF, device=dev()); mean = torch.empty(T, device=dev()); rstd = torch.empty(T, device=dev())
    ops.layernorm_fwd(x.to(dev()), gam.to(dev()), bet.to(dev()), y, mean, rstd)
    x64 = x.double()
    mu = x64.mean(1, keepdim=True); var = x64.var(1, unbiased=False, keepdim=True)
    xh = (x64 - mu) / torch.sqrt(var + 1e-6)
    assert rel(y, xh * gam.double() + bet.double()) < TOL_BF
    dy = rnd(T, D, seed=20)
    dres = rnd(T, D, seed=21, dtype=torch.float32)
    dx = torch.empty(T, D, device=dev()); dg = torch.zeros(D, device=dev()); dbt = torch.zeros(D, device=dev())
    dxa = torch.empty(T, D, dtype=BF, device=dev())
    ops.layernorm_bwd(dy.to(dev()), x.to(dev()), mean, rstd, gam.to(dev()), dres.to(dev()), dx, dg, dbt, beta=0, dx_act=dxa)
    g = dy.double() * gam.double()
    ref = dres.double() + (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True)) / torch.sqrt(var + 1e-6)
    assert rel(dx, ref) < TOL_F32 and rel(dxa, ref) < TOL_BF
    assert rel(dg, (dy.double() * xh).sum(0)) < TOL_F32 and rel(dbt, dy.double().sum(0)) < TOL_F32
    yk = rnd(T * k, D, seed=22); sc = torch.rand(T, k, generator=torch.Generator().manual_seed(23))
    res = rnd(T, D, seed=24, dtype=torch.float32)
    out = torch.empty(T, D, device=dev())
    ops.combine_fwd(yk.to(dev()), sc.to(dev()), res.to(dev()), out)
    want = res.double() + (yk.double().view(T, k, D) * sc.double()[:, :, None]).sum(1)
    assert rel(out, want) < TOL_F32
    dout = rnd(T, D, seed=25, dtype=torch.float32)
    dyk = torch.empty(T * k, D, dtype=BF, device=dev()); dsc = torch.empty(T, k, device=dev())
    ops.combine_bwd(dout.to(dev()), yk.to(dev()), sc.to(dev()), dyk, dsc)
    assert rel(dyk, (dout.double()[:, None, :] * sc.double()[:, :, None]).reshape(T * k, D)) < TOL_BF
    assert rel(dsc, (dout.double()[:, None, :] * yk.double().view(T, k, D)).sum(2)) < TOL_F32


def test_gate_bf16_indices_bit_exact_vs_c_oracle(ops):
    from oracle import c_oracle
    T, D, E, k = 1000, 384, 16, 4
    x = rnd(T, D, seed=26)
    wg = rnd(D, E, scale=0.3, seed=27, dtype=torch.float32)
    g = ops.gate_fwd(x.to(dev()), wg.to(dev()), k)
    want = c_oracle.gate_fwd(x.float().numpy(), wg.numpy(), k)
    assert np.array_equal(g["idx"].cpu().numpy(), want["idx"])
    assert rel(g["score"], torch.from_numpy(want["score"])) < 1e-6


def test_casts_and_gather_bf16(ops):
    w = rnd(3, 40, 72, seed=28, dtype=torch.float32)
    c = ops.cast_matrix(w.to(dev()), torch.empty(3, 40, 72, dtype=BF, device=dev()))
    ct = ops.cast_matrix(w.to(dev()), torch.empty(3, 72, 40, dtype=BF, device=dev()), transpose=True)
    assert torch.equal(c.cpu(), w.to(BF)) and torch.equal(ct.cpu(), w.to(BF).transpose(1, 2).contiguous())
    src = rnd(500, 128, seed=29)
    idx = torch.randint(0, 500 * 2, (700,), generator=torch.Generator().manual_seed(30)).to(torch.int32)
    dst = torch.empty(700, 128, dtype=BF, device=dev())
    ops.gather_rows(src.to(dev()), idx.to(dev()), dst, div=2)
    assert torch.equal(dst.cpu(), src[(idx // 2).long()])


def test_moe_layer_mirror_runs_in_bf16():
    """FMoETransformerMLP mirror (custom_moe_layer.py:161-305) with act_dtype=bfloat16: forward and every gradient
    against the float64 oracle of the same layer, routing taken from the oracle gate on the layer's own gate input"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from m3vit_amd.gate import NoisyGate_VMoE
    from m3vit_amd.moe_layer import FMoETransformerMLP
    torch.manual_seed(31)
    T, D, E, k = 600, 128, 8, 2
    lay = FMoETransformerMLP(num_expert=E, d_model=D, d_gate=D, d_hidden=256, top_k=k, vmoe_noisy_std=0, gate=NoisyGate_VMoE,
                             activation=torch.nn.Sequential(torch.nn.GELU(), torch.nn.Dropout(0.))).cuda()
    ref = lay                                      # same parameters; the activation dtype follows the input's
    x = torch.randn(T, D, device=dev()).to(BF).requires_grad_(True)
    x2 = x.detach().float().requires_grad_(True)
    out = lay(x); outr = ref(x2)
    o, orf = (out[0] if isinstance(out, tuple) else out), (outr[0] if isinstance(outr, tuple) else outr)
    assert rel(o, orf) < 2e-2                       # two bf16 GEMMs deep; a few tokens may route differently at bf16
    o.float().pow(2).sum().backward(); orf.float().pow(2).sum().backward()
    assert rel(x.grad, x2.grad) < 5e-2


def _attn_ref(qkv, B, N, h, dh):
    q, k, v = qkv.view(B, N, 3, h, dh).permute(2, 0, 3, 1, 4)
    a = torch.softmax((q @ k.transpose(-2, -1)) * dh ** -0.5, dim=-1)
    return (a @ v).transpose(1, 2).reshape(B * N, h * dh)


@pytest.mark.parametrize("B,N,h,dh", [(2, 197, 6, 32), (2, 197, 3, 64), (3, 50, 2, 32), (1, 577, 2, 64), (1, 1025, 2, 32)])
def test_attention_bf16(ops, B, N, h, dh):
    """N <= 256: the LDS-resident kernels; longer: the streamed ones (+ the ordered dQ slab sum)"""
    C = h * dh
    qkv = rnd(B * N, 3 * C, seed=40).to(dev())
    o = torch.full((B * N, C), float("nan"), dtype=BF, device=dev())
    lse = torch.empty(B, h, N, device=dev())
    ops.attention_fwd(qkv, B, N, h, dh, o, lse)
    qr = qkv.double().requires_grad_()
    ref = _attn_ref(qr, B, N, h, dh)
    assert rel(o, ref) < TOL_BF
    d_o = rnd(B * N, C, seed=41).to(dev())
    ref.backward(d_o.double())
    dqkv = torch.full_like(qkv, float("nan"))
    need = int(ops.lib().m3_attention_bwd_ws_elems(B, N, h, dh))
    ws = torch.empty(need, device=dev()) if need else None
    ops.attention_bwd(qkv, o, d_o, lse, B, N, h, dh, dqkv, dq_ws=ws)
    assert torch.isfinite(dqkv).all() and rel(dqkv, qr.grad) < 2 * TOL_BF


def test_fused_executor_in_bf16_matches_oracle():
    """BackboneEngine(dtype=bfloat16): tokens, balance loss and every parameter gradient of a 4-block multi-gate MoE-ViT
    against the float64 oracle with the routing the oracle takes on the engine's own gate inputs (bf16 keeps 8
    significant bits: bounds 8x the fp16 ones of tests/test_engine.py)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import test_engine as TE              # (tests/ is on sys.path: rootdir-relative test modules)
    from oracle import ref_torch as R
    cfg = R.BackboneCfg(img_size=(32, 48), embed_dim=64, depth=4, num_heads=2, mlp_ratio=4.0, moe_mlp_ratio=1.0,
                        moe_experts=4, moe_top_k=2, gate_dim=66, multi_gate=True)
    TE._check_backbone(cfg, BF, 8 * TE.F16_TOL, tasks=(0, 1), follow_routing=True, cv_tol=2e-2)
