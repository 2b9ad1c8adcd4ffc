"""GPU: the 256 x 256-tile NT GEMM for long contractions (gemm_nt_big_kernel, csrc/gemm_big.hip: the ViT-Base shapes of
BASELINE configs[3] / configs[4] - K = 768 / 3072) against torch fp64 on the same 16-bit-rounded operands: the four epilogue
kinds (plain + bias, GELU + pre-activation output, GELU' of a stored pre-activation, fp32 residual in / out with a per-row
factor) and the generic one, expert gather / token-major scatter with ragged and EMPTY groups, partial row and column tiles,
and agreement with the 128 x 128 kernels on the same call.  The kernel is forced here (m3_gemm_set_big(1)); the default
dispatch only gives it launches with enough tiles to fill the chip."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}
DTYPES = [torch.float16, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from m3vit_amd import ops as _ops
    _ops.gemm_set_big(1)
    yield _ops
    _ops.gemm_set_big(-1)


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a = a.double().flatten().cpu(); b = b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def gelu64(x):
    return 0.5 * x * (1 + torch.erf(x / 2 ** 0.5))


def gelu_grad64(x):
    return 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * np.pi) ** 0.5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(1000, 768, 768), (256, 256, 512), (513, 2304, 768), (300, 768, 3072), (77, 384, 1536),
                                   (2049, 264, 640)])
def test_big_dense_plain_and_bias(ops, dtype, M, N, K):
    """partial last row tile, partial last column tile (N = 384, 264), K = 8 .. 48 slices"""
    A, B, bias = rnd(M, K, seed=1, dtype=dtype), rnd(N, K, scale=0.05, seed=2, dtype=dtype), rnd(N, seed=3, dtype=torch.float32)
    ref = A.double() @ B.double().t() + bias.double()
    for c_dtype in (dtype, torch.float32):
        C = torch.full((M, N), float("nan"), dtype=c_dtype, device=dev())
        ops.gemm_nt(A.to(dev()), B.to(dev()), C, bias=bias.to(dev()))
        assert torch.isfinite(C).all()
        assert rel(C, ref) < TOL[dtype]


def test_big_agrees_with_the_128_tile_kernel(ops):
    M, N, K = 1333, 1024, 768
    A, B = rnd(M, K, seed=4).to(dev()), rnd(N, K, scale=0.05, seed=5).to(dev())
    C1 = torch.empty(M, N, dtype=torch.float32, device=dev())
    C0 = torch.empty_like(C1)
    ops.gemm_nt(A, B, C1)
    ops.gemm_set_big(0)
    try:
        ops.gemm_nt(A, B, C0)
    finally:
        ops.gemm_set_big(1)
    assert rel(C1, C0) < 2e-6          # same products, fp32 sums in a different order


@pytest.mark.parametrize("dtype", DTYPES)
def test_big_fc1_epilogue_gelu_and_pre(ops, dtype):
    M, N, K = 1100, 3072, 768
    A, B, bias = rnd(M, K, seed=4, dtype=dtype), rnd(N, K, scale=0.05, seed=5, dtype=dtype), rnd(N, scale=0.1, seed=6, dtype=torch.float32)
    C = torch.empty(M, N, dtype=dtype, device=dev())
    pre = torch.empty(M, N, dtype=dtype, device=dev())
    ops.gemm_nt(A.to(dev()), B.to(dev()), C, bias=bias.to(dev()), act=ops.M3_ACT_GELU, pre_out=pre)
    p64 = A.double() @ B.double().t() + bias.double()
    assert rel(pre, p64) < TOL[dtype] and rel(C, gelu64(p64)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_big_dgrad_epilogue_gelu_grad(ops, dtype):
    """d pre = (dy W2) * GELU'(pre): dense FC2 input gradient (K = 768 -> N = 3072), partial last tile; and the generic
    epilogue (GELU' + fp32 residual + per-row factor in one call)"""
    M, N, K = 1201, 3072, 768
    A, B, gp = rnd(M, K, seed=17, dtype=dtype), rnd(N, K, scale=0.05, seed=18, dtype=dtype), rnd(M, N, seed=19, dtype=dtype)
    C = torch.full((M, N), float("nan"), dtype=dtype, device=dev())
    ops.gemm_nt(A.to(dev()), B.to(dev()), C, gelu_grad_pre=gp.to(dev()))
    ref = (A.double() @ B.double().t()) * gelu_grad64(gp.double())
    assert torch.isfinite(C).all() and rel(C, ref) < TOL[dtype]
    div = 1201
    res = rnd(M, N, seed=10, dtype=torch.float32)
    sc = torch.tensor([1.5])
    C2 = torch.empty(M, N, dtype=torch.float32, device=dev())
    ops.gemm_nt(A.to(dev()), B.to(dev()), C2, gelu_grad_pre=gp.to(dev()), residual=res.to(dev()), row_scale=sc.to(dev()),
                row_scale_div=div)
    assert rel(C2, ref * 1.5 + res.double()) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_big_proj_epilogue_fp32_residual_and_droppath_factor(ops, dtype):
    """x1 = x + s[image] * (o Wp + b): fp32 residual stream in and out (in place, as the engine calls it, and out of place)"""
    M, N, K, div = 1970, 768, 768, 197
    A, B, bias = rnd(M, K, seed=20, dtype=dtype), rnd(N, K, scale=0.05, seed=21, dtype=dtype), rnd(N, scale=0.1, seed=22, dtype=torch.float32)
    res = rnd(M, N, seed=23, dtype=torch.float32)
    sc = torch.tensor([0.0, 2.0, 1.0, 0.0, 2.0, 2.0, 1.0, 0.5, 0.0, 1.5])
    ref = (A.double() @ B.double().t() + bias.double()) * sc.double().repeat_interleave(div)[:, None] + res.double()
    C = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev())
    ops.gemm_nt(A.to(dev()), B.to(dev()), C, bias=bias.to(dev()), residual=res.to(dev()), row_scale=sc.to(dev()), row_scale_div=div)
    assert rel(C, ref) < TOL[dtype]
    C2 = res.to(dev()).clone()
    ops.gemm_nt(A.to(dev()), B.to(dev()), C2, bias=bias.to(dev()), residual=C2, row_scale=sc.to(dev()), row_scale_div=div)
    assert rel(C2, ref) < TOL[dtype]


def _route(ops, T, E, k, seed, skip=None):
    g = torch.Generator().manual_seed(seed)
    choices = torch.tensor([e for e in range(E) if e != skip])
    idx = torch.stack([choices[torch.randperm(len(choices), generator=g)[:k]] for _ in range(T)])
    r = ops.route_build(idx.to(torch.int32).to(dev()), E)
    off = r.offsets.cpu().tolist()
    return r, r.row_of_slot.cpu().long(), off


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("E,k,T,D,H", [(16, 4, 1201, 768, 3072), (4, 2, 1500, 768, 768), (64, 4, 788, 768, 768)])
def test_big_grouped_gather_scatter(ops, dtype, E, k, T, D, H):
    """expert FC1 form (gather by row_of_slot / k, bias, GELU, pre-activation) and FC2 form (contraction over H, scatter to
    token-major rows, bias) with ragged groups and an EMPTY expert"""
    r, ros, off = _route(ops, T, E, k, seed=31, skip=2)
    R = T * k
    x = rnd(T, D, seed=32, dtype=dtype)
    w1, b1 = rnd(E, H, D, scale=0.05, seed=33, dtype=dtype), rnd(E, H, scale=0.1, seed=34, dtype=torch.float32)
    w2, b2 = rnd(E, D, H, scale=0.05, seed=35, dtype=dtype), rnd(E, D, scale=0.1, seed=36, dtype=torch.float32)
    hid = torch.full((R, H), float("nan"), dtype=dtype, device=dev())
    pre = torch.full((R, H), float("nan"), dtype=dtype, device=dev())
    ops.gemm_nt(x.to(dev()), w1.to(dev()), hid, M=R, bias=b1.to(dev()), act=ops.M3_ACT_GELU, pre_out=pre, a_row_idx=r.row_of_slot,
                a_row_div=k, group_offsets=r.offsets, tile_starts=r.tile_starts)
    p64 = torch.empty(R, H, dtype=torch.float64)
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        p64[sl] = x.double()[ros[sl] // k] @ w1.double()[e].t() + b1.double()[e]
    assert torch.isfinite(hid).all() and torch.isfinite(pre).all()
    assert rel(pre, p64) < TOL[dtype] and rel(hid, gelu64(p64)) < TOL[dtype]
    y = torch.full((R, D), float("nan"), dtype=dtype, device=dev())
    ops.gemm_nt(hid, w2.to(dev()), y, M=R, bias=b2.to(dev()), c_row_idx=r.row_of_slot, group_offsets=r.offsets,
                tile_starts=r.tile_starts)
    h64 = hid.double().cpu()
    y64 = torch.empty(R, D, dtype=torch.float64)
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        y64[ros[sl]] = h64[sl] @ w2.double()[e].t() + b2.double()[e]
    assert torch.isfinite(y).all() and rel(y, y64) < TOL[dtype]


def test_big_grouped_dgrad_with_the_combine_gradient_unmaterialised(ops):
    """expert FC2 input gradient as the engine calls it: rows gathered from d out [T, D] through row_of_slot / k, scaled by
    the routed row's gate score (row_scale through row_scale_idx), times GELU'(pre)"""
    E, k, T, D, H = 16, 4, 900, 768, 3072
    r, ros, off = _route(ops, T, E, k, seed=41)
    R = T * k
    dout = rnd(T, D, seed=42)
    w2t = rnd(E, H, D, scale=0.05, seed=43)                 # [E, H, D]: the transposed operand copy of W2 [E, D, H]
    pre = rnd(R, H, seed=44)
    score = torch.rand(T * k, generator=torch.Generator().manual_seed(45))
    out = torch.full((R, H), float("nan"), dtype=torch.float16, device=dev())
    ops.gemm_nt(dout.to(dev()), w2t.to(dev()), out, M=R, gelu_grad_pre=pre.to(dev()), a_row_idx=r.row_of_slot, a_row_div=k,
                group_offsets=r.offsets, tile_starts=r.tile_starts, row_scale=score.to(dev()), row_scale_idx=r.row_of_slot)
    ref = torch.empty(R, H, dtype=torch.float64)
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        ref[sl] = (dout.double()[ros[sl] // k] * score.double()[ros[sl]][:, None]) @ w2t.double()[e].t()
    ref = ref * gelu_grad64(pre.double())
    assert torch.isfinite(out).all() and rel(out, ref) < 1e-3
