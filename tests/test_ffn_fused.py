"""GPU parity tests of the fused FFN kernels (m3_ffn_fwd / m3_ffn_bwd, csrc/ffn.hip, ffn_bwd.hip) through the C ABI:
  * against a torch fp64 evaluation of  GELU(x W1^T + b1) W2^T + b2  (models/moe/ckpt/custom_moe_layer.py:36-44,
    vision_transformer_moe.py:255-261) on the same fp16-rounded inputs - tolerance 1e-3 relative L2 (north_star);
  * against the oracle's routed expert FFN (oracle/ref_torch.py::moe_dispatch_ffn) for the grouped call with the row
    gather / token-major scatter fused in (ragged groups, an empty expert, a partial last tile);
  * against the unfused HIP path (two m3_gemm_nt launches), which the fp32 mode keeps using."""
import pytest
import torch

# EXPERIMENTAL kernel (csrc/Makefile: `make EXPERIMENTAL=1`): not in the default build, the engine never takes it.  The marker
# lets `-m "gpu and not experimental"` leave these out; in a default build they skip themselves.
pytestmark = [pytest.mark.gpu, pytest.mark.experimental]


@pytest.fixture(scope="module", autouse=True)
def _experimental_build():
    from m3vit_amd import _lib
    if not _lib.lib().m3_experimental():
        pytest.skip("library built without EXPERIMENTAL=1")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from m3vit_amd import ops as _ops
    return _ops


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a = a.double().flatten().cpu()
    b = b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def gelu64(x):
    return 0.5 * x * (1 + torch.erf(x / 2 ** 0.5))


def perm32(n):
    """source column of position p inside every aligned group of 32 (include/m3vit_hip.h M3_CAST_PERM32)"""
    p = torch.arange(n)
    w = p % 32
    return (p - w) + 16 * ((w & 7) >> 2) + 4 * (w >> 3) + (w & 3)


def test_cast_perm32_plain_and_transposed(ops):
    G, R, C = 3, 64, 96
    src = rnd(G, R, C, seed=1).to(dev())
    dst = torch.empty(G, R, C, dtype=torch.float16, device=dev())
    dst_t = torch.empty(G, C, R, dtype=torch.float16, device=dev())
    from m3vit_amd import _lib
    ops.CastPlan([(src, dst, dst_t, _lib.M3_CAST_PERM32 | _lib.M3_CAST_PERM32_T)], torch.float16).run()
    torch.cuda.synchronize()
    assert torch.equal(dst.cpu(), src.cpu().half()[..., perm32(C)])
    assert torch.equal(dst_t.cpu(), src.cpu().half().transpose(1, 2)[..., perm32(R)])
    # no flags: the plain copies
    ops.CastPlan([(src, dst, dst_t)], torch.float16).run()
    torch.cuda.synchronize()
    assert torch.equal(dst.cpu(), src.cpu().half())
    assert torch.equal(dst_t.cpu(), src.cpu().half().transpose(1, 2))


@pytest.mark.parametrize("T,D,H,out_f32", [(300, 384, 1536, True), (128, 384, 384, False), (1000, 384, 384, True),
                                           (77, 384, 64, False), (200, 768, 768, False), (130, 768, 3072, True)])
def test_ffn_fwd_dense(ops, T, D, H, out_f32):
    x = rnd(T, D, seed=1, dtype=torch.float16)
    w1, b1 = rnd(H, D, scale=0.05, seed=2, dtype=torch.float16), rnd(H, scale=0.1, seed=3)
    w2, b2 = rnd(D, H, scale=0.05, seed=4, dtype=torch.float16), rnd(D, scale=0.1, seed=5)
    res = rnd(T, D, seed=6) if out_f32 else None
    y = torch.full((T, D), float("nan"), dtype=torch.float32 if out_f32 else torch.float16, device=dev())
    w2p = w2[:, perm32(H)].contiguous()
    pre_o = torch.full((T, H), float("nan"), dtype=torch.float16, device=dev())
    act_o = torch.full((T, H), float("nan"), dtype=torch.float16, device=dev())
    ops.ffn_fwd(x.to(dev()), w1.to(dev()), w2p.to(dev()), y, b1=b1.to(dev()), b2=b2.to(dev()),
                residual=None if res is None else res.to(dev()), pre_out=pre_o, act_out=act_o)
    pre = x.double() @ w1.double().t() + b1.double()
    assert rel(pre_o, pre) < 1e-3 and rel(act_o, gelu64(pre)) < 1e-3        # the optional hidden-activation outputs
    hid = gelu64(pre)
    ref = hid @ w2.double().t() + b2.double()
    if res is not None:
        ref = ref + res.double()
    assert torch.isfinite(y).all()
    assert rel(y, ref) < 1e-3


def test_ffn_fwd_no_bias_matches_unfused(ops):
    T, D, H = 260, 384, 384
    x = rnd(T, D, seed=1, dtype=torch.float16).to(dev())
    w1 = rnd(H, D, scale=0.05, seed=2, dtype=torch.float16).to(dev())
    w2 = rnd(D, H, scale=0.05, seed=4, dtype=torch.float16).to(dev())
    y = torch.empty(T, D, dtype=torch.float16, device=dev())
    ops.ffn_fwd(x, w1, w2[:, perm32(H).to(dev())].contiguous(), y)
    hid = torch.empty(T, H, dtype=torch.float16, device=dev())
    y2 = torch.empty(T, D, dtype=torch.float16, device=dev())
    ops.gemm_nt(x, w1, hid, act=ops.M3_ACT_GELU)
    ops.gemm_nt(hid, w2, y2)
    assert rel(y, y2) < 1e-3


@pytest.mark.parametrize("D,H,E,k,T", [(384, 384, 16, 4, 394), (384, 384, 4, 2, 1500), (768, 768, 8, 2, 300)])
def test_ffn_fwd_grouped_gather_scatter_matches_oracle(ops, D, H, E, k, T):
    from oracle import ref_torch as R
    g = torch.Generator().manual_seed(7)
    x = rnd(T, D, seed=1, dtype=torch.float16)
    # ragged routing with an EMPTY expert (expert 1 never chosen) and distinct experts per token
    choices = torch.tensor([e for e in range(E) if e != 1])
    idx = torch.stack([choices[torch.randperm(E - 1, generator=g)[:k]] for _ in range(T)])
    w1, b1 = rnd(E, H, D, scale=0.05, seed=2, dtype=torch.float16), rnd(E, H, scale=0.1, seed=3)
    w2, b2 = rnd(E, D, H, scale=0.05, seed=4, dtype=torch.float16), rnd(E, D, scale=0.1, seed=5)
    r = ops.route_build(idx.to(torch.int32).to(dev()), E)
    y = torch.full((T * k, D), float("nan"), dtype=torch.float16, device=dev())
    act_o = torch.zeros(T * k, H, dtype=torch.float16, device=dev())
    ops.ffn_fwd(x.to(dev()), w1.to(dev()), w2[..., perm32(H)].contiguous().to(dev()), y, b1=b1.to(dev()), b2=b2.to(dev()),
                M=T * k, x_row_idx=r.row_of_slot, x_row_div=k, y_row_idx=r.row_of_slot, group_offsets=r.offsets,
                act_out=act_o)
    # act_out rows are expert-major slots: the unfused FC1 writes the same layout
    hid_u = torch.empty(T * k, H, dtype=torch.float16, device=dev())
    ops.gemm_nt(x.to(dev()), w1.to(dev()), hid_u, M=T * k, bias=b1.to(dev()), act=ops.M3_ACT_GELU, a_row_idx=r.row_of_slot,
                a_row_div=k, group_offsets=r.offsets, tile_starts=r.tile_starts)
    assert rel(act_o, hid_u) < 1e-3
    ref = R.moe_dispatch_ffn(x.double(), idx, w1.double(), b1.double(), w2.double(), b2.double())   # [T*k, D] token-major
    assert torch.isfinite(y).all()
    assert rel(y, ref) < 1e-3


def test_ffn_fwd_rejects_bad_shapes(ops):
    from m3vit_amd._lib import M3Error
    x = torch.zeros(8, 64, dtype=torch.float16, device=dev())
    w1 = torch.zeros(64, 64, dtype=torch.float16, device=dev())
    y = torch.zeros(8, 64, dtype=torch.float16, device=dev())
    with pytest.raises(M3Error):
        ops.ffn_fwd(x, w1, w1, y)
    assert not ops.ffn_supported(64, 64, torch.float16)
    assert not ops.ffn_supported(384, 384, torch.float32)
    assert ops.ffn_supported(384, 1536, torch.float16) and ops.ffn_supported(768, 3072, torch.float16)
