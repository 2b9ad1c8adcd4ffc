"""GPU: the wide-tile weight-gradient kernel (wgrad_wide_kernel in csrc/wgrad.hip: fp16, 128 x 384 tiles for K = 384,
384 x 128 for N = 384) against torch fp64 on the same fp16-rounded operands - dense and grouped, gathered operands on
either side, fused bias gradient, accumulation into dW (beta = 1), explicit splits, ragged / empty / hot experts, row
counts that are not multiples of the 32-row step, short contractions (1-3 steps per split: the peeled loop tails).
Mirrors what the reference gets from autograd for FMoELinear / nn.Linear weights
(models/moe/ckpt/custom_moe_layer.py:32-33, vision_transformer_moe.py:255-261)."""
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
TOL = 2e-3


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from m3vit_amd import ops as _ops
    _ops.wgrad_set_wide(1)            # opt-in kernel: on for this module only
    yield _ops
    _ops.wgrad_set_wide(0)


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a = a.double().flatten().cpu(); b = b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def test_tile_choice(ops):
    h, f = torch.float16, torch.float32
    assert ops.wgrad_tile(1536, 384, h) == (128, 384) and ops.wgrad_tile(1152, 384, h) == (128, 384)
    assert ops.wgrad_tile(384, 1536, h) == (384, 128) and ops.wgrad_tile(384, 768, h) == (384, 128)
    assert ops.wgrad_tile(384, 384, h) == (128, 128) and ops.wgrad_tile(1536, 384, f) == (128, 128)
    assert ops.wgrad_tile(1000, 384, h) == (128, 128)


@pytest.mark.parametrize("M,N,K,splits", [(25216, 1536, 384, None), (3001, 1152, 384, None), (1000, 768, 384, 7),
                                          (25216, 384, 1536, None), (777, 384, 768, 5), (40, 1536, 384, 1),
                                          (96, 384, 1536, 3), (64, 768, 384, 2)])
def test_wide_dense_with_bias_and_accumulate(ops, M, N, K, splits):
    dC, A = rnd(M, N, scale=0.5, seed=1), rnd(M, K, scale=0.5, seed=2)
    dW0, db0 = rnd(N, K, seed=3, dtype=torch.float32), rnd(N, seed=4, dtype=torch.float32)
    dW, db = dW0.to(dev()), db0.to(dev())
    ops.wgrad_tn(dC.to(dev()), A.to(dev()), dW, beta=1, db=db, splits=splits)
    assert rel(dW, dW0.double() + dC.double().t() @ A.double()) < TOL
    assert rel(db, db0.double() + dC.double().sum(0)) < TOL
    dW2 = torch.full((N, K), float("nan"), device=dev())
    ops.wgrad_tn(dC.to(dev()), A.to(dev()), dW2, splits=splits)               # beta = 0 overwrites, no bias
    assert rel(dW2, dC.double().t() @ A.double()) < TOL


@pytest.mark.parametrize("E,k,T,hot", [(16, 4, 1576, False), (4, 2, 3000, True), (8, 4, 40, False)])
def test_wide_grouped_expert_weights(ops, E, k, T, hot):
    """expert FC1 weights (dpre^T x[row_of_slot / k], N = 1536, K = 384: the A side gathered) and FC2 weights
    (dy[row_of_slot]^T hid, N = 384, K = 1536: the dC side gathered), bias gradients fused; one expert empty, and
    with `hot` one expert taking half the tokens (the balanced unit split gives it more workgroups)"""
    D, H = 384, 1536
    g = torch.Generator().manual_seed(5)
    choices = torch.tensor([e for e in range(E) if e != 1])
    idx = torch.stack([choices[torch.randperm(E - 1, generator=g)[:k]] for _ in range(T)])
    if hot:
        idx[: T // 2, 0] = 0
        idx[: T // 2, 1] = 2
    r = ops.route_build(idx.to(torch.int32).to(dev()), E)
    R = T * k
    ros = r.row_of_slot.cpu().long()
    off = r.offsets.cpu().tolist()
    x, dpre = rnd(T, D, scale=0.5, seed=6), rnd(R, H, scale=0.5, seed=7)
    hid, dy = rnd(R, H, scale=0.5, seed=8), rnd(R, D, scale=0.5, seed=9)
    dw1, db1 = torch.zeros(E, H, D, device=dev()), torch.zeros(E, H, device=dev())
    ops.wgrad_tn(dpre.to(dev()), x.to(dev()), dw1, M=R, a_row_idx=r.row_of_slot, a_row_div=k, group_offsets=r.offsets, db=db1)
    dw2, db2 = torch.zeros(E, D, H, device=dev()), torch.zeros(E, D, device=dev())
    ops.wgrad_tn(dy.to(dev()), hid.to(dev()), dw2, M=R, c_row_idx=r.row_of_slot, group_offsets=r.offsets, db=db2)
    w1 = torch.zeros(E, H, D, dtype=torch.float64); b1 = torch.zeros(E, H, dtype=torch.float64)
    w2 = torch.zeros(E, D, H, dtype=torch.float64); b2 = torch.zeros(E, D, dtype=torch.float64)
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        w1[e] = dpre.double()[sl].t() @ x.double()[ros[sl] // k]
        b1[e] = dpre.double()[sl].sum(0)
        w2[e] = dy.double()[ros[sl]].t() @ hid.double()[sl]
        b2[e] = dy.double()[ros[sl]].sum(0)
    assert rel(dw1, w1) < TOL and rel(db1, b1) < TOL
    assert rel(dw2, w2) < TOL and rel(db2, b2) < TOL
    assert float(dw1[1].abs().max()) == 0.0 and float(dw2[1].abs().max()) == 0.0          # the empty expert


def test_wide_and_square_tiles_agree(ops, monkeypatch):
    """same call through both tile shapes (explicit splits so that only the kernel differs): fp32 sums of the same
    products in a different order"""
    M, N, K = 5000, 1536, 384
    dC, A = rnd(M, N, scale=0.5, seed=10).to(dev()), rnd(M, K, scale=0.5, seed=11).to(dev())
    wide = torch.zeros(N, K, device=dev())
    ops.wgrad_tn(dC, A, wide, splits=4)
    # a [M, 1544]-wide view is not a multiple of 128 -> the square tiles take it; compare the common block
    dCp = torch.zeros(M, N + 8, dtype=torch.float16, device=dev()); dCp[:, :N] = dC
    sq = torch.zeros(N + 8, K, device=dev())
    ops.wgrad_tn(dCp, A, sq, splits=4)
    assert rel(wide, sq[:N]) < 1e-5
