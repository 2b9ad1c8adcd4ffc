"""GPU: the streaming weight-gradient kernel for K = 16 / 32 (wgrad_skinny_kernel, csrc/wgrad.hip - the router's weight,
dW_gate = h^T d_logits, custom_moe_layer.py:213-217) against torch fp64 on the same rounded operands: odd row counts, a
partial column tile, explicit and default row splits, accumulation, the queue's ride-along reduce, and that calls the kernel
does not cover (a bias, K = 24, gathers) still go the 128 x 128 way with the same results."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3, torch.float32: 2e-5}
DTYPES = [torch.float16, torch.bfloat16, torch.float32]


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from m3vit_amd import ops as _ops
    return _ops


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a = a.double().flatten().cpu(); b = b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,splits", [(25216, 384, 16, None), (9608, 768, 16, None), (1, 384, 16, 1), (37, 200, 16, 3), (1000, 392, 32, None),
                                          (4097, 768, 32, 5), (130, 8, 16, 4), (64, 384, 16, 128)])
def test_skinny_against_fp64(ops, dtype, M, N, K, splits):
    assert ops.wgrad_skinny(N, K, 1)
    dC, A = rnd(M, N, dtype=dtype, seed=31), rnd(M, K, dtype=dtype, seed=32)
    base = rnd(N, K, seed=33, dtype=torch.float32)
    ref = dC.double().t() @ A.double()
    dW = base.clone()
    ops.wgrad_tn(dC, A, dW, beta=1, splits=splits)
    assert rel(dW - base, ref) < TOL[dtype]
    dW2 = torch.full_like(base, float("nan"))
    ops.wgrad_tn(dC, A, dW2, beta=0, splits=splits)
    assert rel(dW2, ref) < TOL[dtype]


def test_skinny_rule_and_calls_it_does_not_take(ops):
    assert not ops.wgrad_skinny(384, 24, 1) and not ops.wgrad_skinny(384, 16, 4) and not ops.wgrad_skinny(385, 16, 1)
    assert ops.default_wgrad_splits(25216, 384, 16, 1, torch.float16) == 256 and ops.default_wgrad_splits(100, 384, 16, 1, torch.float16) == 1
    M, N, K = 3000, 384, 16
    dC, A = rnd(M, N, seed=41), rnd(M, K, seed=42)
    ref = dC.double().t() @ A.double()
    dW, db = torch.zeros(N, K, device=dev()), torch.zeros(N, device=dev())
    ops.wgrad_tn(dC, A, dW, db=db)                                   # a bias: the 128 x 128 kernels
    assert rel(dW, ref) < 1e-3 and rel(db, dC.double().sum(0)) < 1e-3
    idx = torch.randperm(M, generator=torch.Generator().manual_seed(1)).to(torch.int32).to(dev())
    dWg = torch.zeros(N, K, device=dev())
    ops.wgrad_tn(dC, A, dWg, M=M, c_row_idx=idx)                      # gathered dC rows: the 128 x 128 kernels
    assert rel(dWg, dC[idx.long()].double().t() @ A.double()) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_skinny_in_the_queue_between_other_calls(ops, dtype):
    """engine order around the router's gradient: an expert-sized call, the skinny call, a dense call - every reduction
    rides on the next launch, whichever kernel that is"""
    M = 5000
    calls = [(768, 128, True), (384, 16, False), (384, 384, True), (384, 16, False), (256, 32, False), (128, 128, True)]
    need = max(ops.wgrad_ws_elems(M, N, K, 1, grouped=False, bias=b, dtype=dtype) for N, K, b in calls)
    q = ops.WgradQueue(need, dev())
    outs = []
    for i, (N, K, bias) in enumerate(calls):
        dC, A = rnd(M, N, dtype=dtype, seed=50 + i), rnd(M, K, dtype=dtype, seed=70 + i)
        dW = rnd(N, K, seed=90 + i, dtype=torch.float32)
        base = dW.clone()
        db = torch.zeros(N, device=dev()) if bias else None
        ops.wgrad_tn(dC, A, dW, beta=1, db=db, queue=q)
        outs.append((dW, base, dC.double().t() @ A.double(), db, dC.double().sum(0)))
    q.flush()
    torch.cuda.synchronize()
    for dW, base, ref, db, dbref in outs:
        assert rel(dW - base, ref) < TOL[dtype]
        if db is not None:
            assert rel(db, dbref) < max(TOL[dtype], 1e-4)
