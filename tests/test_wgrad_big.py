"""GPU: the 256 x 256 weight-gradient kernel (wgrad_big_kernel, csrc/wgrad.hip: 16-bit ViT-Base shapes) against torch fp64 on
the same rounded operands and against the 128 x 128 kernels on the same call: dense with row splits and ragged tails, a unit
of ONE step, grouped experts with gathered dC / gathered A rows, ragged and EMPTY experts, the gate score on the rows, the
fused bias sums, slabs and direct accumulation, the balanced (chunked) units, and the queue around it."""
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
    _ops.wgrad_set_big(1)
    yield _ops
    _ops.wgrad_set_big(-1)


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a = a.double().flatten().cpu(); b = b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


def small(ops, fn):
    """the same call through the 128 x 128 kernels"""
    ops.wgrad_set_big(0)
    try:
        return fn()
    finally:
        ops.wgrad_set_big(1)


def test_tile_rule(ops):
    h, b, f = torch.float16, torch.bfloat16, torch.float32
    assert ops.wgrad_tile(768, 768, h) == (256, 256) and ops.wgrad_tile(3072, 768, b) == (256, 256) and ops.wgrad_tile(2304, 768, h) == (256, 256)
    assert ops.wgrad_tile(768, 768, f) == (128, 128) and ops.wgrad_tile(384, 768, h) == (128, 128) and ops.wgrad_tile(768, 1152, h) == (128, 128)
    assert ops.default_wgrad_splits(25216, 2304, 768, 1, h) == 8 and ops.default_wgrad_splits(25216, 768, 768, 1, h) == 28    # (9 -> 8: whole parts per XCD)
    assert ops.default_wgrad_splits(100864, 768, 768, 64, h) == 1 and ops.default_wgrad_splits(38432, 3072, 768, 16, h) == 1
    assert small(ops, lambda: ops.wgrad_tile(768, 768, h)) == (128, 128)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,splits", [(2000, 768, 256, None), (64, 256, 256, 1), (65, 256, 512, 1), (1, 256, 256, 1), (4097, 768, 768, 7),
                                          (9608, 768, 3072, None), (1201, 2304, 768, None), (130, 512, 256, 3), (25216, 768, 768, None)])
def test_dense_against_fp64_and_the_small_tiles(ops, dtype, M, N, K, splits):
    dC, A = rnd(M, N, dtype=dtype, seed=21), rnd(M, K, dtype=dtype, seed=22)
    base = rnd(N, K, seed=23, dtype=torch.float32)
    ref = dC.double().t() @ A.double()

    def run():
        dW = base.clone(); db = torch.zeros(N, device=dev())
        ops.wgrad_tn(dC, A, dW, beta=1, db=db, beta_db=0, splits=splits)
        return dW, db
    dW, db = run()
    assert rel(dW - base, ref) < TOL[dtype]
    assert rel(db, dC.double().sum(0)) < max(TOL[dtype], 1e-4)
    dW0, db0 = small(ops, run)
    assert rel(dW - base, dW0 - base) < 2e-5 and rel(db, db0) < 2e-5        # same products, fp32 sums in another order


def _route(ops, T, E, k, seed, skip=None):
    g = torch.Generator().manual_seed(seed)
    choices = torch.tensor([e for e in range(E) if e != skip])
    idx = torch.stack([choices[torch.randperm(len(choices), generator=g)[:k]] for _ in range(T)])
    r = ops.route_build(idx.to(torch.int32).to(dev()), E)
    return r, r.row_of_slot.cpu().long(), r.offsets.cpu().tolist()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("E,k,T,D,H,splits", [(4, 2, 700, 768, 3072, None), (64, 4, 333, 768, 768, None), (5, 2, 129, 256, 512, None),
                                              (8, 4, 900, 256, 256, 2), (3, 2, 1000, 512, 256, 4)])
def test_grouped_expert_weight_gradients(ops, dtype, E, k, T, D, H, splits):
    """expert FC1: gathered A rows (tokens through row_of_slot / k) + bias; expert FC2: gathered dC rows + bias; ragged groups,
    one EMPTY expert; default splits (direct accumulation where the tiles fill the chip) and explicit ones (balanced units)"""
    r, ros, off = _route(ops, T, E, k, seed=31, skip=1)
    R = T * k
    x = rnd(T, D, dtype=dtype, seed=32)
    dhp = rnd(R, H, dtype=dtype, seed=33)
    hid = rnd(R, H, dtype=dtype, seed=34)
    dy = rnd(R, D, dtype=dtype, seed=35)
    b1, b2 = rnd(E, H, D, seed=36, dtype=torch.float32), rnd(E, D, H, seed=37, dtype=torch.float32)

    def run():
        dW1 = b1.clone(); db1 = torch.zeros(E, H, device=dev())
        ops.wgrad_tn(dhp, x, dW1, M=R, beta=1, beta_db=0, a_row_idx=r.row_of_slot, a_row_div=k, group_offsets=r.offsets, db=db1, splits=splits)
        dW2 = b2.clone(); db2 = torch.zeros(E, D, device=dev())
        ops.wgrad_tn(dy, hid, dW2, M=R, beta=1, beta_db=0, c_row_idx=r.row_of_slot, group_offsets=r.offsets, db=db2, splits=splits)
        return dW1, db1, dW2, db2
    dW1, db1, dW2, db2 = run()
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        if off[e] == off[e + 1]:
            assert torch.equal(dW1[e], b1[e]) and torch.equal(dW2[e], b2[e]) and float(db1[e].abs().max()) == 0.0
            continue
        xs = x.double().cpu()[ros[sl] // k]
        assert rel(dW1[e] - b1[e], dhp.double().cpu()[sl].t() @ xs) < TOL[dtype], e
        assert rel(db1[e], dhp.double().cpu()[sl].sum(0)) < max(TOL[dtype], 1e-4), e
        dys = dy.double().cpu()[ros[sl]]
        assert rel(dW2[e] - b2[e], dys.t() @ hid.double().cpu()[sl]) < TOL[dtype], e
        assert rel(db2[e], dys.sum(0)) < max(TOL[dtype], 1e-4), e
    s1, sb1, s2, sb2 = small(ops, run)
    assert rel(dW1 - b1, s1 - b1) < 2e-5 and rel(dW2 - b2, s2 - b2) < 2e-5 and rel(db1, sb1) < 2e-5 and rel(db2, sb2) < 2e-5


@pytest.mark.parametrize("E,k,T,D,H,splits", [(16, 4, 400, 768, 3072, None), (8, 2, 333, 256, 512, None), (4, 2, 2000, 256, 256, 3)])
def test_expert_fc2_weight_gradient_through_the_gate_score(ops, E, k, T, D, H, splits):
    """dW2[e] = sum_slot score[slot] * d out[slot / k]^T hid[slot] (custom_moe_layer.py:298-305 without d y): the factor
    through the LDS table of the stage, fp16"""
    r, ros, off = _route(ops, T, E, k, seed=81)
    R = T * k
    dout, hid = rnd(T, D, seed=82), rnd(R, H, seed=83)
    score = torch.rand(R, generator=torch.Generator().manual_seed(84)).to(dev())

    def run():
        dW, db = torch.zeros(E, D, H, device=dev()), torch.zeros(E, D, device=dev())
        ops.wgrad_tn(dout, hid, dW, M=R, c_row_idx=r.row_of_slot, c_row_div=k, c_row_scale=score, group_offsets=r.offsets, db=db, splits=splits)
        return dW, db
    dW, db = run()
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        rows = dout.double().cpu()[ros[sl] // k] * score.double().cpu()[ros[sl]][:, None]
        assert rel(dW[e], rows.t() @ hid.double().cpu()[sl]) < 1e-3, e
        assert rel(db[e], rows.sum(0)) < 1e-3, e
    dW0, db0 = small(ops, run)
    assert rel(dW, dW0) < 1e-3 and rel(db, db0) < 1e-3


def test_calls_the_big_kernel_does_not_take_still_work(ops):
    """top-k = 3 (a divisor that is not a power of two), a bf16 per-row factor: the 128 x 128 kernels, planned with the big
    tile's splits"""
    E, k, T, D, H = 6, 3, 300, 256, 256
    r, ros, off = _route(ops, T, E, k, seed=41)
    R = T * k
    x, dhp = rnd(T, D, seed=42), rnd(R, H, seed=43)
    dW = torch.zeros(E, H, D, device=dev())
    ops.wgrad_tn(dhp, x, dW, M=R, a_row_idx=r.row_of_slot, a_row_div=k, group_offsets=r.offsets)
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        assert rel(dW[e], dhp.double().cpu()[sl].t() @ x.double().cpu()[ros[sl] // k]) < 1e-3


def test_in_the_queue_between_other_kernels(ops):
    M = 3000
    calls = [(384, 384), (768, 768), (768, 16), (2304, 768), (384, 1536), (256, 256)]
    need = max(ops.wgrad_ws_elems(M, N, K, 1, grouped=False, dtype=torch.float16) for N, K in calls)
    q = ops.WgradQueue(need, dev())
    outs = []
    for i, (N, K) in enumerate(calls):
        dC, A = rnd(M, N, seed=50 + i), rnd(M, K, seed=60 + i)
        dW = rnd(N, K, seed=90 + i, dtype=torch.float32); base = dW.clone()
        db = torch.zeros(N, device=dev()) if K != 16 else None
        ops.wgrad_tn(dC, A, dW, beta=1, db=db, queue=q)
        outs.append((dW, base, dC.double().t() @ A.double(), db, dC.double().sum(0)))
    q.flush()
    for dW, base, ref, db, dbref in outs:
        assert rel(dW - base, ref) < 1e-3
        if db is not None:
            assert rel(db, dbref) < 1e-3
