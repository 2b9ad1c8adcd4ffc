"""GPU: the LDS-DMA weight-gradient kernel (wgrad_dma_kernel, csrc/wgrad.hip: fp16 / bf16 / fp32 operands, no per-row factor) against
torch fp64 on the same rounded operands and against the register-staged kernel on the same call: dense with row splits,
ragged unit tails (rows past a unit's end read the zero row), partial column tiles, grouped with ragged and EMPTY experts,
gathered dC rows (expert FC2: token-major d y), gathered A rows with a power-of-two divisor (expert FC1: tokens through
row_of_slot / k), the fused bias gradient (v_dot2 column sums), accumulation (beta = 1), and the queue's ride-along reduce."""
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
    _ops.wgrad_set_dma(2)
    yield _ops
    _ops.wgrad_set_dma(-1)


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a = a.double().flatten().cpu(); b = b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,splits", [(2000, 384, 256, None), (64, 128, 128, 1), (65, 136, 72, 1), (1, 8, 8, 1), (4097, 768, 384, 7),
                                          (25216, 384, 384, None), (1201, 3072, 768, None), (130, 200, 264, 3)])
def test_dense_against_fp64_and_the_register_staged_kernel(ops, dtype, M, N, K, splits):
    dC, A = rnd(M, N, dtype=dtype, seed=21), rnd(M, K, dtype=dtype, seed=22)
    base = rnd(N, K, seed=23, dtype=torch.float32)
    ref = dC.double().t() @ A.double()
    dW = base.clone(); db = torch.zeros(N, device=dev())
    ops.wgrad_tn(dC, A, dW, beta=1, db=db, beta_db=0, splits=splits)
    assert rel(dW - base, ref) < TOL[dtype]
    assert rel(db, dC.double().sum(0)) < max(TOL[dtype], 1e-4)
    ops.wgrad_set_dma(0)
    try:
        dW0 = base.clone(); db0 = torch.zeros(N, device=dev())
        ops.wgrad_tn(dC, A, dW0, beta=1, db=db0, beta_db=0, splits=splits)
    finally:
        ops.wgrad_set_dma(2)
    assert rel(dW - base, dW0 - base) < 2e-5 and rel(db, db0) < 2e-5        # same products, fp32 sums in another order


def _route(ops, T, E, k, seed, skip=None):
    g = torch.Generator().manual_seed(seed)
    choices = torch.tensor([e for e in range(E) if e != skip])
    idx = torch.stack([choices[torch.randperm(len(choices), generator=g)[:k]] for _ in range(T)])
    r = ops.route_build(idx.to(torch.int32).to(dev()), E)
    return r, r.row_of_slot.cpu().long(), r.offsets.cpu().tolist()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("E,k,T,D,H", [(16, 4, 1576, 384, 384), (4, 2, 700, 768, 3072), (64, 4, 333, 768, 768), (5, 2, 129, 72, 136)])
def test_grouped_expert_weight_gradients(ops, dtype, E, k, T, D, H):
    """expert FC1: dW1[e] = dhp[e-rows]^T x[row_of_slot / k] (gathered A) + bias; expert FC2: dW2[e] = dy[row_of_slot]^T hid
    (gathered dC: token-major rows) + bias; ragged groups, one EMPTY expert"""
    r, ros, off = _route(ops, T, E, k, seed=31, skip=1)
    R = T * k
    x = rnd(T, D, dtype=dtype, seed=32)
    dhp = rnd(R, H, dtype=dtype, seed=33)
    hid = rnd(R, H, dtype=dtype, seed=34)
    dy = rnd(R, D, dtype=dtype, seed=35)                     # token-major: row t * k + j
    dW1 = torch.zeros(E, H, D, device=dev()); db1 = torch.zeros(E, H, device=dev())
    ops.wgrad_tn(dhp, x, dW1, M=R, a_row_idx=r.row_of_slot, a_row_div=k, group_offsets=r.offsets, db=db1)
    dW2 = torch.zeros(E, D, H, device=dev()); db2 = torch.zeros(E, D, device=dev())
    ops.wgrad_tn(dy, hid, dW2, M=R, c_row_idx=r.row_of_slot, group_offsets=r.offsets, db=db2)
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        if off[e] == off[e + 1]:
            assert float(dW1[e].abs().max()) == 0.0 and float(dW2[e].abs().max()) == 0.0 and float(db1[e].abs().max()) == 0.0
            continue
        xs = x.double().cpu()[ros[sl] // k]
        assert rel(dW1[e], dhp.double().cpu()[sl].t() @ xs) < TOL[dtype], e
        assert rel(db1[e], dhp.double().cpu()[sl].sum(0)) < max(TOL[dtype], 1e-4), e
        dys = dy.double().cpu()[ros[sl]]
        assert rel(dW2[e], dys.t() @ hid.double().cpu()[sl]) < TOL[dtype], e
        assert rel(db2[e], dys.sum(0)) < max(TOL[dtype], 1e-4), e


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("E,k,T,D,H", [(16, 4, 1576, 384, 384), (8, 2, 333, 136, 264)])
def test_expert_fc2_weight_gradient_through_the_gate_score(ops, dtype, E, k, T, D, H):
    """the combine's backward without d y (custom_moe_layer.py:298-305): dW2[e] = sum_slot score[slot] * d out[slot / k]^T hid[slot]
    - the per-row factor travels through a small LDS table and multiplies the dC fragments (fp16: v_pk_mul_f16 with the factor
    rounded to fp16) - against fp64 and against the register-staged kernel (fp32 product, one rounding)"""
    r, ros, off = _route(ops, T, E, k, seed=81)
    R = T * k
    dout, hid = rnd(T, D, dtype=dtype, seed=82), rnd(R, H, dtype=dtype, seed=83)
    score = torch.rand(R, generator=torch.Generator().manual_seed(84)).to(dev())

    def run():
        dW, db = torch.zeros(E, D, H, device=dev()), torch.zeros(E, D, device=dev())
        ops.wgrad_tn(dout, hid, dW, M=R, c_row_idx=r.row_of_slot, c_row_div=k, c_row_scale=score, group_offsets=r.offsets, db=db)
        return dW, db
    dW, db = run()
    ops.wgrad_set_dma(0)
    try:
        dW0, db0 = run()
    finally:
        ops.wgrad_set_dma(2)
    tol = TOL[dtype]
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        rows = dout.double().cpu()[ros[sl] // k] * score.double().cpu()[ros[sl]][:, None]
        assert rel(dW[e], rows.t() @ hid.double().cpu()[sl]) < tol, e
        assert rel(db[e], rows.sum(0)) < max(tol, 1e-4), e
    assert rel(dW, dW0) < (1e-3 if dtype == torch.float16 else 2e-5) and rel(db, db0) < (1e-3 if dtype == torch.float16 else 2e-5)


def test_calls_the_dma_kernel_does_not_take_still_work(ops):
    """a divisor that is not a power of two (top-k = 3) goes to the register-staged kernel, with and without a per-row factor"""
    E, k, T, D, H = 6, 3, 300, 64, 96
    r, ros, off = _route(ops, T, E, k, seed=41)
    R = T * k
    x, dhp = rnd(T, D, seed=42), rnd(R, H, seed=43)
    dW = torch.zeros(E, H, D, device=dev())
    ops.wgrad_tn(dhp, x, dW, M=R, a_row_idx=r.row_of_slot, a_row_div=k, group_offsets=r.offsets)        # k = 3
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        assert rel(dW[e], dhp.double().cpu()[sl].t() @ x.double().cpu()[ros[sl] // k]) < 1e-3
    dout, hid = rnd(T, D, seed=44), rnd(R, H, seed=45)
    score = torch.rand(R, generator=torch.Generator().manual_seed(46)).to(dev())
    dW2 = torch.zeros(E, D, H, device=dev())
    ops.wgrad_tn(dout, hid, dW2, M=R, c_row_idx=r.row_of_slot, c_row_div=k, c_row_scale=score, group_offsets=r.offsets)
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        rows = dout.double().cpu()[ros[sl] // k] * score.double().cpu()[ros[sl]][:, None]
        assert rel(dW2[e], rows.t() @ hid.double().cpu()[sl]) < 1e-3


def test_queue_ride_along_reduce_with_the_dma_kernel(ops):
    M, N, K = 3000, 384, 384
    q = ops.WgradQueue(2 * 32 * N * (K + 1), dev())
    outs, refs = [], []
    for i in range(3):
        dC, A = rnd(M, N, seed=50 + i), rnd(M, K, seed=60 + i)
        dW = torch.zeros(N, K, device=dev()); db = torch.zeros(N, device=dev())
        ops.wgrad_tn(dC, A, dW, db=db, queue=q)
        outs.append((dW, db)); refs.append((dC.double().t() @ A.double(), dC.double().sum(0)))
    q.flush()
    for (dW, db), (rw, rb) in zip(outs, refs):
        assert rel(dW, rw) < 1e-3 and rel(db, rb) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("beta", [0, 1])
def test_direct_mode_accumulates_into_dw_without_slabs(ops, dtype, beta):
    """splits == 1: every (group, tile) belongs to one workgroup, which read-add-writes dW (and db) itself - against fp64 and
    against the slab + reduce path on the same call (both kernels share the epilogue: fp16 takes the LDS-DMA kernel, fp32 the
    register-staged one); ragged groups, an EMPTY expert (its dW must stay / become what beta says), a per-row factor"""
    E, k, T, D, H = 8, 2, 500, 136, 264
    r, ros, off = _route(ops, T, E, k, seed=71, skip=3)
    R = T * k
    x, dhp = rnd(T, D, dtype=dtype, seed=72), rnd(R, H, dtype=dtype, seed=73)
    base, bbase = rnd(E, H, D, seed=74, dtype=torch.float32), rnd(E, H, seed=75, dtype=torch.float32)
    tol = 1e-3 if dtype == torch.float16 else 2e-5

    def run(direct):
        import m3vit_amd.ops as O
        keep = O._WGRAD_DIRECT
        O._WGRAD_DIRECT = direct
        try:
            dW, db = base.clone(), bbase.clone()
            ops.wgrad_tn(dhp, x, dW, M=R, beta=beta, splits=1, a_row_idx=r.row_of_slot, a_row_div=k, group_offsets=r.offsets, db=db)
            return dW, db
        finally:
            O._WGRAD_DIRECT = keep
    dW, db = run(True)
    dW0, db0 = run(False)
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        want = dhp.double().cpu()[sl].t() @ x.double().cpu()[ros[sl] // k] + beta * base[e].double().cpu()
        wantb = dhp.double().cpu()[sl].sum(0) + beta * bbase[e].double().cpu()
        assert rel(dW[e], want) < tol and rel(db[e], wantb) < max(tol, 1e-4), e
    assert rel(dW, dW0) < 1e-6 and rel(db, db0) < 1e-6
    # with the combine's gate score on the rows (register-staged kernel, SC variant) and a queue in front
    dout, hid = rnd(T, D, dtype=dtype, seed=76), rnd(R, H, dtype=dtype, seed=77)
    score = torch.rand(R, generator=torch.Generator().manual_seed(78)).to(dev())
    q = ops.WgradQueue(2 * E * D * (H + 1) + 64, dev())
    dWa = torch.zeros(E, D, H, device=dev())
    ops.wgrad_tn(dout, hid, dWa, M=R, splits=2, c_row_idx=r.row_of_slot, c_row_div=k, c_row_scale=score, group_offsets=r.offsets, queue=q)
    dWb = rnd(E, D, H, seed=79, dtype=torch.float32); b0 = dWb.clone()
    ops.wgrad_tn(dout, hid, dWb, M=R, beta=1, splits=1, c_row_idx=r.row_of_slot, c_row_div=k, c_row_scale=score, group_offsets=r.offsets, queue=q)
    q.flush()
    for e in range(E):
        sl = slice(off[e], off[e + 1])
        rows = dout.double().cpu()[ros[sl] // k] * score.double().cpu()[ros[sl]][:, None]
        want = rows.t() @ hid.double().cpu()[sl]
        assert rel(dWa[e], want) < tol and rel(dWb[e] - b0[e], want) < 2 * tol, e
