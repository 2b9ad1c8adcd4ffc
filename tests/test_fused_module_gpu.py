"""GPU: the drop-in module path - m3vit_amd.vit.VisionTransformerMoE.forward(x, task_id) as ONE autograd node on the fused
executor (m3vit_amd/fused.py) - against the float64 oracle, against the per-op module path, and for torch's gradient
semantics (zero_grad both ways, accumulation, foreign .grad tensors, one task at a time vs the joint backward)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().double().cpu().flatten(); b = b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


KW = dict(img_size=(32, 48), embed_dim=64, depth=4, num_heads=2, moe_top_k=2, gate_dim=66, multi_gate=True)


def _model(std=0.0, E=4, fused="auto", act_dtype=torch.float32, seed=9, model_kw=None, **over):
    from m3vit_amd.vit import VisionTransformerMoE
    from oracle import ref_torch as R
    kw = dict(KW, moe_experts=E)
    kw.update(over)
    cfg = R.BackboneCfg(mlp_ratio=4.0, moe_mlp_ratio=1.0, vmoe_noisy_std=std, **kw)
    P = R.init_backbone_params(cfg, seed=seed)
    m = VisionTransformerMoE(mlp_ratio=4.0, moe_mlp_ratio=1, vmoe_noisy_std=std, fused=fused, act_dtype=act_dtype, **kw,
                             **(model_kw or {})).cuda()
    m.load_state_dict(P)
    m.train()
    return m, cfg


@pytest.mark.parametrize("std,act_dtype,tol", [(0.0, torch.float32, 2e-4), (1.0, torch.float32, 2e-4), (0.0, torch.float16, 1e-3),
                                               (0.0, torch.bfloat16, 8e-3)])
def test_fused_module_joint_multitask_steps_match_oracle(std, act_dtype, tol):
    """The reference's joint multi-task step (models/models.py:299-320 + train/train_utils.py:423-457): backbone(x, task_id)
    for every task, ONE loss.backward(), optimizer.zero_grad(set_to_none=True), a parameter update - four steps, so that the
    eager first use, the hipGraph capture (second use) and two replays are all checked, each step with new images and
    changed weights, against the float64 oracle on the module's current weights."""
    _need_gpu()
    from oracle import ref_torch as R
    m, cfg = _model(std=std, E=8 if std else 4, act_dtype=act_dtype)
    f16 = act_dtype in (torch.float16, torch.bfloat16)       # 16-bit storage: follow the module's routing
    B = 3
    for step in range(4):
        g = torch.Generator().manual_seed(100 + step)
        img = torch.randn(B, 3, 32, 48, generator=g)
        dtok = torch.randn(B, cfg.num_tokens, 64, generator=g) * 0.1
        for p in m.parameters():
            p.grad = None                                              # optimizer.zero_grad(set_to_none=True)
        Pr = {k: v.detach().clone().double().cpu().requires_grad_() for k, v in m.state_dict().items()}
        loss, loss_ref = 0.0, 0.0
        outs = []
        for task in (0, 1):
            tok, cv = m(img.cuda(), task_id=task)
            assert m.fused_fallback_reason is None
            loss = loss + (tok * dtok.cuda()).sum() + 0.01 * cv
            outs.append((tok, cv))
        fb = m._fused
        assert [s.busy for s in fb.slots] == [True, True]              # two forwards wait for their backward
        for task, (tok, cv) in zip((0, 1), outs):
            slot = fb.slots[task]
            noises = None if not std else {i: n.double().cpu() for i, n in slot.noises.items()}
            ovr = None
            if f16:                  # fp16 storage flips near-tied experts: follow the module's routing (tests/test_engine.py)
                ovr = {i: slot.eng.act[i]["gate"]["idx"].cpu() for i in (1, 3)}
            tr, cr, _ = R.backbone_forward(Pr, cfg, img.double(), task, noises=noises, route_override=ovr)
            assert rel(tok, tr) < tol, (step, task, rel(tok, tr))
            assert abs(float(cv) - float(cr)) < 2e-3 * max(1.0, float(cr)), (step, task)
            loss_ref = loss_ref + (tr * dtok.double()).sum() + 0.01 * cr
        loss.backward()
        loss_ref.backward()
        torch.cuda.synchronize()
        assert not any(s.busy for s in fb.slots)
        bad = [(n, rel(p.grad, Pr[n].grad)) for n, p in m.named_parameters()
               if Pr[n].grad is not None and rel(p.grad, Pr[n].grad) > 3 * tol + 7e-4]
        assert not bad, (step, bad)
        with torch.no_grad():                                          # optimizer.step(): the operand copies must follow
            for p in m.parameters():
                p.add_(p.grad, alpha=-0.05)
    assert fb.slots[0].graphs_f and fb.slots[0].graphs_b and fb.slots[1].graphs_b, "steps 2.. must have replayed hipGraphs"


@pytest.mark.parametrize("ckpt", [False, True])
def test_fused_module_matches_per_op_module_path(ckpt):
    """same weights, same images: the fused node and the per-op autograd Functions (fused=False) agree on tokens, balance
    loss and every gradient (fp32: summation order only); ckpt: use_checkpointing=True (the reference's default memory mode:
    the executor keeps block inputs only and re-runs each block in backward).  A deep copy of a model that already ran
    (an EMA twin) starts without executor state and builds its own."""
    _need_gpu()
    import copy
    a, cfg = _model(fused="auto", model_kw=dict(use_checkpointing=ckpt))
    b, _ = _model(fused=False)
    img = torch.randn(4, 3, 32, 48).cuda()
    dtok = (torch.randn(4, cfg.num_tokens, 64) * 0.1).cuda()
    for rep in range(3):
        for m in (a, b):
            m.zero_grad(set_to_none=True)
            loss = 0.0
            for task in (0, 1):
                tok, cv = m(img, task_id=task)
                loss = loss + (tok * dtok).sum() + 0.01 * cv
            loss.backward()
            m.last = (tok.detach(), cv.detach())
        assert a.fused_fallback_reason is None and b._fused is None
        assert a._fused.slots[0].eng.checkpoint == ckpt
        assert rel(a.last[0], b.last[0]) < 1e-5 and abs(float(a.last[1]) - float(b.last[1])) < 1e-5
        bad = [(n, rel(p.grad, q.grad)) for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters())
               if rel(p.grad, q.grad) > 2e-4]
        assert not bad, (rep, bad)
    twin = copy.deepcopy(a)
    assert twin._fused is None and a._fused is not None
    tok, cv = twin(img, task_id=0)
    assert twin._fused is not None and twin._fused is not a._fused and torch.equal(tok, a(img, task_id=0)[0])


def test_fused_module_gradient_semantics():
    """.grad handling of the fused node = torch's: accumulation over backward calls without a zero_grad, zero_grad(set_to_none
    =False), a .grad tensor assigned by someone else (its value is kept and added to), one task at a time
    (train/train_utils.py:373-404 `one_by_one`) equal to the joint backward."""
    _need_gpu()
    m, cfg = _model()
    img = torch.randn(3, 3, 32, 48).cuda()
    dtok = (torch.randn(3, cfg.num_tokens, 64) * 0.1).cuda()

    def joint():
        loss = 0.0
        for task in (0, 1):
            tok, cv = m(img, task_id=task)
            loss = loss + (tok * dtok).sum() + 0.01 * cv
        loss.backward()

    def grads():
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in m.named_parameters()}

    for _ in range(3):                                   # eager, capture, replay
        m.zero_grad(set_to_none=True)
        joint()
    g1 = grads()
    joint()                                              # no zero_grad: accumulates
    g2 = grads()
    assert all(rel(g2[n], 2 * g1[n]) < 1e-6 for n in g1 if float(g1[n].abs().max()) > 0)
    m.zero_grad(set_to_none=False)                       # zeroes in place: the views stay installed
    ptr = m.cls_token.grad.data_ptr()
    joint()
    g3 = grads()
    assert m.cls_token.grad.data_ptr() == ptr
    assert all(torch.equal(g3[n], g1[n]) for n in g1), "deterministic: same step, same bits"
    # a foreign .grad on some parameters, None on others, the view on the rest
    names = [n for n, _ in m.named_parameters()]
    for i, (n, p) in enumerate(m.named_parameters()):
        if i % 3 == 0:
            p.grad = torch.full_like(p, 0.5)
        elif i % 3 == 1:
            p.grad = None
        else:
            p.grad.zero_()
    joint()
    g4 = grads()
    for i, n in enumerate(names):
        want = g1[n] + (0.5 if i % 3 == 0 else 0.0)
        assert rel(g4[n], want) < 1e-6, n
    # one task at a time: forward / backward per task (the same slot twice) = the joint step
    m.zero_grad(set_to_none=True)
    for task in (0, 1):
        tok, cv = m(img, task_id=task)
        ((tok * dtok).sum() + 0.01 * cv).backward()
    g5 = grads()
    assert all(rel(g5[n], g1[n]) < 1e-5 for n in g1 if float(g1[n].abs().max()) > 0)
    assert len(m._fused.slots) == 2


def test_fused_module_eval_and_no_grad_forward():
    _need_gpu()
    a, cfg = _model(std=1.0, E=8)
    b, _ = _model(std=1.0, E=8, fused=False)
    img = torch.randn(2, 3, 32, 48).cuda()
    a.eval(); b.eval()
    with torch.no_grad():
        ta, ca = a(img, task_id=1)
        tb, cb = b(img, task_id=1)
    assert a.fused_fallback_reason is None
    assert rel(ta, tb) < 1e-5 and float(ca) == 0.0 and float(cb) == 0.0          # eval: no noise, no balance loss
    with torch.no_grad():                                                        # second and third use: captured, replayed
        for _ in range(2):
            img2 = torch.randn(2, 3, 32, 48).cuda()
            t2, _ = a(img2, task_id=1)
            assert rel(t2, b(img2, task_id=1)[0]) < 1e-5
    assert a._fused.slots[0].graphs_e
    tc, _ = a(img, task_id=1)                                                    # eval with autograd on: per-op path
    assert a.fused_fallback_reason == "eval mode with autograd on" and rel(tc, tb) < 1e-5
    # a dropped forward (no backward) gives its slot back
    a.train()
    tok, cv = a(img, task_id=0)
    assert a._fused.slots[0].busy
    del tok, cv
    import gc
    gc.collect()
    assert not a._fused.slots[0].busy


def test_fused_module_evaluation_loop_overlaps_the_task_passes():
    """model.eval() under no_grad, the backbone called once per task on every batch (a new tensor each time): from the second
    batch on the other task's pass is started with the first call; same tokens as the per-op path, no context left busy;
    a training step in between and a short last batch are taken in stride."""
    _need_gpu()
    a, cfg = _model(std=1.0, E=8)
    b, _ = _model(std=1.0, E=8, fused=False)
    a.eval(); b.eval()
    g = torch.Generator().manual_seed(33)
    with torch.no_grad():
        for i, B in enumerate((4, 4, 4, 4, 2, 4)):
            x = torch.randn(B, 3, 32, 48, generator=g).cuda()
            for task in (0, 1):
                ta, ca = a(x, task_id=task)
                tb, _ = b(x, task_id=task)
                assert a.fused_fallback_reason is None and rel(ta, tb) < 1e-5 and float(ca) == 0.0, (i, task)
            del x
    fb = a._fused
    assert fb.prefetch_hits >= 3 and not any(s.busy for s in fb.slots) and not fb.e_spec
    a.train()                                               # a training step on the same model afterwards
    img = torch.randn(4, 3, 32, 48).cuda()
    tok, cv = a(img, task_id=0)
    (tok.sum() + cv).backward()
    assert not any(s.busy for s in fb.slots)


def test_fused_module_drop_path_matches_oracle():
    """stochastic depth on the fused node: the per-sample factors it drew (slot.path_scales) fed to the oracle"""
    _need_gpu()
    from oracle import ref_torch as R
    m, cfg = _model(model_kw=dict(drop_path_rate=0.5), img_size=(32, 32))
    img = torch.randn(6, 3, 32, 32)
    dtok = torch.randn(6, cfg.num_tokens, 64) * 0.1
    for step in range(3):
        m.zero_grad(set_to_none=True)
        torch.manual_seed(step)
        tok, cv = m(img.cuda(), task_id=1)
        assert m.fused_fallback_reason is None
        ps = m._fused.slots[0].path_scales
        assert sorted(ps) == [1, 2, 3]
        scales = {i: (sa.double().cpu(), sm.double().cpu()) for i, (sa, sm) in ps.items()}
        allv = torch.cat([torch.cat(v) for v in scales.values()])
        keep = {1: 1 - 0.5 / 3, 2: 1 - 1.0 / 3, 3: 0.5}
        for i, (sa, sm) in scales.items():
            for v in (sa, sm):
                assert all(abs(float(x)) < 1e-6 or abs(float(x) - 1 / keep[i]) < 1e-5 for x in v)
        assert float((allv == 0).float().mean()) > 0.05, "some branches must have been dropped"
        Pr = {k: v.detach().clone().double().cpu().requires_grad_() for k, v in m.state_dict().items()}
        tr, cr, _ = R.backbone_forward(Pr, cfg, img.double(), 1, path_scales=scales)
        assert rel(tok, tr) < 2e-4
        ((tok * dtok.cuda()).sum() + 0.01 * cv).backward()
        ((tr * dtok.double()).sum() + 0.01 * cr).backward()
        bad = [(n, rel(p.grad, Pr[n].grad)) for n, p in m.named_parameters()
               if Pr[n].grad is not None and rel(p.grad, Pr[n].grad) > 1e-3]
        assert not bad, (step, bad)


def test_fused_module_origin_convention_and_task_conditioned_gate():
    """origin convention (train_fastmoe.py:425-435 with --use_checkpointing False): tokens only, the balance loss through
    utils/moe_utils.py::collect_noisy_gating_loss; on a task-conditioned gate (configs[2] structure), whose task-embedding
    MLP gets its gradient through the fused node too."""
    _need_gpu()
    from m3vit_amd.moe_utils import collect_noisy_gating_loss
    from oracle import ref_torch as R
    over = dict(gate_dim=69, multi_gate=False, gate_task_specific_dim=16)
    m, cfg = _model(E=8, model_kw=dict(convention="origin"), **over)
    img = torch.randn(3, 3, 32, 48)
    dtok = torch.randn(3, cfg.num_tokens, 64) * 0.1
    for step in range(3):
        m.zero_grad(set_to_none=True)
        Pr = {k: v.detach().clone().double().cpu().requires_grad_() for k, v in m.state_dict().items()}
        loss, loss_ref = 0.0, 0.0
        for task in (0, 2, 4):
            tok = m(img.cuda(), task_id=task)
            assert torch.is_tensor(tok) and m.fused_fallback_reason is None
            loss = loss + (tok * dtok.cuda()).sum() + collect_noisy_gating_loss(m, 0.01)
            tr, cr, _ = R.backbone_forward(Pr, cfg, img.double(), task)
            assert rel(tok, tr) < 2e-4
            loss_ref = loss_ref + (tr * dtok.double()).sum() + 0.01 * cr
        loss.backward()
        loss_ref.backward()
        bad = [(n, rel(p.grad, Pr[n].grad)) for n, p in m.named_parameters()
               if Pr[n].grad is not None and rel(p.grad, Pr[n].grad) > 1e-3]
        assert not bad, (step, bad)
        assert float(m.gate_task_represent.fc1.weight.grad.abs().max()) > 0
    assert len(m._fused.slots) == 3


def test_fused_module_prefetches_the_other_task_passes_and_survives_a_wrong_guess():
    """After a step that called forward(x, 0), forward(x, 1) on one image tensor, the next step's forward(x, 0) also starts
    the pass of task 1 on its own stream (m3vit_amd/fused.py _prefetch); its call finds the result under way - same tokens,
    loss and gradients as without the prefetch.  A step that then does something else (other images for the second task)
    drops the started pass, gets the right result all the same, and the prediction pauses."""
    _need_gpu()
    a, cfg = _model()
    b, _ = _model()
    imgs = [torch.randn(3, 3, 32, 48).cuda() for _ in range(6)]
    dtok = (torch.randn(3, cfg.num_tokens, 64) * 0.1).cuda()

    def step(m, x0, x1):
        m.zero_grad(set_to_none=True)
        loss = 0.0
        for task, x in ((0, x0), (1, x1)):
            tok, cv = m(x, task_id=task)
            if m is b:
                m._fused.prefetch = False
            loss = loss + (tok * dtok).sum() + 0.01 * cv
        loss.backward()
        torch.cuda.synchronize()
        return tok.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}

    for i in range(4):                                      # same tensor for both tasks: prefetch from the third step on
        ta, ga = step(a, imgs[i], imgs[i])
        tb, gb = step(b, imgs[i], imgs[i])
        assert torch.equal(ta, tb) and all(torch.equal(ga[n], gb[n]) for n in ga), i
    fa = a._fused
    assert fa.pattern == [0, 1] and fa.prefetch_misses == 0 and b._fused.prefetch_misses == 0
    assert not fa.spec and not any(s.busy for s in fa.slots)
    ta, ga = step(a, imgs[4], imgs[5])                     # the guess (task 1 on imgs[4]) is wrong
    tb, gb = step(b, imgs[4], imgs[5])
    assert fa.prefetch_misses == 1 and fa.backoff > 0
    assert torch.equal(ta, tb) and all(torch.equal(ga[n], gb[n]) for n in ga)
    import gc
    gc.collect()
    assert not any(s.busy for s in fa.slots)
    ta, ga = step(a, imgs[0], imgs[0])                     # and on it goes
    tb, gb = step(b, imgs[0], imgs[0])
    assert torch.equal(ta, tb) and all(torch.equal(ga[n], gb[n]) for n in ga)
    # a trainer's loop: a NEW batch tensor every step, the old one dropped before the next step's first call - the step's
    # "same images for every task" property must have been noted while the tensor was alive
    fa.backoff = 0
    hits = fa.prefetch_hits
    for i in range(4):
        cpu = torch.randn(3, 3, 32, 48, generator=torch.Generator().manual_seed(900 + i))
        xa = cpu.cuda()
        ta, ga = step(a, xa, xa)
        del xa
        xb = cpu.cuda()
        tb, gb = step(b, xb, xb)
        del xb
        assert torch.equal(ta, tb) and all(torch.equal(ga[n], gb[n]) for n in ga), i
    assert fa.prefetch_hits >= hits + 2 and fa.pattern == [0, 1], (fa.prefetch_hits, hits, fa.pattern)


def test_fused_module_one_task_at_a_time_prefetches_the_next_task():
    """`one_by_one` (train/train_utils.py:373-404): forward / backward per task on the same images, the optimizer step after
    the last task.  The step boundary is the parameter change, so the step is seen as [t0, t1] and from the third step on
    the forward of t1 runs under t0's forward and backward; same gradients as without."""
    _need_gpu()
    a, cfg = _model()
    b, _ = _model()
    img = torch.randn(3, 3, 32, 48).cuda()
    dtok = (torch.randn(3, cfg.num_tokens, 64) * 0.1).cuda()
    for step in range(5):
        res = []
        for m in (a, b):
            m.zero_grad(set_to_none=True)
            for task in (0, 1):
                tok, cv = m(img, task_id=task)
                if m is b:
                    m._fused.prefetch = False
                ((tok * dtok).sum() + 0.01 * cv).backward()
            torch.cuda.synchronize()
            res.append({n: p.grad.detach().clone() for n, p in m.named_parameters()})
            with torch.no_grad():
                torch._foreach_mul_(list(m.parameters()), 1.0)          # optimizer.step(): every parameter written in place
        assert all(torch.equal(res[0][n], res[1][n]) for n in res[0]), step
    assert a._fused.pattern == [0, 1] and a._fused.prefetch_hits >= 3 and a._fused.prefetch_misses == 0
    assert b._fused.prefetch_hits == 0


def test_fused_module_batch_size_changes():
    """the short last batch of an epoch: another batch size gets its own contexts on the same parameters and the same
    gradient buffer, the usual size's contexts (and their graphs) are kept for when it comes back"""
    _need_gpu()
    a, cfg = _model()
    b, _ = _model(fused=False)
    g = torch.Generator().manual_seed(21)
    ctx_of_3 = None
    for step, B in enumerate((3, 3, 3, 2, 3, 3, 5, 3)):
        img = torch.randn(B, 3, 32, 48, generator=g).cuda()
        dtok = (torch.randn(B, cfg.num_tokens, 64, generator=g) * 0.1).cuda()
        for m in (a, b):
            m.zero_grad(set_to_none=True)
            loss = 0.0
            for task in (0, 1):
                tok, cv = m(img, task_id=task)
                loss = loss + (tok * dtok).sum() + 0.01 * cv
            loss.backward()
        torch.cuda.synchronize()
        assert a.fused_fallback_reason is None and a._fused.batch == B
        bad = [(n, rel(p.grad, q.grad)) for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters())
               if rel(p.grad, q.grad) > 2e-4]
        assert not bad, (step, B, bad)
        if step == 2:
            ctx_of_3 = a._fused.slots[0]
            assert ctx_of_3.graphs_f and ctx_of_3.graphs_b
        if step in (4, 7):
            assert a._fused.slots[0] is ctx_of_3, "the contexts (and graphs) of the usual batch size must have been kept"
    assert sorted(a._fused.slot_sets) == [2, 5]


def _dgdp_worker(rank, world, port, q):
    """the reference's data-parallel wrapper around the fused module path (train_fastmoe.py:460 DistributedGroupedDataParallel,
    train/train_utils.py:414 model.allreduce_params()): every rank its own images, the averaged gradients of the dense
    parameters equal the mean of the per-rank gradients, the expert gradients (dp_comm "none") stay local; twice, so that the
    second step runs on gradients that alias the wrapper's flat buffer."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import m3vit_amd
        m3vit_amd.install_fmoe_shim()
        from fmoe import DistributedGroupedDataParallel                     # the reference's import, bound to this repository
        torch.cuda.set_device(0)
        m, cfg = _model()
        ref, _ = _model(fused=False)
        w = DistributedGroupedDataParallel(m, device_ids=[0], find_unused_parameters=True)
        g = torch.Generator().manual_seed(300 + rank)
        for step in range(3):
            img = torch.randn(3, 3, 32, 48, generator=g).cuda()
            dtok = (torch.randn(3, cfg.num_tokens, 64, generator=g) * 0.1).cuda()
            for model in (w, ref):
                for p in model.parameters():
                    p.grad = None                                            # optimizer.zero_grad(set_to_none=True)
                loss = 0.0
                for task in (0, 1):
                    tok, cv = model(img, task_id=task)
                    loss = loss + (tok * dtok).sum() + 0.01 * cv
                loss.backward()
            assert m.fused_fallback_reason is None and m._fused is not None
            w.allreduce_params()
            torch.cuda.synchronize()
            for (n, p), (_, r) in zip(m.named_parameters(), ref.named_parameters()):
                want = r.grad.detach().clone()
                if getattr(p, "dp_comm", "dp") != "none":
                    dist.all_reduce(want)
                    want /= world
                else:
                    assert ".experts." in n
                assert rel(p.grad, want) < 2e-4, (step, n, rel(p.grad, want))
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_fused_module_under_the_grouped_data_parallel_wrapper_two_ranks_one_gpu():
    _need_gpu()
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dgdp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=480) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res


def _ep_module_worker(rank, world, port, q):
    """the module path with SHARDED experts (world_size = 2: the module holds moe_experts // world_size experts per rank, the gate
    scores all of them - utils/common_config.py:179-185, custom_moe_layer.py:263-265) on the fused executor's expert-parallel
    path, against the same model with every expert local (world_size = 1) on this rank's images: tokens, balance loss, dense
    gradients (own images), expert gradients (this rank's experts saw the rows of BOTH ranks), two steps."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from m3vit_amd.vit import VisionTransformerMoE
        from oracle import ref_torch as R
        torch.cuda.set_device(0)
        E, e_loc = 8, 4
        kw = dict(KW)
        cfg = R.BackboneCfg(mlp_ratio=4.0, moe_mlp_ratio=1.0, vmoe_noisy_std=0.0, moe_experts=E, **kw)
        P = R.init_backbone_params(cfg, seed=11)
        full = VisionTransformerMoE(mlp_ratio=4.0, moe_mlp_ratio=1, vmoe_noisy_std=0.0, moe_experts=E, **kw).cuda()
        full.load_state_dict(P)
        ep = VisionTransformerMoE(mlp_ratio=4.0, moe_mlp_ratio=1, vmoe_noisy_std=0.0, moe_experts=e_loc, world_size=world, **kw).cuda()
        lo = rank * e_loc
        ep.load_state_dict({n: (v[lo:lo + e_loc] if ".mlp.experts." in n else v) for n, v in P.items()})
        full.train(); ep.train()
        g = torch.Generator().manual_seed(400 + rank)
        for step in range(2):
            img = torch.randn(3, 3, 32, 48, generator=g).cuda()
            dtok = (torch.randn(3, cfg.num_tokens, 64, generator=g) * 0.1).cuda()
            outs = {}
            for m in (full, ep):
                m.zero_grad(set_to_none=True)
                loss = 0.0
                for task in (0, 1):
                    tok, cv = m(img, task_id=task)
                    loss = loss + (tok * dtok).sum() + 0.01 * cv
                loss.backward()
                assert m.fused_fallback_reason is None, m.fused_fallback_reason
                outs[m] = (tok.detach(), cv.detach())
            torch.cuda.synchronize()
            assert ep._fused.ep and ep._fused.slots[0].eng.ep_world == world and not ep._fused.slots[0].graphs_f
            assert rel(outs[ep][0], outs[full][0]) < 1e-5 and abs(float(outs[ep][1]) - float(outs[full][1])) < 1e-5
            fg = dict(full.named_parameters())
            for n, p in ep.named_parameters():
                want = fg[n].grad.detach().clone()
                if ".mlp.experts." in n:
                    dist.all_reduce(want)
                    want = want[lo:lo + e_loc]
                    assert getattr(p, "dp_comm", None) == "none"
                assert rel(p.grad, want) < 2e-4, (step, n, rel(p.grad, want))
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_fused_module_with_sharded_experts_two_ranks_one_gpu():
    _need_gpu()
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ep_module_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=480) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res


def test_fused_module_backward_capture_after_a_dropped_forward_and_an_evaluation_pass():
    """a forward is captured, its output dropped (no backward), another Python-level forward runs on the same
    context (an evaluation pass, another task) - the backward graph captured afterwards must not read that other forward's
    routing tensors: the step after the interleave equals the per-op path's."""
    _need_gpu()
    m, cfg = _model()
    ref, _ = _model(fused=False)
    ref.load_state_dict(m.state_dict())
    img = torch.randn(3, 3, 32, 48).cuda()
    img2 = torch.randn(3, 3, 32, 48).cuda()
    dtok = (torch.randn(3, cfg.num_tokens, 64) * 0.1).cuda()

    def one(model, x, task):
        tok, cv = model(x, task_id=task)
        ((tok * dtok).sum() + 0.01 * cv).backward()

    m.zero_grad(set_to_none=True)
    one(m, img, 0)                                        # call 1 of task 0 on slot 0: eager
    m.zero_grad(set_to_none=True)
    tok, cv = m(img, task_id=0)                           # call 2: the forward graph is captured ...
    del tok, cv                                           # ... and its output dropped: no backward, no backward graph
    m.eval()
    with torch.no_grad():
        m(img2, task_id=1)                                # another Python-level forward on the freed context
        m(img2, task_id=1)                                # (and its evaluation graph)
    m.train()
    tok, cv = m(img2, task_id=1)                          # yet another one: task 1, eager, dropped
    del tok, cv
    m.zero_grad(set_to_none=True)
    one(m, img, 0)                                        # task 0 again: replayed forward, backward captured NOW
    torch.cuda.synchronize()
    got = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    ref.zero_grad(set_to_none=True)
    one(ref, img, 0)
    for n, p in ref.named_parameters():
        if p.grad is not None and float(p.grad.abs().max()) > 0:
            assert rel(got[n], p.grad) < 1e-5, n
    m.zero_grad(set_to_none=True)
    one(m, img, 0)                                        # and the replayed pair stays right
    torch.cuda.synchronize()
    for n, p in ref.named_parameters():
        if p.grad is not None and float(p.grad.abs().max()) > 0:
            assert rel(dict(m.named_parameters())[n].grad, p.grad) < 1e-5, n


def test_fused_module_survives_an_exception_in_the_backward_pass_and_frozen_parameters():
    """(1) a backward pass that raises behind the fused node never runs the end-of-backward callback: the next
    step must still hand finished gradients to the caller's stream; (2) frozen parameters keep .grad = None; (3) images that
    require a gradient take the per-op path."""
    _need_gpu()
    m, cfg = _model()
    ref, _ = _model(fused=False)
    ref.load_state_dict(m.state_dict())
    img = torch.randn(3, 3, 32, 48).cuda()
    dtok = (torch.randn(3, cfg.num_tokens, 64) * 0.1).cuda()

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    def step(model, fail=False):
        loss = 0.0
        for task in (0, 1):
            tok, cv = model(img, task_id=task)
            if fail and task == 0:
                tok = Boom.apply(tok)
            loss = loss + (tok * dtok).sum() + 0.01 * cv
        loss.backward()

    for _ in range(2):
        m.zero_grad(set_to_none=True)
        step(m)
    m.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match="boom"):
        step(m, fail=True)
    m.zero_grad(set_to_none=True)
    step(m)                                               # a trainer that caught the exception and went on
    g = {n: p.grad.detach().clone() for n, p in m.named_parameters()}       # (read on the caller's stream, no synchronize)
    ref.zero_grad(set_to_none=True)
    step(ref)
    for n, p in ref.named_parameters():
        if p.grad is not None and float(p.grad.abs().max()) > 0:
            assert rel(g[n], p.grad) < 1e-5, n
    # frozen parameters
    frozen = [p for n, p in m.named_parameters() if n.startswith("blocks.0.")]
    for p in frozen:
        p.requires_grad_(False)
    m.zero_grad(set_to_none=True)
    step(m)
    assert all(p.grad is None for p in frozen)
    assert all(p.grad is not None for n, p in m.named_parameters() if p.requires_grad)
    for p in frozen:
        p.requires_grad_(True)
    # images that require a gradient: per-op path, with a gradient
    x = img.clone().requires_grad_()
    tok, cv = m(x, task_id=0)
    assert m.fused_fallback_reason is not None and "images require" in m.fused_fallback_reason
    (tok * dtok).sum().backward()
    assert x.grad is not None and float(x.grad.abs().max()) > 0
