"""GPU box, 2 ranks on ONE GPU over gloo (RCCL refuses two ranks on one device; the exchange code is
backend-agnostic): the engine's expert-parallel path against the same engine with all experts local."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def rel(a, b):
    a = a.detach().double().cpu().flatten(); b = b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


# layer shapes the expert-parallel engine is run at (two ranks): the toy width every other test of this file uses, and the
# layer shapes of BASELINE configs[3] (ViT-Base, E = 64 -> 32 experts per rank, utils/common_config.py:179-185) and
# configs[4] (ViT-Base, E = 16, moe_mlp_ratio 4 -> H = 3072, 480 x 640 -> N = 1201) at depth 2, in the benchmarked fp16
EP_CASES = {
    "toy_f32": dict(cfg=dict(img_size=(32, 48), embed_dim=64, depth=4, num_heads=2, moe_experts=4, moe_top_k=2, gate_dim=66,
                             multi_gate=True), B=3, dtype="float32", tol=2e-4),
    "config3_vit_base_e64_f16": dict(cfg=dict(img_size=(224, 224), embed_dim=768, depth=2, num_heads=12, mlp_ratio=4.0,
                                              moe_mlp_ratio=1.0, moe_experts=64, moe_top_k=4, gate_dim=770, multi_gate=True),
                                     B=2, dtype="float16", tol=1e-3),
    "config4_vit_base_ratio4_n1201_f16": dict(cfg=dict(img_size=(480, 640), embed_dim=768, depth=2, num_heads=12, mlp_ratio=4.0,
                                                       moe_mlp_ratio=4.0, moe_experts=16, moe_top_k=4, gate_dim=770,
                                                       multi_gate=True), B=1, dtype="float16", tol=1e-3),
}


def _worker(rank, world, port, q, case="toy_f32"):
    """The expert-parallel engine (experts sharded over two ranks, custom_moe_layer.py:263-265 behind
    utils/common_config.py:179-185) against the same engine with every expert local, on this rank's own images: tokens,
    balance loss, every gradient (this rank's experts saw the rows of BOTH ranks), and the dense-only gradient sync.
    Both engines run the same kernels on the same rows (a routed row's result does not depend on its slot): the forward is
    compared at 1e-4 in fp16 too.  The backward rounds at different points by design - with local experts d y = score * d x
    is never materialised (FC2's backward GEMMs take fp16(d x) and apply the score on the way: fp16(score * fp16(d x))),
    with sharded experts it must cross the wire (fp16(score * d x) from the fp32 d x) - so in fp16 the expert FC2
    gradients differ by one fp16 rounding per element (measured 3.5e-4 relative L2 at ~25 rows per expert); bound 1e-3,
    the fp16 bound of tests/test_engine.py."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")       # two processes on one device (profiles/r05_dp_two_rank_stream_count.txt)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from m3vit_amd.config import BackboneConfig, init_params
        from m3vit_amd.engine import BackboneEngine
        torch.cuda.set_device(0)
        c = EP_CASES[case]
        cfg = BackboneConfig(**c["cfg"])
        dtype, tol, B = getattr(torch, c["dtype"]), c["tol"], c["B"]
        D, E = cfg.embed_dim, cfg.moe_experts
        e_loc = E // world
        P = init_params(cfg, seed=3, zero_bias=False)
        g = torch.Generator().manual_seed(50 + rank)
        img = torch.randn(B, 3, *cfg.img_size, generator=g).cuda()
        dtok = (torch.randn(B, cfg.num_tokens, D, generator=g) * 0.1).cuda()
        ref = BackboneEngine(cfg, P, batch=B, dtype=dtype)                                   # all experts local
        ep = BackboneEngine(cfg, P, batch=B, dtype=dtype, ep_world=world, ep_rank=rank)
        assert ep.params["blocks.1.mlp.experts.htoh4.weight"].shape[0] == e_loc
        for task in (0, 1):
            t_ref, cv_ref = ref.forward(img, task)
            t_ep, cv_ep = ep.forward(img, task)
            for i in range(cfg.depth):
                if cfg.is_moe(i):                                   # same routing on both engines
                    assert torch.equal(ep.act[i]["gate"]["idx"], ref.act[i]["gate"]["idx"]), (task, i)
            assert rel(t_ep, t_ref) < (1e-5 if dtype == torch.float32 else 1e-4), ("tokens", task, rel(t_ep, t_ref))
            assert abs(float(cv_ep) - float(cv_ref)) < 1e-5 * max(1.0, abs(float(cv_ref)))
            ref.backward(dtok, cv_weight=0.01)
            ep.backward(dtok, cv_weight=0.01)
        lo = rank * e_loc
        bad = []
        for n, gr in ep.grads.items():
            gref = ref.grads[n]
            if ".mlp.experts." in n:
                tot = gref.clone(); dist.all_reduce(tot)            # this rank's experts saw tokens of BOTH ranks
                e = rel(gr, tot[lo:lo + e_loc])
                assert float(gr.abs().max()) > 0, n
            else:
                e = rel(gr, gref)                                   # before the DP sync: own images only
            if e > tol:
                bad.append((n, e))
        assert not bad, bad
        # DP sync touches only the non-expert slice under EP
        before = ep.flat_grads.clone()
        ep.sync_grads(world=world)
        assert torch.equal(ep.flat_grads[ep.n_dense:], before[ep.n_dense:])
        m = before[: ep.n_dense].clone(); dist.all_reduce(m); m /= world
        assert torch.allclose(ep.flat_grads[: ep.n_dense], m, rtol=1e-6, atol=1e-8)
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", list(EP_CASES))
def test_engine_expert_parallel_two_ranks_one_gpu(case):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, case)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res


def _dp_worker(rank, world, port, q):
    """data parallel, replicated experts: the step cut into parts with an all-reduce behind each (MultiTaskStep) must
    leave on EVERY rank the mean over the ranks of the serial per-rank gradients, eagerly and as replayed graphs"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")       # two processes on one device (profiles/r05_dp_two_rank_stream_count.txt)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from m3vit_amd.config import BackboneConfig, init_params
        from m3vit_amd.step import MultiTaskStep
        torch.cuda.set_device(0)
        cfg = BackboneConfig(img_size=(32, 48), embed_dim=64, depth=4, num_heads=2, moe_experts=4, moe_top_k=2,
                             gate_dim=66, multi_gate=True)
        P = init_params(cfg, seed=3, zero_bias=False)
        B = 3
        g = torch.Generator().manual_seed(70 + rank)                      # every rank its own images
        img = torch.randn(B, 3, 32, 48, generator=g).cuda()
        dtok = (torch.randn(B, cfg.num_tokens, 64, generator=g) * 0.1).cuda()
        ref = MultiTaskStep(cfg, P, batch=B, dtype=torch.float32, cv_weight=0.01)          # world 1: this rank alone
        ref.bind(img, dtok)
        ref.serial_step()
        torch.cuda.synchronize()
        want = {n: v.clone() for n, v in ref.eng.grads.items()}
        for v in want.values():                                            # mean over the ranks
            dist.all_reduce(v)
            v /= world
        for parts in (1, 3):
            run = MultiTaskStep(cfg, P, batch=B, dtype=torch.float32, cv_weight=0.01, world=world, rank=rank, dp_parts=parts)
            run.bind(img, dtok)
            run.step()
            torch.cuda.synchronize()
            bad = [(n, rel(v, want[n])) for n, v in run.eng.grads.items() if rel(v, want[n]) > 1e-5]
            assert not bad, (parts, "eager", bad[:3])
            assert run.capture() and len(run.graphs) == parts
            run.flat.fill_(5.0)
            run.step()
            torch.cuda.synchronize()
            bad = [(n, rel(v, want[n])) for n, v in run.eng.grads.items() if rel(v, want[n]) > 1e-5]
            assert not bad, (parts, "graph", bad[:3])
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_parts_two_ranks_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=480) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res


def _ep_step_worker(rank, world, port, q):
    """expert-parallel step runner: the task passes on their own streams with their blocks interleaved on the host
    (MultiTaskStep._ep_interleaved) must leave the gradients of the one-pass-after-the-other expert-parallel step"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")       # two processes on one device (profiles/r05_dp_two_rank_stream_count.txt)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from m3vit_amd.config import BackboneConfig, init_params
        from m3vit_amd.step import MultiTaskStep
        torch.cuda.set_device(0)
        cfg = BackboneConfig(img_size=(32, 48), embed_dim=64, depth=4, num_heads=2, moe_experts=4, moe_top_k=2,
                             gate_dim=67, multi_gate=True)                                   # 3 task passes
        P = init_params(cfg, seed=3, zero_bias=False)
        B = 3
        g = torch.Generator().manual_seed(70 + rank)
        img = torch.randn(B, 3, 32, 48, generator=g).cuda()
        dtok = (torch.randn(B, cfg.num_tokens, 64, generator=g) * 0.1).cuda()
        ser = MultiTaskStep(cfg, P, batch=B, dtype=torch.float32, world=world, rank=rank, expert_parallel=True,
                            parallel_tasks=False)
        par = MultiTaskStep(cfg, P, batch=B, dtype=torch.float32, world=world, rank=rank, expert_parallel=True)
        assert ser.use_ep and not ser.par_ep and par.par_ep and len(par.engs) == 3 and not par.want_graph
        for run in (ser, par):
            run.bind(img, dtok)
        for _ in range(2):                                   # twice: buffers and streams are re-used step after step
            ser.step(); par.step()
            torch.cuda.synchronize()
            e = rel(par.flat, ser.flat)
            assert e < 1e-5, e
        assert float(ser.flat.abs().max()) > 0
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_expert_parallel_step_with_interleaved_task_streams_two_ranks_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ep_step_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=480) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res


def _ep_fixed_worker(rank, world, port, q):
    """fixed-capacity exchange (BackboneEngine ep_capacity: equal-split all-to-all of ceil(1.25 R / W) rows per pair, plan on
    the device, no host read inside the step) against the exact a2a-v path: identical tokens / loss / gradients when every
    pair fits; with a routing skewed so that one pair overflows, the flag is raised on the device, the step runner repeats
    the step on the exact path and again leaves the exact gradients.  The gate's arithmetic is the same on both paths, so
    the comparison is bit-exact except for the slab order of the grouped weight gradients (different M bound)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")       # two processes on one device (profiles/r05_dp_two_rank_stream_count.txt)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from m3vit_amd.config import BackboneConfig, init_params
        from m3vit_amd.engine import BackboneEngine
        from m3vit_amd.step import MultiTaskStep
        torch.cuda.set_device(0)
        cfg = BackboneConfig(img_size=(32, 48), embed_dim=64, depth=4, num_heads=2, moe_experts=8, moe_top_k=2, gate_dim=66,
                             multi_gate=True)
        P = init_params(cfg, seed=3, zero_bias=False)
        B = 4
        g = torch.Generator().manual_seed(50 + rank)
        img = torch.randn(B, 3, 32, 48, generator=g).cuda()
        dtok = (torch.randn(B, cfg.num_tokens, 64, generator=g) * 0.1).cuda()
        for dtype in (torch.float32, torch.float16):
            exact = BackboneEngine(cfg, P, batch=B, dtype=dtype, ep_world=world, ep_rank=rank)
            fixed = BackboneEngine(cfg, P, batch=B, dtype=dtype, ep_world=world, ep_rank=rank, ep_capacity=1.5)
            assert fixed.ep_cap >= 1.5 * fixed.R / world and fixed.ep_cap % 8 == 0
            orig_item = torch.Tensor.item
            for task in (0, 1):
                t0, c0 = exact.forward(img, task)
                exact.backward(dtok, cv_weight=0.01)
                reads = []
                torch.Tensor.item = lambda self_, *a, **k: (reads.append(1), orig_item(self_, *a, **k))[1]
                sync = torch.cuda.Stream.synchronize
                torch.cuda.Stream.synchronize = lambda self_: (reads.append(1), sync(self_))[1]
                try:
                    t1, c1 = fixed.forward(img, task)
                    fixed.backward(dtok, cv_weight=0.01)
                finally:
                    torch.Tensor.item = orig_item
                    torch.cuda.Stream.synchronize = sync
                assert not reads, "the fixed-capacity pass must not read anything on the host"
                assert torch.equal(t0, t1) and torch.equal(c0, c1), (dtype, task, rel(t1, t0))
            assert not fixed.ep_overflowed()
            torch.cuda.synchronize()
            tol = 1e-5 if dtype == torch.float32 else 1e-4
            bad = [(n, rel(fixed.grads[n], exact.grads[n])) for n in exact.grads if rel(fixed.grads[n], exact.grads[n]) > tol]
            assert not bad, (dtype, bad)
            assert float(fixed.flat_grads.abs().max()) > 0
        # overflow: a logit bias sends every token of BOTH ranks to experts 0 and 1 (rank 0's): the pair (r -> 0) carries
        # all R rows against a capacity of 0.75 R, (r -> 1) none
        bias = torch.zeros(cfg.moe_experts); bias[0] = 30.0; bias[1] = 20.0
        lb = {i: bias.cuda() for i in range(cfg.depth) if cfg.is_moe(i)}
        ref = MultiTaskStep(cfg, P, batch=B, dtype=torch.float32, world=world, rank=rank, expert_parallel=True)
        cap = MultiTaskStep(cfg, P, batch=B, dtype=torch.float32, world=world, rank=rank, expert_parallel=True, ep_capacity=1.5)
        for run in (ref, cap):
            run.bind(img, dtok, logit_bias=lb)
            run.step()
        torch.cuda.synchronize()
        assert cap.ep_repeats == 1, cap.ep_repeats
        assert rel(cap.flat, ref.flat) < 1e-5, rel(cap.flat, ref.flat)
        cap.bind(img, dtok)                                   # balanced routing again: no repeat, same gradients as exact
        ref.bind(img, dtok)
        ref.step(); cap.step()
        torch.cuda.synchronize()
        assert cap.ep_repeats == 1 and rel(cap.flat, ref.flat) < 1e-5
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_expert_parallel_fixed_capacity_exchange_two_ranks_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ep_fixed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=480) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res


def _ep_ckpt_worker(rank, world, port, q):
    """expert parallel x activation checkpointing (the reference's default combination: train_fastmoe.py:178 with sharded
    experts): the recompute of a block in backward repeats no collective - the exchange plan, the received rows and the
    returned expert outputs are kept - and the gradients are those of the keep-everything expert-parallel engine, bit for
    bit (same kernels on the same inputs)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")       # two processes on one device (profiles/r05_dp_two_rank_stream_count.txt)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from m3vit_amd.config import BackboneConfig, init_params
        from m3vit_amd.engine import BackboneEngine
        torch.cuda.set_device(0)
        cfg = BackboneConfig(img_size=(32, 48), embed_dim=64, depth=4, num_heads=2, moe_experts=4, moe_top_k=2,
                             gate_dim=66, multi_gate=True)
        P = init_params(cfg, seed=3, zero_bias=False)
        B = 3
        g = torch.Generator().manual_seed(90 + rank)
        img = torch.randn(B, 3, 32, 48, generator=g).cuda()
        dtok = (torch.randn(B, cfg.num_tokens, 64, generator=g) * 0.1).cuda()
        for dtype in (torch.float32, torch.float16):
            keep = BackboneEngine(cfg, P, batch=B, dtype=dtype, ep_world=world, ep_rank=rank)
            ck = BackboneEngine(cfg, P, batch=B, dtype=dtype, ep_world=world, ep_rank=rank, checkpoint=True)
            calls = []
            orig = dist.all_to_all_single

            def counted(*a, **kw):
                calls.append(1)
                return orig(*a, **kw)
            for task in (0, 1):
                t0, c0 = keep.forward(img, task)
                keep.backward(dtok, cv_weight=0.01)
                t1, c1 = ck.forward(img, task)
                dist.all_to_all_single = counted            # from here on only the backward's own exchanges may run
                try:
                    n0 = len(calls)
                    ck.backward(dtok, cv_weight=0.01)
                    n_bwd = len(calls) - n0
                finally:
                    dist.all_to_all_single = orig
                assert torch.equal(t0, t1) and torch.equal(c0, c1), (dtype, task)
                # backward of an MoE layer: d y out + d x back = 2 row exchanges; the recompute adds none
                assert n_bwd == 2 * sum(ck.is_moe), (n_bwd, sum(ck.is_moe))
            torch.cuda.synchronize()
            assert torch.equal(keep.flat_grads, ck.flat_grads), (dtype, rel(ck.flat_grads, keep.flat_grads))
            assert float(ck.flat_grads.abs().max()) > 0
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_expert_parallel_with_activation_checkpointing_two_ranks_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ep_ckpt_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=480) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res


def _chunk_worker(rank, world, port, q, case, chunks, checkpoint):
    """the exchange overlapped inside one pass (BackboneEngine ep_chunks) against the one-exchange expert-parallel engine:
    same routing, tokens, balance loss and EVERY gradient bit for bit (a row's result does not depend on which chunk
    carried it, and the weight gradients run on the same expert-major rows in the same launches)"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")       # two processes on one device (profiles/r05_dp_two_rank_stream_count.txt)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from m3vit_amd.config import BackboneConfig, init_params
        from m3vit_amd.engine import BackboneEngine
        torch.cuda.set_device(0)
        c = EP_CASES[case]
        cfg = BackboneConfig(**c["cfg"])
        dtype, B = getattr(torch, c["dtype"]), c["B"]
        P = init_params(cfg, seed=3, zero_bias=False)
        g = torch.Generator().manual_seed(60 + rank)
        img = torch.randn(B, 3, *cfg.img_size, generator=g).cuda()
        dtok = (torch.randn(B, cfg.num_tokens, cfg.embed_dim, generator=g) * 0.1).cuda()
        one = BackboneEngine(cfg, P, batch=B, dtype=dtype, ep_world=world, ep_rank=rank, checkpoint=checkpoint)
        cut = BackboneEngine(cfg, P, batch=B, dtype=dtype, ep_world=world, ep_rank=rank, ep_chunks=chunks, checkpoint=checkpoint)
        assert cut.ep_chunks == chunks
        for task in (0, 1):
            t1, cv1 = one.forward(img, task)
            t2, cv2 = cut.forward(img, task)
            assert torch.equal(t1, t2) and torch.equal(cv1, cv2), ("forward", task)
            one.backward(dtok, cv_weight=0.01)
            cut.backward(dtok, cv_weight=0.01)
        torch.cuda.synchronize()
        bad = [n for n in one.grads if not torch.equal(one.grads[n], cut.grads[n])]
        assert not bad, bad
        assert float(one.flat_grads.abs().max()) > 0
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,chunks,checkpoint", [("toy_f32", 2, False), ("toy_f32", 2, True), ("config3_vit_base_e64_f16", 2, False),
                                                    ("config3_vit_base_e64_f16", 4, False), ("config4_vit_base_ratio4_n1201_f16", 2, False)])
def test_expert_parallel_exchange_overlapped_in_chunks_two_ranks_one_gpu(case, chunks, checkpoint):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(r, 2, port, q, case, chunks, checkpoint)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res


class _ExchangeDouble:
    """m3vit_amd.ep_native.NativeExchange's interface over torch.distributed: RCCL refuses two ranks on one device, so on the
    one-GPU build box the engine's ep_native hooks (which exchange goes where, argument order, the work objects' wait()) are
    rehearsed with this stand-in; the entry points themselves run on a one-rank communicator in tests/test_ep_rccl_gpu.py"""
    calls = 0

    def __init__(self, rank, world, group=None, device=None):
        self.rank, self.world, self.group = rank, world, group

    def exchange_counts(self, send_counts):
        import torch.distributed as dist
        assert send_counts.dtype == torch.int64 and send_counts.numel() % self.world == 0
        recv = torch.empty_like(send_counts)
        dist.all_to_all_single(recv, send_counts, group=self.group)
        _ExchangeDouble.calls += 1
        return recv

    def dispatch_async(self, out, x, out_splits, in_splits):
        import torch.distributed as dist
        assert x.is_contiguous() and out.is_contiguous() and sum(in_splits) == x.shape[0] and sum(out_splits) == out.shape[0]
        _ExchangeDouble.calls += 1
        return dist.all_to_all_single(out, x, output_split_sizes=list(out_splits), input_split_sizes=list(in_splits),
                                      group=self.group, async_op=True)

    return_async = dispatch_async

    def close(self):
        pass


def _native_hook_worker(rank, world, port, q, chunks):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import m3vit_amd.ep_native as ep_native
        from m3vit_amd.config import BackboneConfig, init_params
        from m3vit_amd.engine import BackboneEngine
        ep_native.NativeExchange = _ExchangeDouble
        torch.cuda.set_device(0)
        c = EP_CASES["toy_f32"]
        cfg = BackboneConfig(**c["cfg"])
        dtype, B = getattr(torch, c["dtype"]), c["B"]
        P = init_params(cfg, seed=3, zero_bias=False)
        g = torch.Generator().manual_seed(60 + rank)
        img = torch.randn(B, 3, *cfg.img_size, generator=g).cuda()
        dtok = (torch.randn(B, cfg.num_tokens, cfg.embed_dim, generator=g) * 0.1).cuda()
        one = BackboneEngine(cfg, P, batch=B, dtype=dtype, ep_world=world, ep_rank=rank, ep_chunks=chunks)
        nat = BackboneEngine(cfg, P, batch=B, dtype=dtype, ep_world=world, ep_rank=rank, ep_chunks=chunks, ep_native=True)
        assert one.ep_native is None and isinstance(nat.ep_native, _ExchangeDouble)
        for task in (0, 1):
            t1, cv1 = one.forward(img, task)
            before = _ExchangeDouble.calls
            t2, cv2 = nat.forward(img, task)
            assert _ExchangeDouble.calls > before, "the native hooks were not taken"
            assert torch.equal(t1, t2) and torch.equal(cv1, cv2), ("forward", task)
            one.backward(dtok, cv_weight=0.01)
            nat.backward(dtok, cv_weight=0.01)
        torch.cuda.synchronize()
        bad = [n for n in one.grads if not torch.equal(one.grads[n], nat.grads[n])]
        assert not bad, bad
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("chunks", [1, 2])
def test_engine_native_exchange_hooks_two_ranks_one_gpu(chunks):
    """BackboneEngine(ep_native=True): every count / row exchange goes through the NativeExchange object and the results
    are those of the torch.distributed exchange bit for bit (stand-in exchange: see _ExchangeDouble)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_native_hook_worker, args=(r, 2, port, q, chunks)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res
