"""CPU, world_size 2, gloo: the expert-parallel exchange logic (m3vit_amd.ep) and the
DistributedGroupedDataParallel gradient sync.  The HIP row-movement kernels are replaced by
CPU stand-ins injected through ep's callback arguments (test-only); the expert FFN is the
oracle's.  Checked against the single-process oracle with all experts local."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_torch as R


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class CpuGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, div, inv, kk):
        ctx.save_for_backward(inv)
        ctx.kk, ctx.n = kk, src.shape[0]
        return src[(index.long() // div)]

    @staticmethod
    def backward(ctx, g):
        inv, = ctx.saved_tensors
        return g[inv.long()].view(ctx.n, ctx.kk, -1).sum(1), None, None, None, None


def cpu_route(gate_idx, e_tot):
    counts, offsets, pos, ros = R.route_build(gate_idx, e_tot)
    return ros.to(torch.int32), pos.to(torch.int32), counts


def _ep_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from m3vit_amd import ep
        torch.manual_seed(100)                       # same expert weights everywhere
        E_loc, D, H, k, T = 2, 8, 12, 2, 13 + 3 * rank
        E = E_loc * world
        w1 = torch.randn(E, H, D, dtype=torch.float64) * 0.3; b1 = torch.randn(E, H, dtype=torch.float64) * 0.1
        w2 = torch.randn(E, D, H, dtype=torch.float64) * 0.3; b2 = torch.randn(E, D, dtype=torch.float64) * 0.1
        g = torch.Generator().manual_seed(7 + rank)  # rank-specific tokens and routing
        x = torch.randn(T, D, dtype=torch.float64, generator=g).requires_grad_()
        idx = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(T)])
        if rank == 1:
            idx[idx == 0] = 1                        # expert 0 gets nothing from rank 1 (ragged / empty segments)
            idx[:, 1] = torch.where(idx[:, 0] == idx[:, 1], (idx[:, 1] + 1) % E, idx[:, 1])
        lo = rank * E_loc
        w1l = w1[lo:lo + E_loc].clone().requires_grad_()

        def expert_fn(rows, cnt):
            return R.experts_ffn(rows, cnt.tolist(), w1l, b1[lo:lo + E_loc], w2[lo:lo + E_loc], b2[lo:lo + E_loc])

        y = ep.general_global_forward_ep(x, idx, expert_fn, E_loc, world, route_fn=cpu_route,
                                         gather_fn=CpuGather.apply)
        xr = x.detach().clone().requires_grad_()
        ref = R.moe_dispatch_ffn(xr, idx, w1, b1, w2, b2)
        assert torch.allclose(y, ref, rtol=1e-10, atol=1e-12), float((y - y.new_tensor(ref)).abs().max())
        gy = torch.randn(y.shape, dtype=torch.float64, generator=g)
        y.backward(gy); ref.backward(gy)
        assert torch.allclose(x.grad, xr.grad, rtol=1e-10, atol=1e-12)
        # expert weight grads: this rank's experts see tokens of BOTH ranks -> compare with the sum of
        # per-rank single-process grads gathered over ranks
        w1f = w1.clone().requires_grad_()
        R.moe_dispatch_ffn(x.detach(), idx, w1f, b1, w2, b2).backward(gy)
        tot = w1f.grad.clone(); dist.all_reduce(tot)
        assert torch.allclose(w1l.grad, tot[lo:lo + E_loc], rtol=1e-9, atol=1e-11)
        # plan invariants
        p = ep.ExchangePlan([3, 0, 2, 5], [1, 4, 0, 2], 2, 2)
        assert p.in_splits == [3, 7] and p.out_splits == [5, 2] and p.fwd_expert_count == [1, 6]
        assert sorted(p.regroup) == list(range(7)) and [p.regroup[i] for i in p.regroup_inv] == list(range(7))
        q.put((rank, "ok"))
    except Exception as e:                           # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _dgdp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from m3vit_amd.fmoe import DistributedGroupedDataParallel
        from m3vit_amd.fmoe.layers import mark_module_parallel_comm
        torch.manual_seed(rank)                      # different init per rank: the wrapper must broadcast rank 0's
        m = torch.nn.ModuleDict({"dense": torch.nn.Linear(4, 3), "experts": torch.nn.Linear(4, 3)})
        mark_module_parallel_comm(m["experts"], "none")
        ref_dense = m["dense"].weight.detach().clone()
        w = DistributedGroupedDataParallel(m, device_ids=[0], find_unused_parameters=True)
        t = m["dense"].weight.detach().clone(); dist.broadcast(t, 0)
        assert torch.equal(m["dense"].weight.detach(), t)                   # synced to rank 0
        if rank == 1:
            assert not torch.equal(m["dense"].weight.detach(), ref_dense)
        for p in m.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
        w.allreduce_params()
        assert torch.allclose(m["dense"].weight.grad, torch.full_like(m["dense"].weight, 1.5))   # mean of 1, 2
        assert torch.allclose(m["experts"].weight.grad, torch.full_like(m["experts"].weight, float(rank + 1)))  # untouched
        assert list(w.state_dict().keys())[0].startswith("module.")
        # the dense gradients now alias ONE flat buffer: a second step that accumulates in place (as autograd does
        # into an existing .grad) needs no gather / scatter passes ...
        flat = w._flat[("dp", torch.float32, m["dense"].weight.device)][1]
        assert m["dense"].weight.grad.data_ptr() == flat.data_ptr() and flat.numel() == 4 * 3 + 3
        ptr = m["dense"].weight.grad.data_ptr()
        for p in m["dense"].parameters():
            p.grad.zero_(); p.grad.add_(float(2 * rank + 1))                 # 1 on rank 0, 3 on rank 1
        w.allreduce_params()
        assert m["dense"].weight.grad.data_ptr() == ptr and torch.allclose(m["dense"].bias.grad, torch.full((3,), 2.0))
        # ... and a .grad the trainer re-created (zero_grad(set_to_none=True) + backward) is taken back into the buffer
        x = torch.ones(2, 4) * (rank + 1)
        for p in m["dense"].parameters():
            p.grad = None
        m["dense"](x).sum().backward()
        w.allreduce_params()
        assert m["dense"].weight.grad.data_ptr() == ptr
        assert torch.allclose(m["dense"].weight.grad, torch.full((3, 4), 3.0))    # mean of 2 rows x (1 | 2) = (2 + 4) / 2
        # the wrapper's own zero_grad keeps the views installed (one memset): the next backward accumulates straight into the
        # flat buffer and allreduce_params moves nothing; the expert (dp_comm "none") gradients are zeroed the torch way
        w.zero_grad()
        assert m["dense"].weight.grad.data_ptr() == ptr and float(flat.abs().sum()) == 0.0
        assert float(m["experts"].weight.grad.abs().sum()) == 0.0
        m["dense"](x).sum().backward()
        assert m["dense"].weight.grad.data_ptr() == ptr                       # autograd accumulated in place
        w.allreduce_params()
        assert m["dense"].weight.grad.data_ptr() == ptr
        assert torch.allclose(m["dense"].weight.grad, torch.full((3, 4), 3.0))
        # a MIX of kept views and re-created gradients (an optimizer whose zero_grad(set_to_none=True) covers only part of the
        # parameters): about 60 % of the gradients are foreign, the rest still alias the flat buffer - the gather may not
        # write the buffer through tensors that overlap it (torch.cat(out=flat) raises on that)
        m2 = torch.nn.Sequential(*[torch.nn.Linear(4, 4) for _ in range(5)])          # 10 parameters
        w2 = DistributedGroupedDataParallel(m2)
        for p in m2.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
        w2.allreduce_params()
        ps = list(m2.parameters())
        flat2 = w2._flat[("dp", torch.float32, ps[0].device)][1]
        assert all(p.grad.data_ptr() == flat2.data_ptr() + 4 * o for p, o in zip(ps, [0, 16, 20, 36, 40, 56, 60, 76, 80, 96]))
        for i, p in enumerate(ps):
            if i % 5 < 3:                                             # 6 of 10 re-created
                p.grad = torch.full_like(p, float(10 * (rank + 1) + i))
            else:                                                     # 4 of 10 accumulate in place
                p.grad.zero_(); p.grad.add_(float(rank + 1))
        w2.allreduce_params()
        for i, p in enumerate(ps):
            want = (15.0 + i) if i % 5 < 3 else 1.5                   # means of (10 + i, 20 + i) and of (1, 2)
            assert torch.allclose(p.grad, torch.full_like(p, want)), (i, p.grad.flatten()[:2])
            assert flat2.data_ptr() <= p.grad.data_ptr() < flat2.data_ptr() + 4 * flat2.numel()
        q.put((rank, "ok"))
    except Exception:                                # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run(worker):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res


def test_expert_parallel_exchange_world2_gloo():
    _run(_ep_worker)


def test_dgdp_allreduce_params_world2_gloo():
    _run(_dgdp_worker)


def test_device_plan_matches_the_host_plan_on_cpu():
    """ep.device_plan (vectorised torch ops on CPU tensors; m3_ep_plan on the GPU - tests/test_hip_kernels.py) against
    the plain-loop ExchangePlan for random count matrices incl. empty blocks, empty experts and empty sources."""
    import torch
    from m3vit_amd.ep import ExchangePlan, device_plan
    g = torch.Generator().manual_seed(5)
    for world, e_loc in [(1, 4), (2, 2), (4, 4), (8, 2), (8, 8), (3, 5)]:
        for trial in range(3):
            send = torch.randint(0, 40, (world * e_loc,), generator=g)
            recv = torch.randint(0, 40, (world * e_loc,), generator=g)
            if trial == 1:
                recv.view(world, e_loc)[:, 0] = 0            # an expert nobody routes to
                recv.view(world, e_loc)[world - 1] = 0       # a source that sends nothing
            if trial == 2:
                recv.zero_()
            want = ExchangePlan(send.tolist(), recv.tolist(), world, e_loc)
            got = device_plan(send, recv, world, e_loc)
            assert got.in_splits == want.in_splits and got.out_splits == want.out_splits and got.n_recv == want.n_recv
            assert got.regroup.tolist() == want.regroup and got.regroup_inv.tolist() == want.regroup_inv
            assert got.fwd_expert_count.tolist() == want.fwd_expert_count
            off = [0]
            for c in want.fwd_expert_count:
                off.append(off[-1] + c)
            assert got.offsets.tolist() == off
            ts = [0]
            for c in want.fwd_expert_count:
                ts.append(ts[-1] + (c + 127) // 128)
            assert got.tile_starts.tolist() == ts
