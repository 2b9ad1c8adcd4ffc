"""Expert parallelism over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests): rank r owns experts [r*E_loc, (r+1)*E_loc) (utils/common_config.py:179-185),
the gate scores all E_tot = E_loc * W experts, and per MoE layer there is ONE exchange each way:

    local route_build (rows sorted by GLOBAL expert id = by destination rank, then local expert)
 -> all-to-all of the per-expert counts                  (fastmoe expert_exchange, 8*E_tot bytes)
 -> all-to-all-v of the routed rows, expert-major        (fastmoe global_scatter)
 -> local regroup (src-rank major -> local-expert major), expert_fn on the local experts
 -> inverse regroup, all-to-all-v back                   (fastmoe global_gather), MOEGather to token-major

which is what _fmoe_general_global_forward does for world_size > 1 behind
models/moe/ckpt/custom_moe_layer.py:263-265.  The exchange plan (regroup index, expert-major offsets, tile prefix) is
built ON THE DEVICE from the two count vectors (m3_ep_plan; torch ops on CPU tensors for the gloo rehearsals); the
host reads only the 2 * W split sizes per layer (torch.distributed's a2a-v takes python lists; fastmoe syncs here
too) - no per-row host work.
xGMI is a full point-to-point mesh, so the single large a2a-v per direction (one distinct peer per link)
is the right collective; nothing is chunked into ring steps.

The row movement callbacks default to the HIP kernels; tests inject CPU stand-ins to exercise the
exchange logic under gloo without a GPU.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
import torch.distributed as dist


# ------------------------------------------------------------------ device-side plan
class DevicePlan:
    """in_splits / out_splits (python lists: the one host read), n_recv, and device tensors regroup (i32 [n_recv]:
    expert-major slot -> received row), regroup_inv, fwd_expert_count (i64 [E_loc]), offsets / tile_starts (i32)."""
    __slots__ = ("in_splits", "out_splits", "n_recv", "regroup", "regroup_inv", "fwd_expert_count", "offsets", "tile_starts")


def device_plan(send_counts: torch.Tensor, recv_counts: torch.Tensor, world: int, e_loc: int, regroup_buf=None,
                want_inverse: bool = True) -> DevicePlan:
    """The exchange plan without per-row host work.  CUDA tensors: one m3_ep_plan launch; CPU tensors (gloo
    rehearsals of the logic): the same arithmetic in vectorised torch ops."""
    p = DevicePlan()
    dev = send_counts.device
    recv = recv_counts.view(world, e_loc)
    p.fwd_expert_count = recv.sum(0)
    if send_counts.is_cuda:
        from . import ops
        assert regroup_buf is not None, "device_plan on the GPU needs a regroup buffer of world * (rows routed per rank) int32"
        dp = ops.ep_plan(send_counts, recv_counts, world, e_loc, regroup_buf)
        p.in_splits, p.out_splits, p.n_recv = dp.in_splits, dp.out_splits, dp.n_recv
        p.regroup, p.offsets, p.tile_starts = dp.regroup, dp.offsets, dp.tile_starts
    else:
        sp = torch.cat((send_counts.view(world, e_loc).sum(1), recv.sum(1))).tolist()        # the 2 W split sizes
        p.in_splits, p.out_splits = sp[:world], sp[world:]
        n = p.n_recv = sum(p.out_splits)
        sizes_src = recv.reshape(-1)                                          # blocks in arrival order (s, e)
        src_start = torch.cumsum(sizes_src, 0) - sizes_src
        perm = torch.arange(world * e_loc).view(world, e_loc).t().reshape(-1)   # block ids in (e, s) order
        sizes_em = sizes_src[perm]
        em_start = torch.cumsum(sizes_em, 0) - sizes_em
        blk = torch.repeat_interleave(torch.arange(world * e_loc), sizes_em, output_size=n)
        p.regroup = (src_start[perm][blk] + (torch.arange(n) - em_start[blk])).to(torch.int32)
        z = torch.zeros(1, dtype=torch.int64)
        p.offsets = torch.cat((z, torch.cumsum(p.fwd_expert_count, 0))).to(torch.int32)
        p.tile_starts = torch.cat((z, torch.cumsum((p.fwd_expert_count + 127) // 128, 0))).to(torch.int32)
    if want_inverse:
        inv = torch.empty_like(p.regroup)
        inv[p.regroup.long()] = torch.arange(p.n_recv, dtype=torch.int32, device=dev)
        p.regroup_inv = inv
    else:
        p.regroup_inv = None
    return p


# ------------------------------------------------------------------ pure host-side plan (the checker of device_plan)
class ExchangePlan:
    """Everything the exchange needs, derived from the two count vectors (host ints) with plain Python loops: the
    readable statement of the plan, kept as the reference the device plan is tested against (tests/test_ep_gloo.py,
    tests/test_hip_kernels.py); the product path uses device_plan.

    send_counts[d*E_loc + e]: rows this rank routes to local expert e of rank d.
    recv_counts[s*E_loc + e]: rows rank s routes to this rank's local expert e.
    """

    def __init__(self, send_counts: List[int], recv_counts: List[int], world: int, e_loc: int):
        assert len(send_counts) == world * e_loc == len(recv_counts)
        self.world, self.e_loc = world, e_loc
        self.in_splits = [sum(send_counts[d * e_loc:(d + 1) * e_loc]) for d in range(world)]
        self.out_splits = [sum(recv_counts[s * e_loc:(s + 1) * e_loc]) for s in range(world)]
        self.n_recv = sum(self.out_splits)
        # received rows arrive ordered (src, e); expert_fn wants (e, src): regroup[i] = position in the
        # received buffer of the i-th row of the expert-major buffer
        starts = []
        o = 0
        for s in range(world):
            for e in range(e_loc):
                starts.append(o)
                o += recv_counts[s * e_loc + e]
        regroup = []
        for e in range(e_loc):
            for s in range(world):
                st = starts[s * e_loc + e]
                regroup.extend(range(st, st + recv_counts[s * e_loc + e]))
        self.regroup = regroup                                    # expert-major <- received order
        inv = [0] * len(regroup)
        for i, r in enumerate(regroup):
            inv[r] = i
        self.regroup_inv = inv                                    # received order <- expert-major
        self.fwd_expert_count = [sum(recv_counts[s * e_loc + e] for s in range(world)) for e in range(e_loc)]


class _A2ARows(torch.autograd.Function):
    """all_to_all_single on rows with variable splits; backward is the reverse exchange."""

    @staticmethod
    def forward(ctx, x, in_splits, out_splits, group):
        x = x.contiguous()
        out = x.new_empty((sum(out_splits),) + tuple(x.shape[1:]))
        dist.all_to_all_single(out, x, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        ctx.splits = (in_splits, out_splits, group)
        return out

    @staticmethod
    def backward(ctx, g):
        in_splits, out_splits, group = ctx.splits
        g = g.contiguous()
        out = g.new_empty((sum(in_splits),) + tuple(g.shape[1:]))
        dist.all_to_all_single(out, g, output_split_sizes=in_splits, input_split_sizes=out_splits, group=group)
        return out, None, None, None


def exchange_counts(send_counts: torch.Tensor, group=None) -> torch.Tensor:
    """int64 [W*E_loc] -> int64 [W*E_loc]: entry s*E_loc+e = rows rank s sends to my local expert e."""
    recv = torch.empty_like(send_counts)
    dist.all_to_all_single(recv, send_counts.contiguous(), group=group)
    return recv


# ----------------------------------------------------------------------- row movement
def _hip_gather(src, index_i32, div, inv_i32, kk):
    from .functional import GatherRowsFn
    return GatherRowsFn.apply(src, index_i32, div, inv_i32, kk)


def _hip_route(gate_idx, e_tot):
    from . import ops
    idx32 = gate_idx.reshape(-1, 1).to(torch.int32).contiguous()
    r = ops.route_build(idx32, e_tot, want_counts64=True)
    return r.row_of_slot, r.pos, r.counts64


def general_global_forward_ep(inp, gate_idx, expert_fn, num_expert, world_size, group=None,
                              route_fn: Optional[Callable] = None, gather_fn: Optional[Callable] = None,
                              make_count: Optional[Callable] = None):
    """EP version of _fmoe_general_global_forward: (moe_inp [T,D], gate_top_k_idx [T,k] with ids in
    [0, num_expert*world_size)) -> [T*k, D_out] token-major."""
    route_fn = route_fn or _hip_route
    gather_fn = gather_fn or _hip_gather
    k = gate_idx.shape[1] if gate_idx.dim() > 1 else 1
    e_tot = num_expert * world_size
    row_of_slot, pos, counts64 = route_fn(gate_idx, e_tot)
    send_counts = counts64.to(torch.int64)
    recv_counts = exchange_counts(send_counts, group)
    # a rank can receive at most what all ranks route (every rank routes gate_idx.numel() rows)
    buf = torch.empty(world_size * gate_idx.numel(), dtype=torch.int32, device=inp.device) if inp.is_cuda else None
    plan = device_plan(send_counts, recv_counts, world_size, num_expert, regroup_buf=buf)   # host reads the 2 W split sizes only
    x_send = gather_fn(inp, row_of_slot, k, pos, k)                         # MOEScatter (local part)
    x_recv = _A2ARows.apply(x_send, plan.in_splits, plan.out_splits, group)  # global_scatter
    rg, rgi = plan.regroup, plan.regroup_inv
    x_exp = gather_fn(x_recv, rg, 1, rgi, 1) if plan.n_recv else x_recv
    cnt = plan.fwd_expert_count
    if make_count is not None:
        cnt = make_count(cnt)
    y_exp = expert_fn(x_exp, cnt)
    y_recv = gather_fn(y_exp, rgi, 1, rg, 1) if plan.n_recv else y_exp
    y_send = _A2ARows.apply(y_recv, plan.out_splits, plan.in_splits, group)  # global_gather
    return gather_fn(y_send, pos, 1, row_of_slot, 1)                         # MOEGather (local part)


def prepare_forward_ep(gate, num_expert, world_size, group=None):
    row_of_slot, pos, counts64 = _hip_route(gate, num_expert * world_size)
    recv = exchange_counts(counts64, group)
    fwd = recv.view(world_size, num_expert).sum(0)
    return pos, counts64, recv, fwd, int(fwd.sum())
