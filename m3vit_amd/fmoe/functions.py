"""fmoe.functions names the reference imports (custom_moe_layer.py:13-15).  Only ensure_comm,
Slice and AllGather are ever CALLED by the reference (and the latter two only with a slice group
that no config sets up); prepare_forward / MOEScatter / MOEGather are provided for code that
composes its own dispatch."""
import torch

from .. import ops
from ..functional import GatherRowsFn


class ExpertCount(torch.Tensor):
    """int64 [E] device tensor of rows per expert that also carries the device-resident
    offsets / m-tile prefix the grouped GEMMs read (no host sync, unlike fastmoe's .cpu())."""

    @staticmethod
    def wrap(counts64, route):
        t = counts64.as_subclass(ExpertCount)
        t._m3_route = route
        return t


def route_meta_from_counts(fwd_expert_count, device):
    r = getattr(fwd_expert_count, "_m3_route", None)
    if r is not None:
        return r.offsets, r.tile_starts
    # plain tensor of counts (any device): metadata plumbing only
    c = torch.as_tensor(fwd_expert_count).to(device=device, dtype=torch.int32)
    z = torch.zeros(1, dtype=torch.int32, device=device)
    offsets = torch.cat((z, torch.cumsum(c, 0).to(torch.int32)))
    tile_starts = torch.cat((z, torch.cumsum((c + 127) // 128, 0).to(torch.int32)))
    return offsets.contiguous(), tile_starts.contiguous()


def ensure_comm(t, comm):
    """fastmoe creates its NCCL communicator here; torch.distributed (RCCL) already owns ours."""
    return None


def prepare_forward(gate, num_expert, world_size):
    """-> (pos, local_expert_count, global_expert_count, fwd_expert_count, fwd_batch_size) with the
    fastmoe meaning; world_size > 1 goes through m3vit_amd.ep."""
    if world_size > 1:
        from ..ep import prepare_forward_ep
        return prepare_forward_ep(gate, num_expert, world_size)
    idx32 = gate.reshape(-1, 1).to(torch.int32).contiguous()
    r = ops.route_build(idx32, num_expert, want_counts64=True)
    cnt = ExpertCount.wrap(r.counts64, r)
    return r, cnt, cnt, cnt, idx32.numel()


class MOEScatter:
    """x_e[s] = inp[row_of_slot[s] // k]  (local part of fastmoe's MOEScatter)."""

    @staticmethod
    def apply(inp, route, k):
        return GatherRowsFn.apply(inp, route.row_of_slot, k, route.pos, k)


class MOEGather:
    """y[i] = y_e[pos[i]]  (local part of fastmoe's MOEGather), token-major [T*k, D]."""

    @staticmethod
    def apply(y_e, route):
        return GatherRowsFn.apply(y_e, route.pos, 1, route.row_of_slot, 1)


class Slice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, rank, world_size, group):
        B = inp.shape[0]
        local = B // world_size
        ctx.args = (B, rank, world_size, group)
        return inp[rank * local:(rank + 1) * local].contiguous()

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        B, rank, world_size, group = ctx.args
        out = [torch.empty_like(g) for _ in range(world_size)]
        dist.all_gather(out, g.contiguous(), group=group)
        return torch.cat(out, 0), None, None, None


class AllGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, rank, world_size, group):
        import torch.distributed as dist
        out = [torch.empty_like(inp) for _ in range(world_size)]
        dist.all_gather(out, inp.contiguous(), group=group)
        ctx.args = (inp.shape[0], rank)
        return torch.cat(out, 0)

    @staticmethod
    def backward(ctx, g):
        n, rank = ctx.args
        return g[rank * n:(rank + 1) * n].contiguous(), None, None, None
