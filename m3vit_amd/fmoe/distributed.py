"""fmoe.DistributedGroupedDataParallel as train_fastmoe.py:460 / train_utils.py:414,461 use it:
wraps a module, forwards *args/**kwargs, and `.allreduce_params()` averages the gradients of every
parameter whose dp_comm is not "none" over the data-parallel group - one flat RCCL all-reduce per
dtype (xGMI is point-to-point: a few large collectives beat many small ones)."""
import torch
import torch.nn as nn


class DistributedGroupedDataParallel(nn.Module):
    def __init__(self, module, auto_allreduce=False, need_sync=True, device_ids=None, find_unused_parameters=False,
                 **kwargs):
        super().__init__()
        self.module = module
        self.comms = {k: kwargs[k] for k in kwargs if k.endswith("_group")}
        if need_sync:
            self._sync_params()

    def _dist(self):
        import torch.distributed as dist
        return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None

    def _group(self, dp_comm):
        return self.comms.get(dp_comm + "_group", None)

    def _sync_params(self):
        dist = self._dist()
        if dist is None:
            return
        for p in self.module.parameters():
            if getattr(p, "dp_comm", "dp") == "none":
                continue
            dist.broadcast(p.data, 0, group=self._group(getattr(p, "dp_comm", "dp")))

    def allreduce_params(self, no_scale=False, reduce_after=False, fp32_allreduce=False):
        dist = self._dist()
        if dist is None:
            return
        buckets = {}
        for p in self.module.parameters():
            if not p.requires_grad or p.grad is None:
                continue
            comm = getattr(p, "dp_comm", "dp")
            if comm == "none":
                continue
            buckets.setdefault((comm, p.grad.dtype), []).append(p.grad)
        for (comm, dtype), grads in buckets.items():
            group = self._group(comm)
            flat = torch.cat([g.reshape(-1) for g in grads])
            if fp32_allreduce and dtype != torch.float32:
                flat = flat.float()
            world = dist.get_world_size(group=group)
            if not no_scale and not reduce_after:
                flat /= world
            dist.all_reduce(flat, group=group)
            if not no_scale and reduce_after:
                flat /= world
            o = 0
            for g in grads:
                g.copy_(flat[o:o + g.numel()].view_as(g))
                o += g.numel()

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
