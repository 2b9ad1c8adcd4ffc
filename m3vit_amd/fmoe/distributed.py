"""fmoe.DistributedGroupedDataParallel as train_fastmoe.py:460 / train_utils.py:414,461 use it:
wraps a module, forwards *args/**kwargs, and `.allreduce_params()` averages the gradients of every
parameter whose dp_comm is not "none" over the data-parallel group - one flat RCCL all-reduce per
dtype (xGMI is point-to-point: a few large collectives beat many small ones).  The flat buffer is laid out ONCE:
every such parameter's .grad becomes a view of it, autograd then accumulates in place and the collective runs on
the buffer directly - no gather / scatter passes over the gradients per step (a .grad that was re-created, e.g. by
zero_grad(set_to_none=True), is copied into its view once and re-pointed)."""
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
            buckets.setdefault((comm, p.grad.dtype, p.grad.device), []).append(p)
        for (comm, dtype, dev), params in buckets.items():
            group = self._group(comm)
            world = dist.get_world_size(group=group)
            if fp32_allreduce and dtype != torch.float32:            # reduced in fp32: a converted copy either way
                flat = torch.cat([q.grad.reshape(-1) for q in params]).float()
                views = None
            else:
                flat, views = self._flat_views((comm, dtype, dev), params)
            if not no_scale and not reduce_after:
                flat /= world
            dist.all_reduce(flat, group=group)
            if not no_scale and reduce_after:
                flat /= world
            if views is None:
                o = 0
                for q in params:
                    q.grad.copy_(flat[o:o + q.numel()].view_as(q.grad))
                    o += q.numel()

    def _flat_views(self, key, params):
        """(flat buffer, views): the bucket's gradients as ONE contiguous tensor that the .grad attributes alias."""
        if not hasattr(self, "_flat"):
            self._flat = {}
        ids = tuple(id(q) for q in params)
        ent = self._flat.get(key)
        if ent is None or ent[0] != ids:
            total = sum(q.numel() for q in params)
            flat = torch.zeros(total, dtype=key[1], device=key[2])
            views, o = [], 0
            for q in params:
                views.append(flat[o:o + q.numel()].view_as(q))
                o += q.numel()
            ent = (ids, flat, views)
            self._flat[key] = ent
        _, flat, views = ent
        for q, v in zip(params, views):
            if q.grad.data_ptr() != v.data_ptr():                     # first time, or the trainer dropped the old .grad
                v.copy_(q.grad)
                q.grad = v
        return flat, views

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
