"""fmoe.DistributedGroupedDataParallel as train_fastmoe.py:460 / train_utils.py:414,461 use it:
wraps a module, forwards *args/**kwargs, and `.allreduce_params()` averages the gradients of every
parameter whose dp_comm is not "none" over the data-parallel group - one flat RCCL all-reduce per
dtype (xGMI is point-to-point: a few large collectives beat many small ones).  The flat buffer is laid out ONCE:
every such parameter's .grad becomes a view of it, autograd then accumulates in place and the collective runs on
the buffer directly - no gather / scatter passes over the gradients per step, PROVIDED the .grad attributes survive between
steps: call this wrapper's `zero_grad()` (it zeroes the flat buffers and keeps the views installed) or the optimizer's
`zero_grad(set_to_none=False)`.  After torch's default `zero_grad(set_to_none=True)`, as the reference trainer calls it, every
.grad is re-created by autograd; `allreduce_params` then gathers them with ONE `torch.cat` into the flat buffer (the pass it
was written to avoid, but not hundreds of small copies) and re-points the .grad attributes at the views again."""
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
        foreign = [i for i, (q, v) in enumerate(zip(params, views)) if q.grad.data_ptr() != v.data_ptr()]
        if len(foreign) == len(params):
            # first time, or the trainer dropped every .grad (zero_grad(set_to_none=True)): one gather pass.  Only when NO
            # gradient aliases the buffer - torch.cat(out=flat) refuses inputs that overlap its output
            torch.cat([q.grad.reshape(-1) for q in params], out=flat)
        elif foreign:
            # a mix (an optimizer whose zero_grad covers part of the parameters, re-created gradients next to kept views):
            # gather only the foreign ones, one batched copy
            torch._foreach_copy_([views[i] for i in foreign], [params[i].grad for i in foreign])
        for i in foreign:
            params[i].grad = views[i]
        return flat, views

    def zero_grad(self, set_to_none: bool = False):
        """Zero the gradients WITHOUT dropping the flat-buffer views (one memset per bucket); parameters outside the buckets
        (dp_comm "none", or not yet seen by allreduce_params) are zeroed / dropped the torch way."""
        bucketed = set()
        for _, flat, views in getattr(self, "_flat", {}).values():
            flat.zero_()
        for ids, _, views in getattr(self, "_flat", {}).values():
            bucketed.update(ids)
        for q in self.module.parameters():
            if id(q) in bucketed:
                continue
            if q.grad is not None:
                if set_to_none:
                    q.grad = None
                else:
                    q.grad.zero_()
        # re-install the views on parameters whose .grad was replaced meanwhile
        for ids, _, views in getattr(self, "_flat", {}).values():
            by_id = {id(q): q for q in self.module.parameters()}
            for pid, v in zip(ids, views):
                q = by_id.get(pid)
                if q is not None and (q.grad is None or q.grad.data_ptr() != v.data_ptr()):
                    q.grad = v

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
