"""The expert-parallel exchange through the library's own RCCL entry points (include/m3vit_hip.h: m3_ep_init,
m3_ep_exchange_counts, m3_ep_dispatch, m3_ep_return, m3_ep_destroy) instead of torch.distributed: what fastmoe's
expert_exchange / global_scatter / global_gather do behind _fmoe_general_global_forward
(models/moe/ckpt/custom_moe_layer.py:263-265 with world_size > 1).

torch.distributed is only used once, to hand rank 0's 128-byte unique id to the other ranks (any backend: it is an object
broadcast).  The row exchanges run on a side stream of their own, ordered behind the caller's stream by an event, and hand
back a work object whose wait() makes the caller's stream wait - the same contract as `dist.all_to_all_single(async_op=True)`,
so BackboneEngine's chunked exchange overlaps them with the experts' GEMMs the same way.

Opt-in (`BackboneEngine(ep_native=True)`): this build box has one GPU, so the entry points are exercised with a one-rank
communicator only (tests/test_ep_rccl_gpu.py); the default exchange is torch.distributed, which the two-rank gloo rehearsals
cover."""
from __future__ import annotations

import ctypes
from ctypes import byref, c_int, c_void_p

import torch

from ._lib import check, lib


class _Work:
    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class NativeExchange:
    def __init__(self, rank: int, world: int, group=None, device=None):
        self.rank, self.world = int(rank), int(world)
        self.device = torch.device(device if device is not None else torch.cuda.current_device())
        uid = (ctypes.c_char * 128)()
        if self.rank == 0:
            check(lib().m3_ep_unique_id(ctypes.cast(uid, c_void_p)), "m3_ep_unique_id")
        if self.world > 1:
            import torch.distributed as dist
            box = [bytes(uid)]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            uid = (ctypes.c_char * 128).from_buffer_copy(box[0])
        h = c_int(-1)
        with torch.cuda.device(self.device):
            check(lib().m3_ep_init(ctypes.cast(uid, c_void_p), self.rank, self.world, byref(h)), "m3_ep_init")
        self.handle = h.value
        self.stream = torch.cuda.Stream(device=self.device)

    def close(self):
        if self.handle >= 0:
            torch.cuda.synchronize(self.device)
            check(lib().m3_ep_destroy(self.handle), "m3_ep_destroy")
            self.handle = -1

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 - interpreter shutdown
            pass

    # ---- on the CALLER's stream (small, and the plan kernel behind it needs the result at once)
    def exchange_counts(self, send_counts: torch.Tensor) -> torch.Tensor:
        assert send_counts.dtype == torch.int64 and send_counts.is_cuda and send_counts.numel() % self.world == 0
        recv = torch.empty_like(send_counts)
        check(lib().m3_ep_exchange_counts(self.handle, c_void_p(send_counts.data_ptr()), c_void_p(recv.data_ptr()),
                                          send_counts.numel() // self.world, c_void_p(torch.cuda.current_stream().cuda_stream)),
              "m3_ep_exchange_counts")
        return recv

    # ---- on the exchange stream, behind everything queued on the caller's stream so far
    def _rows(self, fn, name, out, x, out_splits, in_splits):
        assert x.is_cuda and out.is_cuda and x.is_contiguous() and out.is_contiguous()
        assert len(in_splits) == self.world == len(out_splits)
        assert sum(in_splits) == x.shape[0] and sum(out_splits) == out.shape[0]
        row_bytes = x[0].numel() * x.element_size() if x.shape[0] else (out[0].numel() * out.element_size() if out.shape[0] else 1)
        ins = (ctypes.c_int64 * self.world)(*[int(v) for v in in_splits])
        outs = (ctypes.c_int64 * self.world)(*[int(v) for v in out_splits])
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        x.record_stream(self.stream); out.record_stream(self.stream)
        check(fn(self.handle, c_void_p(x.data_ptr()), ins, c_void_p(out.data_ptr()), outs, row_bytes, c_void_p(self.stream.cuda_stream)),
              name)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return _Work(ev)

    def dispatch_async(self, out, x, out_splits, in_splits):
        """rows x [sum(in_splits), ...] (grouped by destination) -> out [sum(out_splits), ...] (grouped by source)"""
        return self._rows(lib().m3_ep_dispatch, "m3_ep_dispatch", out, x, out_splits, in_splits)

    def return_async(self, out, x, out_splits, in_splits):
        """the way home; argument order as dispatch_async (x's rows leave by in_splits, out's arrive by out_splits)"""
        return self._rows(lib().m3_ep_return, "m3_ep_return", out, x, out_splits, in_splits)
