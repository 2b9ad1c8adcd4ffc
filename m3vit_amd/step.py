"""Multi-task training-step executor around BackboneEngine: what one reference iteration does to the backbone
(train/train_utils.py:423-457 joint multi-task path; models/models.py:299-301: one backbone pass per task, gradients
accumulated), laid out for the GPU:

* the task passes are independent until their gradients are added, so each gets its own engine context (activations,
  scratch, flat gradient buffer; parameters and operand copies shared) and its own HIP stream; `m3_add_f32` sums the
  gradient buffers at the end.  Kernels of different passes then share the GPU: one pass's bandwidth-bound phases run
  under another's MFMA phases, and the ragged last round of workgroups of one kernel is filled by the others';
* the whole step (no host decisions) is captured into hipGraphs and replayed;
* data parallel (replicated experts, the reference's --moe_data_distributed mode, utils/common_config.py:179-181):
  the step is split in two graphs - A: forward + backward of the upper blocks, B: backward of the lower blocks - and
  the all-reduce of the upper blocks' gradients (one contiguous slice of the flat buffer) is issued between them, so
  it runs on RCCL's stream under graph B; the rest follows B.  Two large collectives per step.

Expert-parallel runs (ep_world > 1) read per-layer counts on the host and therefore execute eagerly on one stream."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .engine import BackboneEngine


class MultiTaskStep:
    def __init__(self, cfg, params, batch: int, dtype=torch.float16, device="cuda:0", tasks=None, cv_weight: float = 0.01,
                 parallel_tasks: bool = True, graph: bool = True, world: int = 1, rank: int = 0, expert_parallel: bool = False,
                 wgrad_streams: bool = False):
        self.cfg, self.dev, self.world, self.cv_weight = cfg, torch.device(device), int(world), float(cv_weight)
        if tasks is None:
            tasks = list(range(cfg.num_tasks)) if (cfg.multi_gate or cfg.gate_task_specific_dim >= 0) else [None]
        self.tasks = list(tasks)
        self.use_ep = bool(expert_parallel) and self.world > 1
        wg = bool(wgrad_streams) and not self.use_ep
        self.eng = BackboneEngine(cfg, params, batch=batch, dtype=dtype, device=str(self.dev),
                                  ep_world=self.world if self.use_ep else 1, ep_rank=rank if self.use_ep else 0,
                                  wgrad_stream=wg)
        self.par = bool(parallel_tasks) and not self.use_ep and len(self.tasks) > 1
        self.engs = [self.eng] + ([BackboneEngine(cfg, None, batch=batch, dtype=dtype, device=str(self.dev), share=self.eng,
                                                   wgrad_stream=wg) for _ in self.tasks[1:]] if self.par else [])
        self.streams = [torch.cuda.Stream(device=self.dev) for _ in self.engs[1:]]
        self.flat = self.eng.flat_grads
        # two halves only make sense when there is a collective to hide and the passes run side by side
        self.two_parts = self.par and self.world > 1
        self.n_up = self.eng.n_upper if self.two_parts else 0
        self.want_graph = bool(graph) and not self.use_ep and not wg     # ROCm 7.2 crashes capturing the wgrad-stream pattern
        self.graph_a = self.graph_b = None
        self.images = self.dtok = None

    # ------------------------------------------------------------------ pieces
    def _run_tasks(self, fn):
        """fn(engine, task) for every task pass, each on its own stream (or one after the other), joined on the
        current stream"""
        if not self.par:
            for t in self.tasks:
                fn(self.eng, t)
            return
        main = torch.cuda.current_stream()
        for st in self.streams:
            st.wait_stream(main)
        for i, t in enumerate(self.tasks):
            with torch.cuda.stream(main if i == 0 else self.streams[i - 1]):
                fn(self.engs[i], t)
        for st in self.streams:
            main.wait_stream(st)

    def _add(self, lo, hi):
        for e in self.engs[1:]:
            ops.add_f32(self.flat[lo:hi], e.flat_grads[lo:hi])          # flat += gradients of the other passes

    def _full(self, e, t):
        e.forward(self.images, t)
        e.backward(self.dtok, cv_weight=self.cv_weight)

    def _upper(self, e, t):
        e.zero_grad()
        e.forward(self.images, t)
        e.backward_begin(self.dtok, cv_weight=self.cv_weight)
        e.backward_blocks(e.depth - 1, e.split_block)
        e.backward_sync_wgrad()

    @staticmethod
    def _lower(e, t):
        e.backward_blocks(e.split_block - 1, 0)
        e.backward_end()

    def serial_step(self):
        """the whole step on the current stream with one engine context (reference order)"""
        self.eng.prepare_weights()
        self.eng.zero_grad()
        for t in self.tasks:
            self._full(self.eng, t)

    def part_a(self):
        if not self.par:
            return self.serial_step()
        self.eng.prepare_weights()
        if self.two_parts:
            self._run_tasks(self._upper)
            self._add(0, self.n_up)
        else:
            self._run_tasks(lambda e, t: (e.zero_grad(), self._full(e, t)))
            self._add(0, self.flat.numel())

    def part_b(self):
        self._run_tasks(self._lower)
        self._add(self.n_up, self.flat.numel())

    def compute(self):
        self.part_a()
        if self.two_parts:
            self.part_b()

    # --------------------------------------------------------------- execution
    def bind(self, images: torch.Tensor, d_tokens: torch.Tensor):
        """the (device-resident) batch and upstream token gradients the step reads; graphs replay on these buffers"""
        self.images, self.dtok = images, d_tokens

    def _collective_step(self, a, b):
        import torch.distributed as dist
        a()
        if not self.two_parts:
            self.eng.sync_grads(world=self.world)       # mean over ranks; the experts stay local under expert parallelism
            return
        w1 = dist.all_reduce(self.flat[: self.n_up], async_op=True)      # overlaps part B
        b()
        w2 = dist.all_reduce(self.flat[self.n_up:], async_op=True)
        w1.wait()
        w2.wait()
        self.flat.div_(self.world)

    def step_eager(self):
        self._collective_step(self.part_a, self.part_b)

    def capture(self) -> bool:
        """Capture the compute of a step into hipGraph(s).  Returns False (and stays eager) if capture is unavailable."""
        if not self.want_graph:
            return False
        try:
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.compute()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            # thread_local: other threads of the process (the RCCL watchdog of torch.distributed polls events)
            # may keep making HIP calls while this thread captures
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                self.part_a()
            gb = None
            if self.two_parts:
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb, capture_error_mode="thread_local"):
                    self.part_b()
            self.graph_a, self.graph_b = ga, gb
            return True
        except Exception:          # capture is an optimisation, never a requirement
            self.graph_a = self.graph_b = None
            torch.cuda.synchronize()
            return False

    def step(self):
        """one step: replay the captured graphs if there are any, else launch eagerly; collectives stay outside the graphs"""
        if self.graph_a is None:
            return self.step_eager()
        self._collective_step(self.graph_a.replay, self.graph_b.replay if self.graph_b is not None else None)

    @property
    def launch(self) -> str:
        return "hipGraph replay" if self.graph_a is not None else "eager"
