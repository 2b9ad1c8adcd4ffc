"""Multi-task training-step executor around BackboneEngine: what one reference iteration does to the backbone
(train/train_utils.py:423-457 joint multi-task path; models/models.py:299-301: one backbone pass per task, gradients
accumulated), laid out for the GPU:

* the task passes are independent until their gradients are added, so each gets its own engine context (activations,
  scratch, flat gradient buffer; parameters and operand copies shared) and its own HIP stream; `m3_add_f32` sums the
  gradient buffers at the end.  Kernels of different passes then share the GPU: one pass's bandwidth-bound phases run
  under another's MFMA phases, and the ragged last round of workgroups of one kernel is filled by the others';
* the whole step (no host decisions) is captured into hipGraphs and replayed;
* data parallel (replicated experts, the reference's --moe_data_distributed mode, utils/common_config.py:179-181):
  the step is cut into `dp_parts` graphs - the first is the forward + the backward of the top blocks, the others the
  backward of the next blocks down - and after each part the all-reduce of the gradients that part completed (one
  contiguous slice of the flat buffer, ordered top block first) is issued asynchronously, so it runs on RCCL's
  stream under the following parts; only the last slice's all-reduce is exposed.  `dp_parts` large collectives per step
  (replicated experts make the gradient buffer ~0.5 GB: with 6 parts one MoE block's ~80 MB are left for the end).

Expert-parallel runs (ep_world > 1) read each MoE layer's exchange split sizes on the host and therefore execute eagerly;
their task passes still get a stream each, with their blocks interleaved on the host (_ep_interleaved)."""
from __future__ import annotations

import torch

from . import ops
from .engine import BackboneEngine


_STREAMS = {}


def _shared_stream(dev, key):
    """one HIP stream per (device, role) for the life of the process - see MultiTaskStep.__init__"""
    k = (str(dev), key)
    if k not in _STREAMS:
        _STREAMS[k] = torch.cuda.Stream(device=dev)
    return _STREAMS[k]


class MultiTaskStep:
    def __init__(self, cfg, params, batch: int, dtype=torch.float16, device="cuda:0", tasks=None, cv_weight: float = 0.01,
                 parallel_tasks: bool = True, graph: bool = True, world: int = 1, rank: int = 0, expert_parallel: bool = False,
                 wgrad_streams: bool = False, dp_parts: int = 6, checkpoint: bool = False, share_stem: bool = False,
                 ep_capacity: float = 0.0, ep_chunks: int = 1, ep_native: bool = False):
        """ep_capacity (expert parallel only; 0 = the exact exchange): fixed row capacity of the exchange as a multiple of
        the uniform share R / W per (source, destination) pair (BackboneEngine ep_capacity).  The step then reads ONE flag
        on the host, at its end, instead of 2 W split sizes per MoE layer and pass, and repeats itself on the exact path
        when some pair overflowed (same gradients either way: tests/test_ep_engine_gpu.py).
        ep_native (expert parallel, exact exchange): the exchanges go through the library's RCCL entry points
        (m3vit_amd/ep_native.py) instead of torch.distributed - opt-in, see BackboneEngine."""
        self.cfg, self.dev, self.world, self.cv_weight = cfg, torch.device(device), int(world), float(cv_weight)
        if tasks is None:
            tasks = list(range(cfg.num_tasks)) if (cfg.multi_gate or cfg.gate_task_specific_dim >= 0) else [None]
        self.tasks = list(tasks)
        self.use_ep = bool(expert_parallel) and self.world > 1
        wg = bool(wgrad_streams) and not self.use_ep
        self.ep_capacity = float(ep_capacity) if self.use_ep else 0.0
        # exact exchange cut into chunks of local experts and overlapped with the experts' GEMMs inside one pass
        # (BackboneEngine ep_chunks; only where the experts per rank divide)
        E_loc = cfg.moe_experts // max(1, int(world))
        self.ep_chunks = int(ep_chunks) if (self.use_ep and not self.ep_capacity and int(ep_chunks) > 1 and
                                            E_loc % int(ep_chunks) == 0) else 1
        self.ep_repeats = 0                              # steps repeated on the exact path after a capacity overflow
        self.eng = BackboneEngine(cfg, params, batch=batch, dtype=dtype, device=str(self.dev),
                                  ep_world=self.world if self.use_ep else 1, ep_rank=rank if self.use_ep else 0,
                                  wgrad_stream=wg, checkpoint=checkpoint, ep_capacity=self.ep_capacity, ep_chunks=self.ep_chunks,
                                  ep_native=bool(ep_native) and self.use_ep)
        self.par = bool(parallel_tasks) and not self.use_ep and len(self.tasks) > 1
        # expert parallel: the task passes still get their own engine contexts and streams, but their blocks are
        # interleaved on the host (_ep_interleaved): each pass stops once per MoE layer to read its exchange's split
        # sizes, and its all-to-alls run on RCCL's stream - both under the other passes' queued kernels
        self.par_ep = bool(parallel_tasks) and self.use_ep and len(self.tasks) > 1
        self.engs = [self.eng] + ([BackboneEngine(cfg, None, batch=batch, dtype=dtype, device=str(self.dev), share=self.eng,
                                                   ep_world=self.world if self.use_ep else 1, ep_rank=rank if self.use_ep else 0,
                                                   wgrad_stream=wg, checkpoint=checkpoint, ep_capacity=self.ep_capacity, ep_chunks=self.ep_chunks,
                                                   ep_native=bool(ep_native) and self.use_ep)
                                    for _ in self.tasks[1:]]
                                    if (self.par or self.par_ep) else [])
        # (measured and dropped in round 3, profiles/r03_stream_experiments.txt: a high-priority side stream serialises the
        # passes - 18.6 -> 24 ms - and starting pass 1 a few forward blocks behind pass 0 only lengthens the step)
        # The task streams are shared by every runner of the process (one step runs at a time): HIP deals a process's streams
        # round-robin onto a handful of hardware queues (GPU_MAX_HW_QUEUES, default 4), and a runner built after a few others had
        # consumed streams could get a task stream on the SAME queue as the stream it forks from - its passes then ran one after the
        # other (bench.py's configs[4] sub-run behind the shared-stem sub-run: 42.6 ms per step, the serial time, against 37.8
        # alone).  The first streams of the process are the ones every measurement of the two-stream step has used.
        self.streams = [_shared_stream(self.dev, i) for i in range(len(self.engs) - 1)]
        self.flat = self.eng.flat_grads
        # cutting the step only makes sense when there is a collective to hide and the passes run side by side
        depth = self.eng.depth
        nparts = max(1, min(int(dp_parts), depth)) if (self.par and self.world > 1) else 1
        # part j runs blocks [lo_j, hi_j] of the backward (top down); its gradients are flat[seg_j[0]:seg_j[1]]
        cuts = [depth - (depth * j) // nparts for j in range(1, nparts)]          # first block of parts 0..nparts-2
        self.block_ranges = [(hi - 1, lo) for hi, lo in zip([depth] + cuts, cuts + [0])]
        ends = [self.eng.grad_prefix(lo) for lo in cuts] + [self.flat.numel()]
        self.segments = list(zip([0] + ends[:-1], ends))
        self.two_parts = nparts > 1
        # share_stem: the patch embedding and the blocks below the first MoE block see neither task id nor gate, and every
        # pass of the step reads the same images (train/train_utils.py:248-256: model(images, single_task=...) per task), so
        # their forward is computed once (pass 0's context, before the streams fork) and their backward once, on the sum of
        # the passes' d x at the first MoE block's input (after the streams join) - the same gradients with one stem
        # forward + backward instead of one per task.  Side-by-side passes without expert parallelism only.
        self.share_stem = bool(share_stem) and self.par
        self.stem = self.eng.stem_blocks if self.share_stem else 0
        # the gradient add runs under the stem's backward on the (then idle) first task stream: no stream of its own
        self.add_stream = self.streams[0] if self.share_stem else None
        self.late_names = [n for n in self.eng.params if self.eng._block_of(n) < 0 and
                           not n.startswith(("patch_embed.", "cls_token", "pos_embed"))]
        # the part whose block range first reaches below the stem boundary takes the other passes' d x
        self.accept_part = next((j for j, (_, lo) in enumerate(self.block_ranges) if lo < self.stem), len(self.block_ranges) - 1)
        # hipGraph capture: expert-parallel steps read split sizes on the host (eager).  Weight-gradient streams capture
        # fine when forked from the capturing stream itself (tools/wgrad_capture_probe.py: capture + replay bit-exact),
        # but forked from a stream that is ITSELF a fork of the capturing stream (task streams x wgrad streams: a nested
        # fork) hipStreamEndCapture of ROCm 7.2 segfaults - also with plain torch streams, events and elementwise
        # kernels and no engine at all (tools/nested_capture_probe.py, profiles/r03_nested_capture_probe.txt; first seen
        # in profiles/r02_wgrad_capture_segv.txt): the runtime's fault, not the engine's event use.  That combination is
        # refused here and runs eagerly.
        self.capture_refused = None
        import os
        if self.use_ep:
            import torch.distributed as dist
            # the fixed-capacity exchange reads nothing on the host inside the step, so a collective library whose calls
            # can be captured (RCCL) could replay it; never tried on hardware (the build box has one GPU), hence opt-in
            if not (self.ep_capacity and os.environ.get("M3_EP_CAPTURE") == "1" and dist.get_backend() == "nccl"):
                self.capture_refused = ("expert-parallel steps read the exchange's split sizes on the host" if not self.ep_capacity
                                        else "fixed-capacity expert-parallel step: capture is opt-in (M3_EP_CAPTURE=1, backend nccl)")
        elif wg and self.par and (self.share_stem or not (os.environ.get("M3_LINEAR_GRAPHS", "auto") == "1" or (os.environ.get("M3_LINEAR_GRAPHS", "auto") == "auto" and self.world == 1)) or
                                  os.environ.get("M3_WGRAD_STREAMS_CAPTURE", "0") != "1"):
            # (with one linear graph per task pass - _capture_linear - a pass's wgrad stream is forked from that pass's OWN
            # capturing stream, which is the pattern that works; opt-in M3_WGRAD_STREAMS_CAPTURE=1 until measured)
            self.capture_refused = ("wgrad streams forked from forked task streams: hipStreamEndCapture (ROCm 7.2) segfaults on "
                                    "a nested fork (engine-free reproducer: tools/nested_capture_probe.py); use serial tasks or "
                                    "no wgrad streams to replay a graph")
        self.want_graph = bool(graph) and self.capture_refused is None
        # side-by-side passes are captured as one linear graph per pass and stream (_capture_linear); the shared stem, whose
        # passes meet in the middle of a part, keeps the one-graph-per-part form with the fork inside the capture
        import os as _os
        # (also at N > 1 since round 5, one linear graph per (part, pass).  Round 4 turned this off there because the only
        # multi-rank run this build box allows - two PROCESSES on one GPU over gloo - stalled for seconds per step with it;
        # profiles/r05_dp_two_rank_stream_count.txt found the cause: the two processes' streams outnumber the hardware queues
        # HIP maps them to by default (one task stream per step is slower still, 8 757 ms; GPU_MAX_HW_QUEUES=8 -> 103 ms per
        # step against 130 for the one-graph-per-part form).  One rank per GPU never shares queues; the one-device rehearsals
        # (bench.py M3_BENCH_ONE_DEVICE, tests) raise the queue count.  M3_LINEAR_GRAPHS=0 restores one graph per part.)
        lin = _os.environ.get("M3_LINEAR_GRAPHS", "auto")
        self.linear_graphs = self.par and lin != "0" and \
            not (self.share_stem and nparts > 1)            # (a shared stem cut into data-parallel parts keeps the old form)
        self.graphs = None
        self.capture_error = None
        self.images = self.dtok = self.noises = self.logit_bias = None

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

    def _forward(self, e, t):
        e.forward(self.images, t, tsf_bias=self.logit_bias, noises=None if self.noises is None else self.noises.get(t),
                  stem_of=self.eng if self.share_stem else None)

    def _full(self, e, t):
        self._forward(e, t)
        e.backward(self.dtok, cv_weight=self.cv_weight)

    def _part(self, j):
        hi, lo = self.block_ranges[j]
        last = j == len(self.block_ranges) - 1
        share = self.share_stem
        top_lo = max(lo, self.stem)                     # shared stem: the passes' own backward stops at the first MoE block

        def fn(e, t):
            mine = not share or e is self.eng           # does this pass own a stem?
            if j == 0:
                if not (share and e is self.eng):       # (pass 0 zeroed its gradients before its stem forward)
                    e.zero_grad()
                self._forward(e, t)
                e.backward_begin(self.dtok, cv_weight=self.cv_weight)
            if hi >= top_lo:
                e.backward_blocks(hi, top_lo)
            if last and not share:
                e.backward_end()
            elif last and not mine:
                e.backward_end(stem=False)
            else:
                e.backward_sync_wgrad()
        return fn

    def _stem_backward(self, j):
        """shared stem, after the passes of part j joined on the current stream: the stem blocks of this part's range (and,
        in the last part, the embeddings) once, on the sum of the passes' d x.  (Sending this single pass's weight-gradient
        GEMMs to the idle task stream was measured and dropped: 17.10 -> 17.22 ms/step, profiles/r03_shared_stem.txt.)"""
        hi, lo = self.block_ranges[j]
        e = self.eng
        if j == self.accept_part:
            e.accept_dx(self.engs[1:])
        if lo < self.stem:
            e.backward_blocks(min(hi, self.stem - 1), lo)
        if j == len(self.block_ranges) - 1:
            e.backward_end()
        else:
            e.backward_sync_wgrad()

    def serial_step(self):
        """the whole step on the current stream with one engine context (reference order)"""
        self.eng.prepare_weights()
        self.eng.zero_grad()
        for t in self.tasks:
            self._full(self.eng, t)

    def _ep_interleaved(self):
        """expert-parallel step with the task passes on their own streams, block by block: forward block i of every
        pass, then block i + 1 ...; backward likewise from the top.  One host thread issues everything in the same
        order on every rank, so the collectives of the passes line up across ranks."""
        main = torch.cuda.current_stream()
        sts = [main] + self.streams
        self.eng.prepare_weights()
        for st in self.streams:
            st.wait_stream(main)
        xs = []
        for e, t, st in zip(self.engs, self.tasks, sts):
            with torch.cuda.stream(st):
                e.zero_grad()
                xs.append(e.forward_begin(self.images, t, tsf_bias=self.logit_bias,
                                          noises=None if self.noises is None else self.noises.get(t)))
        depth = self.eng.depth
        for i in range(depth):
            for n, (e, st) in enumerate(zip(self.engs, sts)):
                with torch.cuda.stream(st):
                    xs[n] = e._block_forward(i, xs[n], e.cv_acc)
        for e, st in zip(self.engs, sts):
            with torch.cuda.stream(st):
                e.backward_begin(self.dtok, cv_weight=self.cv_weight)
        for i in range(depth - 1, -1, -1):
            for e, st in zip(self.engs, sts):
                with torch.cuda.stream(st):
                    e.backward_blocks(i, i)
        for e, st in zip(self.engs, sts):
            with torch.cuda.stream(st):
                e.backward_end()
        for st in self.streams:
            main.wait_stream(st)
        self._add(0, self.flat.numel())

    def _ep_exact_repeat(self, run):
        """fixed-capacity exchange: the ONE host read of the step - did any pair of any layer of any pass overflow?  Every
        rank must take the same branch (the repeat issues collectives): the flags are OR-ed over the ranks first."""
        if not self.ep_capacity:
            return
        import torch.distributed as dist
        flag = self.engs[0].ep_overflow
        for e in self.engs[1:]:
            flag = torch.maximum(flag, e.ep_overflow)
        flag = flag.clone()
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if not bool(int(flag.item())):
            return
        self.ep_repeats += 1
        for e in self.engs:
            e.ep_overflow.zero_()
            e.ep_fixed = False
        try:
            run()
        finally:
            for e in self.engs:
                e.ep_fixed = True

    def part(self, j: int):
        """part j of the step on the current stream (part 0 alone is the whole step unless the step is cut)"""
        if self.par_ep:
            return self._ep_interleaved()
        if not self.par:
            return self.serial_step()
        if j == 0:
            self.eng.prepare_weights()
            if self.share_stem:
                self.eng.zero_grad()
                self.eng.forward_stem(self.images)
        self._run_tasks(self._part(j))
        if not self.share_stem:
            self._add(*self.segments[j])
            return
        # the other passes' gradients of this part (final now; none of them below the stem boundary - those contexts never
        # ran a stem, their slices there are zero) are added on a side stream while the stem's backward runs
        lo, hi = self.segments[j]
        cut = min(hi, max(lo, self.eng.grad_prefix(self.stem)))
        main = torch.cuda.current_stream()
        self.add_stream.wait_stream(main)
        with torch.cuda.stream(self.add_stream):
            if cut > lo:
                self._add(lo, cut)
        self._stem_backward(j)
        if j == len(self.block_ranges) - 1:
            # parameters outside the blocks that are NOT part of the stem (the task-embedding MLP of a task-conditioned gate):
            # every pass has its own (small) gradient there; pass 0's was completed by the backward_end() just above
            for n in self.late_names:
                for e in self.engs[1:]:
                    ops.add_f32(self.eng.grads[n].view(-1), e.grads[n].view(-1))
        main.wait_stream(self.add_stream)

    def compute(self):
        for j in range(len(self.block_ranges)):
            self.part(j)

    # --------------------------------------------------------------- execution
    def bind(self, images: torch.Tensor, d_tokens: torch.Tensor, noises=None, logit_bias=None):
        """the (device-resident) batch and upstream token gradients the step reads; graphs replay on these buffers.
        noises: {task: {block: N(0,1) [T, E]}} caller-supplied gate noise (noisy-gate training, std = vmoe_noisy_std / E);
        logit_bias: {block: [E]} added to the gate logits of every pass (routing-skew experiments; overrides the
        task-conditioned bias)"""
        self.images, self.dtok, self.noises, self.logit_bias = images, d_tokens, noises, logit_bias

    def _collective_step(self, parts):
        if not self.two_parts:
            parts[0]()
            self._ep_exact_repeat(lambda: self.part(0))  # fixed-capacity exchange only: one flag read, a repeat on overflow
            self.eng.sync_grads(world=self.world)       # mean over ranks; the experts stay local under expert parallelism
            return
        import torch.distributed as dist
        # RCCL queues the collectives in order on its own stream; gloo (CPU rehearsals) serves them from a pool of two
        # worker threads and stalls with more than two outstanding
        inflight = 2 if dist.get_backend() == "gloo" else len(parts)
        works = []
        for j, (run, (lo, hi)) in enumerate(zip(parts, self.segments)):
            run()
            if j >= inflight:
                works[j - inflight].wait()
            works.append(dist.all_reduce(self.flat[lo:hi], async_op=True))      # runs under the parts that follow
        for w in works:
            w.wait()
        self.flat.div_(self.world)

    def step_eager(self):
        self._collective_step([lambda j=j: self.part(j) for j in range(len(self.block_ranges))])

    def _capture_linear(self):
        """One LINEAR hipGraph per (part, task pass), replayed on that pass's own stream, instead of one graph per part with
        the passes forked inside the capture.  Same kernels in the same per-stream order, but the host's replay cost falls
        from ~14 ms to ~0.25 ms per step (a graph whose branches run on several streams is launched node by node with
        events between the branches; a linear one is one submission - tools/graph_per_stream_probe.py): with a trainer that
        synchronises every step (it reads the loss) the step is 0.4 ms shorter, the host never becomes the limiter, and at
        N > 1 the all-reduce behind a part is issued the moment the part is queued."""
        nparts = len(self.block_ranges)
        G = lambda: torch.cuda.CUDAGraph()                                        # noqa: E731
        prep = G()
        with torch.cuda.graph(prep, capture_error_mode="thread_local"):
            self.eng.prepare_weights()
            if self.share_stem:                  # the task-independent stem once, before the passes fork (see part())
                self.eng.zero_grad()
                self.eng.forward_stem(self.images)
        per = []
        for j in range(nparts):
            fn, gs = self._part(j), []
            for e, t in zip(self.engs, self.tasks):
                g = G()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    fn(e, t)
                gs.append(g)
            per.append(gs)
        add_g = post_g = None
        if self.share_stem:
            # behind the join (single part: asserted by the caller): the other passes' gradients above the stem boundary are
            # added on the idle first task stream while the stem's backward runs once on the summed d x - as two more linear
            # graphs (same launches as part()'s tail)
            lo, hi = self.segments[0]
            cut = min(hi, max(lo, self.eng.grad_prefix(self.stem)))
            if cut > lo:
                add_g = G()
                with torch.cuda.graph(add_g, capture_error_mode="thread_local"):
                    self._add(lo, cut)
            post_g = G()
            with torch.cuda.graph(post_g, capture_error_mode="thread_local"):
                self._stem_backward(0)
                for n in self.late_names:
                    for e in self.engs[1:]:
                        ops.add_f32(self.eng.grads[n].view(-1), e.grads[n].view(-1))
        step = self

        class PartReplay:
            def __init__(self, j):
                self.j = j

            def replay(self):
                j = self.j
                main = torch.cuda.current_stream()
                if j == 0:
                    prep.replay()
                for st in step.streams:
                    st.wait_stream(main)
                per[j][0].replay()
                for st, g in zip(step.streams, per[j][1:]):
                    with torch.cuda.stream(st):
                        g.replay()
                for st in step.streams:
                    main.wait_stream(st)
                if not step.share_stem:
                    step._add(*step.segments[j])
                    return
                step.add_stream.wait_stream(main)
                if add_g is not None:
                    with torch.cuda.stream(step.add_stream):
                        add_g.replay()
                post_g.replay()
                main.wait_stream(step.add_stream)

        self._linear_keep = (prep, per, add_g, post_g)
        return [PartReplay(j) for j in range(nparts)]

    def capture(self) -> bool:
        """Capture the compute of a step into hipGraph(s).  Returns False (and stays eager) if capture is unavailable."""
        if not self.want_graph:
            self.capture_error = self.capture_refused
            return False
        try:
            side = _shared_stream(self.dev, "capture")
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.compute()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if self.linear_graphs:
                self.graphs = self._capture_linear()
                return True
            # thread_local: other threads of the process (the RCCL watchdog of torch.distributed polls events)
            # may keep making HIP calls while this thread captures
            graphs = []
            for j in range(len(self.block_ranges)):
                gj = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gj, capture_error_mode="thread_local"):
                    self.part(j)
                graphs.append(gj)
            self.graphs = graphs
            return True
        except Exception as exc:   # capture is an optimisation, never a requirement - but say why it was lost
            import sys
            print(f"[m3vit_amd.step] hipGraph capture failed, running eagerly: {type(exc).__name__}: {exc}", file=sys.stderr,
                  flush=True)
            self.capture_error = f"{type(exc).__name__}: {exc}"
            self.graphs = None
            for e in self.engs:                       # launches that were only captured never ran: nothing may ride on them
                if e.wq is not None:
                    e.wq.reset()
            torch.cuda.synchronize()
            return False

    def step(self):
        """one step: replay the captured graphs if there are any, else launch eagerly; collectives stay outside the graphs"""
        if self.graphs is None:
            return self.step_eager()
        self._collective_step([g.replay for g in self.graphs])

    @property
    def launch(self) -> str:
        if self.graphs is None:
            return "eager"
        return "hipGraph replay (one linear graph per task pass and stream)" if self.linear_graphs else "hipGraph replay"
