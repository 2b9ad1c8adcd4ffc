"""VisionTransformerMoE.forward as ONE autograd node per call, on the straight-line executor.

What the reference's trainer calls (models/models.py:299-320: `self.backbone(x, task_id=...)` once per task, then one
`loss.backward()`, train/train_utils.py:423-457) is the module API of m3vit_amd.vit.  Built from per-op autograd Functions
(m3vit_amd.functional) that API pays per call for what the fused executor (m3vit_amd.engine.BackboneEngine) does once per
step or not at all: operand copies of every weight, separate bias-gradient passes, materialised d y, ~700 tensor
allocations and ~1500 Python-issued launches.  FusedBackbone puts the executor behind the same `forward(x, task_id)`:

  * the module's nn.Parameters ARE the executor's fp32 masters (same storage); their `.grad` attributes are views of the
    executor's flat gradient buffer, so `optimizer.step()`, `DistributedGroupedDataParallel.allreduce_params()`
    (fmoe/distributed.py) and `zero_grad(set_to_none=True/False)` work as they do on any module;
  * one forward = one torch.autograd.Function node (tokens [B, N, D] fp32 and the summed balance loss come out, d tokens
    and d cv_loss go in); every call that is still waiting for its backward owns an executor context ("slot": activations,
    scratch, gradient buffer, HIP stream).  Forward and backward of a slot are captured into hipGraphs at their second
    use and replayed from then on (the upstream gradient of cv_loss reaches the gate's backward kernel as a device
    scalar: m3_gate_bwd_args.balance_scale_dev);
  * autograd runs a node's backward on the stream of its forward, so the task passes of a joint multi-task step
    (all forwards, one backward) run their backward passes side by side on the GPU like m3vit_amd.step.MultiTaskStep's
    task streams; every slot's gradient buffer is added, as soon as its backward is done, into the ONE buffer the `.grad`
    views alias, on a stream of its own (the only stream that ever writes that buffer - a slot's add runs under the other
    slots' backward kernels), and the stream that called forward() waits for all of it.

The per-op path stays for everything the executor does not cover (see `unsupported()`).
"""
from __future__ import annotations

import weakref

import torch

from . import ops
from .config import BackboneConfig
from .engine import BackboneEngine


class _Slot:
    """one executor context + stream + captured graphs; owned by one forward until its backward ran"""

    def __init__(self, index, eng, device):
        self.index = index
        self.eng = eng
        self.stream = torch.cuda.Stream(device=device)
        self.busy = False
        self.result = None
        self.calls_f, self.calls_b = {}, {}          # per task: number of forward / backward calls seen
        self.graphs_f, self.graphs_b = {}, {}
        self.calls_e, self.graphs_e = {}, {}         # forward without autograd (evaluation): per (task, training flag)
        self.images = self.dtok = None               # static inputs of the graphs
        self.dcv = torch.zeros(1, dtype=torch.float32, device=device)
        self.noises = self.path_scales = None
        self.add_done = None                          # event: slot 0's stream has read this slot's gradient buffer
        self.main = None                              # the stream forward() was called on
        self.out = {}                                 # per task: the forward graph's output tensors
        self.py_owner = None                          # which Python-level eng.forward ran last on this context: ("eager" | "fcap"
                                                      # | "eval", key) - the engine's per-call tensors are that call's


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, fb, slot, task_id, images):
        tok, cv = fb._run_forward(slot, task_id, images)
        ctx.fb, ctx.slot, ctx.task_id = fb, slot, task_id
        ctx.token = _Release(fb, slot)                # returns the slot if this node dies without a backward
        return tok, cv

    @staticmethod
    def backward(ctx, g_tok, g_cv):
        ctx.token.done = True
        ctx.fb._run_backward(ctx.slot, ctx.task_id, g_tok, g_cv)
        return None, None, None, None, None


class _Release:
    def __init__(self, fb, slot):
        self.fb, self.slot, self.done = weakref.ref(fb), slot, False

    def __del__(self):
        if not self.done:
            self.slot.busy = False


class FusedBackbone:
    def __init__(self, model, graph: bool = True, max_slots: int = 8):
        self.model = weakref.ref(model)
        self.graph = bool(graph)
        self.max_slots = max_slots
        self.slots = []
        self.batch = None
        self.device = None
        self.sig = None
        self.dirty = True
        self.anchor = None
        self.names = None
        self._join_queued = False
        import os
        self.prefetch = os.environ.get("M3VIT_PREFETCH", "1") != "0"
        self.hist, self.pattern, self.spec, self.spec_images, self.spec_version = [], None, {}, None, 0
        self.hist_same = True
        self.backoff, self.prefetch_misses, self.prefetch_hits = 0, 0, 0
        self.backward_seen = False
        self.versions_changed = False
        self.e_first, self.e_hist, self.e_pattern, self.e_spec, self.e_version, self.e_backoff = None, [], None, {}, 0, 0

    # a copy of the model (copy.deepcopy for an EMA twin, torch.save of the whole module) gets no executor state - contexts,
    # streams and graphs are rebuilt at its first call
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    # ------------------------------------------------------------------ eligibility
    @staticmethod
    def unsupported(model, x, gate_inp, task_id, sem):
        """None when this call can run on the executor, else the reason it takes the per-op path"""
        if not x.is_cuda:
            return "CPU tensor"
        if model.world_size > 1:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                return "expert parallel layer without an initialised process group"
        if gate_inp is not None:
            return "caller-supplied gate input"
        if not model._fused_static_ok:
            return model._fused_static_why
        if torch.is_grad_enabled() and not model.training:
            return "eval mode with autograd on"
        if torch.is_grad_enabled() and x.requires_grad:
            return "the images require a gradient (the executor's node returns none for them)"
        if x.dim() != 4 or tuple(x.shape[2:]) != tuple(model.img_size):
            return "image size differs from the constructor's"
        if model.multi_gate and task_id is None:
            return "multi-gate model called without a task id"
        for blk in model.blocks:
            if blk.moe and (blk.mlp.gate_hook is not None or blk.mlp.mask is not None):
                return "gate hook / expert mask installed"
        if sem is not None and any(blk.moe and blk.mlp.sem_force for blk in model.blocks):
            return "sem_force routing override"
        return None

    # ------------------------------------------------------------------ set-up
    def _build(self, B, device):
        model = self.model()
        cfg = BackboneConfig(**model._cfg_kwargs)
        named = dict(model.named_parameters())
        self.names = list(named)
        params = {n: p.detach() for n, p in named.items()}
        # experts sharded over the ranks (world_size > 1, utils/common_config.py:179-185): the module holds this rank's
        # moe_experts // world_size experts, the executor exchanges the routed rows (engine._experts_fwd_ep).  The exchange
        # reads its split sizes on the host, so these passes run eagerly (no hipGraph)
        self.ep = {}
        if model.world_size > 1:
            import torch.distributed as dist
            group = next((blk.mlp.moe_group for blk in model.blocks if blk.moe), None)
            self.ep = dict(ep_group=group, ep_world=model.world_size, ep_rank=dist.get_rank(group), experts_are_local=True)
            assert dist.get_world_size(group) == model.world_size, "the layer's world_size must be the expert group's size"
            self.graph = False
        eng0 = BackboneEngine(cfg, params, batch=B, dtype=model.act_dtype, device=str(device),
                              checkpoint=bool(model.use_checkpointing), **self.ep)
        for n, p in named.items():
            if eng0.params[n].data_ptr() != p.data_ptr():
                raise RuntimeError(f"fused backbone: parameter {n} must be a contiguous fp32 CUDA tensor")
        self.cfg, self.batch, self.device = cfg, B, device
        self.base_eng = eng0                     # owns the operand copies every context shares
        self.slot_sets = {}                      # batch size -> its contexts (a smaller last batch of an epoch comes back)
        self.slots = [_Slot(0, eng0, device)]
        self.plist = [named[n] for n in eng0.params]                 # in the flat buffer's order
        # the buffer the parameters' .grad alias: the sum over the slots' own gradient buffers (same layout)
        self.gsum = torch.zeros_like(eng0.flat_grads)
        self.gstream = torch.cuda.Stream(device=device)
        self.views, o = [], 0
        for n, p in eng0.params.items():
            self.views.append(self.gsum[o:o + p.numel()].view_as(p))
            o += p.numel()
        self.view_ptrs = [v.data_ptr() for v in self.views]
        self.ptrs = [p.data_ptr() for p in self.plist]
        self.anchor = torch.zeros((), device=device, requires_grad=True)
        self.drop = {i: float(getattr(blk.drop_path, "drop_prob", 0.0)) for i, blk in enumerate(model.blocks)}
        self.drop = {i: p for i, p in self.drop.items() if p > 0.0}
        self.moe_blocks = [i for i in range(cfg.depth) if cfg.is_moe(i)]
        self.dirty = True

    def _switch_batch(self, B):
        """another batch size (the short last batch of an epoch): its own contexts - activations are sized by the batch - on
        the SAME parameters, operand copies and gradient buffer; the contexts of the sizes seen before are kept (three sizes
        at most), so that going back to the usual size costs nothing"""
        for slot, _, _ in self.e_spec.values():           # evaluation passes started ahead and never asked for
            slot.busy = False
        self.e_spec, self.e_hist, self.e_first = {}, [], None
        if self.spec:
            self._drop_prefetched()
            import gc
            gc.collect()                                  # (their autograd nodes hold the contexts until collected)
        if any(s.busy for s in self.slots):
            raise RuntimeError("fused backbone: batch size changed while a forward waits for its backward")
        self.slot_sets[self.batch] = self.slots
        slots = self.slot_sets.pop(B, None)
        if slots is None:
            e0 = self.base_eng
            eng = e0 if e0.B == B else BackboneEngine(self.cfg, None, batch=B, dtype=e0.dt, device=str(self.device), share=e0,
                                                      checkpoint=e0.checkpoint, **self.ep)
            slots = [_Slot(0, eng, self.device)]
        while len(self.slot_sets) > 2:
            self.slot_sets.pop(next(iter(self.slot_sets)))
        self.slots, self.batch = slots, B
        self.hist, self.hist_same, self.pattern = [], True, None

    def _slot(self):
        for s in self.slots:
            if not s.busy:
                return s
        if len(self.slots) >= self.max_slots:
            raise RuntimeError(f"fused backbone: {self.max_slots} forward passes are waiting for their backward; "
                               "call backward() (or drop the outputs) before running more")
        e0 = self.base_eng
        eng = BackboneEngine(self.cfg, None, batch=self.batch, dtype=e0.dt, device=str(self.device), share=e0,
                             checkpoint=e0.checkpoint, **self.ep)
        s = _Slot(len(self.slots), eng, self.device)
        self.slots.append(s)
        return s

    def _check_params(self):
        """the executor reads the parameters' storage in place: a re-allocated parameter (model.to(), p.data = ...) needs
        new contexts; a changed value (optimizer.step(), load_state_dict()) needs fresh operand copies"""
        if [p.data_ptr() for p in self.plist] != self.ptrs:
            self.slots, self.batch = [], None
            return False
        sig = [p._version for p in self.plist]
        if sig != self.sig:
            self.sig, self.dirty, self.versions_changed = sig, True, True
        return True

    # ------------------------------------------------------------------ forward
    def forward(self, images, task_id):
        model = self.model()
        B = images.shape[0]
        dev = images.device
        if not self.slots or dev != self.device or not self._check_params():
            if any(s.busy for s in self.slots):
                raise RuntimeError("fused backbone: parameter storage changed while a forward waits for its backward")
            self._build(B, dev)
            self._check_params()
        elif self.batch != B:
            self._switch_batch(B)
        train = torch.is_grad_enabled()
        main = torch.cuda.current_stream()
        refreshed = self.versions_changed
        self.versions_changed = False
        if self.dirty and (refreshed or not self.spec):
            # first forward after a backward (an optimizer step came in between) or after the values changed: refresh the
            # activation-dtype operand copies W, W^T of every Linear - one launch, ~0.1 ms (engine.prepare_weights).  (Not
            # while passes started ahead are reading the copies, unless the parameters really changed - then those passes
            # are dropped below anyway)
            if refreshed and self.spec:
                self._drop_prefetched()
                torch.cuda.current_stream().wait_stream(self.gstream)
                for sl in self.slots:
                    torch.cuda.current_stream().wait_stream(sl.stream)
            self.base_eng.prepare_weights()
            self.dirty = False
        if not train:
            return self._forward_nograd(images, task_id, model, main, refreshed)
        if self.e_spec:                                  # evaluation passes nobody claimed keep their contexts no longer
            for sl, _, _ in self.e_spec.values():
                sl.busy = False
            self.e_spec = {}
        # a new step begins with the first training forward after the parameters changed (an optimizer step), or after a
        # backward when this call cannot belong to the calls before it (other images, or a task that was already run): so
        # both the joint schedule (all forwards, one backward) and one task at a time (forward / backward per task on the
        # same images, train/train_utils.py:373-404) are seen as ONE step of [t0, t1, ..]
        first = self.hist[0][1]() if self.hist else None
        new_step = (not self.hist) or refreshed or (self.backward_seen and (first is not images or
                                                                            task_id in [t for t, _ in self.hist]))
        self.backward_seen = False
        if new_step:
            self._close_step()
        # a pass that was started ahead of its call (see _prefetch): hand it over
        hit = self.spec.pop(task_id, None) if (self.spec and images is self.spec_images and
                                               images._version == self.spec_version) else None
        if hit is None and self.spec:
            self._drop_prefetched()                   # the caller did something else than last step: stop predicting for a while
        # (whether the step's calls share ONE image tensor is noted now, while the tensor is alive: by the time the step is
        # closed - at the next step's first call - a trainer has usually dropped the batch)
        self.hist_same = self.hist_same and (not self.hist or self.hist[0][1]() is images)
        self.hist.append((task_id, weakref.ref(images)))
        if not self.spec:
            self.spec_images = None
        if hit is not None:
            slot, tok, cv = hit
            slot.main = main
            self.prefetch_hits += 1
        else:
            slot = self._launch(task_id, images, main)
            tok, cv = slot.result
            slot.result = None
            if new_step:
                self._prefetch(task_id, images, main)
        main.wait_stream(slot.stream)
        tok.record_stream(main)
        cv.record_stream(main)
        return tok, cv

    # ------------------------------------------------------------------ forward without autograd (evaluation)
    def _nograd_pass(self, slot, task_id, images, model):
        """one forward on `slot`, on the current stream: eager at its first use, a hipGraph replay from then on (an evaluation
        loop is ~100 launches a call)"""
        key = (task_id, bool(model.training))
        n = slot.calls_e.get(key, 0)
        slot.calls_e[key] = n + 1
        if not self.graph or n == 0:
            tok, cv = self._forward_eager(slot, task_id, images.float().contiguous(), False)
            slot.py_owner = ("eval", key)
        else:
            if slot.images is None:
                slot.images = torch.empty(images.shape, dtype=torch.float32, device=self.device)
            slot.images.copy_(images)
            noises, ps = self._draw(slot, slot.eng)
            if key not in slot.graphs_e:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    out = slot.eng.forward(slot.images, task_id, noises=noises, path_scales=ps)
                slot.graphs_e[key] = (g, out)
                slot.py_owner = ("eval", key)
            g, (tok, cv) = slot.graphs_e[key]
            g.replay()
        return tok.clone(), (tok.new_zeros(()) if not model.training else cv.clone())

    def _forward_nograd(self, images, task_id, model, main, refreshed=False):
        """Evaluation (model.eval() under torch.no_grad()) or a no_grad probe: every call runs on a free context's stream and
        the caller's stream waits for it.  An evaluation loop calls the backbone once per task on the same batch
        (models/models.py:299-320) just like training: when the previous batch saw tasks [t0, t1, ..] on one tensor, the first
        call on a new batch starts the other tasks' passes as well (same rule, same fall-back as _prefetch)."""
        first = self.e_first() if self.e_first is not None else None
        if first is not images:                          # a new batch: what was called on the last one is the prediction
            tasks = self.e_hist
            self.e_pattern = tasks if (len(tasks) >= 2 and len(set(tasks)) == len(tasks)) else None
            for slot, _, _ in self.e_spec.values():      # (passes nobody asked for)
                slot.busy = False
            if self.e_spec:
                self.e_backoff = self.PREFETCH_BACKOFF
            self.e_spec, self.e_hist, self.e_first = {}, [], weakref.ref(images)
            if self.e_backoff > 0:
                self.e_backoff -= 1
            new_batch = True
        else:
            new_batch = False
        self.e_hist.append(task_id)
        if refreshed and self.e_spec:                    # the parameters changed since those passes were started: stale
            for slot, _, _ in self.e_spec.values():
                slot.busy = False
            self.e_spec = {}
        hit = self.e_spec.pop(task_id, None) if images._version == self.e_version else None
        if hit is None:
            if self.e_spec:                              # the batch tensor was written to: started passes are stale
                for slot, _, _ in self.e_spec.values():
                    slot.busy = False
                self.e_spec = {}

            def start(t):
                slot = self._slot()
                slot.busy = True
                slot.stream.wait_stream(main)
                with torch.cuda.stream(slot.stream):
                    tok, cv = self._nograd_pass(slot, t, images, model)
                return slot, tok, cv
            slot, tok, cv = start(task_id)
            if new_batch and self.prefetch and self.e_backoff == 0 and self.e_pattern and self.e_pattern[0] == task_id:
                self.e_version = images._version
                for t in self.e_pattern[1:]:
                    self.e_spec[t] = start(t)
        else:
            slot, tok, cv = hit
            self.prefetch_hits += 1
        main.wait_stream(slot.stream)
        tok.record_stream(main)
        cv.record_stream(main)
        slot.busy = False
        return tok, cv

    def _launch(self, task_id, images, main):
        """start one task pass on a free slot's stream, ordered behind everything queued on `main` so far; main is NOT made to
        wait for it here"""
        slot = self._slot()
        slot.busy, slot.main = True, main
        s = slot.stream
        s.wait_stream(main)
        try:
            with torch.cuda.stream(s):
                slot.result = _BackboneFn.apply(self.anchor, self, slot, task_id, images)
        except BaseException:
            slot.busy = False                         # a failed launch must not keep the context
            raise
        return slot

    # ---- task passes started ahead of their call
    # The reference's multi-task forward calls the backbone once per task on the SAME images (models/models.py:299-320), and
    # between two such calls the caller's stream has to wait for the pass just returned (its tokens are consumed right
    # away), so pass t + 1 cannot start before pass t is done: the forward passes of a step run one after the other where
    # m3vit_amd.step.MultiTaskStep runs them side by side.  When the previous step called forward(x, t0), forward(x, t1), ..
    # on one image tensor, the first call of this step also starts the passes of t1, .. on their own streams (they depend on
    # x and the parameters only); their calls then find the result under way.  A call that does not match (other images,
    # other task order) drops what was started - wasted GPU time, never a wrong result - and switches the prediction off
    # for the next PREFETCH_BACKOFF steps.
    PREFETCH_BACKOFF = 16

    def _close_step(self):
        """first training forward after a backward: the calls since the last boundary were one step"""
        # The end-of-backward callback (_join_caller) is queued once per backward pass and clears its own latch - unless
        # the pass raised (an out-of-memory in a head's backward ...): autograd then never runs the callback, the latch would
        # stay set and no later backward would make the caller's stream wait for the gradient stream.  A step boundary
        # clears it, and - as a backstop for exactly that case - joins the gradient stream here (a no-op when the callback ran).
        if self._join_queued:
            self._join_queued = False
            torch.cuda.current_stream().wait_stream(self.gstream)
        if self.spec:
            self._drop_prefetched()
        tasks = [t for t, _ in self.hist]
        same = len(self.hist) >= 2 and self.hist_same and len(set(tasks)) == len(tasks)
        self.pattern = tasks if same else None
        self.hist, self.hist_same = [], True
        if self.backoff > 0:
            self.backoff -= 1

    def _drop_prefetched(self):
        self.spec.clear()                             # the nodes die with their outputs; _Release frees the slots
        self.spec_images = None
        self.backoff = self.PREFETCH_BACKOFF
        self.prefetch_misses += 1

    def _prefetch(self, task_id, images, main):
        if not self.prefetch or self.backoff > 0 or not self.pattern or self.pattern[0] != task_id:
            return
        self.spec_images, self.spec_version = images, images._version
        for t in self.pattern[1:]:
            slot = self._launch(t, images, main)
            tok, cv = slot.result
            slot.result = None
            self.spec[t] = (slot, tok, cv)

    def _draw(self, slot, eng):
        """per-step random inputs of the pass, drawn outside the graphs into the slot's static buffers: gate noise
        (noisy_gate_vmoe.py:168: randn_like(clean) * std, training only) and DropPath factors
        (vision_transformer_moe.py:167-185: floor(keep + U) / keep per sample and residual branch)"""
        model = self.model()
        noises = ps = None
        if self.cfg.vmoe_noisy_std > 0 and model.training:
            if slot.noises is None:
                slot.noises = {i: torch.empty(eng.T, eng.E, device=self.device) for i in self.moe_blocks}
            for n in slot.noises.values():
                n.normal_()
            noises = slot.noises
        if self.drop and model.training:
            if slot.path_scales is None:
                slot.path_scales = {i: (torch.empty(self.batch, device=self.device), torch.empty(self.batch, device=self.device))
                                    for i in self.drop}
            for i, (sa, sm) in slot.path_scales.items():
                keep = 1.0 - self.drop[i]
                for t in (sa, sm):
                    t.uniform_().add_(keep).floor_().div_(keep)
            ps = slot.path_scales
        return noises, ps

    def _forward_eager(self, slot, task_id, images, zero):
        eng = slot.eng
        noises, ps = self._draw(slot, eng)
        if zero:
            eng.zero_grad()
        return eng.forward(images, task_id, noises=noises, path_scales=ps)

    def _run_forward(self, slot, task_id, images):
        eng = slot.eng
        zero = True                               # the slot's own buffer holds this pass only; .grad aliases the sum (self.gsum)
        if slot.add_done is not None:             # the previous add of this slot's buffer into the sum has read it
            torch.cuda.current_stream().wait_event(slot.add_done)
        n = slot.calls_f.get(task_id, 0)
        slot.calls_f[task_id] = n + 1
        if not self.graph or n == 0:
            img = images.float().contiguous()
            img.record_stream(torch.cuda.current_stream())
            tok, cv = self._forward_eager(slot, task_id, img, zero)      # (the backward re-reads eng.rows, not the images)
            slot.py_owner = ("eager", task_id)
            return tok.clone(), cv
        if slot.images is None:
            slot.images = torch.empty(images.shape, dtype=torch.float32, device=self.device)
        images.record_stream(torch.cuda.current_stream())
        slot.images.copy_(images)
        self._draw(slot, eng)
        g = slot.graphs_f.get(task_id)
        if g is not None and task_id not in slot.graphs_b and slot.py_owner != ("fcap", task_id):
            # The backward graph of this (slot, task) is still to be captured, and capturing it runs eng.backward at Python
            # level: it reads the engine's per-call tensors (gate outputs, routes) by reference - those of the LAST Python-level
            # eng.forward on this slot.  That was this task's forward capture only if nothing else ran on the slot since (a
            # dropped prefetched pass, another task, an evaluation pass): if not, the replayed forward would write the old
            # addresses while the new backward graph reads another forward's tensors.  Capture the forward again.
            slot.graphs_f.pop(task_id)
            slot.out.pop(task_id, None)
            g = None
        if g is None:
            noises = slot.noises if (self.cfg.vmoe_noisy_std > 0 and self.model().training) else None
            ps = slot.path_scales if (self.drop and self.model().training) else None
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                if zero:
                    eng.zero_grad()
                out = eng.forward(slot.images, task_id, noises=noises, path_scales=ps)
            slot.graphs_f[task_id] = g
            slot.out[task_id] = out
            slot.py_owner = ("fcap", task_id)
        g.replay()
        tok, cv = slot.out[task_id]
        return tok.clone(), cv.clone()

    # ------------------------------------------------------------------ backward
    def _install_grads(self):
        """Make every parameter's .grad the view of the sum buffer, with torch's accumulation semantics: a parameter
        whose .grad is None starts from zero, one that already holds the view accumulates in place, one that holds another
        tensor (assigned by the trainer or another wrapper) contributes that tensor's value.  All on the gradient stream -
        the one stream that ever writes the buffer."""
        plist, views, vptrs = self.plist, self.views, self.view_ptrs
        fresh, foreign = [], []
        for i, p in enumerate(plist):
            if not p.requires_grad:                   # frozen parameters keep .grad = None, as on the per-op path: an optimizer
                continue                              # then skips them (no weight decay / momentum on a zero gradient)
            g = p.grad
            if g is None:
                fresh.append(i)
            elif g.data_ptr() != vptrs[i] or g.dtype != torch.float32:
                foreign.append(i)
        if not fresh and not foreign:
            return
        flat = self.gsum
        gs = self.gstream
        with torch.cuda.stream(gs):
            if len(fresh) == len(plist):
                flat.zero_()
            else:
                if fresh:
                    torch._foreach_zero_([views[i] for i in fresh])
                if foreign:
                    for i in foreign:
                        plist[i].grad.record_stream(gs)
                    torch._foreach_copy_([views[i] for i in foreign], [plist[i].grad for i in foreign])
        for i in fresh + foreign:
            plist[i].grad = views[i]

    def _run_backward(self, slot, task_id, g_tok, g_cv):
        eng = slot.eng
        cur = torch.cuda.current_stream()             # autograd: the stream of the node's forward = slot.stream
        gs = self.gstream
        if slot.main is not None and cur != slot.stream:
            # (a caller that ran backward under another stream context: order this pass behind the slot's own stream)
            cur.wait_stream(slot.stream)
        # the trainer may have consumed / zeroed the gradients on its own stream since the last backward
        gs.wait_stream(slot.main)
        self._install_grads()
        if slot.dtok is None:
            slot.dtok = torch.zeros(eng.B, eng.N, eng.D, dtype=torch.float32, device=self.device)
        if g_tok is None:
            slot.dtok.zero_()
        else:
            g_tok.record_stream(cur)
            slot.dtok.copy_(g_tok.reshape(slot.dtok.shape))
        if g_cv is None:
            slot.dcv.zero_()
        else:
            g_cv.record_stream(cur)
            slot.dcv.copy_(g_cv.reshape(1))
        n = slot.calls_b.get(task_id, 0)
        slot.calls_b[task_id] = n + 1
        # a backward graph reads the forward graph's tensors by address: only behind a replayed forward
        replayed_fwd = self.graph and slot.calls_f.get(task_id, 0) >= 2
        if not replayed_fwd:
            eng.backward(slot.dtok, cv_weight=slot.dcv)
        else:
            g = slot.graphs_b.get(task_id)
            if g is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    eng.backward(slot.dtok, cv_weight=slot.dcv)
                slot.graphs_b[task_id] = g
            g.replay()
        gs.wait_stream(cur)
        with torch.cuda.stream(gs):
            ops.add_f32(self.gsum, eng.flat_grads)
            if slot.add_done is None:
                slot.add_done = torch.cuda.Event()
            slot.add_done.record(gs)
        # The caller's stream must see the finished gradients - but not from here: nodes that autograd runs AFTER this one
        # on the caller's stream (the loss arithmetic of the other task passes, which produces their d tokens) would then
        # wait for this pass, and the passes' backward would run one after the other.  The wait is queued as a callback
        # of the backward pass: it runs once, on the thread and stream that called backward(), when every node is done.
        if not self._join_queued:
            self._join_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._join_caller)
        slot.busy = False
        self.dirty = True
        self.backward_seen = True

    def _join_caller(self):
        self._join_queued = False
        torch.cuda.current_stream().wait_stream(self.gstream)
