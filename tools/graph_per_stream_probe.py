#!/usr/bin/env python3
"""One hipGraph for the whole two-stream step (MultiTaskStep: the task passes fork / join INSIDE the capture) against one
LINEAR hipGraph per task pass replayed on that pass's own stream (what m3vit_amd/fused.py does): GPU time and host launch
time per step.
    python tools/graph_per_stream_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402
from m3vit_amd.config import VIT_SMALL_MOE, BackboneConfig, init_params  # noqa: E402
from m3vit_amd.step import MultiTaskStep  # noqa: E402

cfg = BackboneConfig(**VIT_SMALL_MOE)
P = init_params(cfg, seed=1)
B = 128
img = torch.randn(B, 3, 224, 224).cuda()
dtok = (torch.randn(B, cfg.num_tokens, cfg.embed_dim) * 0.05).cuda()


def timeit(fn, n=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3


run = MultiTaskStep(cfg, P, batch=B, dtype=torch.float16)
run.bind(img, dtok)
run.step(); torch.cuda.synchronize()
assert run.capture()
ms, host = timeit(run.step)
print(f"one graph, passes forked inside the capture : {ms:6.2f} ms/step   host {host:5.2f} ms/step", flush=True)

# linear graph per pass
engs, streams = run.engs, [torch.cuda.Stream() for _ in run.engs]
graphs = []
for e, t in zip(engs, run.tasks):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        e.zero_grad()
        e.forward(img, t)
        e.backward(dtok, cv_weight=0.01)
    graphs.append(g)
prep = torch.cuda.CUDAGraph()
with torch.cuda.graph(prep):
    run.eng.prepare_weights()


def step2():
    main = torch.cuda.current_stream()
    prep.replay()
    for s, g in zip(streams, graphs):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            g.replay()
    for s in streams:
        main.wait_stream(s)
    for e in engs[1:]:
        ops.add_f32(run.flat, e.flat_grads)


ref = run.flat.clone()
step2(); torch.cuda.synchronize()
print("same gradients:", torch.equal(ref, run.flat))
ms, host = timeit(step2)
print(f"one LINEAR graph per pass on its own stream : {ms:6.2f} ms/step   host {host:5.2f} ms/step", flush=True)
# the same with a host synchronisation every step (a trainer that reads the loss): the host cannot run ahead


def synced(fn):
    def f():
        fn()
        torch.cuda.synchronize()
    return f


ms, _ = timeit(synced(run.step))
print(f"one graph + a host sync per step            : {ms:6.2f} ms/step", flush=True)
ms, _ = timeit(synced(step2))
print(f"linear graphs + a host sync per step        : {ms:6.2f} ms/step", flush=True)
