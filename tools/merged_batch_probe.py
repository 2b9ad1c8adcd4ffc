#!/usr/bin/env python3
"""is ONE pass over a merged 2*B batch (gate weight picked per image half; half the launches,
weight fetches and slab reduces, no gradient-buffer add) faster than the two task passes of B images on two HIP
streams?  Timing stand-in for the merged pass: a single task pass at batch 2*B (the per-half gate choice only changes
which w_gate a token row is multiplied with - same kernels, same bytes).  Same box, same process, hipGraph replay.

    python tools/merged_batch_probe.py [--batch 128] [--steps 20]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd.config import VIT_SMALL_MOE, BackboneConfig, init_params  # noqa: E402
from m3vit_amd.step import MultiTaskStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = BackboneConfig(**VIT_SMALL_MOE)
params = init_params(cfg, seed=1)


def make(batch, tasks, **kw):
    r = MultiTaskStep(cfg, params, batch=batch, dtype=torch.float16, device=str(dev), tasks=tasks, **kw)
    g = torch.Generator().manual_seed(1000)
    img = torch.randn(batch, 3, *cfg.img_size, generator=g).to(dev)
    dtok = (torch.randn(batch, cfg.num_tokens, cfg.embed_dim, generator=g) * 0.05).to(dev)
    r.bind(img, dtok)
    r.step_eager()
    torch.cuda.synchronize()
    r.capture()
    return r


def time_it(r):
    for _ in range(5):
        r.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r.step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / a.steps


B = a.batch
variants = [
    ("two task passes of B on two streams (shipped)", lambda: make(B, [0, 1])),
    ("ONE pass at 2B on one stream (merged stand-in)", lambda: make(2 * B, [0])),
    ("ONE pass at 2B + wgrad stream", lambda: make(2 * B, [0], wgrad_streams=True)),
    ("two task passes of B on one stream", lambda: make(B, [0, 1], parallel_tasks=False)),
]
runners = [(n, f()) for n, f in variants]
for rd in range(a.rounds):
    for n, r in runners:
        print(f"round {rd}  {time_it(r):7.3f} ms/step  [{r.launch}]  {n}", flush=True)
