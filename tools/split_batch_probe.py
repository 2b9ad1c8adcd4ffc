#!/usr/bin/env python3
"""Would MORE, smaller concurrent passes beat two full-batch task streams?  Probe only (the balance loss of a task pass is
over its whole batch, which split passes would have to combine): the same total work - 2 tasks x 128 images - run as
2 streams x 128 images, 4 streams x 64 images and 8 streams x 32 images of MultiTaskStep (tasks repeated).
    python tools/split_batch_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd.config import VIT_SMALL_MOE, BackboneConfig, init_params  # noqa: E402
from m3vit_amd.step import MultiTaskStep  # noqa: E402

cfg = BackboneConfig(**VIT_SMALL_MOE)
P = init_params(cfg, seed=1)
for split in (1, 2, 4):
    B = 128 // split
    run = MultiTaskStep(cfg, P, batch=B, dtype=torch.float16, tasks=[0, 1] * split)
    run.bind(torch.randn(B, 3, 224, 224).cuda(), (torch.randn(B, cfg.num_tokens, cfg.embed_dim) * 0.05).cuda())
    run.step(); torch.cuda.synchronize()
    ok = run.capture()
    for _ in range(5):
        run.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        run.step()
    torch.cuda.synchronize()
    print(f"{2 * split} streams x {B} images: {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step (graph {ok})", flush=True)
    del run
    torch.cuda.empty_cache()
