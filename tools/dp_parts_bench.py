#!/usr/bin/env python3
"""What cutting the step into graph parts costs on one GPU (no collectives: the part graphs are replayed back to back).
    python tools/dp_parts_bench.py [--batch 128]"""
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
a = ap.parse_args()
cfg = BackboneConfig(**VIT_SMALL_MOE)
params = init_params(cfg, seed=1)
dev = torch.device("cuda:0")
img = torch.randn(a.batch, 3, *cfg.img_size, device=dev)
dtok = torch.randn(a.batch, cfg.num_tokens, cfg.embed_dim, device=dev) * 0.01
for parts in (1, 2, 4, 6, 12):
    run = MultiTaskStep(cfg, params, batch=a.batch, dtype=torch.float16, device="cuda:0", cv_weight=0.01, world=2, dp_parts=parts)
    run.bind(img, dtok)
    run.compute()
    torch.cuda.synchronize()
    assert run.capture()
    for _ in range(3):
        for g in run.graphs:
            g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        for g in run.graphs:
            g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    seg = [round((hi - lo) * 4 / 1e6, 1) for lo, hi in run.segments]
    print(f"dp_parts {parts:2d}: {ms:6.2f} ms/step   gradient slices (MB): {seg}", flush=True)
    del run
    torch.cuda.empty_cache()
