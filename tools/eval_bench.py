#!/usr/bin/env python3
"""Evaluation throughput of the module path at configs[1] (model.eval(), torch.no_grad(), one backbone call per task and batch):
fused executor (hipGraph replay from the second call on) against the per-op path."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd.config import VIT_SMALL_MOE, BackboneConfig, init_params  # noqa: E402
from m3vit_amd.vit import VisionTransformerMoE  # noqa: E402

cfg = BackboneConfig(**VIT_SMALL_MOE)
base = torch.randn(128, 3, 224, 224).cuda()
for fused in ("auto", False):
    m = VisionTransformerMoE(vmoe_noisy_std=1.0, act_dtype=torch.float16, fused=fused, **VIT_SMALL_MOE).cuda().eval()
    m.load_state_dict(init_params(cfg, seed=1))
    with torch.no_grad():
        for _ in range(3):
            img = base.clone()                       # a new batch tensor every iteration, as a data loader hands them over
            for t in (0, 1):
                m(img, task_id=t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            img = base.clone()
            for t in (0, 1):
                tok, _ = m(img, task_id=t)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"fused={fused}: {dt * 1e3:.2f} ms per batch (two task passes, forward only) = {128 / dt:.0f} images/s", flush=True)
