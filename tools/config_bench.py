#!/usr/bin/env python3
"""One-GPU step times of the OTHER BASELINE configs' shapes (they are parity-test cases, not bench lines; this is
orientation for the next rounds).  One step = refresh operand copies + forward + backward of every task pass
(m3vit_amd.step.MultiTaskStep: one stream per task pass, hipGraph replay), fp16 activations.
    python tools/config_bench.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd.config import BackboneConfig, init_params  # noqa: E402
from m3vit_amd.step import MultiTaskStep  # noqa: E402

CASES = [
    ("configs[2] ViT-S task-conditioned, 5 tasks, 8 x 512x512", 8, dict(img_size=(512, 512), embed_dim=384, depth=12,
     num_heads=12, moe_experts=16, moe_top_k=4, gate_dim=389, multi_gate=False, gate_task_specific_dim=64), 5),
    ("configs[3] ViT-B E=64, 128 x 224x224 (all 64 experts on one GPU)", 128, dict(img_size=(224, 224), embed_dim=768,
     depth=12, num_heads=12, moe_experts=64, moe_top_k=4, gate_dim=768, multi_gate=False), 1),
    ("configs[4] ViT-B E=16 moe_mlp_ratio 4, 2 tasks, 8 x 480x640", 8, dict(img_size=(480, 640), embed_dim=768, depth=12,
     num_heads=12, moe_experts=16, moe_top_k=4, moe_mlp_ratio=4.0, gate_dim=770, multi_gate=True), 2),
]
for name, B, kw, ntasks in CASES:
    cfg = BackboneConfig(**kw)
    tasks = list(range(ntasks)) if (cfg.multi_gate or cfg.gate_task_specific_dim >= 0) else [None]
    run = MultiTaskStep(cfg, init_params(cfg, seed=1), batch=B, dtype=torch.float16, tasks=tasks,
                        share_stem="--no-share-stem" not in sys.argv)
    run.bind(torch.randn(B, 3, *cfg.img_size).cuda(), (torch.randn(B, cfg.num_tokens, cfg.embed_dim) * 0.05).cuda())
    run.step()
    torch.cuda.synchronize()
    run.capture()
    for _ in range(2):
        run.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        run.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fl = 3.0 * cfg.fwd_flops_per_image() * B * ntasks
    print(f"{name}: {dt * 1e3:7.1f} ms/step  {B / dt:7.0f} img/s  model {fl / dt / 1e12:5.0f} TFLOP/s  ({run.launch}, "
          f"{len(run.engs)} stream(s){', shared stem' if run.share_stem else ''})", flush=True)
    del run
    torch.cuda.empty_cache()
