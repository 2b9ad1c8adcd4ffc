#!/usr/bin/env python3
"""Step time of the DROP-IN module path at BASELINE configs[1]: install_fmoe_shim() + m3vit_amd.vit.VisionTransformerMoE +
torch.autograd, the reference's joint multi-task step (models/models.py:299-320: backbone(x, task_id) per task;
train/train_utils.py:423-457: one loss.backward(); optimizer.zero_grad(set_to_none=True); parameters touched in place the
way an optimizer's foreach step does).
    python tools/module_bench.py [--dtype f16|f32] [--fused auto|off] [--steps 20] [--batch 128] [--one-by-one]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# the other BASELINE configs' shapes on one GPU (parity-test cases, not bench lines: tools/config_bench.py has the executor's times)
OTHER_CONFIGS = {
    2: (8, dict(img_size=(512, 512), embed_dim=384, depth=12, num_heads=12, moe_experts=16, moe_top_k=4, gate_dim=389,
                multi_gate=False, gate_task_specific_dim=64)),
    3: (128, dict(img_size=(224, 224), embed_dim=768, depth=12, num_heads=12, moe_experts=64, moe_top_k=4, gate_dim=770,
                  multi_gate=True)),
    4: (8, dict(img_size=(480, 640), embed_dim=768, depth=12, num_heads=12, moe_experts=16, moe_top_k=4, moe_mlp_ratio=4.0,
                gate_dim=770, multi_gate=True)),
}


def module_path_step_time(dtype_name="f16", fused="auto", steps=20, warmup=5, batch=128, one_by_one=False, log=None, config=1):
    import m3vit_amd
    m3vit_amd.install_fmoe_shim()                      # the reference's `from fmoe...` imports bind to this repository
    from m3vit_amd.config import VIT_SMALL_MOE, BackboneConfig, init_params
    from m3vit_amd.vit import VisionTransformerMoE
    kw = dict(VIT_SMALL_MOE)
    if config != 1:
        batch, kw = OTHER_CONFIGS[config]
        kw = dict(mlp_ratio=4.0, moe_mlp_ratio=1.0, **kw) if "moe_mlp_ratio" not in kw else dict(mlp_ratio=4.0, **kw)
    cfg = BackboneConfig(**kw)
    dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[dtype_name]
    model = VisionTransformerMoE(vmoe_noisy_std=0.0, act_dtype=dt, fused=("auto" if fused == "auto" else False), **kw).cuda()
    model.load_state_dict(init_params(cfg, seed=1))
    model.train()
    g = torch.Generator().manual_seed(1000)
    images = torch.randn(batch, 3, *cfg.img_size, generator=g).cuda()
    dtok = (torch.randn(batch, cfg.num_tokens, cfg.embed_dim, generator=g) * 0.05).cuda()
    params = [p for p in model.parameters()]
    tasks = list(range(cfg.num_tasks))

    def step():
        for p in params:                               # optimizer.zero_grad(set_to_none=True), train/train_utils.py:360
            p.grad = None
        if one_by_one:                                 # train/train_utils.py:373-404
            for t in tasks:
                tok, cv = model(images, task_id=t)
                ((tok * dtok).sum() + 0.01 * cv).backward()
        else:
            loss = 0.0
            for t in tasks:
                tok, cv = model(images, task_id=t)
                loss = loss + (tok * dtok).sum() + 0.01 * cv
            loss.backward()
        with torch.no_grad():                          # stands for optimizer.step(): every parameter written in place
            torch._foreach_mul_(params, 1.0)

    for _ in range(max(warmup, 3)):                    # first use eager, second use captures the hipGraphs
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / steps
    if log:
        log(f"module path {dtype_name} fused={fused}: host {1e3 * t_host / steps:.2f} ms/step, {1e3 * dt_s:.2f} ms/step")
    reason = model.fused_fallback_reason
    return {"value": round(batch / dt_s, 2), "ms_per_step": round(1e3 * dt_s, 3),
            "host_ms_per_step": round(1e3 * t_host / steps, 3),
            "model_tflops": round(3.0 * cfg.fwd_flops_per_image() * batch * len(tasks) / dt_s / 1e12, 2),
            "path": ("one autograd node per backbone call on the fused executor (m3vit_amd/fused.py), hipGraph replay"
                     if (fused == "auto" and reason is None) else "per-op autograd Functions (m3vit_amd/functional.py)"),
            "schedule": "one task at a time" if one_by_one else "joint: all task forwards, one backward"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--fused", default="auto", choices=["auto", "off"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--one-by-one", action="store_true")
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4], help="BASELINE configs[N] shapes on one GPU")
    a = ap.parse_args()
    import json
    print(json.dumps(module_path_step_time(a.dtype, a.fused, a.steps, 5, a.batch, a.one_by_one,
                                           log=lambda m: print(m, file=sys.stderr, flush=True), config=a.config)))
