#!/usr/bin/env python3
"""What each kernel class really costs INSIDE the two-stream, graph-replayed training step: the step is captured with
that class's launches skipped (the C entry points return at once; results are garbage, only the time is read) and
timed against the full step - same process, same box.  The serial kernel trace (profiles/*_serial_kernel_stats.csv) says
how long a kernel runs by itself; this says how much of that is NOT hidden under the other task stream's kernels.
Only entry points whose outputs are plain data are skipped (never the gate / routing kernels, whose outputs are indices).

    python tools/ablate_step.py [--steps 20] [--rounds 2]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import _lib  # noqa: E402
from m3vit_amd.config import VIT_SMALL_MOE, BackboneConfig, init_params  # noqa: E402
from m3vit_amd.step import MultiTaskStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--serial", action="store_true", help="one stream instead of two")
ap.add_argument("--only", default="", help="comma-separated substrings: run only the classes whose name contains one (plus the full step)")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = BackboneConfig(**VIT_SMALL_MOE)
params = init_params(cfg, seed=1)
L = _lib.lib()

CLASSES = {
    "full step": [],
    "m3_gemm_nt": ["m3_gemm_nt"],
    "m3_wgrad_tn": ["m3_wgrad_tn"],
    "wgrad slab reduces": ["m3_wgrad_reduce", "m3_wgrad_reduce_grouped", "m3_wgrad_bias_reduce"],
    "layernorm_bwd (+ batched reduce)": ["m3_layernorm_bwd", "m3_layernorm_bwd_reduce"],
    "layernorm_fwd": ["m3_layernorm_fwd"],
    "attention_bwd": ["m3_attention_bwd"],
    "attention_fwd": ["m3_attention_fwd"],
    "combine_fwd": ["m3_combine_fwd"],
    "combine_bwd": ["m3_combine_bwd"],
    "cast_batch (weight copies)": ["m3_cast_batch"],
    "add_f32 (gradient buffers)": ["m3_add_f32"],
    "balance + gate_bwd_logits + cast_f32": ["m3_balance_loss", "m3_gate_bwd_logits", "m3_cast_f32"],
    "gate_bwd_logits alone": ["m3_gate_bwd_logits"],
    "combine_gate_bwd": ["m3_combine_gate_bwd"],
    "im2row + assemble + tokens_bwd": ["m3_im2row", "m3_assemble_tokens", "m3_tokens_bwd"],
}
# finer cuts of m3_gemm_nt by launch shape: (label, predicate on the m3_gemm_args struct)
GEMM_CUTS = {
    "m3_gemm_nt, dense N = 384 launches only (591 workgroups: proj fwd/dgrad, qkv dgrad, fc2 fwd, fc1 dgrad)":
        lambda g: g.N == 384 and not g.group_offsets,
    "m3_gemm_nt, expert grouped launches only": lambda g: bool(g.group_offsets),
    "m3_gemm_nt, dense N >= 1152 launches only (qkv fwd, fc1 fwd, fc2 dgrad)": lambda g: g.N >= 1152 and not g.group_offsets,
}
for label in GEMM_CUTS:
    CLASSES[label] = ["m3_gemm_nt:" + label]
orig = {n: getattr(L, n) for names in CLASSES.values() for n in names if ":" not in n}
orig.setdefault("m3_gemm_nt", getattr(L, "m3_gemm_nt"))


def build(skip):
    for n, f in orig.items():
        setattr(L, n, f)
    r = MultiTaskStep(cfg, params, batch=128, dtype=torch.float16, device=str(dev), parallel_tasks=not a.serial)
    g = torch.Generator().manual_seed(1000)
    img = torch.randn(128, 3, *cfg.img_size, generator=g).to(dev)
    dtok = (torch.randn(128, cfg.num_tokens, cfg.embed_dim, generator=g) * 0.05).to(dev)
    r.bind(img, dtok)
    r.step_eager()                       # a COMPLETE step first: every buffer a skipped kernel would write keeps realistic
    torch.cuda.synchronize()             # (stale) values - zeros or NaNs downstream would change the clock the chip holds
    for n in skip:
        if ":" in n:                     # a shape cut of m3_gemm_nt: skip the launches the predicate picks
            pred, real = GEMM_CUTS[n.split(":", 1)[1]], orig["m3_gemm_nt"]
            setattr(L, "m3_gemm_nt", lambda args, stream, pred=pred, real=real: 0 if pred(args._obj) else real(args, stream))
        else:
            setattr(L, n, lambda *args: 0)
    assert r.capture()                   # (capture's own warm-up run and the captured graph lack the skipped launches)
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


res = {}
for name, skip in CLASSES.items():
    if a.only and name != "full step" and not any(k in name for k in a.only.split(",")):
        continue
    r = build(skip)
    res[name] = min(time_it(r) for _ in range(a.rounds))
    del r
    torch.cuda.empty_cache()
    full = res["full step"]
    print(f"{res[name]:7.3f} ms/step   saves {full - res[name]:6.3f} ms   without: {name}", flush=True)
