#!/usr/bin/env python3
"""One-shot probe: capture a step whose weight-gradient GEMMs run on their own HIP stream (BackboneEngine(wgrad_stream=
True): fork by event from the capturing stream, join by wait_stream) into a hipGraph, replay it and compare with the
eager gradients.  Prints the HIP error text if the capture fails.  Run it in its own process (a hard fault in the
runtime must not take a test session down):   python tools/wgrad_capture_probe.py [--batch 4]
The NESTED pattern (task streams x wgrad streams) has its own engine-free reproducer: tools/nested_capture_probe.py
(result: profiles/r03_nested_capture_probe.txt - it faults in hipStreamEndCapture with plain torch streams)."""
import argparse
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd.engine import BackboneEngine  # noqa: E402
from m3vit_amd.config import BackboneConfig, init_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
a = ap.parse_args()
cfg = BackboneConfig(img_size=(64, 64), embed_dim=64, depth=4, num_heads=2, mlp_ratio=4.0, moe_mlp_ratio=1.0,
                    moe_experts=8, moe_top_k=2, gate_dim=66, multi_gate=True)
P = init_params(cfg, seed=4)
torch.manual_seed(9)
img = torch.randn(a.batch, 3, 64, 64).cuda()
dtok = (torch.randn(a.batch, cfg.num_tokens, 64) * 0.1).cuda()
eng = BackboneEngine(cfg, P, batch=a.batch, dtype=torch.float16, wgrad_stream=True)


def step():
    eng.zero_grad()
    eng.forward(img, 1)
    eng.backward(dtok, cv_weight=0.01)


step()
torch.cuda.synchronize()
want = eng.flat_grads.clone()
print("eager step with a wgrad stream: ok", flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        step()
    print("capture: ok", flush=True)
    for _ in range(3):
        eng.flat_grads.fill_(5.0)
        g.replay()
        torch.cuda.synchronize()
        err = float((eng.flat_grads - want).norm() / want.norm())
        print(f"replay: rel diff to the eager gradients {err:.2e}", flush=True)
        assert err < 1e-5
    print("PROBE RESULT: capture + replay of the wgrad-stream pattern works", flush=True)
except Exception:
    traceback.print_exc()
    print("PROBE RESULT: capture failed (text above)", flush=True)
    sys.exit(3)
