#!/usr/bin/env python3
"""Decoder head (VisionTransformerUpHead, 4 conv + 4 upsample stages) at the NYUD resolution (8 images, 480 x 640, D = 384,
40 classes): forward + backward time of the torch / MIOpen stages, fp32 and under fp16 autocast, per stage."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd.heads import VisionTransformerUpHead  # noqa: E402

dev = torch.device("cuda:0")
B, D = 8, 384
FUSED = os.environ.get("M3_HEAD_FUSED", "1") != "0"       # ReLU + x2 resize of a stage as one hand-written kernel each way
head = VisionTransformerUpHead(img_size=(480, 640), embed_dim=D, num_classes=40, fused_resize=FUSED).to(dev).train()
print(f"fused_resize={FUSED}")
tok = torch.randn(B, 30 * 40 + 1, D, device=dev, requires_grad=True)


def run(amp):
    with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        y = head(tok)
    y.float().square().mean().backward()


for amp in (False, True):
    for _ in range(2):
        run(amp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        run(amp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fl = 0
    for (h, w, ci) in ((30, 40, 384), (60, 80, 256), (120, 160, 256), (240, 320, 256)):
        fl += 2 * B * h * w * 256 * 9 * ci
    print(f"{'fp16 autocast' if amp else 'fp32'}: {dt * 1e3:.1f} ms fwd+bwd  (3x3 conv FLOPs fwd {fl / 1e9:.0f} G, x3 for fwd+bwd: {3 * fl / dt / 1e12:.0f} TFLOP/s)", flush=True)
# per conv at the largest stage
x = torch.randn(B, 256, 240, 320, device=dev)
conv = head.conv_3
for amp, xx in ((False, x), (True, x.half())):
    cv = conv if not amp else conv.half()
    for _ in range(2):
        cv(xx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        cv(xx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"conv_3 forward alone {'fp16' if amp else 'fp32'} NCHW: {dt * 1e3:.2f} ms = {2 * B * 240 * 320 * 256 * 2304 / dt / 1e12:.0f} TFLOP/s", flush=True)
xh = x.half().to(memory_format=torch.channels_last)
cvh = conv.half().to(memory_format=torch.channels_last)
for _ in range(2):
    cvh(xh)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    cvh(xh)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"conv_3 forward alone fp16 channels_last: {dt * 1e3:.2f} ms = {2 * B * 240 * 320 * 256 * 2304 / dt / 1e12:.0f} TFLOP/s", flush=True)
