#!/usr/bin/env python3
"""decoder head forward + backward at 8 x 480 x 640 under fp16 autocast: where the time goes (torch profiler, top kernels),
with torch's relu + resize stages (M3_HEAD_FUSED=0) or the fused ReLU + x2 resize kernels (default)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd.heads import VisionTransformerUpHead  # noqa: E402

dev = torch.device("cuda:0")
fused = os.environ.get("M3_HEAD_FUSED", "1") != "0"
head = VisionTransformerUpHead(img_size=(480, 640), embed_dim=384, num_classes=40, amp=True, fused_resize=fused).to(dev).train()
tok = torch.randn(8, 30 * 40 + 1, 384, device=dev, requires_grad=True)


def run():
    y = head(tok)
    y.float().square().mean().backward()


for _ in range(3):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        run()
    torch.cuda.synchronize()
print(f"fused_resize={fused}: 3 iterations")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=16, max_name_column_width=64))
