#!/usr/bin/env python3
"""decoder head forward + backward at 480 x 640, fp16 autocast: NCHW against channels_last, and where the time goes
(torch profiler, top kernels)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd.heads import VisionTransformerUpHead  # noqa: E402

dev = torch.device("cuda:0")
B, D = 8, 384
tok = torch.randn(B, 30 * 40 + 1, D, device=dev, requires_grad=True)
for cl in (False, True):
    head = VisionTransformerUpHead(img_size=(480, 640), embed_dim=D, num_classes=40, amp=True).to(dev).train()
    if cl:
        head = head.to(memory_format=torch.channels_last)
        orig = head._stages
        head._stages = lambda x, o=orig: o(x.contiguous(memory_format=torch.channels_last))

    def run():
        y = head(tok)
        y.float().square().mean().backward()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    print(f"channels_last={cl}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms fwd+bwd", flush=True)
    if not cl:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(3):
                run()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
