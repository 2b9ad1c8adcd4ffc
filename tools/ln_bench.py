#!/usr/bin/env python3
"""LayerNorm backward at the configs[1] shape (T = 25 216 rows, D = 384, fp16 gradient in, fp32 residual-gradient
in / out + fp16 copy out), operands rotating over a ring larger than the Infinity Cache.  M3_LN_ROWS (rows per wave, read
once per process) selects the geometry:   for r in 1 2 4 8 16; do M3_LN_ROWS=$r python tools/ln_bench.py; done"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
T, D, ring = 25216, 384, 8
xs = [torch.randn(T, D, device=dev) for _ in range(ring)]
dys = [torch.randn(T, D, device=dev).half() for _ in range(ring)]
res = [torch.randn(T, D, device=dev) for _ in range(ring)]
dxs = [torch.empty(T, D, device=dev) for _ in range(ring)]
dxa = [torch.empty(T, D, device=dev, dtype=torch.float16) for _ in range(ring)]
mean = torch.randn(T, device=dev); rstd = torch.rand(T, device=dev) + 0.5
gamma = torch.randn(D, device=dev)
nblk = int(ops.lib().m3_ln_bwd_blocks(T, D))
ws = torch.empty(2, nblk, D, device=dev)


def run(n):
    for i in range(n):
        j = i % ring
        ops.layernorm_bwd(dys[j], xs[j], mean, rstd, gamma, res[j], dxs[j], None, None, ws=ws, dx_act=dxa[j])


run(2 * ring)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10 * ring
e0.record(); run(n); e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
byts = T * D * (2 + 4 + 4 + 4 + 2)
print(f"M3_LN_ROWS={os.environ.get('M3_LN_ROWS', 'default')}: {nblk} workgroups, {us:.1f} us per launch, "
      f"{byts / us / 1e6:.2f} TB/s algorithmic ({byts / 1e6:.0f} MB)")
