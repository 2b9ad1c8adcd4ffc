#!/usr/bin/env python3
"""d h2 at an MoE layer's branch point (T = 128 x 197 tokens, k = 4, D = 384, E = 16): the gather-sum pass + the K = E GEMM pass
against the fused m3_combine_gate_bwd, operands streamed through a ring larger than the Infinity Cache."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
T, k, D, E, RING = 128 * 197, 4, 384, 16, 6
dxe = [torch.randn(T * k, D, device=dev).half() for _ in range(RING)]
dl = torch.randn(T, E, device=dev)
dl_t = dl.half()
wg = torch.randn(D, E, device=dev)
wg_t = wg.half()
out = [torch.empty(T, D, device=dev) for _ in range(RING)]
ones = torch.ones(T, k, device=dev)


def timeit(fn, n=30):
    for i in range(RING):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fn(i % RING)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


a = timeit(lambda i: ops.combine_fwd(dxe[i], ones, None, out[i]))
b = timeit(lambda i: ops.gemm_nt(dl_t, wg_t, out[i], residual=out[i]))
c = timeit(lambda i: ops.combine_gate_bwd(dxe[i], k, dl, wg, out[i]))
byts = T * k * D * 2 + T * D * 4
print(f"gather-sum {a:.1f} us + gate GEMM {b:.1f} us = {a + b:.1f} us | fused {c:.1f} us ({byts / c / 1e6:.2f} TB/s algorithmic, {byts / 1e6:.0f} MB)")
