#!/usr/bin/env python3
"""d w_gate = h2^T d_logits (T = 128 x 197, D = 384, E = 16): m3_gate_bwd_params (VALU kernel + partial reduce) against the
TN MFMA GEMM on an activation-dtype copy of d_logits."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
T, D, E, RING = 128 * 197, 384, 16, 8
xs = [torch.randn(T, D, device=dev).half() for _ in range(RING)]
dl = torch.randn(T, E, device=dev)
dl_t = torch.empty(T, E, device=dev, dtype=torch.float16)
w = torch.randn(D, E, device=dev)
dw = torch.zeros(D, E, device=dev)
part = torch.empty(ops.lib().m3_gate_dw_blocks(T) * D * E, device=dev)
ws = torch.empty(ops.wgrad_ws_elems(T, D, E, 1, grouped=False, dtype=torch.float16), device=dev)


def timeit(fn, n=40):
    for i in range(RING):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fn(i % RING)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


a = timeit(lambda i: ops.gate_bwd_params(xs[i], w, dl, d_w_gate=dw, beta_dw=1, part_dw=part))


def gemm(i):
    ops.cast_f32(dl, dl_t)
    ops.wgrad_tn(xs[i], dl_t, dw, beta=1, ws=ws)


b = timeit(gemm)
print(f"gate_bwd_params (kernel + reduce) {a:.1f} us | cast + wgrad_tn + reduce {b:.1f} us")
