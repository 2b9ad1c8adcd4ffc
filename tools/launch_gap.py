#!/usr/bin/env python3
"""What a dependent kernel launch costs inside a replayed hipGraph: a chain of N tiny kernels (m3_add_f32 on 1 K floats),
one stream vs the same chain split over two forked streams."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
a = torch.zeros(1024, device=dev); b = torch.ones(1024, device=dev)
a2 = torch.zeros(1024, device=dev)
N = 2000


def chain(n, x):
    for _ in range(n):
        ops.add_f32(x, b)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


side = torch.cuda.Stream()
cap = torch.cuda.Stream()
with torch.cuda.stream(cap):
    chain(10, a); torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=cap):
        chain(N, a)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=cap):
        side.wait_stream(cap)
        chain(N // 2, a)
        with torch.cuda.stream(side):
            chain(N // 2, a2)
        cap.wait_stream(side)
torch.cuda.synchronize()
t1 = timed(g1.replay); t2 = timed(g2.replay)
print(f"graph replay, {N} dependent tiny kernels on one stream: {t1:.0f} us = {t1 / N:.2f} us per kernel")
print(f"graph replay, two forked chains of {N // 2}: {t2:.0f} us = {t2 / (N // 2):.2f} us per kernel of a chain")
te = timed(lambda: chain(N, a), reps=3)
print(f"eager, {N} launches: {te:.0f} us = {te / N:.2f} us per kernel")
