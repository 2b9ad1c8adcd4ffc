#!/usr/bin/env python3
"""Phase timeline of the fused FFN forward from in-kernel s_memtime stamps (diagnostic build of the library).

    (cd m3vit_amd/csrc && make CXXFLAGS="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -DM3_FFN_STAMPS" \
         OBJDIR=../../build/fstamps OUT=../../build/fstamps/libm3vit_hip.so)
    M3VIT_LIB=$PWD/build/fstamps/libm3vit_hip.so python tools/ffn_stamps.py [expert|dense]

Operands streamed (ring of (X, Y) pairs); the last launch is analysed: median ticks (= shader cycles) per phase of a
workgroup and the share of its life."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import _lib, ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "expert"
dev = torch.device("cuda:0")
T, D, E, k, ring = 128 * 197, 384, 16, 4, 6


def perm32(n):
    p = torch.arange(n)
    w = p % 32
    return (p - w) + 16 * ((w & 7) >> 2) + 4 * (w >> 3) + (w & 3)


if mode == "expert":
    H, R = 384, T * k
    Xs = [torch.randn(T, D, device=dev).half() for _ in range(ring)]
    Ys = [torch.empty(R, D, dtype=torch.float16, device=dev) for _ in range(ring)]
    w1 = (torch.randn(E, H, D, device=dev) * 0.05).half()
    w2p = (torch.randn(E, D, H, device=dev) * 0.05).half()
    b1, b2 = torch.zeros(E, H, device=dev), torch.zeros(E, D, device=dev)
    idx = torch.stack([torch.randperm(E)[:k] for _ in range(T)]).to(torch.int32).to(dev)
    r = ops.route_build(idx, E)
    run = lambda i: ops.ffn_fwd(Xs[i % ring], w1, w2p, Ys[i % ring], b1=b1, b2=b2, M=R, x_row_idx=r.row_of_slot,   # noqa: E731
                                x_row_div=k, y_row_idx=r.row_of_slot, group_offsets=r.offsets)
    wgs = (R + 127) // 128 + E
else:
    H = 1536
    Xs = [torch.randn(T, D, device=dev).half() for _ in range(ring)]
    Rs = [torch.randn(T, D, device=dev) for _ in range(ring)]
    Ys = [torch.empty(T, D, device=dev) for _ in range(ring)]
    w1 = (torch.randn(H, D, device=dev) * 0.05).half()
    w2p = (torch.randn(D, H, device=dev) * 0.05).half()
    b1, b2 = torch.zeros(H, device=dev), torch.zeros(D, device=dev)
    run = lambda i: ops.ffn_fwd(Xs[i % ring], w1, w2p, Ys[i % ring], b1=b1, b2=b2, residual=Rs[i % ring])   # noqa: E731
    wgs = (T + 127) // 128

for i in range(2 * ring):
    run(i)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
run(1)
ev1.record()
torch.cuda.synchronize()
us = ev0.elapsed_time(ev1) * 1e3
L = _lib.lib()
SN = 16
buf = np.zeros((min(wgs, 2048), SN), dtype=np.uint64)
fn = L.m3_debug_ffn_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf.ctypes.data, buf.shape[0]) == 0
st = buf.astype(np.int64)
live = st[:, 6] > 0
st = st[live]
names = ["tile search, loads issued", "X, biases, slice 0 in LDS", "X fragments, row ids", "main loop", "epilogue issue",
         "store ack"]
life = st[:, 6] - st[:, 0]
print(f"{mode}: launch {us:.1f} us, {live.sum()} live workgroups, median life {np.median(life):.0f} ticks "
      f"(p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f})")
for i, n in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print(f"  {n:26s} median {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}   {100 * np.median(d) / np.median(life):5.1f} %")
# s_memtime counters of the XCCs are not aligned: cluster by entry stamp to get the span per clock domain
order = np.argsort(st[:, 0])
gaps = np.diff(st[order, 0])
cl = np.zeros(len(order), dtype=np.int64)
cl[order[1:]] = np.cumsum(gaps > 2000000)
spans = [int(st[cl == c, 6].max() - st[cl == c, 0].min()) for c in np.unique(cl)]
print(f"  clock domains {len(spans)}; span per domain (ticks): median {np.median(spans):.0f} -> {np.median(spans) / us:.0f} ticks/us")
