#!/usr/bin/env python3
"""Which role the weight-stationary GEMM waits for: barrier-arrival stamps of its two roles (diagnostic build).

    (cd m3vit_amd/csrc && make CXXFLAGS="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -DM3_GEMM_STAMPS" \
         OBJDIR=../../build/stamps OUT=../libm3vit_hip_stamps.so)
    M3_GEMM_WS=1 M3VIT_LIB=$PWD/m3vit_amd/libm3vit_hip_stamps.so python tools/gemm_ws_stamps.py [qkv|proj|fc1]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import _lib, ops  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "qkv"
N, K, two = {"qkv": (1152, 384, False), "proj": (384, 384, False), "fc1": (1536, 384, True)}[shape]
M, ring = 128 * 197, 10
dev = torch.device("cuda:0")
As = [torch.randn(M, K, device=dev).half() for _ in range(ring)]
B = (torch.randn(N, K, device=dev) * 0.05).half()
Cs = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(ring)]
Ps = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(ring)] if two else None
bias = torch.zeros(N, device=dev)


def run(j):
    if two:
        ops.gemm_nt(As[j], B, Cs[j], bias=bias, act=ops.M3_ACT_GELU, pre_out=Ps[j])
    else:
        ops.gemm_nt(As[j], B, Cs[j])


for i in range(2 * ring + 1):
    run(i % ring)
torch.cuda.synchronize()
buf = np.zeros((256, 2, 64), dtype=np.uint64)
fn = _lib.lib().m3_debug_gemm_ws_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(buf.ctypes.data) == 0
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for i in range(ring):
    run(i)
ev1.record(); torch.cuda.synchronize()
us = ev0.elapsed_time(ev1) * 1e3 / ring
st = buf.astype(np.int64)
live = st[:, 0, 0] > 0
st = st[live]
nb = int((st[:, 0] > 0).sum(axis=1).min())
print(f"{shape}: launch {us:.1f} us by events; {live.sum()} workgroups stamped, {nb} barriers each")
t0 = st[:, :, 0].min(axis=1)                                  # first arrival at barrier 0 per workgroup
rel = st[:, :, :nb] - t0[:, None, None]
release = rel.max(axis=1)                                     # a barrier opens when its last role arrives
step = np.diff(release, axis=1)
print(f"ticks from barrier to barrier (mean over workgroups): {np.round(step.mean(axis=0)).astype(int).tolist()}")
span = release[:, -1] - release[:, 0]
print(f"span barrier 0 -> barrier {nb - 1}: mean {span.mean():.0f} ticks; whole launch {us:.1f} us")
last = rel.argmax(axis=1)                                     # role that arrived last
names = ["mfma", "helper"]
for r in range(2):
    print(f"  {names[r]:6s} arrives last at {100 * (last == r).mean():5.1f} % of barriers; "
          f"mean slack before release {np.mean(release - rel[:, r, :]):.0f} ticks")
# per position within a round (6 barriers: one per slice)
for pos in range(6):
    sel = np.arange(pos, nb, 6)
    sel = sel[sel > 0]
    if len(sel) == 0:
        continue
    d = release[:, sel] - release[:, sel - 1]
    who = [(last[:, sel] == r).mean() for r in range(2)]
    print(f"  barrier {pos} of a round: {d.mean():7.0f} ticks since the previous one; last to arrive "
          + ", ".join(f"{names[r]} {100 * who[r]:.0f} %" for r in range(2)))
