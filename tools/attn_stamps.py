#!/usr/bin/env python3
"""Phase timeline of the short-sequence attention backward (attention_bwd_res_kernel, N <= 256) from in-kernel s_memtime
stamps (diagnostic build of the library), at the configs[1] shape (B = 128, N = 197, 12 heads of 32).

    (cd m3vit_amd/csrc && make CXXFLAGS="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -DM3_ATTN_STAMPS" \
         OBJDIR=../../build/astamps OUT=../../build/astamps/libm3vit_hip.so)
    M3VIT_LIB=$PWD/build/astamps/libm3vit_hip.so python tools/attn_stamps.py
Operands rotate over a ring larger than the Infinity Cache; the last launch is analysed (ticks = shader cycles)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
B, N, h, dh, ring = 128, 197, 12, 32, 6
C = h * dh
qkvs = [torch.randn(B * N, 3 * C, device=dev).half() for _ in range(ring)]
dos = [torch.randn(B * N, C, device=dev).half() for _ in range(ring)]
os_ = [torch.empty(B * N, C, dtype=torch.float16, device=dev) for _ in range(ring)]
lses = [torch.empty(B, h, N, device=dev) for _ in range(ring)]
dq = [torch.empty_like(q) for q in qkvs]
for i in range(ring):
    ops.attention_fwd(qkvs[i], B, N, h, dh, os_[i], lses[i])
for i in range(2 * ring):
    j = i % ring
    ops.attention_bwd(qkvs[j], os_[j], dos[j], lses[j], B, N, h, dh, dq[j])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.attention_bwd(qkvs[1], os_[1], dos[1], lses[1], B, N, h, dh, dq[1])
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
SN = 40
wgs = min(B * h, 2048)
buf = np.zeros((wgs, SN), dtype=np.uint64)
fn = _lib.lib().m3_debug_attn_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf.ctypes.data, wgs) == 0
st = buf.astype(np.int64)
nsteps = (N + 31) // 32
life = st[:, 37] - st[:, 0]
med = float(np.median(life))
print(f"attention_bwd_res<32>: launch {us:.1f} us; {B * h} workgroups, 2 per CU; median life {med:.0f} ticks")


def show(name, d):
    print(f"  {name:44s} median {np.median(d):8.0f}   {100 * np.median(d) / med:5.1f} % of a workgroup's life")


show("issue loads + LDS stores (Q, dO, K, O; K/V frags)", st[:, 1] - st[:, 0])
show("wait for the operands (vmcnt + barrier)", st[:, 2] - st[:, 1])
a = np.zeros(len(st)); b = np.zeros(len(st)); c = np.zeros(len(st)); d = np.zeros(len(st))
for s in range(nsteps):
    prev = st[:, 2] if s == 0 else st[:, 6 + 4 * (s - 1)]
    a += st[:, 3 + 4 * s] - prev
    b += st[:, 4 + 4 * s] - st[:, 3 + 4 * s]
    c += st[:, 5 + 4 * s] - st[:, 4 + 4 * s]
    d += st[:, 6 + 4 * s] - st[:, 5 + 4 * s]
show(f"{nsteps} x [S, dP (8 MFMA), exp2, dS]", a)
show(f"{nsteps} x [dV, dK (16 MFMA), dS^T -> LDS]", b)
show(f"{nsteps} x barrier", c)
show(f"{nsteps} x [dQ piece: 7 dependent MFMAs on tr reads] + store", d)
show("dK / dV stores issue", st[:, 36] - st[:, 6 + 4 * (nsteps - 1)])
show("store acknowledge", st[:, 37] - st[:, 36])
