#!/usr/bin/env python3
"""Per-step timeline of wgrad_big_kernel from in-kernel s_memtime stamps (diagnostic build of the library).

    (cd m3vit_amd/csrc && make CXXFLAGS="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -DM3_WGRAD_STAMPS" \
         OBJDIR=../../build/wstamps OUT=../libm3vit_hip_wstamps.so)
    M3VIT_LIB=$PWD/m3vit_amd/libm3vit_hip_wstamps.so python tools/wgrad_big_stamps.py [qkv|fc1|proj]

Dense ViT-Base weight gradient at M = 25 216 rows, operands streamed through a ring; the last launch is analysed: per step, cycles from
the step's start to "DMA issued", "MFMAs done", "vmcnt wait over", "barrier passed" (waves 0 and 4 of every workgroup)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import _lib, ops  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "qkv"
N, K = {"qkv": (2304, 768), "fc1": (3072, 768), "proj": (768, 768)}[shape]
M, ring = 128 * 197, 6
dev = torch.device("cuda:0")
sets = [(torch.randn(M, N, device=dev).half(), torch.randn(M, K, device=dev).half()) for _ in range(ring)]
dW = torch.zeros(N, K, device=dev)
for i in range(2 * ring + 1):
    dC, A = sets[i % ring]
    ops.wgrad_tn(dC, A, dW, beta=1)
torch.cuda.synchronize()
L = _lib.lib()
fn = L.m3_debug_wbig_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
tiles = (N // 256) * (K // 256)
SN = 2 * (2 + 4 * 24)
buf = np.zeros((tiles, SN), dtype=np.uint64)
assert fn(buf.ctypes.data, tiles) == 0
half = SN // 2
sp = ops.default_wgrad_splits(M, N, K, 1, torch.float16)
print(f"# {shape}: N={N} K={K} M={M}, {tiles} tiles x {sp} parts; workgroups (blockIdx.y = z = 0) stamped: {tiles}; cycles (s_memtime)")
for wv in (0, 1):
    st = buf[:, wv * half:(wv + 1) * half].astype(np.int64)
    ok = st[:, 0] > 0
    st = st[ok]
    nsteps = 0
    while nsteps < 24 and np.all(st[:, 5 + 4 * nsteps] > 0):
        nsteps += 1
    print(f"wave {4 * wv}: {st.shape[0]} workgroups, {nsteps} steps stamped; prologue (entry -> step 0 landed) {np.median(st[:, 1] - st[:, 0]):.0f}")
    print("  step   issue  compute  vm-wait  barrier    total")
    prev = st[:, 1]
    for t in range(nsteps):
        a, b, c, d = (st[:, 2 + 4 * t + j] for j in range(4))
        print(f"  {t:4d} {np.median(a - prev):7.0f} {np.median(b - a):8.0f} {np.median(c - b):8.0f} {np.median(d - c):8.0f} {np.median(d - prev):8.0f}")
        prev = d
