#!/usr/bin/env python3
"""The shader clock the wide-tile weight-gradient kernel really runs at (diagnostic build of the library):
    (cd m3vit_amd/csrc && make CXXFLAGS="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -DM3_WGRAD_CLOCK" \
         OBJDIR=../../build/wclk OUT=../libm3vit_hip_wclk.so)
    M3VIT_LIB=$PWD/m3vit_amd/libm3vit_hip_wclk.so python tools/wgrad_clock.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = 25216, 1536, 384
dC = torch.randn(M, N, device=dev).half(); A = torch.randn(M, K, device=dev).half()
dW = torch.zeros(N, K, device=dev)
for reps in (1, 50):
    for _ in range(reps):
        ops.wgrad_tn(dC, A, dW)
    torch.cuda.synchronize()
    buf = np.zeros(4, dtype=np.uint64)
    fn = _lib.lib().m3_debug_wgrad_clock
    fn.argtypes = [ctypes.c_void_p]
    assert fn(buf.ctypes.data) == 0
    cyc, ref, nst = int(buf[0]), int(buf[1]), int(buf[2])
    print(f"after {reps:2d} back-to-back launches: workgroup 0 lived {cyc} cycles = {ref / 100:.1f} us -> {cyc / (ref * 10):.2f} GHz; "
          f"{nst} steps, {cyc / max(nst, 1):.0f} cycles per 32-row step incl. prologue / slab store")
