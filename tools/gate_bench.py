#!/usr/bin/env python3
"""m3_gate_fwd alone (T = 128 x 197 tokens, D = 384, E = 16, k = 4, fp16 rows): with / without the dense outputs."""
import os
import sys
from ctypes import byref

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import _lib, ops  # noqa: E402
from m3vit_amd.ops import _p, _stream, dt_code, lib  # noqa: E402

dev = torch.device("cuda:0")
T, D, E, k, RING = 128 * 197, 384, 16, 4, 8
xs = [torch.randn(T, D, device=dev).half() for _ in range(RING)]
w = torch.randn(D, E, device=dev) * 0.05
f32 = torch.float32
idx = torch.empty(T, k, dtype=torch.int64, device=dev); idx32 = torch.empty(T, k, dtype=torch.int32, device=dev)
nxt = torch.empty(T, dtype=torch.int32, device=dev); score = torch.empty(T, k, device=dev); top = torch.empty(T, k + 1, device=dev)
clean, noisy, gates = (torch.empty(T, E, device=dev) for _ in range(3))
nblk = lib().m3_gate_num_blocks(T)
pi = torch.empty(nblk, E, device=dev); pl = torch.empty(nblk, E, dtype=torch.int32, device=dev)


def run(x, dense):
    a = _lib.GateFwdArgs(_p(x), dt_code(x.dtype), T, D, x.stride(0), _p(w), E, None, None, 0.0, k, _p(idx), _p(idx32), _p(nxt),
                         _p(score), _p(top), _p(clean) if dense else None, _p(noisy) if dense else None,
                         _p(gates) if dense else None, _p(pi), _p(pl), None)
    ops.check(lib().m3_gate_fwd(byref(a), _stream()), "m3_gate_fwd")


for dense in (True, False):
    for i in range(RING):
        run(xs[i], dense)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    n = 40
    for i in range(n):
        run(xs[i % RING], dense)
    e.record(); torch.cuda.synchronize()
    print(f"dense={dense}: {s.elapsed_time(e) * 1e3 / n:.1f} us per launch ({nblk} workgroups)")
