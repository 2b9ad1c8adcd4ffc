#!/usr/bin/env python3
"""Phase timeline of the LDS-DMA GEMM from in-kernel s_memtime stamps (diagnostic build of the library).

    (cd m3vit_amd/csrc && make CXXFLAGS="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -DM3_GEMM_STAMPS" \
         OBJDIR=../../build/stamps OUT=../libm3vit_hip_stamps.so)
    M3VIT_LIB=$PWD/m3vit_amd/libm3vit_hip_stamps.so python tools/gemm_stamps.py [qkv|proj|fc1|fc2]

Operands are streamed (a ring of (A, C) pairs larger than the Infinity Cache), the last launch is analysed."""
import ctypes
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import _lib, ops  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "qkv"
N, K, two = {"qkv": (1152, 384, False), "proj": (384, 384, False), "fc1": (1536, 384, True), "fc2": (384, 1536, False)}[shape]
M, ring = 128 * 197, 10
dev = torch.device("cuda:0")
As = [torch.randn(M, K, device=dev).half() for _ in range(ring)]
B = (torch.randn(N, K, device=dev) * 0.05).half()
Cs = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(ring)]
Ps = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(ring)] if two else None
bias = torch.zeros(N, device=dev)
for i in range(2 * ring + 1):
    j = i % ring
    if two:
        ops.gemm_nt(As[j], B, Cs[j], bias=bias, act=ops.M3_ACT_GELU, pre_out=Ps[j])
    else:
        ops.gemm_nt(As[j], B, Cs[j])
torch.cuda.synchronize()
L = _lib.lib()
nk = K * 2 // 128
wgs = ((M + 127) // 128) * ((N + 127) // 128)
SN, EP = 64, 56
buf = np.zeros((min(wgs, 4096), SN), dtype=np.uint64)
fn = L.m3_debug_gemm_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf.ctypes.data, buf.shape[0]) == 0
st = buf.astype(np.int64)
hw = buf[:, SN - 1]
# s_memtime counters of the XCCs are not aligned (offsets of millions of ticks): cluster the workgroups by the gaps in
# their entry stamps and put every cluster on its own time line (0 = entry of the cluster's first workgroup)
order = np.argsort(st[:, 0])
gaps = np.diff(st[order, 0])
cluster = np.zeros(len(order), dtype=np.int64)
cluster[order[1:]] = np.cumsum(gaps > 300000)
for x in np.unique(cluster):
    m = cluster == x
    st[m, :SN - 1] -= st[m, 0].min()
print(f"clock domains found: {len(np.unique(cluster))}")
t0 = 0
span = int(st[:, EP + 1].max())
# time the same launch with events to convert ticks
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for i in range(ring):
    (ops.gemm_nt(As[i], B, Cs[i], bias=bias, act=ops.M3_ACT_GELU, pre_out=Ps[i]) if two else ops.gemm_nt(As[i], B, Cs[i]))
ev1.record(); torch.cuda.synchronize()
us = ev0.elapsed_time(ev1) * 1e3 / ring
print(f"{shape}: {wgs} workgroups, K steps {nk}; launch {us:.1f} us by events")
setup = st[:, 1] - st[:, 0]
wait = np.stack([st[:, 2 + 2 * k] - (st[:, 1] if k == 0 else st[:, 1 + 2 * k]) for k in range(nk)], 1)
comp = np.stack([st[:, 3 + 2 * k] - st[:, 2 + 2 * k] for k in range(nk)], 1)
ep0 = st[:, EP] - st[:, 1 + 2 * nk]
ep1 = st[:, EP + 1] - st[:, EP]
ack = st[:, EP + 2] - st[:, EP + 1]
life = st[:, EP + 1] - st[:, 0]


def pct(x):
    return f"{x.mean():8.0f} ({100 * x.mean() / life.mean():4.1f} %)"


print(f"per workgroup (ticks, mean; share of its lifetime {life.mean():.0f}):")
print(f"  set-up (tile lookup, row pointers)        {pct(setup)}")
print(f"  DMA issue + wait + barrier, first slice   {pct(wait[:, 0])}")
if nk > 1:
    print(f"  DMA issue + wait + barrier, later slices  {pct(wait[:, 1:].sum(1))}   per slice {wait[:, 1:].mean():.0f}")
print(f"  MFMA phases + barrier                     {pct(comp.sum(1))}   per slice {comp.mean():.0f}")
print(f"  epilogue half 0 (stage, bias, store)      {pct(ep0)}")
print(f"  epilogue half 1                           {pct(ep1)}")
print(f"  (until the stores are acknowledged        {ack.mean():8.0f})")
cu_key = (cluster.astype(np.uint64) << np.uint64(16)) | ((hw >> np.uint64(8)) & np.uint64(0xFF)) | (((hw >> np.uint64(13)) & np.uint64(7)) << np.uint64(8))
groups = defaultdict(list)
for i, k in enumerate(cu_key):
    groups[int(k)].append(i)
def coverage(idx, lo_of, hi_of):
    """share of this CU's own span (its first entry -> its last store issue) covered by the union of the intervals"""
    ev = []
    for i in idx:
        for a_, b_ in zip(lo_of(i), hi_of(i)):
            ev.append((a_, 1)); ev.append((b_, -1))
    ev.sort()
    cur, last, tot = 0, 0, 0
    for t, d in ev:
        if cur > 0:
            tot += t - last
        cur += d; last = t
    cu_span = max(st[i, EP + 1] for i in idx) - min(st[i, 0] for i in idx)
    return tot / cu_span


cu_spans = [max(st[i, EP + 1] for i in idx) - min(st[i, 0] for i in idx) for idx in groups.values()]
print(f"per-CU span (first entry -> last store issue on that CU): mean {np.mean(cu_spans):.0f} ticks, max {max(cu_spans)} "
      f"-> ~{max(cu_spans) / us:.0f} ticks/us if the slowest CU spans the launch")
busy_mfma, busy_ep, busy_any, n_res = [], [], [], []
for k, idx in groups.items():
    busy_mfma.append(coverage(idx, lambda i: [st[i, 2 + 2 * s_] for s_ in range(nk)], lambda i: [st[i, 3 + 2 * s_] for s_ in range(nk)]))
    busy_ep.append(coverage(idx, lambda i: [st[i, 1 + 2 * nk]], lambda i: [st[i, EP + 1]]))
    busy_any.append(coverage(idx, lambda i: [st[i, 0]], lambda i: [st[i, EP + 1]]))
    n_res.append(len(idx))
print(f"CUs seen: {len(groups)}, workgroups per CU {np.mean(n_res):.1f} (min {min(n_res)}, max {max(n_res)})")
print(f"share of a CU's span with >= 1 workgroup resident {100 * np.mean(busy_any):.0f} %, >= 1 in an MFMA phase "
      f"{100 * np.mean(busy_mfma):.0f} % (min {100 * min(busy_mfma):.0f} %, max {100 * max(busy_mfma):.0f} %), >= 1 in its epilogue {100 * np.mean(busy_ep):.0f} %")
