import os, sys, torch
sys.path.insert(0, os.getcwd())
from m3vit_amd import ops
dev = torch.device("cuda:0")
E, k, T = 16, 4, 25216
R = T * k
g = torch.Generator().manual_seed(0)
idx = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(T)]).to(torch.int32).to(dev)
r = ops.route_build(idx, E)
N, K = 1536, 384
dC = torch.randn(R, N, device=dev).half()
x = torch.randn(T, K, device=dev).half()
xR = torch.randn(R, K, device=dev).half()
dW = torch.zeros(E, N, K, device=dev); db = torch.zeros(E, N, device=dev)
splits = 1
ws = torch.empty(ops.wgrad_ws_elems(R, N, K, E, grouped=True, dtype=torch.float16), device=dev)
ident = torch.arange(R, dtype=torch.int32, device=dev)
def t(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:50s} {s.elapsed_time(e) * 100:7.1f} us", flush=True)
t(lambda: ops.wgrad_tn(dC, x, dW, M=R, ws=ws, db=db, group_offsets=r.offsets, splits=splits, a_row_idx=r.row_of_slot, a_row_div=k), "fc1: A = x[row_of_slot / 4]")
t(lambda: ops.wgrad_tn(dC, xR, dW, M=R, ws=ws, db=db, group_offsets=r.offsets, splits=splits, a_row_idx=r.row_of_slot, a_row_div=1), "fc1: A = xR[row_of_slot] (div 1)")
t(lambda: ops.wgrad_tn(dC, xR, dW, M=R, ws=ws, db=db, group_offsets=r.offsets, splits=splits, a_row_idx=ident, a_row_div=1), "fc1: A = xR[identity]")
t(lambda: ops.wgrad_tn(dC, xR, dW, M=R, ws=ws, db=db, group_offsets=r.offsets, splits=splits), "fc1: no gather")
t(lambda: ops.wgrad_tn(dC, xR, dW, M=R, ws=ws, group_offsets=r.offsets, splits=splits), "fc1: no gather, no bias")
