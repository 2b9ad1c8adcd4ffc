import os, sys, torch
sys.path.insert(0, os.getcwd())
from m3vit_amd import ops
dev = torch.device("cuda:0")
N, K = 1536, 384
for M in (3152, 6304, 12608, 25216, 50432):
    dC = torch.randn(M, N, device=dev).half(); A = torch.randn(M, K, device=dev).half()
    dW = torch.zeros(N, K, device=dev)
    for splits in (21,):
        ws = torch.empty(splits * N * K + splits * N, device=dev)
        f = lambda: ops.wgrad_tn(dC, A, dW, ws=ws, splits=splits)
        for _ in range(5): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30): f()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 30
        steps = (M + 31) // 32 / splits
        print(f"M {M:6d} splits {splits} steps/split {steps:6.1f}  {us:6.1f} us (kernel + reduce)")
