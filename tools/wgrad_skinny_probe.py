import sys, torch
sys.path.insert(0, "/root/repo")
from m3vit_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for dt in (torch.float16, torch.float32):
    for M, N, K in ((25216, 384, 16), (9608, 768, 16), (25216, 768, 32)):
        es = 2 if dt == torch.float16 else 4
        n = int(max(2, min(16, -(-320e6 // (M * (N + K) * es)))))
        sets = [(torch.randn(M, N, generator=g).to(dt).to(dev), torch.randn(M, K, generator=g).to(dt).to(dev)) for _ in range(n)]
        dW = torch.zeros(N, K, device=dev)
        for sp in (None, 64, 128, 256, 512):
            q = ops.WgradQueue(2 * 512 * N * K, dev)
            def fn(i):
                dC, A = sets[i % n]
                ops.wgrad_tn(dC, A, dW, beta=1, queue=q, splits=sp)
            for i in range(3): fn(i)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(20): fn(i)
            e.record(); torch.cuda.synchronize()
            print(dt, M, N, K, "splits", sp, f"{s.elapsed_time(e) * 1e3 / 20:.1f} us", f"ring {n}", flush=True)
