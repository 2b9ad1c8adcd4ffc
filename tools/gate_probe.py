import sys, torch
sys.path.insert(0, "/root/repo")
from m3vit_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (T, D, E, k) in ((25216, 768, 64, 4), (25216, 384, 16, 4), (9608, 768, 16, 4)):
    n = 6
    xs = [torch.randn(T, D, generator=g).half().to(dev) for _ in range(n)]
    w = (torch.randn(D, E, generator=g) * 0.02).to(dev)
    outs = [ops.gate_fwd(xs[i % n], w, k) for i in range(3)]
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(30):
        ops.gate_fwd(xs[i % n], w, k)
    e.record(); torch.cuda.synchronize()
    print(f"T={T} D={D} E={E}: {s.elapsed_time(e) * 1e3 / 30:.1f} us per call (host + kernels)")
