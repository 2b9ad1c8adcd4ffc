"""Per-SHAPE time table of the step's two GEMM entry points (m3_gemm_nt, m3_wgrad_tn), measured in the step itself:
one serial eager step of `bench.py`'s workload per repetition, every launch bracketed by HIP events on its stream, grouped by
(entry point, M, N, K, groups, epilogue kind).  What rocprofv3's by-launch view cannot say (it groups by grid size, and a
weight-gradient launch's grid also carries the previous launch's reduction) this says: which SHAPE runs at what fraction of
the MFMA peak.

    python tools/shape_times.py [--dtype f32|f16|bf16] [--config 1|3|4] [--reps 3]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None)
    args = ap.parse_args()
    import bench
    from m3vit_amd import ops
    from m3vit_amd.config import BackboneConfig, init_params
    from m3vit_amd.step import MultiTaskStep
    wl = bench.WORKLOADS[args.config]
    cfg = BackboneConfig(**wl["cfg"])
    batch = args.batch or wl["batch"]
    dtype = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
    peak = 157.3 if args.dtype == "f32" else 2500.0
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    params = init_params(cfg, seed=1)
    run = MultiTaskStep(cfg, params, batch=batch, dtype=dtype, device=str(dev), cv_weight=0.01, parallel_tasks=False, graph=False)
    g = torch.Generator().manual_seed(1000)
    images = torch.randn(batch, 3, *cfg.img_size, generator=g).to(dev)
    dtok = (torch.randn(batch, cfg.num_tokens, cfg.embed_dim, generator=g) * 0.05).to(dev)
    run.bind(images, dtok)
    for _ in range(2):
        run.serial_step()
    torch.cuda.synchronize()

    recs = []
    o_gemm, o_wgrad = ops.gemm_nt, ops.wgrad_tn

    def ev():
        return torch.cuda.Event(enable_timing=True)

    def gemm(A, B, C, **kw):
        s, e = ev(), ev()
        s.record(); r = o_gemm(A, B, C, **kw); e.record()
        M = kw.get("M")
        if M is None:
            M = kw["a_row_idx"].numel() if kw.get("a_row_idx") is not None else A.shape[0]
        kind = "+".join(k for k in ("pre_out", "gelu_grad_pre", "residual", "row_scale", "c_row_idx", "a_row_idx") if kw.get(k) is not None)
        G = B.shape[0] if B.dim() == 3 else 1
        recs.append(("gemm_nt", int(M), int(B.shape[-2]), int(B.shape[-1]), int(G), kind, s, e))
        return r

    def wgrad(dC, A, dW, **kw):
        s, e = ev(), ev()
        s.record(); r = o_wgrad(dC, A, dW, **kw); e.record()
        M = kw.get("M")
        if M is None:
            M = kw["c_row_idx"].numel() if kw.get("c_row_idx") is not None else dC.shape[0]
        kind = "+".join(k for k in ("db", "c_row_idx", "a_row_idx", "c_row_scale") if kw.get(k) is not None)
        G = dW.shape[0] if dW.dim() == 3 else 1
        recs.append(("wgrad_tn", int(M), int(dW.shape[-2]), int(dW.shape[-1]), int(G), kind, s, e))
        return r

    ops.gemm_nt, ops.wgrad_tn = gemm, wgrad
    t0, t1 = ev(), ev()
    t0.record()
    for _ in range(args.reps):
        run.serial_step()
    t1.record()
    torch.cuda.synchronize()
    ops.gemm_nt, ops.wgrad_tn = o_gemm, o_wgrad
    step_ms = t0.elapsed_time(t1) / args.reps

    table = {}
    for name, M, N, K, G, kind, s, e in recs:
        a = table.setdefault((name, M, N, K, G, kind), [0, 0.0])
        a[0] += 1; a[1] += s.elapsed_time(e)
    print(f"# {wl['name']}, {args.dtype}, batch {batch}: serial eager step {step_ms:.2f} ms (with the event pairs); peak {peak} TFLOP/s")
    print(f"{'entry':9s} {'M':>7s} {'N':>5s} {'K':>5s} {'G':>3s} {'n/step':>6s} {'avg us':>8s} {'ms/step':>8s} {'TFLOP/s':>8s} {'frac':>6s}  kind")
    tot = {}
    for (name, M, N, K, G, kind), (n, ms) in sorted(table.items(), key=lambda kv: -kv[1][1]):
        fl = 2.0 * M * N * K
        tf = fl * n / (ms * 1e-3) / 1e12
        print(f"{name:9s} {M:7d} {N:5d} {K:5d} {G:3d} {n / args.reps:6.1f} {1e3 * ms / n:8.1f} {ms / args.reps:8.3f} {tf:8.1f} {tf / peak:6.3f}  {kind}")
        a = tot.setdefault(name, [0.0, 0.0])
        a[0] += ms / args.reps; a[1] += fl * n / args.reps
    for name, (ms, fl) in tot.items():
        tf = fl / (ms * 1e-3) / 1e12
        print(f"# {name}: {ms:.2f} ms/step, {tf:.1f} TFLOP/s = {tf / peak:.3f} of peak")


if __name__ == "__main__":
    main()
