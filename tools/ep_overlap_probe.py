#!/usr/bin/env python3
"""Two ranks on ONE GPU over gloo (the only multi-rank form this build box allows): one forward + backward of a single MoE
block at BASELINE configs[3]'s layer shape (ViT-Base, E = 64 -> 32 experts per rank, k = 4, 128 x 197 tokens per rank, fp16)
with the expert-parallel exchange as ONE all-to-all-v each way (ep_chunks = 1) and cut into 2 / 4 chunks overlapped with the
experts' GEMMs.  Per variant: host wall time of the pass (device synchronised) and the host time spent BLOCKED in the
exchanges (gloo collectives are host calls: `wait()` returns when the rows have arrived), i.e. the exposed exchange.
gloo stages every row through host memory, so the absolute numbers are NOT xGMI numbers; what transfers is the structure:
how much of the exchange sits under the GEMMs.
    python tools/ep_overlap_probe.py            (spawns the two ranks itself)
"""
import os
import socket
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from m3vit_amd.config import BackboneConfig, init_params
        from m3vit_amd.engine import BackboneEngine
        torch.cuda.set_device(0)
        B = int(os.environ.get("PROBE_BATCH", "128"))
        cfg = BackboneConfig(img_size=(224, 224), embed_dim=768, depth=2, num_heads=12, mlp_ratio=4.0, moe_mlp_ratio=1.0,
                             moe_experts=64, moe_top_k=4, gate_dim=768, multi_gate=False)
        P = init_params(cfg, seed=3)
        g = torch.Generator().manual_seed(70 + rank)
        img = torch.randn(B, 3, 224, 224, generator=g).cuda()
        dtok = (torch.randn(B, cfg.num_tokens, 768, generator=g) * 0.1).cuda()
        out = {}
        for chunks in (1, 2, 4):
            eng = BackboneEngine(cfg, P, batch=B, dtype=torch.float16, ep_world=world, ep_rank=rank, ep_chunks=chunks)
            blocked = [0.0]

            class Timed:
                def __init__(self, w):
                    self.w = w

                def wait(self):
                    t0 = time.perf_counter()
                    self.w.wait()
                    blocked[0] += time.perf_counter() - t0
            a2a_async, a2a = eng._a2a_async, eng._a2a
            eng._a2a_async = lambda *a, **k: Timed(a2a_async(*a, **k))

            def timed_a2a(*a, **k):
                t0 = time.perf_counter()
                r = a2a(*a, **k)
                blocked[0] += time.perf_counter() - t0
                return r
            eng._a2a = timed_a2a
            ts, bl = [], []
            for it in range(6):
                dist.barrier()
                torch.cuda.synchronize()
                blocked[0] = 0.0
                t0 = time.perf_counter()
                eng.zero_grad()
                eng.forward(img, None)
                eng.backward(dtok, cv_weight=0.01)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0); bl.append(blocked[0])
            out[chunks] = (min(ts[2:]) * 1e3, min(bl[2:]) * 1e3)
            del eng
            torch.cuda.empty_cache()
        q.put((rank, out))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    print("one MoE block + one dense block, configs[3] layer shape, 2 ranks on one MI355X over gloo, fp16, forward + backward")
    print(f"{'chunks':>6s} | " + " | ".join(f"rank {r}: pass ms / blocked-in-exchange ms" for r in sorted(res)))
    for c in (1, 2, 4):
        print(f"{c:6d} | " + " | ".join((f"{res[r][c][0]:26.1f} / {res[r][c][1]:8.1f}" if isinstance(res[r], dict) else str(res[r])) for r in sorted(res)))
