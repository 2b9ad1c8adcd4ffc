#!/usr/bin/env python3
"""Where does the data-parallel step with a shared stem stall in the two-ranks-on-ONE-GPU gloo rehearsal?
(profiles/r03: bench_2r_dps 901.9 ms/step; dpdbg.log "SHARE=1 async it 1 ... p2 wait 434.7".)

ONE instrumented run, both ranks logging.  For every iteration of MultiTaskStep._collective_step's schedule (graph part j
replayed, all-reduce j issued asynchronously, at most two all-reduces outstanding under gloo) it records, on one clock:
  * host: when part j's replay returned, when all-reduce j was issued, when the wait for all-reduce j returned;
  * GPU : when part j's graph FINISHED on the device (an event recorded behind the replay, converted to the host clock
          through an anchor event taken at the iteration's start).
The question the two readings of the r3 log differ on: while the host sat in the long wait, had the GPU already finished
the parts (then the time is gloo's: staging copies / its worker threads / the transport between two processes sharing a
device), or did the parts themselves finish late (then the device starved them: two processes' queues time-sliced)?
    M3_PROBE_SHARE=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
        tools/dp_stall_probe.py
"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd.config import VIT_SMALL_MOE, BackboneConfig, init_params  # noqa: E402
from m3vit_amd.step import MultiTaskStep  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
share = os.environ.get("M3_PROBE_SHARE", "1") == "1"
iters = int(os.environ.get("M3_PROBE_ITERS", "12"))
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = BackboneConfig(**VIT_SMALL_MOE)
run = MultiTaskStep(cfg, init_params(cfg, seed=1), batch=128, dtype=torch.float16, world=world, rank=rank, dp_parts=6,
                    share_stem=share)
g = torch.Generator().manual_seed(1000 + rank)
run.bind(torch.randn(128, 3, 224, 224, generator=g).cuda(), (torch.randn(128, cfg.num_tokens, cfg.embed_dim, generator=g) * 0.05).cuda())
run.step_eager()
torch.cuda.synchronize()
ok = run.capture()
out = open(f"gpurun_out/r4/dp_stall_rank{rank}.log", "w")


def P(*a):
    print(*a, file=out, flush=True)


P(f"rank {rank}: share_stem={run.share_stem} captured={ok} parts={len(run.graphs or [])} segments(MB)="
  f"{[round((hi - lo) * 4 / 1e6, 1) for lo, hi in run.segments]}")
dist.barrier()
torch.cuda.synchronize()
nparts = len(run.graphs)
for it in range(iters):
    anchor = torch.cuda.Event(enable_timing=True)
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(nparts)]
    torch.cuda.synchronize()
    anchor.record()
    anchor.synchronize()
    t_anchor = time.perf_counter()
    rel = lambda: 1e3 * (time.perf_counter() - t_anchor)            # noqa: E731
    works, rec = [], []
    for j, (gr, (lo, hi)) in enumerate(zip(run.graphs, run.segments)):
        r = {}
        gr.replay()
        ends[j].record()
        r["replayed"] = rel()
        if j >= 2:
            works[j - 2].wait()
            r["waited_for"] = j - 2
            r["wait_done"] = rel()
        works.append(dist.all_reduce(run.flat[lo:hi], async_op=True))
        r["issued"] = rel()
        rec.append(r)
    tail = []
    for j in range(max(0, nparts - 2), nparts):
        works[j].wait()
        tail.append((j, rel()))
    torch.cuda.synchronize()
    total = rel()
    gpu_done = [anchor.elapsed_time(e) for e in ends]
    line = f"it {it}: total {total:7.1f} ms | "
    for j, r in enumerate(rec):
        line += f"p{j}: replayed@{r['replayed']:.1f} gpu-done@{gpu_done[j]:.1f}"
        if "wait_done" in r:
            line += f" AR{r['waited_for']}-done@{r['wait_done']:.1f}"
        line += f" issued@{r['issued']:.1f} | "
    line += " ".join(f"AR{j}-done@{t:.1f}" for j, t in tail)
    P(line)
    # the stall signature: an all-reduce whose wait returned long after the part it follows had finished on the GPU
    for r in rec:
        if "wait_done" in r:
            lag = r["wait_done"] - gpu_done[r["waited_for"]]
            if lag > 100:
                P(f"   STALL: all-reduce {r['waited_for']} returned {lag:.0f} ms after its part finished on the GPU "
                  f"(part done @{gpu_done[r['waited_for']]:.1f}, every part done by @{max(gpu_done):.1f})")
P("done")
dist.destroy_process_group()
