#!/usr/bin/env python3
"""Is the hipStreamEndCapture segfault of profiles/r02_wgrad_capture_segv.txt the runtime's or the engine's?

Plain torch streams / events / elementwise kernels, NO engine: the capturing stream forks two "task" streams, each task
stream forks its own "wgrad" grand-child stream several times by an event, the grand-child records a `done` event that
the task stream waits on before it overwrites the buffer the grand-child reads (engine._fork / _before_write), the
grand-child is joined by wait_stream (engine._join_wgrad) and the task streams are joined into the capturing stream
(step._run_tasks).  Variants, each in its OWN child process (a hard fault must not hide the others):

    fresh    every event is created inside the capture
    pooled   events come from a pool created (and already recorded once, in a warm-up run) BEFORE the capture and are
             re-recorded inside it - what BackboneEngine._event() does
    single   the single-level pattern (grand-children forked from the capturing stream itself), the known-good control
    prefork  nested as `pooled`, but the capturing stream first forks EVERY stream (task and grand-child) itself, so that no
             stream enters the capture through a stream that is itself a fork
    prefork_join  `prefork`, and the grand-children are also joined straight into the capturing stream at the end

    python tools/nested_capture_probe.py            # runs all variants, prints one RESULT line each
"""
import subprocess
import sys

import torch


def pattern(variant: str):
    dev = torch.device("cuda:0")
    nested = variant != "single"
    prefork = variant.startswith("prefork")
    n_task = 2
    bufs = [torch.zeros(1 << 20, device=dev) for _ in range(n_task)]
    outs = [torch.zeros(1 << 20, device=dev) for _ in range(n_task)]
    tasks = [torch.cuda.Stream(device=dev) for _ in range(n_task)]
    wgs = [torch.cuda.Stream(device=dev) for _ in range(n_task)]
    pool, cursor = [], [0]

    def event():
        if variant == "fresh":
            return torch.cuda.Event()
        if cursor[0] == len(pool):
            pool.append(torch.cuda.Event())
        cursor[0] += 1
        return pool[cursor[0] - 1]

    def task_body(i, blocks=6):
        cur = torch.cuda.current_stream()
        readers = None
        for b in range(blocks):
            if readers is not None:                       # _before_write: the grand-child still reads bufs[i]
                cur.wait_event(readers)
            bufs[i].add_(1.0)                             # the "dgrad chain"
            ready = event(); ready.record(cur)            # _fork
            wgs[i].wait_event(ready)
            with torch.cuda.stream(wgs[i]):
                outs[i].add_(bufs[i])                     # the "wgrad"
                done = event(); done.record(wgs[i])
            readers = done
        cur.wait_stream(wgs[i])                           # _join_wgrad

    def step():
        cursor[0] = 0
        main = torch.cuda.current_stream()
        if not nested:
            for i in range(n_task):
                task_body(i)
            return
        for st in tasks + (wgs if prefork else []):
            st.wait_stream(main)
        for i, st in enumerate(tasks):
            with torch.cuda.stream(st):
                task_body(i)
        for st in tasks + (wgs if variant == "prefork_join" else []):
            main.wait_stream(st)

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()                                            # warm-up (fills the event pool)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    want = [o.clone() for o in outs]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        step()
    print(f"{variant}: capture ended without a fault", flush=True)
    for o, b in zip(outs, bufs):
        o.zero_(); b.zero_()
    g.replay()
    torch.cuda.synchronize()
    ok = all(torch.equal(o, w) for o, w in zip(outs, want))
    print(f"{variant}: replay {'matches' if ok else 'DIFFERS from'} the eager run", flush=True)
    return ok


if __name__ == "__main__":
    if len(sys.argv) > 1:
        import faulthandler
        faulthandler.enable()
        sys.exit(0 if pattern(sys.argv[1]) else 4)
    for v in ("single", "fresh", "pooled", "prefork", "prefork_join"):
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True, timeout=300)
        tail = (r.stdout + r.stderr).strip().splitlines()[-12:]
        print(f"RESULT {v}: exit code {r.returncode}" + (" (signal)" if r.returncode < 0 else ""), flush=True)
        for ln in tail:
            print("    " + ln, flush=True)
