#!/usr/bin/env python3
"""Headline benchmark: images/sec, forward+backward, ViT-S/16 + MoE (E=16, top-k=4, multi-gate),
224x224, batch 128 per GPU (BASELINE.json configs[1]), synthetic data, random-init weights.

    python bench.py --gpus N --steps K --warmup W [--dtype f16|f32] [--batch 128]

One "step" = what one reference training iteration does to the backbone for one batch
(train/train_utils.py:423-457, multi_gate joint path; models/models.py:299-301): refresh the
operand copies of the weights, then for EACH of the 2 tasks one full backbone forward (task's
gate) and one full backward (d tokens + 0.01 * cv_loss), gradients accumulated; for N > 1 the
experts are sharded E/N per rank (expert parallel: all-to-all of the routed rows over RCCL, dense gradients
all-reduced; utils/common_config.py:179-185) - the primary value when E % N == 0 - and the replicated-experts
data-parallel form (the reference's --moe_data_distributed mode, :179-181) is timed too (sub-objects "ep" / "dp").
At N = 1 the fp32 run (train_fastmoe.py's arithmetic) follows the fp16 one as the sub-object "f32".  Decoder heads / losses are out
of scope: d tokens is a fixed synthetic tensor.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field definitions).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK = {"f16": 2500.0, "bf16": 2500.0, "f32": 157.3}      # dense MFMA TFLOP/s, MI355X_MICROARCH.md
PEAK_HBM = 8000.0                          # GB/s, MI355X_MICROARCH.md
METRIC = "images/sec fwd+bwd ViT-S/16 MoE(E=16,k=4) 224^2 bs=128"
EP_WATCHDOG_S = 420            # N > 1: the expert-parallel leg may not hang the whole line (see attempt())
DP_WATCHDOG_S = 300            # ... nor the data-parallel one (it takes well under a minute when it works)
AGREE_S = 120                  # after each leg the ranks agree on its outcome over a CPU group, at most this long
SHARED_WATCHDOG_S = 240       # the optional dp_shared_stem leg (the plain dp leg takes well under a minute)
_JSON_OUT = sys.stdout
CV_WEIGHT = 0.01                           # --moe_noisy_gate_loss_weight default (train_fastmoe.py:118; applied at train/train_utils.py:277)

# The workloads a line can be quoted on.  configs[1] is BASELINE.json's metric configuration (the default and the driver's line);
# configs[3] / configs[4] are the ViT-Base configurations in their single-GPU form (all experts local):
# the regime where the expert grouped GEMMs contract over K = 768 / 3072 and the MFMA roof, not operand delivery, is the bound.
# They are timed in the default run as the sub-objects "configs3" / "configs4" (own roofline), or alone with --config 3 / 4.
WORKLOADS = {
    1: dict(name="configs[1]: ViT-Small/16 + MoE E=16 top-k=4 multi_gate, synthetic 224x224", batch=128,
            cfg=dict(img_size=(224, 224), embed_dim=384, depth=12, num_heads=12, mlp_ratio=4.0, moe_mlp_ratio=1.0,
                     moe_experts=16, moe_top_k=4, gate_dim=386, multi_gate=True)),
    3: dict(name="configs[3], single-GPU form (all 64 experts local): ViT-Base/16 + MoE E=64 top-k=4, synthetic 224x224, one pass",
            batch=128,
            cfg=dict(img_size=(224, 224), embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, moe_mlp_ratio=1.0,
                     moe_experts=64, moe_top_k=4, gate_dim=768, multi_gate=False)),
    4: dict(name="configs[4], single-GPU form: ViT-Base/16 + MoE E=16 top-k=4 moe_mlp_ratio=4, NYUD 2-task multi_gate, "
                 "synthetic 480x640", batch=8,
            cfg=dict(img_size=(480, 640), embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, moe_mlp_ratio=4.0,
                     moe_experts=16, moe_top_k=4, gate_dim=770, multi_gate=True)),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # >= 10 warm-up + >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", choices=["f16", "bf16", "f32"], default="f16")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU and step (default: the workload's own - 128, 128, 8)")
    ap.add_argument("--config", type=int, choices=sorted(WORKLOADS), default=1, help="which BASELINE.json configuration the line is "
                    "quoted on: 1 = the metric's (default); 3 / 4 = the ViT-Base configurations, single-GPU form (N = 1 only)")
    ap.add_argument("--no-vitb", action="store_true", help="N = 1, --config 1: skip the configs[3] / configs[4] sub-objects")
    ap.add_argument("--no-skew", action="store_true", help="N = 1, --config 1: skip the skewed-routing sub-object \"skew\"")
    ap.add_argument("--checkpoint", action="store_true",
                    help="the reference's default memory mode: keep block inputs only, recompute each block in backward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--noisy", action="store_true", help="variant: noisy-gate training (vmoe_noisy_std = 1, caller-supplied noise, CDF load loss)")
    ap.add_argument("--skew", action="store_true", help="variant: +8 on expert 0's gate logit - every token routes to it (4x the mean load)")
    ap.add_argument("--dp-parts", type=int, default=6, help="N > 1, replicated experts: cut the step into this many graphs with an all-reduce behind each")
    ap.add_argument("--serial-tasks", action="store_true", help="run the task passes one after the other on one "
                    "stream (default: one HIP stream per task pass, gradients summed at the end)")
    ap.add_argument("--wgrad-streams", action="store_true", help="also launch the weight-gradient GEMMs of each "
                    "pass on their own stream (fork / join by events).  With --serial-tasks the pattern is captured into "
                    "the hipGraph; with the default concurrent task streams the nested fork cannot be captured (ROCm 7.2 "
                    "faults in hipStreamEndCapture: tools/nested_capture_probe.py) and the step runs eagerly "
                    "(config.capture_refused says so); off under expert parallelism")
    ap.add_argument("--share-stem", action="store_true", help="make the shared-stem schedule the primary figure: the "
                    "task-independent stem (patch embedding + the blocks below the first MoE block) computed once per step for "
                    "all task passes instead of once per pass (MultiTaskStep share_stem: same gradients).  Without the flag "
                    "`value` is the reference's schedule - one FULL pass per task - and the shared-stem step is timed as well and "
                    "reported as the sub-object \"shared_stem\"")
    ap.add_argument("--no-share-stem", action="store_true", help="do not time the shared-stem schedule at all")
    ap.add_argument("--ep", action="store_true", help="N > 1: time ONLY the expert-parallel form (experts sharded over the "
                    "ranks, all-to-all over RCCL); default: expert parallel (primary, when E %% N == 0) AND data parallel")
    ap.add_argument("--ep-capacity", type=float, default=0.0, help="N > 1, expert parallel: exchange the routed rows with a fixed "
                    "capacity of this multiple of the uniform share per (source, destination) pair (e.g. 1.25) instead of the exact "
                    "a2a-v: no host read inside the step, one flag read at its end, the step repeated on the exact path on overflow "
                    "(MultiTaskStep ep_capacity); default 0 = exact exchange")
    ap.add_argument("--ep-chunks", type=int, default=2, help="N > 1: the third leg \"ep_overlap\" cuts every expert-parallel exchange "
                    "into this many chunks of local experts and overlaps them with the experts' GEMMs inside one pass (MultiTaskStep "
                    "ep_chunks; same results bit for bit); 1 = no such leg")
    ap.add_argument("--ep-native", action="store_true", help="N > 1 over RCCL: one more leg \"ep_native\", LAST: the expert-parallel "
                    "exchange through the library's own RCCL entry points (m3_ep_dispatch / m3_ep_return; m3vit_amd/ep_native.py) instead "
                    "of torch.distributed, with --ep-chunks chunks.  Opt-in: those entry points have only ever run on a one-rank "
                    "communicator (the build box has one GPU)")
    ap.add_argument("--dp-only", action="store_true", help="N > 1: time only the replicated-experts data-parallel form")
    ap.add_argument("--no-f32", action="store_true", help="N = 1: skip the fp32 run reported as the sub-object \"f32\"")
    ap.add_argument("--module-path", action="store_true", help="N = 1: time ONLY the drop-in module path (install_fmoe_shim() + "
                    "m3vit_amd.vit.VisionTransformerMoE + torch.autograd: backbone(x, task_id) per task, one loss.backward()) and "
                    "report it as `value`; by default it is timed after the executor and reported as the sub-object \"module_path\"")
    ap.add_argument("--no-module-path", action="store_true", help="N = 1: do not time the drop-in module path")
    a = ap.parse_args()
    if a.checkpoint and a.wgrad_streams:
        ap.error("--checkpoint re-uses the activation buffers a wgrad stream may still read: pick one of the two")
    if a.ep and a.dp_only:
        ap.error("--ep and --dp-only exclude each other")
    if a.config != 1 and a.gpus > 1:
        ap.error("--config 3 / 4 are single-GPU forms (the multi-GPU legs are quoted on configs[1])")
    if a.batch is None:
        a.batch = WORKLOADS[a.config]["batch"]
    return a


class GemmTimer:
    """HIP-event timing of every m3_gemm_nt launch (events on the launch stream)."""

    def __init__(self, ops):
        self.ops = ops
        self.orig = ops.gemm_nt
        self.records = []       # (start, end, flops, grouped)

    def __enter__(self):
        def timed(A, B, C, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = self.orig(A, B, C, **kw)
            e.record()
            M = kw.get("M")
            if M is None:
                M = kw["a_row_idx"].numel() if kw.get("a_row_idx") is not None else A.shape[0]
            N, K = B.shape[-2], B.shape[-1]
            # algorithmic bytes: every operand / output element once (a routed row counts once per use)
            byts = M * K * A.element_size() + B.numel() * B.element_size() + M * N * C.element_size()
            for name, esz in (("pre_out", A.element_size()), ("gelu_grad_pre", A.element_size()), ("residual", 4)):
                if kw.get(name) is not None:
                    byts += M * N * esz
            self.records.append((s, e, 2.0 * M * N * K, kw.get("group_offsets") is not None, float(byts), int(K)))
            return r
        self.ops.gemm_nt = timed
        return self

    def __exit__(self, *a):
        self.ops.gemm_nt = self.orig

    def summary(self):
        tot_ms = tot_fl = g_ms = g_fl = tot_by = g_by = 0.0
        by_k = {}
        for s, e, fl, grouped, byts, K in self.records:
            ms = s.elapsed_time(e)
            tot_ms += ms; tot_fl += fl; tot_by += byts
            if grouped:
                g_ms += ms; g_fl += fl; g_by += byts
                a = by_k.setdefault(K, [0.0, 0.0, 0])
                a[0] += ms; a[1] += fl; a[2] += 1
        n = len(self.records)
        return dict(launches=n, avg_us=1e3 * tot_ms / max(n, 1), tflops=tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms else 0.0,
                    flops_per_launch=tot_fl / max(n, 1), bytes_per_launch=tot_by / max(n, 1),
                    gbps=tot_by / (tot_ms * 1e-3) / 1e9 if tot_ms else 0.0,
                    grouped_gbps=g_by / (g_ms * 1e-3) / 1e9 if g_ms else 0.0,
                    grouped_launches=sum(1 for r in self.records if r[3]),
                    grouped_by_k={str(K): {"launches": a[2], "avg_us": round(1e3 * a[0] / a[2], 1),
                                           "tflops": round(a[1] / (a[0] * 1e-3) / 1e12, 1)} for K, a in sorted(by_k.items()) if a[0] > 0},
                    grouped_tflops=g_fl / (g_ms * 1e-3) / 1e12 if g_ms else 0.0)


def cpu_baseline(cfg_kwargs, batch, threads):
    """Oracle (torch fp32 CPU restatement, 'port') timed on the host cores for one step on a bounded
    sample (`batch` images, both task passes, fwd+bwd)."""
    from oracle import ref_torch as R
    torch.set_num_threads(threads)
    cfg = R.BackboneCfg(**cfg_kwargs)
    P = {k: v.requires_grad_() for k, v in R.init_backbone_params(cfg, seed=1).items()}
    img = torch.randn(batch, 3, *cfg.img_size)
    dtok = torch.randn(batch, cfg.num_tokens, cfg.embed_dim)

    def step():
        loss = 0.0
        for task in (range(cfg.num_tasks) if (cfg.multi_gate or cfg.gate_task_specific_dim >= 0) else [None]):
            tok, cv, _ = R.backbone_forward(P, cfg, img, task)
            loss = loss + (tok * dtok).sum() + CV_WEIGHT * cv
        loss.backward()
        for p in P.values():
            p.grad = None
    step()                                  # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 2 or time.perf_counter() - t0 < 10.0:
        step(); n += 1
    dt = (time.perf_counter() - t0) / n
    return batch / dt, n


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly with --gpus N: become the launcher (one rank per GPU), as a CHILD process - nothing has
        # touched the GPU yet in this one
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    # stdout carries exactly ONE line, the JSON: libraries that print there (gloo's "[Gloo] Rank 0 is connected to ..." banner
    # in CPU-side rehearsals) are sent to stderr - file descriptor 1 is pointed at stderr and the JSON goes to a saved copy
    global _JSON_OUT
    sys.stdout.flush()
    _JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: running {world} rank(s)", file=sys.stderr, flush=True)
    dist = None
    # rehearsal knobs (tests only): M3_BENCH_BACKEND=gloo + M3_BENCH_ONE_DEVICE=1 run N ranks on ONE GPU
    backend = os.environ.get("M3_BENCH_BACKEND", "nccl")
    dev_index = 0 if (world == 1 or os.environ.get("M3_BENCH_ONE_DEVICE") == "1") else local_rank
    if world > 1 and os.environ.get("M3_BENCH_ONE_DEVICE") == "1":
        # several PROCESSES on one device share its hardware queues: with HIP's default of 4 per process the ranks' task /
        # wgrad / exchange streams stall one another for seconds per step (profiles/r05_dp_two_rank_stream_count.txt);
        # read by the HIP runtime when it initialises, so set before the first device call
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from m3vit_amd import ops
    from m3vit_amd.config import BackboneConfig, init_params
    from m3vit_amd.step import MultiTaskStep

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    # the library that really moves the bytes: backend "nccl" is RCCL on ROCm; gloo only in CPU-side rehearsals
    coll = {"nccl": "RCCL"}.get(backend, backend)

    def run_mode(dtype_name, expert_parallel, want_roofline, share_stem=False, workload=None, batch=None, skew=None, steps=None,
                 warmup=None, ep_chunks=1, ep_native=False):
        """One timed configuration: W untimed warm-up steps, exactly K timed steps between barriers, max over ranks.
        Returns the fields of the JSON line that depend on the mode."""
        wl = WORKLOADS[args.config if workload is None else workload]
        batch = (args.batch if workload is None else wl["batch"]) if batch is None else batch
        skew = args.skew if skew is None else skew
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        cfg = BackboneConfig(**wl["cfg"])
        torch.cuda.reset_peak_memory_stats(dev)
        dtype = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[dtype_name]
        params = init_params(cfg, seed=1)                       # same weights on every rank
        # m3vit_amd/step.py: one engine context + HIP stream per task pass, hipGraph capture, and for N > 1 the
        # all-reduce of the upper blocks' gradients overlapped with the lower blocks' backward
        runner = MultiTaskStep(cfg, params, batch=batch, dtype=dtype, device=str(dev), cv_weight=CV_WEIGHT,
                               parallel_tasks=not args.serial_tasks, graph=not args.no_graph, world=world, rank=rank,
                               expert_parallel=expert_parallel, wgrad_streams=args.wgrad_streams, dp_parts=args.dp_parts,
                               checkpoint=args.checkpoint, share_stem=share_stem,
                               ep_capacity=args.ep_capacity if expert_parallel else 0.0,
                               ep_chunks=ep_chunks if expert_parallel else 1, ep_native=ep_native and expert_parallel)
        use_ep, par_tasks, ntasks = runner.use_ep, runner.par or runner.par_ep, len(runner.tasks)
        g = torch.Generator().manual_seed(1000 + rank)          # each rank its own images
        images = torch.randn(batch, 3, *cfg.img_size, generator=g).to(dev)
        dtok = (torch.randn(batch, cfg.num_tokens, cfg.embed_dim, generator=g) * 0.05).to(dev)
        noises = bias = None
        if args.noisy:          # second run with std = 1 and a caller-supplied noise tensor (seed 2)
            cfg.vmoe_noisy_std = 1.0
            gn = torch.Generator().manual_seed(2)
            noises = {t: {i: torch.randn(batch * cfg.num_tokens, cfg.moe_experts, generator=gn).to(dev)
                          for i in range(cfg.depth) if i % 2 == 1} for t in runner.tasks}
        if skew:           # routing-skew variant: one expert receives ~4x the mean load
            b = torch.zeros(cfg.moe_experts); b[0] = 8.0
            bias = {i: b.to(dev) for i in range(cfg.depth) if i % 2 == 1}
        runner.bind(images, dtok, noises=noises, logit_bias=bias)     # inputs resident in HBM before anything is timed
        step, serial_step = runner.step_eager, runner.serial_step
        tag = f"{dtype_name}/{'ep' if use_ep else ('dp' if world > 1 else 'single')}"
        log(f"{tag}: engine ready")
        step()
        torch.cuda.synchronize()
        log(f"{tag}: first step done")
        # The step is ~760 dependent kernel launches with no host decisions in between: capture it once
        # into hipGraph(s) and replay (the launch-bound inner loop is the graph, not the Python loop).
        if runner.capture():
            log(f"{tag}: step captured into a hipGraph" + (f" ({len(runner.graphs)} parts, an all-reduce behind each)" if runner.two_parts else ""))
        elif runner.want_graph:
            log(f"{tag}: graph capture unavailable ({runner.capture_error}); running eagerly")
        run = runner.step
        for i in range(warmup):
            run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        t_host = time.perf_counter() - t0
        barrier()
        dt = time.perf_counter() - t0
        log(f"{tag}: host launch time {1e3 * t_host / steps:.2f} ms/step")
        if world > 1:
            t = torch.tensor([dt], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        ms_per_step = 1e3 * dt / steps
        log(f"{tag}: timed region done: {ms_per_step:.2f} ms/step")
        step_flops = 3.0 * cfg.fwd_flops_per_image() * batch * ntasks
        if runner.share_stem:      # FLOPs actually executed: the stem (patch embedding + blocks below the first MoE block) once
            step_flops -= 3.0 * cfg.stem_flops_per_image() * batch * (ntasks - 1)
        res = {"value": round(world * batch * steps / dt, 2), "ms_per_step": round(ms_per_step, 3),
               "model_tflops": round(step_flops * steps / dt / 1e12 * world, 2), "launch": runner.launch,
               "task_streams": ntasks if par_tasks else 1,
               "wgrad_streams": sum(1 for e in runner.engs if e.wg_stream is not None),
               "capture_refused": runner.capture_refused,
               "task_passes": ntasks, "tokens_per_image": cfg.num_tokens,
               "activation_checkpointing": bool(args.checkpoint), "shared_stem": bool(runner.share_stem),
               "peak_hbm_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
               "ep_exchange": (None if not use_ep else ((f"exact a2a-v in {runner.ep_chunks} chunks of local experts, overlapped with the expert "
                                                         "GEMMs inside a pass" if runner.ep_chunks > 1 else
                                                         "exact a2a-v (2 W split sizes read per MoE layer and pass)") if not runner.ep_capacity
                               else f"fixed capacity {runner.ep_capacity} x R / W per pair, {runner.ep_repeats} step(s) repeated on the exact path")),
               "ep_transport": (None if not use_ep else ("library RCCL entry points (m3_ep_dispatch / m3_ep_return: grouped ncclSend / ncclRecv)"
                                                         if runner.eng.ep_native is not None else f"torch.distributed all_to_all_single ({coll})")),
               "parallelism": "single" if world == 1 else (f"dp{world}+ep{world} (experts sharded, {coll} all-to-all + all-reduce)"
                                                            if use_ep else f"dp{world} (replicated experts, {coll} all-reduce)")}
        if want_roofline:
            res["roofline"] = roofline_of(ops, runner, dtype_name, step, serial_step, par_tasks, ntasks,
                                          replay_traffic=(batch == wl["batch"] and not skew and not args.noisy),
                                          traffic_cfg=next(k for k, v in WORKLOADS.items() if v is wl))
        res["workload"], res["batch"], res["steps"], res["warmup"] = wl["name"], batch, steps, warmup
        del runner
        torch.cuda.empty_cache()
        return res

    def roofline_of(ops, runner, dtype_name, step, serial_step, par_tasks, ntasks, replay_traffic=True, traffic_cfg=1):
        # ---- roofline of the dominant kernel (m3_gemm_nt: every Linear / FMoELinear forward + input-gradient GEMM;
        # two device kernels behind it, gemm_nt_dma_kernel for short K and gemm_nt_kernel), measured with HIP events
        # around each launch, on the launch stream, in a few extra instrumented steps.  Primary figures: the launches
        # of a step issued on ONE stream, i.e. each kernel by itself - this is what rocprofv3's per-kernel duration
        # reports too (profiles/).  "as_timed" repeats the measurement with the task passes on concurrent streams as
        # in the timed region: there the event pair also spans the time a launch waits for CU slots held by the other
        # pass's kernels, so it is an upper bound of the kernel's duration, not its resource time.
        n_inst = min(3, args.steps)
        with GemmTimer(ops) as gt1:
            for _ in range(n_inst):
                serial_step()
            torch.cuda.synchronize()
        g1 = gt1.summary()
        gs = None
        if par_tasks:
            with GemmTimer(ops) as gt:
                for _ in range(n_inst):
                    step()
                torch.cuda.synchronize()
            gs = gt.summary()
        peak = PEAK[dtype_name]
        # HBM traffic per launch of that kernel: PMC numbers cannot be read from inside the process; they come from the
        # newest committed rocprofv3 --pmc passes over this same workload (profiles/rNN_pmc_traffic.json, named in
        # traffic_source together with the commit they were taken at) - a replayed measurement, not a live one
        traffic = traffic_source = None
        try:
            import glob
            from m3vit_amd._lib import csrc_sha16
            # (the metric's configuration: r*_pmc_traffic*.json; the ViT-Base configurations: r*_cfgN_hbm_bytes_per_launch.json)
            pat = "r*_pmc_traffic*.json" if traffic_cfg == 1 else f"r*_cfg{traffic_cfg}_hbm_bytes_per_launch.json"
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pat)), reverse=True):
                with open(f) as fh:
                    pm = json.load(fh)
                if pm.get("dtype") != dtype_name or not replay_traffic:
                    continue
                where = f"profiles/{os.path.basename(f)} (rocprofv3 --pmc passes" + \
                        (f", taken at commit {pm['head']}" if pm.get("head") else "")
                if pm.get("csrc_sha16") != csrc_sha16():
                    # measured on other kernels than the ones in this tree: say so instead of replaying it
                    traffic_source = where + "): STALE - the kernel sources changed since; traffic withheld"
                else:
                    ks = pm["kernels"]
                    if "gemm_nt_all" in ks:
                        traffic = round(ks["gemm_nt_all"]["hbm_bytes_per_launch"])
                    else:                      # every kernel behind m3_gemm_nt, weighted by its launches
                        parts = [v for k_, v in ks.items() if k_.startswith("gemm_nt")]
                        traffic = round(sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in parts) /
                                        max(1, sum(v["launches_sampled"] for v in parts)))
                    traffic_source = where + "; not measured in this run)"
                break
        except Exception:
            traffic = traffic_source = None
        # Which roof binds: the launches' arithmetic intensity (algorithmic FLOPs / algorithmic bytes) against the machine
        # balance peak_flops / peak_bandwidth.  fp16: ~170 FLOP/B < 312 -> the HBM roof (the K = 384 GEMMs of
        # this model are on the HBM side of the ridge); fp32: ~85 FLOP/B > 19.7 -> the MFMA roof.
        intensity = g1["flops_per_launch"] / g1["bytes_per_launch"]
        hbm_bound = intensity < peak * 1e12 / (PEAK_HBM * 1e9)
        roofline = {"kernel": "m3_gemm_nt (gemm_nt_dma_kernel + gemm_nt_kernel)",
                    "bound": "hbm" if hbm_bound else "mfma",
                    "achieved": round(g1["gbps"], 1) if hbm_bound else round(g1["tflops"], 2),
                    "peak": PEAK_HBM if hbm_bound else peak,
                    "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": round(g1["gbps"] / PEAK_HBM, 4) if hbm_bound else round(g1["tflops"] / peak, 4),
                    "traffic": traffic, "traffic_source": traffic_source,
                    "avg_launch_us": round(g1["avg_us"], 2), "launches_per_step": g1["launches"] // n_inst,
                    "flops_per_launch": g1["flops_per_launch"], "bytes_per_launch": round(g1["bytes_per_launch"]),
                    "flop_per_byte": round(intensity, 1),
                    "mfma_tflops": round(g1["tflops"], 2), "mfma_frac": round(g1["tflops"] / peak, 4),
                    "hbm_gbps": round(g1["gbps"], 1), "hbm_frac": round(g1["gbps"] / PEAK_HBM, 4),
                    "expert_grouped_gemm_tflops": round(g1["grouped_tflops"], 2),
                    "expert_grouped_gemm_frac": round(g1["grouped_tflops"] / peak, 4),
                    "expert_grouped_gemm_gbps": round(g1["grouped_gbps"], 1),
                    # the expert grouped launches by contraction length K (FC1 / FC2 input gradient: K = D; FC2 / FC1 input
                    # gradient over H: K = H): at ViT-Base width K = 3072 is where the 256 x 256-tile kernel runs
                    "expert_grouped_gemm_by_k": {k_: dict(v, frac=round(v["tflops"] / peak, 4)) for k_, v in g1["grouped_by_k"].items()},
                    "launch_mode": "the step's launches on one stream (kernel by itself)"}
        if gs is not None:
            roofline["as_timed"] = {"launch_mode": f"{ntasks} concurrent task streams; event pairs include waiting for CU slots",
                                    "mfma_tflops": round(gs["tflops"], 2), "hbm_gbps": round(gs["gbps"], 1),
                                    "avg_launch_us": round(gs["avg_us"], 2)}
        return roofline

    # ---- which configurations this invocation times
    #  N = 1: the primary dtype (f16, the reference's AMP arithmetic) with the roofline object, then - unless --dtype
    #         was given explicitly - the fp32 run (the arithmetic of train_fastmoe.py) as the sub-object "f32";
    #  N > 1: north_star's expert-parallel form (experts sharded E/N per rank, all-to-all over RCCL) is the primary
    #         when the experts divide over the ranks, and the replicated-experts data-parallel form (the reference's
    #         --moe_data_distributed mode) is timed as well: both appear as the sub-objects "ep" and "dp".
    E = WORKLOADS[args.config]["cfg"]["moe_experts"]
    extra = {}

    def emit(main_res, final=False):
        """rank 0 prints THE JSON line (once: at the end, or from the watchdog)"""
        out = {
            "metric": METRIC if args.config == 1 else "images/sec fwd+bwd, " + WORKLOADS[args.config]["name"],
            "value": main_res["value"], "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": WORKLOADS[args.config]["name"],
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world, "task_passes": main_res["task_passes"],
                       "tokens_per_image": main_res["tokens_per_image"], "cv_loss_weight": CV_WEIGHT,
                       "launch": main_res["launch"],
                       "routing": ("noisy gate std=1 (supplied noise, CDF load loss)" if args.noisy else "deterministic (std=0)") +
                                  (", skewed: expert 0 in every token's top-k" if args.skew else ""),
                       "task_streams": main_res["task_streams"], "wgrad_streams": main_res["wgrad_streams"],
                       "capture_refused": main_res["capture_refused"],
                       "activation_checkpointing": main_res["activation_checkpointing"],
                       "shared_stem": main_res["shared_stem"],
                       "peak_hbm_gib": main_res["peak_hbm_gib"],
                       "parallelism": main_res["parallelism"]},
            "model_tflops": main_res["model_tflops"],
        }
        if "roofline" in main_res:
            out["roofline"] = main_res["roofline"]
        out.update(extra)
        if final and rank == 0 and world == 1 and not args.no_cpu_baseline:
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            cores = max(1, min(cores, 16))          # the GPU box gives one GPU a 16-core share
            cpu_b = min(args.cpu_batch, args.batch) if args.config != 4 else 1
            v, n = cpu_baseline(WORKLOADS[args.config]["cfg"], cpu_b, cores)
            out["cpu_baseline"] = {"value": round(v, 3), "unit": "images/s", "cores": cores, "kind": "port",
                                   "sample": f"{n} full steps ({main_res['task_passes']} task pass(es) fwd+bwd, fp32 torch CPU oracle) at "
                                             f"batch {cpu_b}, scaled per image"}
        if rank == 0:
            print(json.dumps(out), file=_JSON_OUT, flush=True)


    def attempt(tag, fn, watchdog_s=None, on_timeout=None):
        """run one configuration; an exception (first contact with a collective library, an out-of-memory ...) costs
        that configuration only: the text goes into the JSON line as "<tag>_error".  watchdog_s: a collective that never
        returns cannot be caught - after that many seconds on_timeout() reports what was measured so far and the process
        ends (every rank has its own timer)."""
        import threading
        import traceback
        timer = None
        if watchdog_s and on_timeout is not None:
            timer = threading.Timer(watchdog_s, on_timeout)
            timer.daemon = True
            timer.start()
        try:
            return fn()
        except Exception as exc:      # noqa: BLE001 - whatever it is, the other configuration's line must survive
            log(f"{tag}: FAILED: {type(exc).__name__}: {exc}")
            traceback.print_exc(file=sys.stderr)
            extra[f"{tag}_error"] = f"{type(exc).__name__}: {exc}"[:400]
            try:
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
            except Exception:         # noqa: BLE001
                pass
            return None
        finally:
            if timer is not None:
                timer.cancel()

    def module_leg(dtype_name):
        """the drop-in path: what a trainer that imports the reference's names gets (tools/module_bench.py)"""
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from module_bench import module_path_step_time
        return module_path_step_time(dtype_name, "auto", steps=args.steps, warmup=args.warmup, batch=args.batch, log=log)

    main_res = None
    if world == 1 and args.module_path:
        mp = module_leg(args.dtype)
        cfg0 = BackboneConfig(**WORKLOADS[1]["cfg"])
        main_res = {"value": mp["value"], "ms_per_step": mp["ms_per_step"], "model_tflops": mp["model_tflops"],
                    "launch": mp["path"], "task_streams": cfg0.num_tasks, "wgrad_streams": 0, "capture_refused": None,
                    "task_passes": cfg0.num_tasks, "tokens_per_image": cfg0.num_tokens, "activation_checkpointing": False,
                    "shared_stem": False, "peak_hbm_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                    "parallelism": "single"}
        extra["module_path"] = mp
    elif world == 1:
        # `value` is the reference's schedule: one FULL backbone pass per task.  Every task pass of a step reads the same images
        # (train/train_utils.py:248-256) and the patch embedding + block 0 see neither task id nor gate, so they can be computed
        # once per step with their backward on the summed d x (m3vit_amd/step.py share_stem; identical gradients:
        # tests/test_engine.py::test_shared_stem_step_matches_per_task_stems) - 23 instead of 24 block passes.  That step is
        # timed too and reported BESIDE the headline as "shared_stem" (or as the headline with --share-stem, then with the
        # reference's schedule beside it as "per_task_stems").
        base1 = args.config == 1          # shared stem, fp32, module path and the sub-objects below belong to the metric's configuration
        can_share = not args.serial_tasks and not args.no_share_stem and base1
        share = bool(args.share_stem) and can_share
        main_res = run_mode(args.dtype, False, True, share_stem=share)
        if can_share:
            other = "per_task_stems" if share else "shared_stem"
            ot = attempt(other, lambda: run_mode(args.dtype, False, False, share_stem=not share))
            if ot is not None:
                extra[other] = {"value": ot["value"], "ms_per_step": ot["ms_per_step"], "launch": ot["launch"],
                                "model_tflops": ot["model_tflops"],
                                "note": ("the reference's schedule: patch embedding + block 0 computed by every task pass" if share else
                                         "patch embedding + block 0 computed once per step for both task passes, their backward once on "
                                         "the summed d x: same gradients, 23 instead of 24 block passes (model_tflops counts the FLOPs "
                                         "executed)")}
        if args.dtype == "f16" and not args.no_f32 and base1:
            f32 = attempt("f32", lambda: run_mode("f32", False, True, share_stem=share))
            if f32 is not None:
                extra["f32"] = {k: f32[k] for k in ("value", "ms_per_step", "model_tflops", "launch", "roofline")}
                extra["f32"]["dtype"] = "f32"
        plain = base1 and not args.serial_tasks and not args.checkpoint and not (args.noisy or args.skew) and not args.no_graph
        sub_steps, sub_warm = min(args.steps, 20), min(args.warmup, 5)
        if plain and not args.no_skew and args.batch == WORKLOADS[1]["batch"]:
            # routing-skew variant beside the headline: +8 on expert 0's gate logit puts it into EVERY token's top-k
            # (4x the mean load) - what the device-side load balancing of the grouped weight gradients and the tile prefix of the
            # grouped GEMMs are for; same work, same FLOPs, only the distribution over the experts changes
            sk = attempt("skew", lambda: run_mode(args.dtype, False, False, share_stem=share, skew=True, steps=sub_steps, warmup=sub_warm))
            if sk is not None:
                extra["skew"] = {"value": sk["value"], "ms_per_step": sk["ms_per_step"], "steps": sk["steps"], "warmup": sk["warmup"],
                                 "ratio_to_value": round(sk["value"] / main_res["value"], 4),
                                 "routing": "expert 0 in every token's top-k (4x the mean load), the other three picks as routed"}
        if plain and not args.no_vitb and args.dtype in ("f16", "bf16"):
            # the ViT-Base configurations, single-GPU form: the expert grouped GEMMs contract over K = 768 / 3072 there
            for w in (3, 4):
                vb = attempt(f"configs{w}", lambda w=w: run_mode(args.dtype, False, True, workload=w, skew=False, steps=sub_steps,
                                                                  warmup=sub_warm))
                if vb is not None:
                    extra[f"configs{w}"] = {k: vb[k] for k in ("workload", "batch", "value", "ms_per_step", "model_tflops", "launch",
                                                               "task_passes", "tokens_per_image", "steps", "warmup",
                                                               "peak_hbm_gib", "roofline")}
                    extra[f"configs{w}"]["unit"] = "images/s"
        if not args.no_module_path and plain:
            # The number above is the executor driven directly (m3vit_amd.step.MultiTaskStep).  What a maintainer who drops
            # the library in calls is the MODULE API - install_fmoe_shim() + VisionTransformerMoE.forward(x, task_id) per task +
            # one loss.backward() (models/models.py:299-320, train/train_utils.py:423-457): timed here on the same
            # configuration, same dtype (and fp32, train_fastmoe.py's arithmetic), reported beside the headline
            mp = attempt("module_path", lambda: module_leg(args.dtype))
            if mp is not None:
                mp["ratio_to_value"] = round(mp["value"] / main_res["value"], 4)
                extra["module_path"] = mp
                if "f32" in extra:
                    mp32 = attempt("module_path_f32", lambda: module_leg("f32"))
                    if mp32 is not None:
                        mp32["ratio_to_f32_value"] = round(mp32["value"] / extra["f32"]["value"], 4)
                        extra["module_path"]["f32"] = mp32
    else:
        want_ep = args.ep or (not args.dp_only and E % world == 0)
        want_dp = not args.ep
        results = {}
        sub = ("value", "ms_per_step", "model_tflops", "launch", "parallelism", "ep_exchange", "ep_transport")
        # The legs, in this order: data parallel first - it needs one collective (all-reduce) and its line must survive whatever
        # the expert-parallel leg (count all-to-all + uneven all-to-all-v on several streams) does at its first contact with
        # RCCL; the opt-in shared-stem data-parallel leg (--share-stem; the one-GPU gloo rehearsal stalls in it, DESIGN
        # section 6) last.  EVERY leg runs under a watchdog on every rank (a collective that never returns cannot be caught: on a
        # time-out rank 0 prints the line from what the earlier legs measured and the process ends), and after every leg the
        # ranks AGREE on its outcome over a CPU (gloo) group: a rank that failed alone (out of memory, a RCCL error) must not
        # walk into the next leg's collectives while its peers still sit in the failed leg's - if any rank failed, every rank
        # records the error, no further collective leg is started, and the line is printed from the legs that finished.
        legs = []
        if want_dp:
            legs.append(("dp", False, lambda: run_mode(args.dtype, False, False, share_stem=False), DP_WATCHDOG_S))
        if want_ep:
            legs.append(("ep", True, lambda: run_mode(args.dtype, True, False), EP_WATCHDOG_S))
            if args.ep_chunks > 1 and not args.ep_capacity and (E // world) % args.ep_chunks == 0:
                legs.append(("ep_overlap", "ep_overlap", lambda: run_mode(args.dtype, True, False, ep_chunks=args.ep_chunks), EP_WATCHDOG_S))
        if want_ep and args.ep_native and backend == "nccl" and not args.ep_capacity:
            nat_chunks = args.ep_chunks if (args.ep_chunks > 1 and (E // world) % args.ep_chunks == 0) else 1
            legs.append(("ep_native", "ep_native", lambda: run_mode(args.dtype, True, False, ep_chunks=nat_chunks, ep_native=True),
                         EP_WATCHDOG_S))
        if want_dp and args.share_stem and not args.serial_tasks:
            legs.append(("dp_shared_stem", "dp_shared_stem", lambda: run_mode(args.dtype, False, False, share_stem=True),
                         SHARED_WATCHDOG_S))
        import datetime
        agree_group = None
        if os.environ.get("M3_BENCH_NO_AGREE") != "1":
            try:          # (a CPU group next to the RCCL one; if gloo cannot come up on this host the legs run without agreement)
                agree_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=AGREE_S))
            except Exception as exc:      # noqa: BLE001
                log(f"no CPU agreement group ({type(exc).__name__}: {exc}); legs run without cross-rank agreement")
                extra["agreement"] = "unavailable"

        def finished():
            return [r for r in results.values() if r is not None]

        def bail(code_if_none=1):
            done = finished()
            if rank == 0 and done:
                emit(max(done, key=lambda r: r["value"]))
            _JSON_OUT.flush()
            os._exit(0 if done else code_if_none)

        def agree(ok):
            """True iff the leg succeeded on EVERY rank (min over the ranks on the CPU group; a peer that never arrives - it
            hangs in a collective until its watchdog ends it - counts as a failure after AGREE_S seconds)"""
            if agree_group is None:
                return ok
            try:
                t = torch.tensor([1 if ok else 0], dtype=torch.int32)
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=agree_group)
                return bool(int(t[0]))
            except Exception as exc:      # noqa: BLE001
                log(f"agreement over the CPU group failed: {type(exc).__name__}: {exc}")
                return False

        for tag, key, fn, wd in legs:
            def hung(tag=tag, wd=wd):
                extra[f"{tag}_error"] = f"no result after {wd} s (a collective that never returned); reporting the legs that finished"
                bail()
            results[key] = attempt(tag, fn, watchdog_s=wd, on_timeout=hung)
            if not agree(results[key] is not None):
                extra.setdefault(f"{tag}_error", "failed on another rank (see its stderr)")
                results[key] = None                   # a leg some rank did not finish has no max-over-ranks time
                names = [l[0] for l in legs]
                extra["legs_skipped"] = names[names.index(tag) + 1:]
                bail()
            extra[tag] = {k: results[key][k] for k in sub}
        # primary: the faster of the expert-parallel and data-parallel forms that ran (in the line as "ep" / "dp"; "dp_shared_stem" beside them).  configs[1]'s experts
        # (E = 16 x 0.6 MB) fit one GPU many times over, so sharding them is a choice, not a need: expert parallelism moves
        # ~3.7 GB of routed rows per step and rank through the xGMI links (DESIGN.md section 6 has the predicted table) where
        # data parallelism moves one 172 MB gradient all-reduce - north_star asks for the all-to-all "only where experts shard"
        # (the shared-stem leg is reported, and eligible as the primary only with --share-stem: same policy as at N = 1)
        ran = [r for r in (results.get(True), results.get("ep_overlap"), results.get("ep_native"), results.get(False),
                           results.get("dp_shared_stem") if args.share_stem else None) if r is not None]
        main_res = max(ran, key=lambda r: r["value"]) if ran else None
        if main_res is None:
            if rank == 0:
                print(json.dumps({"metric": METRIC, "value": None, "n_gpus": world, **extra}), file=_JSON_OUT, flush=True)
            sys.exit(1)
    emit(main_res, final=True)
    if world > 1:
        dist.destroy_process_group()



if __name__ == "__main__":
    main()
