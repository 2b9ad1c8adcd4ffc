#!/usr/bin/env python3
"""Host time of ONE expert-parallel exchange plan at the configs[1] size (W = 8 ranks, E = 16 -> 2 experts per rank,
100 864 routed rows per rank), no collectives involved:
  * round 1: counts .tolist() + ExchangePlan (Python loops over ~100 k rows) + torch.tensor(list) uploads;
  * now    : ops.ep_plan (one kernel + one pinned read of 2 W integers).
    python tools/ep_host_bench.py [--world 8] [--iters 20]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402
from m3vit_amd.ep import ExchangePlan  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--experts", type=int, default=16)
ap.add_argument("--rows", type=int, default=128 * 197 * 4)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
W, E = a.world, a.experts
e_loc = E // W
g = torch.Generator().manual_seed(0)
# every rank routes `rows` rows uniformly over the E experts; this rank receives about the same number
send = torch.bincount(torch.randint(0, E, (a.rows,), generator=g), minlength=E)
recv = torch.stack([torch.bincount(torch.randint(0, E, (a.rows,), generator=g), minlength=E)[:e_loc] for _ in range(W)]).reshape(-1)
send_d, recv_d = send.to(dev), recv.to(dev)
buf = torch.empty(W * a.rows, dtype=torch.int32, device=dev)
pin = torch.empty(2 * W, dtype=torch.int64, pin_memory=True)


def old():
    plan = ExchangePlan(send_d.tolist(), recv_d.tolist(), W, e_loc)
    rg = torch.tensor(plan.regroup, dtype=torch.int32, device=dev)
    cnt = torch.tensor(plan.fwd_expert_count, dtype=torch.int32, device=dev)
    z = torch.zeros(1, dtype=torch.int32, device=dev)
    off = torch.cat((z, torch.cumsum(cnt, 0).to(torch.int32)))
    ts = torch.cat((z, torch.cumsum((cnt + 127) // 128, 0).to(torch.int32)))
    return rg, off, ts, plan


def new():
    return ops.ep_plan(send_d, recv_d, W, e_loc, buf, splits_host=pin)


for name, fn, n in (("round-1 host plan", old, max(2, a.iters // 10)), ("device plan (m3_ep_plan)", new, a.iters)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name:28s} {1e6 * (time.perf_counter() - t0) / n:10.1f} us of host time per exchange plan "
          f"(W = {W}, {e_loc} experts per rank, {int(recv.sum())} rows received)", flush=True)
p_old, p_new = old()[3], new()
assert p_new.regroup.cpu().tolist() == p_old.regroup and p_new.in_splits == p_old.in_splits
print("plans identical")
