#!/usr/bin/env python3
"""300 joint multi-task steps of the fused module path with a real optimizer (AdamW) and a NEW batch tensor every step, twice from
the same initial weights and batches: with the other task's forward started ahead of its call (prefetch) and without.  Checks:
step time and device memory stay flat, the slot count does not grow, every step's prefetch is claimed, and the two runs end
with BIT-IDENTICAL weights (the prefetch changes when a pass runs, never what it computes)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd.config import VIT_SMALL_MOE, BackboneConfig, init_params  # noqa: E402
from m3vit_amd.vit import VisionTransformerMoE  # noqa: E402

cfg = BackboneConfig(**VIT_SMALL_MOE)
B, STEPS = 32, 300
target = (torch.randn(B, cfg.num_tokens, cfg.embed_dim, generator=torch.Generator().manual_seed(7)) * 0.5).cuda()


def train(prefetch):
    m = VisionTransformerMoE(vmoe_noisy_std=0.0, act_dtype=torch.float16, **VIT_SMALL_MOE).cuda().train()
    m.load_state_dict(init_params(cfg, seed=1))
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(11)
    marks = []
    for step in range(STEPS):
        img = torch.randn(B, 3, 224, 224, generator=g).cuda()
        opt.zero_grad(set_to_none=True)
        loss = 0.0
        for t in (0, 1):
            tok, cv = m(img, task_id=t)
            m._fused.prefetch = prefetch
            loss = loss + (tok - target).square().mean() + 0.01 * cv
        loss.backward()
        opt.step()
        if step % 50 == 49:
            v = float(loss)                     # a host sync, as a trainer's logging does
            marks.append((step, time.perf_counter(), torch.cuda.memory_allocated() / 2 ** 30, v))
    fb = m._fused
    for (s0, t0, *_), (s1, t1, a, v) in zip(marks, marks[1:]):
        print(f"prefetch={prefetch}: steps {s0 + 1}-{s1}: {(t1 - t0) / (s1 - s0) * 1e3:.2f} ms/step  allocated {a:.2f} GiB  loss {v:.5f}")
    print(f"prefetch={prefetch}: slots {len(fb.slots)}, hits {fb.prefetch_hits} misses {fb.prefetch_misses}, fallback {m.fused_fallback_reason}")
    assert len(fb.slots) == 2 and fb.prefetch_misses == 0 and all(v == v for *_, v in marks)
    assert (fb.prefetch_hits >= STEPS - 3) if prefetch else fb.prefetch_hits == 0
    assert marks[-1][3] < marks[0][3], "the loss must go down"
    return {n: p.detach().clone() for n, p in m.named_parameters()}, [v for *_, v in marks]


wa, la = train(True)
wb, lb = train(False)
same = all(torch.equal(wa[n], wb[n]) for n in wa)
print("losses equal:", la == lb, " final weights bit-identical:", same)
assert same and la == lb
