#!/usr/bin/env python3
"""Per-parameter gradient error of the fp16 engine NEXT TO the error of the reference's own AMP arithmetic
(pretrain/engine/train_one_epoch.py:35: torch autocast(fp16) around the model + a scaled loss), both against the float64
oracle, at BASELINE configs[1] full size (128 x 224^2, ViT-S/16 + MoE E=16 k=4).

The engine runs the whole batch; the float64 oracle (CPU) and the AMP run (the oracle's functions on the GPU under
torch.autocast, i.e. torch's own fp16 kernels with its cast policy: Linear / matmul / conv in fp16, LayerNorm / softmax in
fp32, fp16 weight gradients cast back to fp32) run the two images whose d tokens are non-zero - images only interact through
the balance loss, whose weight is 0 here (tests/test_full_size.py has the argument).  Both follow the ENGINE's routing
(route_override), so the table isolates arithmetic.  Test infrastructure (it imports the oracle, hence it lives under
tests/): used by tests/test_full_size.py::test_fp16_gradient_error_is_bounded_by_the_reference_amp_arithmetic, and as a
script prints the table:
    python tests/amp_error_table.py [--out profiles/r05_amp_error_table.txt]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rel(a, b):
    a = a.detach().double().cpu().flatten(); b = b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def error_table(batch=128, pick=(3, 101), task=1, loss_scale=1024.0, seed=5):
    from m3vit_amd.config import VIT_SMALL_MOE, BackboneConfig, init_params
    from m3vit_amd.engine import BackboneEngine
    from oracle import ref_torch as R
    cfg = BackboneConfig(**VIT_SMALL_MOE)
    P = init_params(cfg, seed=1, zero_bias=False)
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(batch, 3, *cfg.img_size, generator=g)
    pick = list(pick)
    N, D, k = cfg.num_tokens, cfg.embed_dim, cfg.moe_top_k
    eng = BackboneEngine(cfg, P, batch=batch, dtype=torch.float16)
    tok, _ = eng.forward(img.cuda(), task)
    ocfg = R.BackboneCfg(**{kk: getattr(cfg, kk) for kk in ("img_size", "embed_dim", "depth", "num_heads", "mlp_ratio",
                                                             "moe_mlp_ratio", "moe_experts", "moe_top_k", "gate_dim", "multi_gate")})
    moe_blocks = [i for i in range(cfg.depth) if i % 2 == 1]
    ovr = {i: eng.act[i]["gate"]["idx"].view(batch, N, k)[pick].reshape(-1, k).cpu() for i in moe_blocks}
    dsel = torch.randn(len(pick), N, D, generator=torch.Generator().manual_seed(6)) * 0.1
    dtok = torch.zeros(batch, N, D)
    dtok[pick] = dsel
    eng.zero_grad()
    eng.backward(dtok.cuda(), cv_weight=0.0)
    # float64 oracle on the CPU
    P64 = {kk: v.clone().double().requires_grad_() for kk, v in P.items()}
    tok64, _, _ = R.backbone_forward(P64, ocfg, img[pick].double(), task, route_override=ovr)
    (tok64 * dsel.double()).sum().backward()
    # the reference's AMP arithmetic: the same functions on the GPU under autocast, fp32 master parameters, scaled loss
    Pa = {kk: v.clone().cuda().requires_grad_() for kk, v in P.items()}
    with torch.autocast("cuda", dtype=torch.float16):
        toka, _, _ = R.backbone_forward(Pa, ocfg, img[pick].cuda(), task, route_override={i: v.cuda() for i, v in ovr.items()})
    ((toka.float() * dsel.cuda()).sum() * loss_scale).backward()
    rows = []
    for name, gr in eng.grads.items():
        ref = P64[name].grad
        if ref is None:
            continue
        ga = Pa[name].grad / loss_scale
        rows.append((name, rel(gr, ref), rel(ga, ref)))
    return dict(tokens_engine=rel(tok[pick], tok64), tokens_amp=rel(toka, tok64), rows=rows)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    t = error_table()
    lines = [f"tokens vs float64 oracle: engine {t['tokens_engine']:.2e}   torch AMP {t['tokens_amp']:.2e}",
             f"{'parameter gradient':44s} {'engine':>9s} {'torch AMP':>10s} {'ratio':>6s}"]
    for name, e, a_ in sorted(t["rows"], key=lambda r: -r[1] / max(r[2], 1e-30)):
        lines.append(f"{name:44s} {e:9.2e} {a_:10.2e} {e / max(a_, 1e-30):6.2f}")
    worst = max(t["rows"], key=lambda r: r[1] / max(r[2], 1e-30))
    lines.append(f"worst ratio {worst[1] / worst[2]:.2f} ({worst[0]}); worst engine error {max(r[1] for r in t['rows']):.2e}; "
                 f"worst AMP error {max(r[2] for r in t['rows']):.2e}")
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")
