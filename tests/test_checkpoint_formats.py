"""CPU: expert-parallel checkpoint formats (m3vit_amd/checkpoint.py) - rank-shard directories, global single files
with meta, expert-format inference, gate / pos_embed adaptation (utils/moe_utils.py:128-198,
pretrain/utils/moe_checkpoint.py:57-212, utils/common_config.py:31-100)."""
import os

import pytest
import torch

from m3vit_amd import checkpoint as C
from m3vit_amd.config import BackboneConfig, init_params


def _state(E=8, multi_gate=False):
    cfg = BackboneConfig(img_size=(32, 32), embed_dim=32, depth=2, num_heads=2, moe_experts=E, moe_top_k=2,
                         gate_dim=34 if multi_gate else 32, multi_gate=multi_gate)
    return cfg, init_params(cfg, seed=3, zero_bias=False)


def test_rank_shard_directory_round_trip(tmp_path):
    cfg, P = _state()
    W = 4
    d = str(tmp_path / "checkpoint.pth.tar")
    for r in range(W):                                        # what each rank of train_fastmoe would write
        C.save_rank_shard({"state_dict": C.shard_experts(P, r, cfg.moe_experts // W), "epoch": 3}, d, r)
    assert sorted(os.listdir(d)) == [f"{r}.pth" for r in range(W)]
    r1 = torch.load(os.path.join(d, "1.pth"))["state_dict"]
    assert r1 and all(C.is_expert_key(k) for k in r1)         # ranks > 0: expert tensors only
    assert r1["blocks.1.mlp.experts.htoh4.weight"].shape[0] == 2
    base, merged, n = C.merge_rank_shards(d)
    assert n == W and base["epoch"] == 3 and set(merged) == set(P)
    assert all(torch.equal(merged[k], P[k]) for k in P)
    os.remove(os.path.join(d, "2.pth"))
    with pytest.raises(ValueError):
        C.merge_rank_shards(d)


def test_expert_format_inference_and_meta():
    cfg, P = _state()
    loc = C.shard_experts(P, 1, 2)
    assert C.first_expert_dim0(P) == 8 and C.first_expert_dim0(loc) == 2
    assert C.infer_expert_format({"meta": {"expert_format": "local"}}, P) == "local"          # meta wins
    assert C.infer_expert_format({}, P, expected_global_experts=8) == "global"
    assert C.infer_expert_format({}, loc, expected_global_experts=8, expected_world_size=4) == "local"
    assert C.infer_expert_format({"args": {"moe_experts": 8, "world_size": 4}}, loc) == "local"
    assert C.infer_expert_format({}, loc, expected_global_experts=8) == "unknown"
    dense = {k: v for k, v in P.items() if not C.is_expert_key(k)}
    assert C.infer_expert_format({}, dense) == "dense"
    meta = C.build_meta(P, source="train_fastmoe", world_size=4)
    assert meta == {"expert_format": "global", "moe_experts_global": 8, "moe_experts_local": 2, "world_size": 4,
                    "source": "train_fastmoe"}


def test_gate_and_prefix_adaptation():
    cfg, P = _state()
    wrapped = {"module.backbone." + k: v for k, v in P.items()}
    st, fmt = C.to_backbone_state({"state_dict": wrapped}, rank=1, world_size=2, expected_global_experts=8,
                                  multi_gate=True, num_tasks=2)
    assert fmt == "global" and "blocks.1.mlp.gate.0.w_gate" in st and "blocks.1.mlp.gate.w_gate" not in st
    assert torch.equal(st["blocks.1.mlp.gate.1.w_gate"], P["blocks.1.mlp.gate.w_gate"])
    assert torch.equal(st["blocks.1.mlp.experts.h4toh.bias"], P["blocks.1.mlp.experts.h4toh.bias"][4:8])   # rank 1 of 2
    assert torch.equal(st["blocks.0.attn.qkv.weight"], P["blocks.0.attn.qkv.weight"])
    tc = C.adapt_gates(P, multi_gate=False, num_tasks=5, extra_gate_rows=16)               # task-conditioned gate
    w = tc["blocks.1.mlp.gate.w_gate"]
    assert w.shape == (32 + 16, 8) and torch.equal(w[:32], P["blocks.1.mlp.gate.w_gate"]) and float(w[32:].abs().sum()) == 0
    # a local state is left alone
    st2, fmt2 = C.to_backbone_state({"meta": {"expert_format": "local"}, "model": C.shard_experts(P, 0, 4)}, rank=0,
                                    world_size=2)
    assert fmt2 == "local" and st2["blocks.1.mlp.experts.htoh4.weight"].shape[0] == 4


def test_pos_embed_resize():
    pe = torch.randn(1, 1 + 14 * 14, 8)
    same = C.resize_pos_embed(pe, (14, 14))
    assert torch.allclose(same, pe, atol=1e-6)
    big = C.resize_pos_embed(pe, (30, 40))
    assert big.shape == (1, 1 + 30 * 40, 8) and torch.equal(big[:, 0], pe[:, 0])
    const = torch.ones(1, 1 + 4, 3) * 2.5
    assert torch.allclose(C.resize_pos_embed(const, (7, 5)), torch.full((1, 36, 3), 2.5))
    with pytest.raises(ValueError):
        C.resize_pos_embed(torch.randn(1, 1 + 12, 4), (4, 4))


def test_formats_match_the_reference_fixture(golden_dir, tmp_path):
    """Pinned to the reference: tests/golden/g7_checkpoint_formats.npz holds the outputs of
    pretrain/utils/moe_checkpoint.py (to_mtl_backbone_state_dict, get_first_expert_dim0, infer_expert_format over 168
    argument combinations, build_mtl_meta, merge_moe_sharded_directory) made by tests/gen_golden.py::g7 in the build
    container, together with the inputs; the same calls through m3vit_amd.checkpoint must give the same results."""
    import json
    from collections import OrderedDict

    import numpy as np
    z = np.load(os.path.join(golden_dir, "g7_checkpoint_formats.npz"))
    tab = json.loads(str(z["table"]))
    wrapped = OrderedDict((k[3:], torch.from_numpy(z[k])) for k in z.files if k.startswith("in/"))
    out, dropped = C.to_mtl_backbone_state_dict(wrapped)
    assert list(out.keys()) == tab["backbone_keys"] and dropped == tab["dropped"]
    E, W = tab["E"], tab["W"]
    glob = out
    loc = OrderedDict((k, (v[2:4] if C.is_expert_key(k) else v)) for k, v in glob.items())
    dense = OrderedDict((k, v) for k, v in glob.items() if not C.is_expert_key(k))
    states = {"global": glob, "local": loc, "dense": dense}
    assert {k: C.first_expert_dim0(v) for k, v in states.items()} == tab["first_dim0"]
    assert len(tab["infer"]) == 168
    for c in tab["infer"]:
        got = C.infer_expert_format(c["checkpoint"], states[c["state"]], expected_global_experts=c["expected_global_experts"],
                                    expected_world_size=c["expected_world_size"])
        assert got == c["out"], c
    for c in tab["meta"]:
        assert C.build_mtl_meta(states[c["state"]], "gen_golden", **c["kwargs"]) == c["out"], c
    # the shard directory the generator wrote (rank 0: everything with its experts; ranks > 0: expert tensors only)
    d = str(tmp_path / "shards")
    for r in range(W):
        C.save_rank_shard({"state_dict": C.shard_experts(glob, r, E // W), "epoch": 7, "rank": r}, d, r)
    base, merged, n = C.merge_rank_shards(d)
    assert n == tab["n_shards"] and base["epoch"] == tab["base_epoch"] and list(merged.keys()) == tab["merged_keys"]
    for k in merged:
        assert torch.equal(merged[k], torch.from_numpy(z["merged/" + k])), k


def test_a_local_shard_without_meta_is_not_sliced_again():
    cfg, P = _state()
    loc = C.shard_experts(P, 1, 2)                                   # 2 of 8 experts, no meta, no args
    with pytest.raises(ValueError, match="global or rank-local"):
        C.to_backbone_state({"state_dict": loc}, rank=1, world_size=2)
    st, fmt = C.to_backbone_state({"state_dict": loc}, rank=1, world_size=4, expected_global_experts=8)
    assert fmt == "local" and st["blocks.1.mlp.experts.htoh4.weight"].shape[0] == 2
    st, fmt = C.to_backbone_state({"state_dict": P}, rank=1, world_size=4, expected_global_experts=8)
    assert fmt == "global" and st["blocks.1.mlp.experts.htoh4.weight"].shape[0] == 2
    with pytest.raises(ValueError, match="do not divide"):
        C.to_backbone_state({"state_dict": P, "meta": {"expert_format": "global"}}, rank=0, world_size=3)


def test_upcycling_matches_the_reference_fixture(golden_dir):
    """g11: utils/helpers.py:481-713 `_inject_moe_expert_from_deit_mlp` run on tiny dense state dicts (split upcycling at
    ratio 1 with / without the GELU weight scaling, expert parallel local counts, replicate at ratio 4, truncate mode,
    deit_warm_start, granularity 2) - bit-identical expert tensors; :714-753 `_auto_virtual_group_size` over 480 argument
    combinations; :756-867 `_inject_virtual_group_init_for_gates` under the same seed - bit-identical w_gate tensors."""
    import json
    from collections import OrderedDict

    import numpy as np
    g = np.load(os.path.join(golden_dir, "g11_upcycling.npz"))
    tab = json.loads(str(g["table"]))
    assert len(tab["cases"]) == 6
    for c in tab["cases"]:
        sd = OrderedDict((k.split("/in/")[1], torch.tensor(g[k])) for k in g.files if k.startswith(c["name"] + "/in/"))
        dense_before = {k: v.clone() for k, v in sd.items()}
        out = C.upcycle_dense_mlp_to_experts(sd, c["moe"], c["e_local"], c["eh"], total_experts=c["e_local"] * c["world"],
                                             top_k=c["top_k"], moe_mlp_ratio=c["ratio"],
                                             use_weight_scaling=bool(c["cfg"].get("use_weight_scaling")), mode=c["mode"])
        assert out is sd
        want = {k.split("/out/")[1]: g[k] for k in g.files if k.startswith(c["name"] + "/out/")}
        assert set(want) == {k for k in sd if C.is_expert_key(k)}, c["name"]
        for k, v in want.items():
            assert torch.equal(sd[k], torch.tensor(v)), (c["name"], k)
        assert all(torch.equal(sd[k], v) for k, v in dense_before.items())          # the dense tensors stay, untouched
    for s in tab["group_sizes"]:
        got = C.auto_virtual_group_size(s["tot"], local_experts=s["local_experts"], world_size=s["world_size"],
                                        dense_hidden=s["dense_hidden"], expert_hidden=s["expert_hidden"])
        assert got == s["out"], s
    for v in tab["gate_init"]:
        sd = OrderedDict((f"blocks.{i}.mlp.fc1.weight", torch.zeros(v["Hd"], v["D"])) for i in range(v["depth"]))
        keys = [k.split("/out/")[1] for k in g.files if k.startswith(v["name"] + "/out/")]
        shapes = OrderedDict((k, (v["D"], v["e_local"] * v["world"])) for k in keys)
        torch.manual_seed(v["seed"])
        C.virtual_group_gate_init(sd, shapes, local_experts=v["e_local"], world_size=v["world"], expert_hidden=v["expert_hidden"])
        for k in keys:
            w = sd[k]
            assert torch.equal(w, torch.tensor(g[v["name"] + "/out/" + k])), k
            G = C.auto_virtual_group_size(w.shape[1], local_experts=v["e_local"], world_size=v["world"], dense_hidden=v["Hd"],
                                          expert_hidden=v["expert_hidden"])
            assert all(torch.equal(w[:, :G], w[:, j:j + G]) for j in range(0, w.shape[1], G))


def test_upcycled_experts_reproduce_the_dense_mlp():
    """what upcycling is for: with every gate score 1 / G over one whole template group, the split experts' outputs add up
    to the dense MLP's (fc2's bias is repeated per expert, so it appears G times: the reference's 'official-style' choice)"""
    torch.manual_seed(0)
    D, Hd, G = 8, 32, 4
    sd = {"blocks.1.mlp.fc1.weight": torch.randn(Hd, D), "blocks.1.mlp.fc1.bias": torch.randn(Hd),
          "blocks.1.mlp.fc2.weight": torch.randn(D, Hd), "blocks.1.mlp.fc2.bias": torch.randn(D)}
    C.upcycle_dense_mlp_to_experts(sd, [1], 8, Hd // G, total_experts=8, moe_mlp_ratio=1.0)
    x = torch.randn(5, D)
    gelu = torch.nn.functional.gelu
    dense = gelu(x @ sd["blocks.1.mlp.fc1.weight"].t() + sd["blocks.1.mlp.fc1.bias"]) @ sd["blocks.1.mlp.fc2.weight"].t()
    w1, b1 = sd["blocks.1.mlp.experts.htoh4.weight"], sd["blocks.1.mlp.experts.htoh4.bias"]
    w2, b2 = sd["blocks.1.mlp.experts.h4toh.weight"], sd["blocks.1.mlp.experts.h4toh.bias"]
    parts = sum(gelu(x @ w1[e].t() + b1[e]) @ w2[e].t() for e in range(G))
    assert torch.allclose(parts, dense, atol=1e-5)
    assert torch.equal(w1[:G], w1[G:]) and torch.equal(b2[0], sd["blocks.1.mlp.fc2.bias"])
    with pytest.raises(ValueError):
        C.upcycle_dense_mlp_to_experts(dict(sd), [1], 8, 8, total_experts=6, moe_mlp_ratio=1.0)      # 6 experts, granularity 4
    with pytest.raises(ValueError):
        C.upcycle_dense_mlp_to_experts(dict(sd), [1], 4, 16, total_experts=4, mode="deit_warm_start")  # granularity 2
