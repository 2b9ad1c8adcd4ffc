"""CPU: the `fmoe` API surface the reference imports exists with the signatures / attributes it
relies on, and the layer mirrors expose the reference's state_dict keys."""
import inspect

import torch


def test_reference_import_lines_work():
    import m3vit_amd
    m3vit_amd.install_fmoe_shim()
    # models/moe/ckpt/custom_moe_layer.py:6-16, noisy_gate_vmoe.py:4, train_fastmoe.py:35,460
    from fmoe.layers import FMoE, _fmoe_general_global_forward          # noqa: F401
    from fmoe.linear import FMoELinear
    from fmoe.functions import prepare_forward, ensure_comm              # noqa: F401
    from fmoe.functions import MOEScatter, MOEGather                     # noqa: F401
    from fmoe.functions import AllGather, Slice                          # noqa: F401
    from fmoe.gates import NaiveGate                                     # noqa: F401
    from fmoe.gates.base_gate import BaseGate
    import fmoe
    assert hasattr(fmoe, "DistributedGroupedDataParallel")
    lin = FMoELinear(4, 8, 16, bias=True, rank=0)
    assert tuple(lin.weight.shape) == (4, 16, 8) and tuple(lin.bias.shape) == (4, 16)
    g = BaseGate(4, 2)
    assert g.tot_expert == 8 and g.loss is None and not g.has_loss
    g.set_loss(torch.tensor(1.0)); assert g.has_loss and float(g.get_loss()) == 1.0 and not g.has_loss
    sig = inspect.signature(_fmoe_general_global_forward)
    assert list(sig.parameters)[:5] == ["inp", "gate", "expert_fn", "num_expert", "world_size"]


def test_layer_mirror_signature_and_state_dict_keys():
    from m3vit_amd.gate import NoisyGate_VMoE
    from m3vit_amd.moe_layer import FMoETransformerMLP
    from m3vit_amd.vit import VisionTransformerMoE
    want = ["num_expert", "d_model", "d_gate", "d_hidden", "activation", "expert_dp_comm", "expert_rank", "gate",
            "world_size", "top_k", "vmoe_noisy_std", "gate_return_decoupled_activation", "gate_task_specific_dim",
            "multi_gate", "regu_experts_fromtask", "num_experts_pertask", "num_tasks", "regu_sem", "sem_force",
            "regu_subimage", "expert_prune", "prune_threshold"]
    got = list(inspect.signature(FMoETransformerMLP.__init__).parameters)[1:]
    assert got[:len(want)] == want                      # custom_moe_layer.py:73-98
    fwd = list(inspect.signature(FMoETransformerMLP.forward).parameters)[1:]
    assert fwd == ["inp", "gate_inp", "task_id", "task_specific_feature", "sem"]    # :161
    layer = FMoETransformerMLP(num_expert=4, d_model=32, d_gate=34, d_hidden=48, gate=NoisyGate_VMoE, top_k=2,
                               multi_gate=True, activation=torch.nn.Sequential(torch.nn.GELU(), torch.nn.Dropout(0.)))
    keys = set(layer.state_dict().keys())
    assert keys == {"experts.htoh4.weight", "experts.htoh4.bias", "experts.h4toh.weight", "experts.h4toh.bias",
                    "gate.0.w_gate", "gate.1.w_gate"}
    assert tuple(layer.experts.htoh4.weight.shape) == (4, 48, 32)       # utils/helpers.py:645-662
    assert all(getattr(p, "dp_comm") == "none" for p in layer.experts.parameters())   # :159
    assert layer.d_model == 32 and layer.top_k == 2 and layer.world_size == 1 and layer.num_expert == 4
    # multi-gate list length = d_gate - d_model (:143-150)
    assert len(layer.gate) == 2
    # task-conditioned: one gate with input dim d_model + gtsd (:127-130)
    tc = FMoETransformerMLP(num_expert=4, d_model=32, d_gate=37, d_hidden=32, gate=NoisyGate_VMoE, top_k=2,
                            gate_task_specific_dim=8)
    assert tuple(tc.gate.w_gate.shape) == (40, 4)
    vit = VisionTransformerMoE(img_size=(32, 32), embed_dim=64, depth=2, num_heads=2, moe_mlp_ratio=1, moe_experts=4,
                               moe_top_k=2, gate_dim=66, multi_gate=True)
    from oracle import ref_torch as R
    cfg = R.BackboneCfg(img_size=(32, 32), embed_dim=64, depth=2, num_heads=2, moe_mlp_ratio=1.0, moe_experts=4,
                        moe_top_k=2, gate_dim=66, multi_gate=True)
    assert set(vit.state_dict().keys()) == set(R.init_backbone_params(cfg).keys())
    # utils/moe_utils.py:128-134,191-198 key filters keep working
    assert any("mlp.experts.htoh4" in k for k in vit.state_dict()) and any("mlp.experts.h4toh" in k for k in vit.state_dict())


def test_pre_routed_layer_signature_and_keys():
    """TokenFMoETransformerMLP (models/moe/token/custom_moe_layer.py:55-156): gate routing is done by the Block."""
    from m3vit_amd.moe_layer import TokenFMoETransformerMLP
    want = ["num_expert", "d_model", "d_gate", "d_hidden", "activation", "expert_dp_comm", "expert_rank", "world_size",
            "top_k"]
    assert list(inspect.signature(TokenFMoETransformerMLP.__init__).parameters)[1:1 + len(want)] == want     # :64-75
    assert list(inspect.signature(TokenFMoETransformerMLP.forward).parameters)[1:] == ["inp", "gate_top_k_idx", "gate_score"]
    layer = TokenFMoETransformerMLP(num_expert=4, d_model=32, d_hidden=48, top_k=2)
    keys = {k for k in layer.state_dict().keys() if k.startswith("experts.")}
    assert keys == {"experts.htoh4.weight", "experts.htoh4.bias", "experts.h4toh.weight", "experts.h4toh.bias"}
    assert layer.our_d_model == 32 and layer.num_expert == 4 and layer.top_k == 2
    assert all(getattr(p, "dp_comm") == "none" for p in layer.experts.parameters())


def test_reference_layer_and_gate_construct_against_the_shim():
    """Build container only (skipped where /root/reference is absent, e.g. on the GPU box): the reference's OWN
    models/moe/ckpt/custom_moe_layer.py and noisy_gate_vmoe.py import and construct against install_fmoe_shim() - the
    drop-in boundary of SURVEY.md 8(b) from the reference's side.  `tree` (dm-tree, third party, not installed here)
    is given a two-function stand-in for the import line only; nothing of it runs at construction."""
    import os
    import sys
    import types

    import pytest
    if not os.path.isdir("/root/reference/models/moe/ckpt"):
        pytest.skip("reference checkout not present")
    import m3vit_amd
    m3vit_amd.install_fmoe_shim()
    added = []
    if "tree" not in sys.modules:
        t = types.ModuleType("tree")
        t.map_structure = lambda fn, *s: fn(*s)            # single-tensor structures: all the reference ever passes
        t.flatten = lambda x: [x]
        sys.modules["tree"] = t
        added.append("tree")
    sys.path.insert(0, "/root/reference")
    try:
        from models.moe.ckpt.custom_moe_layer import FMoETransformerMLP as RefLayer
        from models.moe.ckpt.noisy_gate_vmoe import NoisyGate_VMoE as RefGate
        from fmoe.gates.base_gate import BaseGate
        from fmoe.layers import FMoE
        assert issubclass(RefLayer, FMoE) and issubclass(RefGate, BaseGate)            # the shim's classes are the bases
        act = torch.nn.Sequential(torch.nn.GELU(), torch.nn.Dropout(0.))
        layer = RefLayer(num_expert=4, d_model=32, d_gate=34, d_hidden=48, gate=RefGate, top_k=2, multi_gate=True,
                         activation=act, vmoe_noisy_std=0)
        assert set(layer.state_dict().keys()) == {"experts.htoh4.weight", "experts.htoh4.bias", "experts.h4toh.weight",
                                                  "experts.h4toh.bias", "gate.0.w_gate", "gate.1.w_gate"}
        assert tuple(layer.experts.htoh4.weight.shape) == (4, 48, 32) and tuple(layer.experts.h4toh.bias.shape) == (4, 32)
        assert all(getattr(p, "dp_comm") == "none" for p in layer.experts.parameters())      # mark_parallel_comm, :159
        assert all(getattr(p, "dp_comm") == "gate" for g in layer.gate for p in g.parameters())
        assert layer.d_model == 32 and layer.top_k == 2 and layer.world_size == 1 and layer.num_expert == 4
        assert layer.gate_hook is None and layer.mask is None and layer.slice_size == 1 and layer.moe_group is None
        assert callable(layer.expert_fn)
        tc = RefLayer(num_expert=4, d_model=32, d_gate=37, d_hidden=32, gate=RefGate, top_k=2, gate_task_specific_dim=8,
                      activation=act)
        assert tuple(tc.gate.w_gate.shape) == (40, 4) and tc.gate.tot_expert == 4 and tc.gate.loss is None
        # the mirror exposes the same keys / shapes, so a state_dict moves between the two
        from m3vit_amd.gate import NoisyGate_VMoE
        from m3vit_amd.moe_layer import FMoETransformerMLP
        mine = FMoETransformerMLP(num_expert=4, d_model=32, d_gate=34, d_hidden=48, gate=NoisyGate_VMoE, top_k=2,
                                  multi_gate=True, activation=act, vmoe_noisy_std=0)
        mine.load_state_dict(layer.state_dict())
        layer.load_state_dict(mine.state_dict())
    finally:
        sys.path.remove("/root/reference")
        for m in added:
            sys.modules.pop(m, None)
        for m in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            sys.modules.pop(m, None)


def test_origin_convention_signatures():
    """convention="origin" mirrors models/moe/origin/*: same ctor arguments, tensor / 2-tuple returns (checked on the
    GPU in tests/test_modules_gpu.py); here: flags are plumbed from the backbone down to every gate."""
    from m3vit_amd.gate import NoisyGate_VMoE
    from m3vit_amd.vit import VisionTransformerMoE
    vit = VisionTransformerMoE(img_size=(32, 32), embed_dim=64, depth=4, num_heads=2, moe_mlp_ratio=1, moe_experts=4,
                               moe_top_k=2, gate_dim=66, multi_gate=True, convention="origin")
    gates = [m for m in vit.modules() if isinstance(m, NoisyGate_VMoE)]
    assert len(gates) == 4 and all(g.convention == "origin" for g in gates)
    assert all(b.convention == "origin" for b in vit.blocks) and vit.blocks[1].mlp.convention == "origin"
    ck = VisionTransformerMoE(img_size=(32, 32), embed_dim=64, depth=2, num_heads=2, moe_mlp_ratio=1, moe_experts=4,
                              moe_top_k=2, gate_dim=66, multi_gate=True)
    assert set(ck.state_dict().keys()) == {k for k in vit.state_dict().keys() if not k.startswith(("blocks.2", "blocks.3"))}


def test_fused_backbone_eligibility_rules():
    """which calls VisionTransformerMoE hands to the fused executor (m3vit_amd/fused.py) and which go through the per-op
    autograd Functions - decided without touching the GPU"""
    import torch
    from m3vit_amd.fused import FusedBackbone
    from m3vit_amd.vit import VisionTransformerMoE
    kw = dict(img_size=(32, 32), embed_dim=64, depth=2, num_heads=2, moe_mlp_ratio=1, moe_experts=4, moe_top_k=2, gate_dim=66,
              multi_gate=True)
    m = VisionTransformerMoE(**kw)
    assert m.fused == "auto" and m._fused_static_ok
    x = torch.zeros(2, 3, 32, 32)
    assert FusedBackbone.unsupported(m, x, None, 0, None) == "CPU tensor"
    mm = VisionTransformerMoE(**{**kw, "world_size": 2})          # sharded experts: eligible once a process group exists
    assert mm._fused_static_ok and mm._cfg_kwargs["moe_experts"] == 8
    for bad_kw, why in ((dict(qkv_bias=False), "qkv_bias"), (dict(world_size=2, use_checkpointing=True), "expert parallel"),
                        (dict(expert_prune=True), "routing edits"), (dict(gate_input_ahead=True), "routing edits"),
                        (dict(num_heads=4), "head dim")):
        mm = VisionTransformerMoE(**{**kw, **bad_kw})
        assert not mm._fused_static_ok and why in mm._fused_static_why, (bad_kw, mm._fused_static_why)

    class FakeCuda(torch.Tensor):                     # (is_cuda is all `unsupported` looks at before the shape)
        is_cuda = True
    xc = torch.zeros(2, 3, 32, 32).as_subclass(FakeCuda)
    assert FusedBackbone.unsupported(m, xc, None, 0, None) is None
    assert FusedBackbone.unsupported(m, xc, xc, 0, None) == "caller-supplied gate input"
    assert FusedBackbone.unsupported(m, xc, None, None, None) == "multi-gate model called without a task id"
    assert "image size" in FusedBackbone.unsupported(m, torch.zeros(2, 3, 64, 32).as_subclass(FakeCuda), None, 0, None)
    m.eval()
    assert FusedBackbone.unsupported(m, xc, None, 0, None) == "eval mode with autograd on"
    with torch.no_grad():
        assert FusedBackbone.unsupported(m, xc, None, 0, None) is None
    m.train()
    m.blocks[1].mlp.gate_hook = lambda *a: None
    assert "gate hook" in FusedBackbone.unsupported(m, xc, None, 0, None)
    import os
    os.environ["M3VIT_FUSED"] = "0"
    try:
        assert VisionTransformerMoE(**kw).fused is False
    finally:
        del os.environ["M3VIT_FUSED"]
