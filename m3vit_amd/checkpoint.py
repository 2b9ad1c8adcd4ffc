"""Checkpoint formats of the expert-parallel path, so that state_dicts written by the
reference's trainers load here and vice versa.  Pure host-side dict / file logic on CPU tensors.

Formats (reference file:line):
  * rank-shard directory `<dir>/{rank}.pth` of train_fastmoe.py (utils/moe_utils.py:164-175): rank 0 stores the
    whole state (its E/W experts + every shared tensor), every other rank only its expert tensors
    (`filter_state` :128-134).  Expert tensors are `...mlp.experts.htoh4.*` / `...mlp.experts.h4toh.*`, experts
    along dim 0 (utils/helpers.py:645-662).
  * single file with GLOBAL experts (all E along dim 0) plus `meta = {expert_format, moe_experts_global,
    moe_experts_local, world_size, source}` (pretrain/utils/moe_checkpoint.py:81-112); `expert_format` may also
    have to be inferred from shapes (:131-171).
  * loading a global state on rank r of W keeps experts [r*E/W, (r+1)*E/W) (`read_specific_group_experts`,
    utils/moe_utils.py:191-198).
  * gate adaptation when the checkpoint and the model differ in gate layout (utils/common_config.py:47-68):
    a single `mlp.gate.w_gate` is copied to every `mlp.gate.{t}.w_gate` of a multi-gate model; for task-one-hot /
    task-conditioned gates zero rows are appended to w_gate for the extra gate-input dimensions.
  * position-embedding resize for a different token grid (utils/common_config.py:71-92).
  * DeiT dense-MLP -> experts upcycling and the virtual-group gate initialisation of the pretrained-weight loader
    (utils/helpers.py:481-713, :714-867): `upcycle_dense_mlp_to_experts`, `auto_virtual_group_size`,
    `virtual_group_gate_init` (pinned to the reference functions' outputs by tests/golden/g11_upcycling.npz).
"""
from collections import OrderedDict
import os
import re
from typing import Dict, Optional, Tuple

import torch

_EXPERT_MARKS = ("mlp.experts.htoh4", "mlp.experts.h4toh")
_WRAPPERS = ("module.", "backbone.", "encoder.")


def is_expert_key(key: str) -> bool:
    return any(m in key for m in _EXPERT_MARKS)


def strip_wrapper_prefix(key: str) -> str:
    """drop DDP / wrapper prefixes ('module.', 'backbone.', 'encoder.') however they are stacked"""
    changed = True
    while changed:
        changed = False
        for w in _WRAPPERS:
            if key.startswith(w):
                key = key[len(w):]
                changed = True
    return key


def first_expert_dim0(state: Dict[str, torch.Tensor]) -> Optional[int]:
    for k, v in state.items():
        if torch.is_tensor(v) and is_expert_key(k) and v.dim() >= 1:
            return int(v.shape[0])
    return None


def unwrap(checkpoint, model_key: Optional[str] = None):
    """(checkpoint dict, its model state_dict): under `model_key`, 'state_dict', 'model', or the dict itself"""
    if not isinstance(checkpoint, dict):
        raise ValueError(f"checkpoint must be a dict, got {type(checkpoint)}")
    for k in ([model_key] if model_key else []) + ["state_dict", "model"]:
        if k in checkpoint and isinstance(checkpoint[k], dict):
            return checkpoint, checkpoint[k]
    return checkpoint, checkpoint


# ------------------------------------------------------------------ expert sharding
def shard_experts(state, rank: int, num_local: int):
    """global state -> what rank `rank` holds: experts [rank*num_local, (rank+1)*num_local), the rest as is"""
    out = OrderedDict()
    for k, v in state.items():
        out[k] = v[rank * num_local:(rank + 1) * num_local] if (torch.is_tensor(v) and is_expert_key(k)) else v
    return out


def expert_only(state):
    return OrderedDict((k, v) for k, v in state.items() if is_expert_key(k))


def save_rank_shard(checkpoint: dict, dirname: str, rank: int, state_key: str = "state_dict") -> str:
    """train_fastmoe-style shard: `<dirname>/<rank>.pth`; ranks > 0 keep only their expert tensors.  The caller
    synchronises the ranks (rank 0 creates the directory first)."""
    os.makedirs(dirname, exist_ok=True)
    ck = dict(checkpoint)
    if rank != 0:
        ck[state_key] = expert_only(ck[state_key])
    path = os.path.join(dirname, f"{rank}.pth")
    torch.save(ck, path)
    return path


def merge_rank_shards(dirname: str) -> Tuple[dict, "OrderedDict[str, torch.Tensor]", int]:
    """(rank-0 checkpoint, merged GLOBAL state, number of shards): expert tensors concatenated in rank order"""
    ranks = sorted(int(m.group(1)) for m in (re.fullmatch(r"(\d+)\.pth", n) for n in os.listdir(dirname)) if m)
    if not ranks or ranks[0] != 0:
        raise ValueError(f"{dirname}: expected rank shards 0.pth, 1.pth, ...")
    if ranks != list(range(len(ranks))):
        raise ValueError(f"{dirname}: rank shards are not contiguous: {ranks}")
    base, state0 = unwrap(torch.load(os.path.join(dirname, "0.pth"), map_location="cpu"))
    merged = OrderedDict(state0)
    for r in ranks[1:]:
        _, st = unwrap(torch.load(os.path.join(dirname, f"{r}.pth"), map_location="cpu"))
        for k, v in st.items():
            if is_expert_key(k):
                merged[k] = torch.cat([merged[k], v], dim=0) if k in merged else v
            elif k not in merged:
                merged[k] = v
    return base, merged, len(ranks)


def build_mtl_meta(state, source, world_size: Optional[int] = None, moe_experts_global: Optional[int] = None,
                   moe_experts_local: Optional[int] = None) -> dict:
    """`meta` of a global single-file checkpoint, pretrain/utils/moe_checkpoint.py:81-112 (same argument names and
    defaults; pinned to the reference's outputs by tests/golden/g7_checkpoint_formats.npz): the global expert count
    defaults to the first expert tensor's dim 0, the local one to dim0 // world_size when that divides, else dim0."""
    if world_size is None:
        import torch.distributed as dist
        world_size = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    d0 = first_expert_dim0(state)
    if moe_experts_global is None and d0 is not None:
        moe_experts_global = int(d0)
    if moe_experts_local is None:
        if d0 is None:
            moe_experts_local = 0
        elif world_size > 0 and d0 % world_size == 0:
            moe_experts_local = int(d0 // world_size)
        else:
            moe_experts_local = int(d0)
    return {"expert_format": "global", "moe_experts_global": int(moe_experts_global) if moe_experts_global is not None else 0,
            "moe_experts_local": int(moe_experts_local), "world_size": int(world_size), "source": str(source)}


def build_meta(state, source: str, world_size: int = 1, moe_experts_global: Optional[int] = None) -> dict:
    return build_mtl_meta(state, source, world_size=world_size, moe_experts_global=moe_experts_global)


def to_mtl_backbone_state_dict(state):
    """model / wrapper state_dict -> the backbone-only key space of the MTL loader
    (pretrain/utils/moe_checkpoint.py:24-50): ONE leading `module.` stripped, then `encoder.` stripped, the wrapper's own
    top-level `head.*` / `norm.*` dropped.  Returns (state, dropped keys)."""
    out, dropped = OrderedDict(), []
    for key, value in state.items():
        if key.startswith("module."):
            key = key[len("module."):]
        if key.startswith("encoder."):
            out[key[len("encoder."):]] = value
        elif key.startswith("head.") or key.startswith("norm."):
            dropped.append(key)
        else:
            out[key] = value
    return out, dropped


def infer_expert_format(checkpoint, state, expected_global_experts: Optional[int] = None,
                        expected_world_size: Optional[int] = None) -> str:
    """'global' | 'local' | 'dense' | 'unknown' (pretrain/utils/moe_checkpoint.py:131-171): an explicit
    meta.expert_format wins; else the first expert tensor's dim 0 against the expected global expert count.
    checkpoint['args'] is consulted ONLY when expected_global_experts is not given (and its world_size only when
    expected_world_size is not given either), exactly as the reference does."""
    if isinstance(checkpoint, dict):
        meta = checkpoint.get("meta", {})
        if isinstance(meta, dict) and meta.get("expert_format", None) in ("global", "local"):
            return meta["expert_format"]
    d0 = first_expert_dim0(state)
    if d0 is None:
        return "dense"
    if expected_global_experts is None and isinstance(checkpoint, dict):
        args = checkpoint.get("args", {})
        if isinstance(args, dict):
            expected_global_experts = args.get("moe_experts", None)
            if expected_world_size is None:
                expected_world_size = args.get("world_size", None)
    if expected_global_experts is not None:
        if d0 == int(expected_global_experts):
            return "global"
        if expected_world_size is not None and int(expected_world_size) > 1 and \
                d0 * int(expected_world_size) == int(expected_global_experts):
            return "local"
    return "unknown"


# -------------------------------------------------------------------- key adaptation
def adapt_gates(state, multi_gate: bool, num_tasks: int, extra_gate_rows: int = 0):
    """single-gate checkpoint -> the model's gate layout: `extra_gate_rows` zero rows appended to every
    `mlp.gate.w_gate` (task one-hot: num_tasks rows; task-conditioned: gate_task_specific_dim rows), or one copy
    per task under `mlp.gate.{t}.w_gate` for a multi-gate model."""
    out = OrderedDict()
    for k, v in state.items():
        if k.endswith("mlp.gate.w_gate"):
            if multi_gate:
                for t in range(num_tasks):
                    out[k[:-len("w_gate")] + f"{t}.w_gate"] = v.clone()
                continue
            if extra_gate_rows > 0:
                v = torch.cat((v, torch.zeros(extra_gate_rows, v.shape[-1], dtype=v.dtype)), dim=0)
        out[k] = v
    return out


def resize_pos_embed(pos_embed: torch.Tensor, grid_hw: Tuple[int, int], align_corners: bool = False) -> torch.Tensor:
    """[1, 1 + h0*w0, C] -> [1, 1 + h*w, C]: cls row kept, the patch grid resized bilinearly"""
    n, tokens, c = pos_embed.shape
    side = int(round((tokens - 1) ** 0.5))
    if side * side != tokens - 1:
        raise ValueError(f"pos_embed with {tokens - 1} patch positions is not a square grid")
    grid = pos_embed[:, 1:].transpose(1, 2).reshape(n, c, side, side)
    grid = torch.nn.functional.interpolate(grid, size=tuple(grid_hw), mode="bilinear", align_corners=align_corners)
    return torch.cat((pos_embed[:, :1], grid.reshape(n, c, -1).transpose(1, 2)), dim=1)


def to_backbone_state(checkpoint, rank: int = 0, world_size: int = 1, expected_global_experts: Optional[int] = None,
                      multi_gate: bool = False, num_tasks: int = 0, extra_gate_rows: int = 0,
                      grid_hw: Optional[Tuple[int, int]] = None, model_key: Optional[str] = None):
    """Everything above in the order the reference applies it (utils/common_config.py:31-100): unwrap, strip
    prefixes, adapt gates, resize pos_embed, then keep this rank's experts when the state is global."""
    ck, state = unwrap(checkpoint, model_key)
    state = OrderedDict((strip_wrapper_prefix(k), v) for k, v in state.items() if torch.is_tensor(v))
    fmt = infer_expert_format(ck, state, expected_global_experts, world_size)
    state = adapt_gates(state, multi_gate, num_tasks, extra_gate_rows)
    if grid_hw is not None and "pos_embed" in state and state["pos_embed"].shape[1] != 1 + grid_hw[0] * grid_hw[1]:
        state["pos_embed"] = resize_pos_embed(state["pos_embed"], grid_hw)
    if world_size > 1 and fmt != "local":
        d0 = first_expert_dim0(state)
        if d0 is not None:
            # shard a state that is KNOWN to be global: meta / shapes say so, or its expert count is the expected one.
            # Anything else (e.g. a rank-local shard without meta or args) must not be sliced a second time.
            known_global = fmt == "global" or (fmt == "unknown" and expected_global_experts is not None
                                               and d0 == int(expected_global_experts))
            if not known_global:
                raise ValueError(f"cannot tell whether the checkpoint's {d0} experts per tensor are global or rank-local "
                                 f"(format '{fmt}'): pass expected_global_experts or write meta.expert_format")
            if d0 % world_size != 0:
                raise ValueError(f"{d0} global experts do not divide over {world_size} ranks")
            state = shard_experts(state, rank, d0 // world_size)
    return state, fmt


# ------------------------------------------------------------- dense checkpoint -> experts (upcycling)
def upcycle_dense_mlp_to_experts(state, moe_blocks, num_local_experts: int, expert_hidden: Optional[int] = None, *,
                                 total_experts: Optional[int] = None, top_k: int = 4, moe_mlp_ratio: float = 1.0,
                                 use_weight_scaling: bool = False, mode: str = "deit_upcycling"):
    """Fill `blocks.{i}.mlp.experts.{htoh4,h4toh}.{weight,bias}` of every block in `moe_blocks` from that block's dense
    `mlp.fc1 / fc2` tensors (a DeiT checkpoint), in place, the way the reference's pretrained-weight loader does
    (utils/helpers.py:481-713 `_inject_moe_expert_from_deit_mlp`; tensor layout :579-662):

      * split upcycling (moe_mlp_ratio == 1, or mode == "deit_warm_start"): the dense hidden dimension is cut into
        G = dense_hidden / expert_hidden chunks (G = 4 when expert_hidden is unknown) - fc1 rows, fc1 bias, fc2 columns -
        giving a template group of G experts; fc2's bias is repeated whole.  With `use_weight_scaling` fc1.weight, fc1.bias
        and fc2.weight (NOT fc2.bias) are multiplied by sqrt((total_experts / G) * G^2 / top_k) first (:617-634, the GELU
        form).  `num_local_experts` a multiple of G: the template is repeated; otherwise the first `num_local_experts`
        template experts are taken (:655-670).  total_experts must be a multiple of G; deit_warm_start insists on G == 4.
      * otherwise (moe_mlp_ratio == 4): fc1 / fc2 are copied to every local expert (:672-681).
    Blocks without dense MLP keys are skipped (:528-535).  The dense keys stay in the dict, as in the reference (the model's
    non-strict load ignores them).  Returns `state`."""
    mode = str(mode).lower()
    if mode not in ("deit_upcycling", "deit_warm_start"):
        raise ValueError(f"upcycling mode '{mode}': expected deit_upcycling or deit_warm_start")
    forced = mode == "deit_warm_start"
    E = int(num_local_experts)
    tot = int(total_experts) if total_experts else E
    for i in moe_blocks:
        kf1w, kf1b, kf2w, kf2b = (f"blocks.{i}.mlp.{n}" for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"))
        if kf1w not in state or kf2w not in state:
            continue
        f1w, f1b, f2w, f2b = state[kf1w], state[kf1b], state[kf2w], state[kf2b]
        if forced or float(moe_mlp_ratio) == 1.0:
            hidden = f1w.shape[0]
            G = hidden // int(expert_hidden) if expert_hidden else 4
            if G <= 0 or hidden % G != 0:
                raise ValueError(f"block {i}: granularity {G} does not divide the dense hidden dimension {hidden}")
            if tot % G != 0:
                raise ValueError(f"block {i}: total_experts = {tot} must be a multiple of the granularity {G}")
            if forced and G != 4:
                raise ValueError(f"deit_warm_start needs dense_hidden / expert_hidden == 4, got {G}")
            scale = ((tot // G) * G * G / float(max(int(top_k), 1))) ** 0.5 if use_weight_scaling else 1.0
            t1w = torch.stack((f1w * scale).chunk(G, dim=0), dim=0)            # [G, hidden / G, D]
            t1b = torch.stack((f1b * scale).chunk(G, dim=0), dim=0)            # [G, hidden / G]
            t2w = torch.stack((f2w * scale).chunk(G, dim=1), dim=0)            # [G, D, hidden / G]
            if E % G == 0:
                r = E // G
                e1w, e1b, e2w = t1w.repeat(r, 1, 1), t1b.repeat(r, 1), t2w.repeat(r, 1, 1)
            else:
                e1w, e1b, e2w = t1w[:E], t1b[:E], t2w[:E]
            e2b = f2b.unsqueeze(0).repeat(E, 1)
        else:
            e1w, e1b = f1w.unsqueeze(0).repeat(E, 1, 1), f1b.unsqueeze(0).repeat(E, 1)
            e2w, e2b = f2w.unsqueeze(0).repeat(E, 1, 1), f2b.unsqueeze(0).repeat(E, 1)
        state[f"blocks.{i}.mlp.experts.htoh4.weight"] = e1w.contiguous()
        state[f"blocks.{i}.mlp.experts.htoh4.bias"] = e1b.contiguous()
        state[f"blocks.{i}.mlp.experts.h4toh.weight"] = e2w.contiguous()
        state[f"blocks.{i}.mlp.experts.h4toh.bias"] = e2b.contiguous()
    return state


def auto_virtual_group_size(tot_experts, *, local_experts=None, world_size=None, dense_hidden=None, expert_hidden=None) -> int:
    """How many experts form one virtual group of the gate initialisation (utils/helpers.py:714-753): the upcycling
    granularity dense_hidden / expert_hidden when that divides, else the local expert count, else tot / world_size, else 1;
    then reduced to a common divisor of the local and the total expert count."""
    import math
    tot = int(tot_experts)
    if tot <= 0:
        return 1
    g = None
    if dense_hidden is not None and expert_hidden is not None and expert_hidden > 0 and dense_hidden % expert_hidden == 0:
        g = int(dense_hidden // expert_hidden)
    if g is None or g <= 0:
        if local_experts is not None and int(local_experts) > 0:
            g = int(local_experts)
        elif world_size is not None and int(world_size) > 0 and tot % int(world_size) == 0:
            g = tot // int(world_size)
        else:
            g = 1
    if local_experts is not None and int(local_experts) > 0:
        g = math.gcd(g, int(local_experts))
    g = math.gcd(g, tot)
    return g if (g > 0 and tot % g == 0) else 1


_GATE_KEY = re.compile(r"^blocks\.(\d+)\.(?:mlp\.)?(?:gate(?:\.\d+)?|shared_gate)\.w_gate$")


def virtual_group_gate_init(state, gate_shapes, *, local_experts=None, world_size: int = 1, expert_hidden=None,
                            std: float = 0.02, generator: Optional[torch.Generator] = None):
    """Virtual-group initialisation of every gate matrix (utils/helpers.py:756-867, used with deit_upcycling when
    `use_virtual_group_initialization` is set, :453-458): w_gate [d_gate, E_tot] ~ N(0, std), then the first group of
    `auto_virtual_group_size` expert columns is repeated over all groups - experts that start as copies of the same dense
    MLP chunk start with the same routing logits.  `gate_shapes`: ordered {key: (d_gate, E_tot)} (or tensors) of the MODEL's
    gate parameters, in state_dict order - the normal draws are consumed key by key, so the same seed gives the reference's
    tensors; `local_experts` / `expert_hidden`: ints, or {block index: int}.  Writes `state[key]`, returns `state`."""
    def per_block(v, b):
        return v.get(b) if isinstance(v, dict) else v
    keys = [k for k in gate_shapes if _GATE_KEY.search(k)]
    if not keys:
        raise KeyError("no gate w_gate keys (blocks.{i}[.mlp].gate[.{t}].w_gate / blocks.{i}.shared_gate.w_gate) given")
    for k in keys:
        shp = gate_shapes[k]
        d_gate, tot = (int(shp.shape[0]), int(shp.shape[1])) if torch.is_tensor(shp) else (int(shp[0]), int(shp[1]))
        b = int(_GATE_KEY.search(k).group(1))
        kf1 = f"blocks.{b}.mlp.fc1.weight"
        dense_hidden = int(state[kf1].shape[0]) if kf1 in state else None
        ws = max(1, int(world_size or 1))
        G = auto_virtual_group_size(tot, local_experts=per_block(local_experts, b), world_size=ws, dense_hidden=dense_hidden,
                                    expert_hidden=per_block(expert_hidden, b))
        w = torch.empty(d_gate, tot)
        torch.nn.init.normal_(w, mean=0.0, std=std, generator=generator)
        if G > 1:
            w = torch.cat([w[:, :G]] * (tot // G), dim=1).contiguous()
        state[k] = w
    return state
