"""ctypes loader for libm3vit_hip.so (the C ABI declared in include/m3vit_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails the
error is raised, never papered over by a torch/CPU path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M3VIT_LIB") or os.path.join(_HERE, "libm3vit_hip.so")   # M3VIT_LIB: diagnostic builds

M3_F32, M3_F16, M3_BF16 = 0, 1, 2
M3_ACT_NONE, M3_ACT_GELU = 0, 1


class M3Error(RuntimeError):
    pass


class GemmArgs(Structure):
    _fields_ = [
        ("A", c_void_p), ("lda", c_int64),
        ("a_row_idx", c_void_p), ("a_row_div", c_int32),
        ("B", c_void_p), ("ldb", c_int64),
        ("C", c_void_p), ("ldc", c_int64), ("c_dtype", c_int32),
        ("c_row_idx", c_void_p),
        ("bias", c_void_p),
        ("pre_out", c_void_p), ("ld_pre", c_int64),
        ("gelu_grad_pre", c_void_p), ("ld_gpre", c_int64),
        ("residual", c_void_p), ("ld_res", c_int64),
        ("act", c_int32),
        ("M", c_int64), ("N", c_int32), ("K", c_int32),
        ("G", c_int32),
        ("group_offsets", c_void_p),
        ("tile_starts", c_void_p),
        ("dtype", c_int32),
        ("row_scale", c_void_p), ("row_scale_div", c_int32),
        ("row_scale_idx", c_void_p),
    ]


class WgradReduceDesc(Structure):
    _fields_ = [
        ("ws", c_void_p), ("splits", c_int32), ("elems", c_int64),
        ("group_offsets", c_void_p), ("G", c_int32), ("chunk_rows", c_int32),
        ("dW", c_void_p), ("beta", c_int32),
        ("bias_ws", c_void_p), ("bias_elems", c_int64), ("db", c_void_p), ("beta_db", c_int32),
    ]


class WgradArgs(Structure):
    _fields_ = [
        ("dC", c_void_p), ("lddc", c_int64), ("c_row_idx", c_void_p),
        ("A", c_void_p), ("lda", c_int64), ("a_row_idx", c_void_p), ("a_row_div", c_int32),
        ("M", c_int64), ("N", c_int32), ("K", c_int32), ("G", c_int32),
        ("group_offsets", c_void_p),
        ("splits", c_int32),
        ("ws", c_void_p),
        ("dtype", c_int32),
        ("bias_ws", c_void_p),
        ("chunk_rows", c_int32), ("units", c_int32),
        ("c_row_div", c_int32), ("c_row_scale", c_void_p),
        ("prev", POINTER(WgradReduceDesc)),
        ("direct_dW", c_void_p), ("direct_db", c_void_p), ("direct_beta", c_int32), ("direct_beta_db", c_int32),
    ]


class FfnArgs(Structure):
    _fields_ = [
        ("X", c_void_p), ("ldx", c_int64), ("x_row_idx", c_void_p), ("x_row_div", c_int32),
        ("W1", c_void_p), ("W2p", c_void_p),
        ("b1", c_void_p), ("b2", c_void_p),
        ("Y", c_void_p), ("ldy", c_int64), ("y_dtype", c_int32), ("y_row_idx", c_void_p),
        ("residual", c_void_p), ("ld_res", c_int64),
        ("pre_out", c_void_p), ("act_out", c_void_p),
        ("M", c_int64), ("D", c_int32), ("H", c_int32), ("G", c_int32),
        ("group_offsets", c_void_p),
        ("dtype", c_int32),
    ]


M3_CAST_PERM32, M3_CAST_PERM32_T = 1, 2


class CastDesc(Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("dst_t", c_void_p), ("G", c_int32), ("rows", c_int32),
                ("cols", c_int32), ("tile_start", c_int32), ("flags", c_int32), ("pad1", c_int32)]


class GateFwdArgs(Structure):
    _fields_ = [
        ("x", c_void_p), ("x_dtype", c_int32), ("T", c_int64), ("D", c_int32), ("ldx", c_int64),
        ("w_gate", c_void_p), ("E", c_int32),
        ("logit_bias", c_void_p), ("noise", c_void_p), ("noise_std", c_float), ("k", c_int32),
        ("idx", c_void_p), ("idx32", c_void_p), ("idx_next", c_void_p),
        ("score", c_void_p), ("top_logits", c_void_p),
        ("clean", c_void_p), ("noisy", c_void_p), ("gates", c_void_p),
        ("part_importance", c_void_p), ("part_load", c_void_p), ("part_load_prob", c_void_p), ("part_count", c_void_p),
    ]


class GateBwdArgs(Structure):
    _fields_ = [
        ("noisy", c_void_p), ("clean", c_void_p), ("top_logits", c_void_p),
        ("idx", c_void_p), ("idx_next", c_void_p),
        ("d_score", c_void_p), ("d_top", c_void_p), ("d_importance", c_void_p), ("d_load_prob", c_void_p),
        ("balance_scale", c_float), ("noise_std", c_float),
        ("T", c_int64), ("E", c_int32), ("k", c_int32),
        ("d_logits", c_void_p), ("balance_scale_dev", c_void_p), ("d_logits_act", c_void_p), ("act_dtype", c_int32),
    ]


_V, _I, _L, _F = c_void_p, c_int, c_int64, c_float

# name -> (restype, argtypes); every symbol include/m3vit_hip.h declares
SIGNATURES = {
    "m3_version": (c_int, []),
    "m3_last_error": (c_char_p, []),
    "m3_device_query": (c_int, [c_char_p, _I]),
    "m3_gate_num_blocks": (c_int, [_L]),
    "m3_gate_dw_blocks": (c_int, [_L]),
    "m3_gate_fwd": (c_int, [POINTER(GateFwdArgs), _V]),
    "m3_gate_reduce": (c_int, [_V, _V, _I, _I, _V, _V, _V]),
    "m3_balance_loss": (c_int, [_V, _V, _V, _I, _I, _V, _V, _V, _V, _V, _V, _V, _V]),
    "m3_balance_route": (c_int, [_V, _V, _V, _I, _I, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V]),
    "m3_route_assign": (c_int, [_V, _L, _I, _I, _V, _V, _V, _V, _V]),
    "m3_gate_bwd_logits": (c_int, [POINTER(GateBwdArgs), _V]),
    "m3_gate_bwd_params": (c_int, [_V, _I, _L, _I, _L, _V, _I, _V, _V, _V, _I, _V, _L, _I, _V]),
    "m3_route_ws_elems": (c_int64, [_L, _I]),
    "m3_route_build": (c_int, [_V, _L, _I, _V, _V, _V, _V, _V, _V, _V, _V]),
    "m3_ep_plan": (c_int, [_V, _V, _I, _I, _V, _V, _L, _V, _V, _V]),
    "m3_relu_up2x_fwd": (c_int, [_V, _I, _L, _I, _I, _I, _I, _V, _I, _V]),
    "m3_relu_up2x_bwd": (c_int, [_V, _I, _V, _I, _L, _I, _I, _I, _I, _V, _V]),
    "m3_ep_plan_fixed": (c_int, [_V, _V, _I, _I, _I, _V, _V, _L, _V, _V, _V, _V, _V, _V, _V, _V]),
    "m3_ep_unique_id": (c_int, [_V]),
    "m3_ep_init": (c_int, [_V, _I, _I, POINTER(c_int)]),
    "m3_ep_destroy": (c_int, [_I]),
    "m3_ep_exchange_counts": (c_int, [_I, _V, _V, _I, _V]),
    "m3_ep_dispatch": (c_int, [_I, _V, _V, _V, _V, _L, _V]),
    "m3_ep_return": (c_int, [_I, _V, _V, _V, _V, _L, _V]),
    "m3_gemm_nt": (c_int, [POINTER(GemmArgs), _V]),
    "m3_experimental": (c_int, []),
    "m3_gemm_set_variant": (c_int, [_I]),
    "m3_gemm_set_big": (c_int, [_I]),
    "m3_ffn_fwd": (c_int, [POINTER(FfnArgs), _V]),
    "m3_wgrad_tn": (c_int, [POINTER(WgradArgs), _V]),
    "m3_wgrad_tile": (c_int, [_I, _I, _I, POINTER(c_int), POINTER(c_int)]),
    "m3_wgrad_skinny": (c_int, [_I, _I, _I]),
    "m3_wgrad_set_wide": (c_int, [_I]),
    "m3_wgrad_set_dma": (c_int, [_I]),
    "m3_wgrad_set_big": (c_int, [_I]),
    "m3_wgrad_reduce": (c_int, [_V, _I, _L, _V, _I, _V, _L, _V, _I, _V]),
    "m3_wgrad_reduce_grouped": (c_int, [_V, _V, _I, _I, _L, _V, _I, _V, _L, _V, _I, _V]),
    "m3_wgrad_bias_reduce": (c_int, [_V, _I, _L, _V, _I, _V]),
    "m3_colsum_ws_elems": (c_int64, [_L, _I, _I]),
    "m3_colsum": (c_int, [_V, _I, _L, _V, _L, _I, _I, _V, _V, _V, _I, _V]),
    "m3_combine_fwd": (c_int, [_V, _I, _V, _V, _L, _I, _I, _V, _V]),
    "m3_combine_bwd": (c_int, [_V, _V, _I, _V, _L, _I, _I, _V, _V, _V]),
    "m3_combine_gate_bwd": (c_int, [_V, _I, _L, _I, _I, _V, _V, _I, _V, _I, _V]),
    "m3_gather_rows": (c_int, [_V, _I, _V, _I, _L, _I, _I, _V, _V]),
    "m3_layernorm_fwd": (c_int, [_V, _L, _I, _V, _V, _F, _V, _I, _V, _V, _V]),
    "m3_ln_bwd_blocks": (c_int, [_L, _I]),
    "m3_layernorm_bwd": (c_int, [_V, _I, _V, _V, _V, _V, _V, _L, _I, _V, _V, _V, _V, _I, _V, _I, _V]),
    "m3_layernorm_bwd_reduce": (c_int, [_V, _L, _I, _I, _V, _I, _I, _I, _V]),
    "m3_attention_fwd": (c_int, [_V, _I, _I, _I, _I, _I, _V, _V, _V]),
    "m3_attention_bwd_ws_elems": (c_int64, [_I, _I, _I, _I]),
    "m3_attention_bwd": (c_int, [_V, _V, _V, _V, _I, _I, _I, _I, _I, _V, _V, _V]),
    "m3_cast_matrix": (c_int, [_V, _I, _I, _I, _I, _V, _I, _V]),
    "m3_cast_batch": (c_int, [_V, _I, _I, _I, _V]),
    "m3_add_f32": (c_int, [_V, _V, _L, _V]),
    "m3_cast_f32": (c_int, [_V, _L, _V, _I, _V]),
    "m3_scale_rows_cast": (c_int, [_V, _L, _I, _V, _I, _V, _I, _V]),
    "m3_im2row": (c_int, [_V, _I, _I, _I, _I, _I, _V, _I, _V]),
    "m3_assemble_tokens": (c_int, [_V, _V, _V, _I, _I, _I, _V, _V]),
    "m3_tokens_bwd": (c_int, [_V, _I, _I, _I, _V, _I, _V, _V, _I, _V]),
}

def csrc_sha16() -> str:
    """first 16 hex digits of the sha256 over the kernel sources (csrc/*.hip, *.h and the C header, by name): stamps a
    PMC traffic file with the kernels it was measured on, so that bench.py can tell a stale replayed measurement
    (the GPU box has no .git to ask)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "m3vit_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_lib = None


def lib():
    """Load the library once.  Raises M3Error if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise M3Error(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C m3vit_amd/csrc` (hipcc, --offload-arch=gfx950). There is no CPU fallback.")
        # torch ships its own HIP runtime (libamdhip64): import it FIRST so that this library binds to the
        # same runtime instance (and therefore the same device context / streams) instead of a second copy
        # from /opt/rocm, which would see "no ROCm-capable device" for torch-allocated memory.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)       # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().m3_last_error()
        raise M3Error(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
