"""Tensor-level wrappers over the C ABI (include/m3vit_hip.h).

torch is used here only as the owner of device memory and streams: every function
hands raw device pointers + sizes to libm3vit_hip.so and launches on torch's current
HIP stream.  No function in this module computes anything with torch ops.
"""
from __future__ import annotations

import ctypes
from ctypes import byref, c_void_p
from typing import Optional

import torch

from . import _lib
from ._lib import M3_ACT_GELU, M3_ACT_NONE, M3_BF16, M3_F16, M3_F32, GemmArgs, WgradArgs, WgradReduceDesc, check, lib

_DT = {torch.float32: M3_F32, torch.float16: M3_F16, torch.bfloat16: M3_BF16}      # bf16: every entry point except the opt-in fused m3_ffn_fwd


def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise _lib.M3Error(f"unsupported activation dtype {dtype}; use float32, float16 or bfloat16")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t: torch.Tensor, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise _lib.M3Error(f"{name} must live on the GPU (no CPU path)")
    if dtype is not None and t.dtype != dtype:
        raise _lib.M3Error(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.M3Error(f"{name} must be contiguous")
    return t


# ----------------------------------------------------------------------------- gate
def gate_fwd(x, w_gate, k, logit_bias=None, noise=None, noise_std=0.0, dense=True, want_idx32=True,
             loss_acc=None, route=False, want_counts64=False):
    """x [T,D] f32/f16, w_gate [D,E] f32 (rows beyond D are ignored: pass w_gate[:D] + bias for
    task conditioning).  Returns dict with idx i64 [T,k], idx32, idx_next i32 [T], score, top_logits,
    clean, noisy, gates (dense only), importance f32 [E], load i64 [E], and the block's balance loss
    cv_loss = cv^2(importance) + cv^2(load) (0-dim) with its gradients d_importance / d_load_prob [E].
    For noisy training (noise given, noise_std != 0, k < E; needs dense) load is the Normal-CDF form
    load_prob f32 [E] (vision_transformer_moe.py:456-457), otherwise the count and d_load_prob is None.
    loss_acc: optional 1-element f32 tensor the loss is also added to.
    route: also build the dispatch metadata (result["route"]: a Route as route_build(idx32, E) returns it) from the gate
    kernel's own per-block counts - the histogram pass is not launched and the scan rides in the balance launch
    (m3_balance_route); needs k | 16."""
    _req(x, name="x"); _req(w_gate, torch.float32, "w_gate")
    T, D = x.shape
    E = w_gate.shape[1]
    assert w_gate.shape[0] == D
    dev = x.device
    kp = min(k + 1, E)
    f32 = torch.float32
    idx = torch.empty((T, k), dtype=torch.int64, device=dev)
    idx32 = torch.empty((T, k), dtype=torch.int32, device=dev) if want_idx32 else None
    idx_next = torch.empty((T,), dtype=torch.int32, device=dev) if kp > k else None
    score = torch.empty((T, k), dtype=f32, device=dev)
    top = torch.empty((T, kp), dtype=f32, device=dev)
    clean = torch.empty((T, E), dtype=f32, device=dev) if dense else None
    noisy = torch.empty((T, E), dtype=f32, device=dev) if dense else None
    gates = torch.empty((T, E), dtype=f32, device=dev) if dense else None
    nblk = lib().m3_gate_num_blocks(T)
    # one allocation for the partials and the [E]-sized results
    prob = noise is not None and float(noise_std) != 0.0 and k < E and dense
    pi = torch.empty((max(nblk, 1), E), dtype=f32, device=dev)
    pl = torch.empty((max(nblk, 1), E), dtype=torch.int32, device=dev)
    pp = torch.empty((max(nblk, 1), E), dtype=f32, device=dev) if prob else None
    small = torch.empty((4, E), dtype=f32, device=dev)
    imp, load_prob, d_imp, d_lp = small[0], (small[1] if prob else None), small[2], (small[3] if prob else None)
    load = torch.empty(E, dtype=torch.int64, device=dev)
    loss = torch.empty((), dtype=f32, device=dev)
    if logit_bias is not None:
        _req(logit_bias, f32, "logit_bias")
    if noise is not None:
        _req(noise, f32, "noise")
    route = bool(route) and T > 0 and 16 % k == 0 and want_idx32
    pc = torch.empty((2, max(nblk, 1), E), dtype=torch.int32, device=dev) if route else None     # counts, then their prefix
    a = _lib.GateFwdArgs(_p(x), dt_code(x.dtype), T, D, x.stride(0), _p(w_gate), E, _p(logit_bias), _p(noise),
                         float(noise_std), k, _p(idx), _p(idx32), _p(idx_next), _p(score), _p(top), _p(clean), _p(noisy),
                         _p(gates), _p(pi), _p(pl), _p(pp), _p(pc))
    check(lib().m3_gate_fwd(byref(a), _stream()), "m3_gate_fwd")
    r = None
    if route:
        n = T * k
        r = Route()
        r.n, r.E, r.k = n, E, k
        meta = torch.empty(3 * E + 2, dtype=torch.int32, device=dev)
        r.counts, r.offsets, r.tile_starts = meta[:E], meta[E:2 * E + 1], meta[2 * E + 1:]
        r.pos = torch.empty(n, dtype=torch.int32, device=dev)
        r.row_of_slot = torch.empty(n, dtype=torch.int32, device=dev)
        r.counts64 = torch.empty(E, dtype=torch.int64, device=dev) if want_counts64 else None
        check(lib().m3_balance_route(_p(pi), _p(pl), _p(pp), nblk, E, _p(imp), _p(load), _p(load_prob), _p(loss),
                                     _p(loss_acc), _p(d_imp), _p(d_lp), _p(pc[0]), _p(pc[1]), _p(r.counts), _p(r.offsets),
                                     _p(r.tile_starts), _p(r.counts64), _stream()), "m3_balance_route")
        check(lib().m3_route_assign(_p(idx32), n, E, k, _p(pc[1]), _p(r.offsets), _p(r.pos), _p(r.row_of_slot), _stream()),
              "m3_route_assign")
    else:
        check(lib().m3_balance_loss(_p(pi), _p(pl), _p(pp), nblk, E, _p(imp), _p(load),
                                    _p(load_prob), _p(loss), _p(loss_acc), _p(d_imp), _p(d_lp), _stream()),
              "m3_balance_loss")
    return dict(route=r, idx=idx, idx32=idx32, idx_next=idx_next, score=score, top_logits=top, clean=clean, noisy=noisy,
                gates=gates, importance=imp, load=load, load_prob=load_prob, cv_loss=loss, d_importance=d_imp,
                d_load_prob=d_lp, noise_std=float(noise_std) if prob else 0.0)


def gate_bwd_logits(noisy, idx, d_score, d_importance, k, *, balance_scale=1.0, d_top=None, idx_next=None,
                    d_load_prob=None, clean=None, top_logits=None, noise_std=0.0, out=None, balance_scale_dev=None,
                    out_act=None):
    """d_logits [T,E] from d_score [T,k], d_top [T,k+1], balance_scale * (d_importance, d_load_prob) [E].
    balance_scale_dev: optional 1-element f32 device tensor multiplied onto balance_scale inside the kernel.
    out_act: optional [T,E] tensor (activation dtype) that receives a second copy of the result."""
    T, E = noisy.shape
    dl = torch.empty_like(noisy) if out is None else out
    if balance_scale_dev is not None:
        _req(balance_scale_dev, torch.float32, "balance_scale_dev")
    a = _lib.GateBwdArgs(_p(noisy), _p(clean), _p(top_logits), _p(idx), _p(idx_next), _p(d_score), _p(d_top),
                         _p(d_importance), _p(d_load_prob), float(balance_scale), float(noise_std), T, E, k, _p(dl),
                         _p(balance_scale_dev), _p(out_act), dt_code(out_act.dtype) if out_act is not None else M3_F32)
    check(lib().m3_gate_bwd_logits(byref(a), _stream()), "m3_gate_bwd_logits")
    return dl


def gate_bwd_params(x, w_gate, d_logits, d_w_gate=None, beta_dw=0, dx=None, beta_dx=0, part_dw=None):
    T, D = x.shape
    E = w_gate.shape[1]
    if d_w_gate is not None and part_dw is None:
        part_dw = torch.empty((lib().m3_gate_dw_blocks(T), D, E), dtype=torch.float32, device=x.device)
    check(lib().m3_gate_bwd_params(_p(x), dt_code(x.dtype), T, D, x.stride(0), _p(w_gate), E, _p(d_logits),
                                   _p(part_dw) if d_w_gate is not None else None, _p(d_w_gate), beta_dw,
                                   _p(dx), dx.stride(0) if dx is not None else 0, beta_dx, _stream()),
          "m3_gate_bwd_params")


# ---------------------------------------------------------------------------- route
class Route:
    """Device-resident dispatch metadata (no host sync).  Expert ids outside [0, E) are a caller error: such entries
    get no slot (their token-major output rows are never written) while pos / row_of_slot stay in range; check()
    - a host sync, for tests and debugging - raises when any entry was dropped."""
    __slots__ = ("counts", "offsets", "pos", "row_of_slot", "tile_starts", "counts64", "n", "E", "k")

    def check(self):
        routed = int(self.offsets[-1])
        if routed != self.n:
            raise _lib.M3Error(f"route_build: {self.n - routed} of {self.n} expert ids lie outside [0, {self.E})")
        return self


def route_build(idx32: torch.Tensor, E: int, want_counts64=False) -> Route:
    _req(idx32, torch.int32, "idx32")
    n = idx32.numel()
    dev = idx32.device
    r = Route()
    r.n, r.E, r.k = n, E, idx32.shape[-1] if idx32.dim() > 1 else 1
    r.counts = torch.empty(E, dtype=torch.int32, device=dev)
    r.offsets = torch.empty(E + 1, dtype=torch.int32, device=dev)
    r.pos = torch.empty(n, dtype=torch.int32, device=dev)
    r.row_of_slot = torch.empty(n, dtype=torch.int32, device=dev)
    r.tile_starts = torch.empty(E + 1, dtype=torch.int32, device=dev)
    r.counts64 = torch.empty(E, dtype=torch.int64, device=dev) if want_counts64 else None
    ws = torch.empty(int(lib().m3_route_ws_elems(n, E)), dtype=torch.int32, device=dev)
    if n == 0:                                   # nothing routed: all-zero metadata, no launch
        for t_ in (r.counts, r.offsets, r.tile_starts):
            t_.zero_()
        if r.counts64 is not None:
            r.counts64.zero_()
        return r
    check(lib().m3_route_build(_p(idx32), n, E, _p(r.counts), _p(r.offsets), _p(r.pos), _p(r.row_of_slot),
                               _p(r.tile_starts), _p(r.counts64), _p(ws), _stream()), "m3_route_build")
    return r


class EpPlan:
    """Device-resident expert-parallel exchange plan (m3_ep_plan) + the two split lists the a2a-v API needs."""
    __slots__ = ("splits", "regroup", "offsets", "tile_starts", "in_splits", "out_splits", "n_recv")


def ep_plan(send_counts64, recv_counts64, world: int, e_loc: int, regroup_buf: torch.Tensor, splits_host=None) -> EpPlan:
    """Plan of one expert-parallel exchange from the two count vectors, computed on the device; the host reads only the
    2 * world split sizes (one small copy into pinned memory; torch.distributed's a2a-v takes python lists)."""
    _req(send_counts64, torch.int64, "send_counts"); _req(recv_counts64, torch.int64, "recv_counts")
    _req(regroup_buf, torch.int32, "regroup_buf")
    dev = send_counts64.device
    p = EpPlan()
    p.splits = torch.empty(2 * world, dtype=torch.int64, device=dev)
    p.offsets = torch.empty(e_loc + 1, dtype=torch.int32, device=dev)
    p.tile_starts = torch.empty(e_loc + 1, dtype=torch.int32, device=dev)
    check(lib().m3_ep_plan(_p(send_counts64), _p(recv_counts64), world, e_loc, _p(p.splits), _p(regroup_buf),
                           regroup_buf.numel(), _p(p.offsets), _p(p.tile_starts), _stream()), "m3_ep_plan")
    if splits_host is None:
        splits_host = torch.empty(2 * world, dtype=torch.int64, pin_memory=True)
    splits_host.copy_(p.splits, non_blocking=True)
    torch.cuda.current_stream().synchronize()          # the one host read of the exchange: 2 * world integers
    sp = splits_host.tolist()
    p.in_splits, p.out_splits = sp[:world], sp[world:]
    p.n_recv = sum(p.out_splits)
    if p.n_recv > regroup_buf.numel():
        raise _lib.M3Error(f"ep_plan: {p.n_recv} received rows exceed the regroup buffer ({regroup_buf.numel()})")
    p.regroup = regroup_buf[: p.n_recv]
    return p


def ep_plan_chunks(send_counts64, recv_counts64, world: int, e_chunk: int, regroup_bufs: torch.Tensor, splits_host=None):
    """The plans of an exchange cut into C chunks of e_chunk local experts each (send / recv counts laid out
    [C][world][e_chunk]: the routing keys are chunk-major, BackboneEngine ep_chunks): C m3_ep_plan launches, then ONE host read
    of all C * 2 * world split sizes.  regroup_bufs: i32 [C, capacity].  Returns a list of EpPlan."""
    _req(send_counts64, torch.int64, "send_counts"); _req(recv_counts64, torch.int64, "recv_counts")
    _req(regroup_bufs, torch.int32, "regroup_bufs")
    C = regroup_bufs.shape[0]
    assert send_counts64.numel() == C * world * e_chunk == recv_counts64.numel()
    dev = send_counts64.device
    send, recv = send_counts64.view(C, world * e_chunk), recv_counts64.view(C, world * e_chunk)
    splits = torch.empty(C, 2 * world, dtype=torch.int64, device=dev)
    plans = []
    for c in range(C):
        p = EpPlan()
        p.splits = splits[c]
        p.offsets = torch.empty(e_chunk + 1, dtype=torch.int32, device=dev)
        p.tile_starts = torch.empty(e_chunk + 1, dtype=torch.int32, device=dev)
        check(lib().m3_ep_plan(_p(send[c]), _p(recv[c]), world, e_chunk, _p(p.splits), _p(regroup_bufs[c]),
                               regroup_bufs.shape[1], _p(p.offsets), _p(p.tile_starts), _stream()), "m3_ep_plan")
        plans.append(p)
    if splits_host is None:
        splits_host = torch.empty(C, 2 * world, dtype=torch.int64, pin_memory=True)
    splits_host.copy_(splits, non_blocking=True)
    torch.cuda.current_stream().synchronize()          # the one host read of the exchange: C * 2 * world integers
    sp = splits_host.tolist()
    for c, p in enumerate(plans):
        p.in_splits, p.out_splits = sp[c][:world], sp[c][world:]
        p.n_recv = sum(p.out_splits)
        if p.n_recv > regroup_bufs.shape[1]:
            raise _lib.M3Error(f"ep_plan_chunks: {p.n_recv} received rows exceed the regroup buffer ({regroup_bufs.shape[1]})")
        p.regroup = regroup_bufs[c, : p.n_recv]
    return plans


class EpPlanFixed:
    """Device-resident plan of a fixed-capacity exchange (m3_ep_plan_fixed): nothing of it is read by the host."""
    __slots__ = ("cap", "world", "regroup", "offsets", "tile_starts", "pad_idx", "unpad_idx", "splits")


def ep_plan_fixed(send_counts64, recv_counts64, world: int, e_loc: int, cap: int, route: "Route", overflow: torch.Tensor,
                  bufs: Optional[EpPlanFixed] = None) -> EpPlanFixed:
    """Plan of one expert-parallel exchange with `cap` rows per (source, destination) pair, entirely on the device.
    overflow: i32 [1] flag the kernel raises (and never clears) when a pair routes more than cap rows.
    bufs: a previous plan whose index buffers are overwritten (static addresses for a captured step)."""
    _req(send_counts64, torch.int64, "send_counts"); _req(recv_counts64, torch.int64, "recv_counts")
    _req(overflow, torch.int32, "overflow")
    dev = send_counts64.device
    p = bufs
    if p is None:
        p = EpPlanFixed()
        p.cap, p.world = int(cap), int(world)
        p.regroup = torch.empty(world * cap, dtype=torch.int32, device=dev)
        p.pad_idx = torch.empty(world * cap, dtype=torch.int32, device=dev)
        p.unpad_idx = torch.empty(max(route.n, 1), dtype=torch.int32, device=dev)
        p.offsets = torch.empty(e_loc + 1, dtype=torch.int32, device=dev)
        p.tile_starts = torch.empty(e_loc + 1, dtype=torch.int32, device=dev)
        p.splits = torch.empty(2 * world, dtype=torch.int64, device=dev)
    assert p.cap == cap and p.world == world and p.unpad_idx.numel() >= route.n
    check(lib().m3_ep_plan_fixed(_p(send_counts64), _p(recv_counts64), world, e_loc, cap, _p(route.row_of_slot), _p(route.pos),
                                 route.n, _p(p.splits), _p(p.regroup), _p(p.offsets), _p(p.tile_starts), _p(p.pad_idx),
                                 _p(p.unpad_idx), _p(overflow), _stream()), "m3_ep_plan_fixed")
    return p


# ----------------------------------------------------------------------------- GEMM
def gemm_nt(A, B, C, *, M=None, bias=None, act=M3_ACT_NONE, pre_out=None, gelu_grad_pre=None, residual=None,
            a_row_idx=None, a_row_div=1, c_row_idx=None, group_offsets=None, tile_starts=None, row_scale=None,
            row_scale_div=1, row_scale_idx=None):
    """C[m,n] = epi(sum_k A[arow(m),k] B[g][n,k]).  A [rows,K]; B [N,K] or [G,N,K]; C [rows,N] (f32 or A.dtype)."""
    _req(A, name="A"); _req(B, A.dtype, "B"); _req(C, name="C")
    G = 1 if B.dim() == 2 else B.shape[0]
    N, K = B.shape[-2], B.shape[-1]
    a = GemmArgs()
    a.A = A.data_ptr(); a.lda = A.stride(0)
    a.a_row_idx = a_row_idx.data_ptr() if a_row_idx is not None else None
    a.a_row_div = a_row_div
    a.B = B.data_ptr(); a.ldb = B.stride(-2)
    a.C = C.data_ptr(); a.ldc = C.stride(0); a.c_dtype = dt_code(C.dtype)
    a.c_row_idx = c_row_idx.data_ptr() if c_row_idx is not None else None
    a.bias = bias.data_ptr() if bias is not None else None
    a.pre_out = pre_out.data_ptr() if pre_out is not None else None
    a.ld_pre = pre_out.stride(0) if pre_out is not None else 0
    a.gelu_grad_pre = gelu_grad_pre.data_ptr() if gelu_grad_pre is not None else None
    a.ld_gpre = gelu_grad_pre.stride(0) if gelu_grad_pre is not None else 0
    a.residual = residual.data_ptr() if residual is not None else None
    a.ld_res = residual.stride(0) if residual is not None else 0
    a.act = act
    if row_scale is not None:
        _req(row_scale, torch.float32, "row_scale")
    a.row_scale = row_scale.data_ptr() if row_scale is not None else None
    a.row_scale_div = row_scale_div
    if row_scale_idx is not None:
        _req(row_scale_idx, torch.int32, "row_scale_idx")
        assert row_scale is not None
    a.row_scale_idx = row_scale_idx.data_ptr() if row_scale_idx is not None else None
    if M is None:
        M = a_row_idx.numel() if a_row_idx is not None else A.shape[0]
    a.M = M; a.N = N; a.K = K; a.G = G
    a.group_offsets = group_offsets.data_ptr() if group_offsets is not None else None
    a.tile_starts = tile_starts.data_ptr() if tile_starts is not None else None
    a.dtype = dt_code(A.dtype)
    if bias is not None:
        _req(bias, torch.float32, "bias")
    if residual is not None:
        _req(residual, torch.float32, "residual")
    if M == 0:
        return C
    check(lib().m3_gemm_nt(byref(a), _stream()), "m3_gemm_nt")
    return C


def gemm_set_variant(ws_mask: int):
    """which gemm_nt calls may take the weight-stationary kernel (include/m3vit_hip.h: m3_gemm_set_variant); 0 = none"""
    check(lib().m3_gemm_set_variant(int(ws_mask)), "m3_gemm_set_variant")


def gemm_set_big(mode: int):
    """which gemm_nt calls take the 256 x 256-tile kernel for long contractions (include/m3vit_hip.h: m3_gemm_set_big):
    0 never, 1 every call it can run, 2 (default) those with enough tiles to fill the chip, -1 re-read M3_GEMM_BIG"""
    check(lib().m3_gemm_set_big(int(mode)), "m3_gemm_set_big")


def ffn_supported(D: int, H: int, dtype: torch.dtype, G: int = 1) -> bool:
    """shapes the fused FFN kernels take (anything else runs the unfused m3_gemm_nt pair)"""
    return dtype == torch.float16 and D in (384, 768) and H % 64 == 0 and 64 <= H <= 8192 and G <= 64 and \
        49152 + 64 * (2 if D == 384 else 1) * D * 2 + (H + D) * 4 <= 163840


def ffn_fwd(X, W1, W2p, Y, *, b1=None, b2=None, M=None, residual=None, x_row_idx=None, x_row_div=1, y_row_idx=None,
            group_offsets=None, pre_out=None, act_out=None):
    """Y[crow(m)] = (residual +) GELU(X[arow(m)] W1[g]^T + b1[g]) W2[g]^T + b2[g] in one launch (m3_ffn_fwd).
    X [rows, D] f16; W1 [H, D] or [G, H, D]; W2p = W2 [.., D, H] with the PERM32 column order (CastPlan perm flag);
    Y [rows, D] f16 or f32.  pre_out / act_out (optional, f16 [M, H], slot order): x W1^T + b1 and its GELU."""
    _req(X, torch.float16, "X"); _req(W1, torch.float16, "W1"); _req(W2p, torch.float16, "W2p"); _req(Y, name="Y")
    G = 1 if W1.dim() == 2 else W1.shape[0]
    H, D = W1.shape[-2], W1.shape[-1]
    assert W2p.shape[-2] == D and W2p.shape[-1] == H
    a = _lib.FfnArgs()
    a.X = X.data_ptr(); a.ldx = X.stride(0)
    a.x_row_idx = x_row_idx.data_ptr() if x_row_idx is not None else None
    a.x_row_div = x_row_div
    a.W1 = W1.data_ptr(); a.W2p = W2p.data_ptr()
    for b, n in ((b1, "b1"), (b2, "b2")):
        if b is not None:
            _req(b, torch.float32, n)
    a.b1 = b1.data_ptr() if b1 is not None else None
    a.b2 = b2.data_ptr() if b2 is not None else None
    a.Y = Y.data_ptr(); a.ldy = Y.stride(0); a.y_dtype = dt_code(Y.dtype)
    a.y_row_idx = y_row_idx.data_ptr() if y_row_idx is not None else None
    if residual is not None:
        _req(residual, torch.float32, "residual")
    a.residual = residual.data_ptr() if residual is not None else None
    a.ld_res = residual.stride(0) if residual is not None else 0
    if M is None:
        M = x_row_idx.numel() if x_row_idx is not None else X.shape[0]
    for o, n in ((pre_out, "pre_out"), (act_out, "act_out")):
        if o is not None:
            _req(o, torch.float16, n)
            assert o.shape[-1] == H and o.stride(0) == H and o.shape[0] >= M
    a.pre_out = pre_out.data_ptr() if pre_out is not None else None
    a.act_out = act_out.data_ptr() if act_out is not None else None
    a.M = M; a.D = D; a.H = H; a.G = G
    a.group_offsets = group_offsets.data_ptr() if group_offsets is not None else None
    a.dtype = M3_F16
    if M == 0:
        return Y
    check(lib().m3_ffn_fwd(byref(a), _stream()), "m3_ffn_fwd")
    return Y


class WgradQueue:
    """Weight-gradient calls of ONE stream whose slab reductions ride in front of the next call's launch
    (m3_wgrad_args.prev) instead of being launched by themselves: two slab workspaces used in turn, the reduction of the
    latest call pending until the next wgrad_tn(.., queue=self) or flush().  Whoever reads the gradients (an all-reduce,
    the optimizer, a test) calls flush() first."""

    def __init__(self, ws_elems: int, device):
        self.ws = [torch.empty(ws_elems, dtype=torch.float32, device=device) for _ in range(2)]
        self.i = 0
        self.pending = None          # (WgradReduceDesc, tensors it points at)

    def reset(self):
        """drop a pending reduction without running it (step boundaries: see BackboneEngine.zero_grad)"""
        self.pending = None
        self.i = 0

    def flush(self):
        if self.pending is None:
            return
        d, _keep = self.pending
        self.pending = None
        if d.elems == 0:
            return
        if d.chunk_rows:
            check(lib().m3_wgrad_reduce_grouped(d.ws, d.group_offsets, d.G, d.chunk_rows, d.elems, d.dW, d.beta, d.bias_ws,
                                                d.bias_elems, d.db, d.beta_db, _stream()), "m3_wgrad_reduce_grouped")
        else:
            check(lib().m3_wgrad_reduce(d.ws, d.splits, d.elems, d.dW, d.beta, d.bias_ws, d.bias_elems, d.db, d.beta_db,
                                        _stream()), "m3_wgrad_reduce")


def wgrad_tn(dC, A, dW, *, M=None, beta=0, splits=None, ws=None, c_row_idx=None, a_row_idx=None, a_row_div=1,
             group_offsets=None, db=None, beta_db=None, c_row_div=1, c_row_scale=None, queue: Optional[WgradQueue] = None):
    """dW[g][n,k] (+)= sum_m dC[crow(m),n] A[arow(m),k].  dW f32 [N,K] or [G,N,K].
    db (optional, f32 [N] / [G,N]): bias gradient = column sums of dC, fused into the same pass.
    c_row_div / c_row_scale (with c_row_idx): slot m reads c_row_scale[c_row_idx[m]] * dC[c_row_idx[m] // c_row_div].
    queue: this call's slab reduction is left pending in the queue (dW is complete only after the queue's next call or
    flush()) and the queue's previous pending reduction runs in front of this launch."""
    _req(dC, name="dC"); _req(A, dC.dtype, "A"); _req(dW, torch.float32, "dW")
    G = 1 if dW.dim() == 2 else dW.shape[0]
    N, K = dW.shape[-2], dW.shape[-1]
    if M is None:
        M = c_row_idx.numel() if c_row_idx is not None else dC.shape[0]
    if M == 0:                                   # nothing to contract over: no launch (an empty tensor has no address)
        if not beta:
            dW.zero_()
        if db is not None and not (beta if beta_db is None else beta_db):
            db.zero_()
        return dW
    if splits is None:
        splits = default_wgrad_splits(M, N, K, G, dC.dtype)
    # direct mode (include/m3vit_hip.h: m3_wgrad_args.direct_dW): with ONE part per group every (group, tile) belongs to one
    # workgroup, which adds its tile into dW itself - no slabs, no reduction.  The default split rule says 1 exactly when the
    # tiles alone fill the chip (the ViT-Base experts: 2304 tiles, 151 MB of gradient per layer)
    direct = _WGRAD_DIRECT and splits == 1 and wgrad_tile(N, K, dC.dtype) in ((128, 128), (256, 256)) and dW.data_ptr() % 16 == 0
    balanced = group_offsets is not None and not direct
    chunk, units = wgrad_plan(M, G, splits, balanced)
    balanced = chunk > 0
    need = 0 if direct else units * N * K + (units * N if db is not None else 0)
    if queue is not None:
        ws = queue.ws[queue.i]
        assert ws.numel() >= need, "WgradQueue workspace too small for this shape"
    elif ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 4), dtype=torch.float32, device=dW.device)
    a = WgradArgs()
    a.dC = dC.data_ptr(); a.lddc = dC.stride(0)
    a.c_row_idx = c_row_idx.data_ptr() if c_row_idx is not None else None
    a.c_row_div = c_row_div
    if c_row_scale is not None:
        _req(c_row_scale, torch.float32, "c_row_scale")
        assert c_row_idx is not None
    a.c_row_scale = c_row_scale.data_ptr() if c_row_scale is not None else None
    a.A = A.data_ptr(); a.lda = A.stride(0)
    a.a_row_idx = a_row_idx.data_ptr() if a_row_idx is not None else None
    a.a_row_div = a_row_div
    a.M = M; a.N = N; a.K = K; a.G = G
    a.group_offsets = group_offsets.data_ptr() if group_offsets is not None else None
    a.splits = splits
    a.chunk_rows = chunk; a.units = units
    a.ws = ws.data_ptr()
    a.dtype = dt_code(dC.dtype)
    bias_ws = ws[units * N * K:] if (db is not None and not direct) else None
    a.bias_ws = bias_ws.data_ptr() if bias_ws is not None else None
    bdb = beta if beta_db is None else beta_db
    if direct:
        if db is not None:
            _req(db, torch.float32, "db")
        a.direct_dW = dW.data_ptr(); a.direct_beta = 1 if beta else 0
        a.direct_db = db.data_ptr() if db is not None else None
        a.direct_beta_db = 1 if bdb else 0
        if queue is not None and queue.pending is not None:      # the previous call's reduction still rides in front
            if queue.pending[0].dW == dW.data_ptr():              # ... unless it writes the tensor this launch read-add-writes
                queue.flush()
            else:
                a.prev = ctypes.pointer(queue.pending[0])
        check(lib().m3_wgrad_tn(byref(a), _stream()), "m3_wgrad_tn")
        if queue is not None:
            queue.pending = None                                  # (this call left nothing to reduce; its slab buffer was not used)
        return dW
    fuse = False
    if db is not None:
        _req(db, torch.float32, "db")
        fuse = db.data_ptr() % 16 == 0 and bias_ws.data_ptr() % 16 == 0      # both slab reductions in one launch
    if queue is not None:
        assert db is None or fuse, "queued wgrad: db and the bias slabs must be 16-byte aligned"
        prev = queue.pending
        if prev is not None:
            a.prev = ctypes.pointer(prev[0])
        check(lib().m3_wgrad_tn(byref(a), _stream()), "m3_wgrad_tn")
        d = WgradReduceDesc()
        d.ws = ws.data_ptr(); d.splits = splits
        d.elems = (N * K) if balanced else (G * N * K)
        d.group_offsets = group_offsets.data_ptr() if balanced else None
        d.G = G; d.chunk_rows = chunk if balanced else 0
        d.dW = dW.data_ptr(); d.beta = beta
        d.bias_ws = bias_ws.data_ptr() if fuse else None
        d.bias_elems = (N if balanced else G * N) if fuse else 0
        d.db = db.data_ptr() if fuse else None
        d.beta_db = bdb
        queue.pending = (d, (ws, dW, db, group_offsets))
        queue.i ^= 1
        return dW
    check(lib().m3_wgrad_tn(byref(a), _stream()), "m3_wgrad_tn")
    if balanced:
        assert db is None or fuse, "balanced grouped wgrad: db and the bias slabs must be 16-byte aligned"
        check(lib().m3_wgrad_reduce_grouped(_p(ws), _p(group_offsets), G, chunk, N * K, _p(dW), beta,
                                            _p(bias_ws) if fuse else None, N, _p(db) if fuse else None, bdb, _stream()),
              "m3_wgrad_reduce_grouped")
        return dW
    check(lib().m3_wgrad_reduce(_p(ws), splits, G * N * K, _p(dW), beta, _p(bias_ws) if fuse else None, G * N,
                                _p(db) if fuse else None, bdb, _stream()), "m3_wgrad_reduce")
    if db is not None and not fuse:
        check(lib().m3_wgrad_bias_reduce(_p(bias_ws), splits, G * N, _p(db), bdb, _stream()), "m3_wgrad_bias_reduce")
    return dW


def wgrad_plan(M, G, splits, grouped):
    """(chunk_rows, slab slots) of a weight-gradient call.  Grouped calls: work units of equal row counts dealt to the
    groups by their (device-resident) sizes, so that a hot expert gets more workgroups instead of longer ones; `splits`
    is the average number of units per group and the chunk sits 1/8 above the mean part, so that groups near the mean keep
    `splits` units.  chunk 0: every group in `splits` equal parts."""
    if grouped and 1 < G <= 64 and M > 0:
        chunk = max(64, (-(-M * 9 // (8 * splits * G)) + 63) // 64 * 64)
        return chunk, M // chunk + G
    return 0, splits * G


def wgrad_ws_elems(M, N, K, G, grouped, bias=True, dtype=None):
    """fp32 elements of workspace wgrad_tn needs for this shape with the default splits"""
    _, units = wgrad_plan(M, G, default_wgrad_splits(M, N, K, G, dtype), grouped)
    return units * N * (K + (1 if bias else 0))


_WGRAD_MIN_STEPS = 16      # 32-row steps per split at least (measured on the 8-image configs; no effect at batch 128)
import os as _os
_WGRAD_DIRECT = _os.environ.get("M3_WGRAD_DIRECT", "1") != "0"      # splits == 1: the kernel accumulates into dW itself (no slabs)
# workgroup slots a weight-gradient launch is split to fill: the LDS-DMA kernel runs four workgroups per CU, the
# register-staged one two; which kernel takes a launch is m3_wgrad_tn's rule (include/m3vit_hip.h: m3_wgrad_set_dma),
# mirrored in _wgrad_uses_dma (measured with streamed operands: tools/wgrad_ab_bench.py, profiles/r05_wgrad_ab_streamed.txt)
_WGRAD_MOST16 = int(_os.environ.get("M3_WGRAD_MOST16", "32"))       # most row parts of a 16-bit weight gradient (A/B knob; 44 / 56 measured level to +0.5 % at configs[1])
_WGRAD_XCD_ALIGN = _os.environ.get("M3_WGRAD_XCD_ALIGN", "1") != "0"
_WGRAD_SLOTS = int(_os.environ.get("M3_WGRAD_SLOTS", "0"))         # 0: by kernel (1024 / 512)
_WGRAD_DMA = int(_os.environ.get("M3_WGRAD_DMA", "1"))              # 0 never, 1 where it pays (default), 2 wherever it can run


def _wgrad_uses_dma(N, K, G, dtype, tiles_total):
    if _WGRAD_DMA == 0:
        return False
    if _WGRAD_DMA == 2 or dtype == torch.float32:
        return True
    return N * K >= 1500000 or tiles_total >= 1024


def _wgrad_slots(dtype, N=0, K=0, G=1, tiles_total=0):
    if _WGRAD_SLOTS:
        return _WGRAD_SLOTS
    return 1024 if _wgrad_uses_dma(N, K, G, dtype, tiles_total) else 512


def wgrad_set_wide(on: int):
    """wide weight-gradient tiles on / off (include/m3vit_hip.h: m3_wgrad_set_wide); switch before sizing workspaces"""
    check(lib().m3_wgrad_set_wide(int(on)), "m3_wgrad_set_wide")


def wgrad_set_big(on: int):
    """256 x 256 weight-gradient tiles for the 16-bit ViT-Base shapes: 1 on (default) / 0 off / -1 from M3_WGRAD_BIG
    (include/m3vit_hip.h: m3_wgrad_set_big); switch before sizing workspaces"""
    check(lib().m3_wgrad_set_big(int(on)), "m3_wgrad_set_big")


def wgrad_set_dma(on: int):
    """LDS-DMA weight-gradient kernel: 0 never / 1 where it pays (default) / 2 wherever it can run / -1 from M3_WGRAD_DMA
    (include/m3vit_hip.h: m3_wgrad_set_dma);
    switch before sizing workspaces (the default row splits follow the kernel's workgroups per CU)"""
    global _WGRAD_DMA
    check(lib().m3_wgrad_set_dma(int(on)), "m3_wgrad_set_dma")
    _WGRAD_DMA = int(_os.environ.get("M3_WGRAD_DMA", "1")) if int(on) < 0 else int(on)


def wgrad_tile(N, K, dtype=None):
    """(tn, tk): the output tile m3_wgrad_tn uses for this shape (m3_wgrad_tile in include/m3vit_hip.h)"""
    if dtype is None:
        return 128, 128
    tn, tk = ctypes.c_int(0), ctypes.c_int(0)
    check(lib().m3_wgrad_tile(N, K, dt_code(dtype), byref(tn), byref(tk)), "m3_wgrad_tile")
    return tn.value, tk.value


def wgrad_skinny(N, K, G=1) -> bool:
    """m3_wgrad_tn takes plain calls of this shape (K = 16 / 32, one group) with the streaming kernel (m3_wgrad_skinny)"""
    return bool(lib().m3_wgrad_skinny(int(N), int(K), int(G)))


def _xcd_aligned(splits):
    """Row parts of a dense call in whole multiples of the 8 XCDs where that costs at most 1/8 of the parts: the tiles of a part
    read the same rows, and the XCD remap hands every XCD an equal run of consecutive workgroups - with a multiple of 8 parts no
    part straddles two XCDs (its rows then come into one L2, not two).  M3_WGRAD_XCD_ALIGN=0 switches it off."""
    if not _WGRAD_XCD_ALIGN or splits < 8:
        return splits
    down = splits - splits % 8
    return down if 8 * down >= 7 * splits else splits


def default_wgrad_splits(M, N, K, G, dtype=None):
    """Row splits of the TN GEMM.  128 x 128 tiles: fill the 512 resident workgroup slots (2 per CU) exactly once - more
    splits only add slab traffic and a ragged second wave of workgroups - but keep at least _WGRAD_MIN_STEPS 32-row
    steps per split, so that short contractions (few tokens) do not pay a 64 KiB slab write + reduce per handful of
    steps.  Wide tiles (one 512-thread workgroup per CU, 256 slots): as many splits as fill the slots once; with more
    tiles than a third of the slots (grouped experts) the split count whose workgroups come closest to whole rounds."""
    if wgrad_skinny(N, K, G):                       # the router's weight: a stream over dC, 64+ rows per part (16 per wave)
        return int(max(1, min(256, M // 64)))
    tn, tk = wgrad_tile(N, K, dtype)
    tiles = ((N + tn - 1) // tn) * ((K + tk - 1) // tk) * G
    steps = max(1, (M // max(G, 1) + 31) // 32)
    cap = max(1, steps // _WGRAD_MIN_STEPS)
    if (tn, tk) == (128, 128):
        nslots = _wgrad_slots(dtype, N, K, G, tiles)
        # fp32 is MFMA-bound (1/16 of the fp16 rate): its workgroup slots matter more than its slab bytes, so small weights
        # (proj: 9 tiles) may be cut into as many parts as fill them; 16-bit stays at 32 (slab traffic)
        most = 128 if dtype == torch.float32 else _WGRAD_MOST16
        sp = int(max(1, min(cap, most, nslots // tiles if tiles <= nslots else 1)))
        return _xcd_aligned(sp) if G == 1 else sp
    slots = 256
    if (tn, tk) == (256, 256):
        # one 8-wave workgroup per CU: fill the 256 slots once; with more tiles than slots (grouped experts) one part per
        # group - the kernel then accumulates into dW itself (direct mode), no slabs
        sp = int(max(1, min(cap, 32, slots // tiles))) if tiles < slots else 1
        return _xcd_aligned(sp) if G == 1 else sp
    if 3 * tiles <= slots:
        return int(max(1, min(cap, slots // tiles)))
    best, best_cost = 1, None
    for s_ in range(1, min(cap, 8) + 1):
        cost = -(-tiles * s_ // slots) / s_ * (1 + 0.15 * s_)      # rounds x length of a unit (+ its slab: measured, tools/wgrad_bench.py --splits)
        if best_cost is None or cost < best_cost:
            best, best_cost = s_, cost
    return int(best)


def colsum(dC, db, *, M=None, beta=0, c_row_idx=None, group_offsets=None, ws=None):
    _req(dC, name="dC"); _req(db, torch.float32, "db")
    G = 1 if db.dim() == 1 else db.shape[0]
    N = db.shape[-1]
    if M is None:
        M = dC.shape[0]
    need = int(lib().m3_colsum_ws_elems(M, N, G))
    if ws is None:
        ws = torch.empty(need, dtype=torch.float32, device=dC.device)
    assert ws.numel() >= need
    check(lib().m3_colsum(_p(dC), dt_code(dC.dtype), dC.stride(0), _p(c_row_idx), M, N, G, _p(group_offsets),
                          _p(ws), _p(db), beta, _stream()), "m3_colsum")
    return db


# -------------------------------------------------------------------- combine / LN
def combine_fwd(y, score, residual, out):
    T, k = score.shape
    D = y.shape[-1]
    check(lib().m3_combine_fwd(_p(y), dt_code(y.dtype), _p(score), _p(residual), T, k, D, _p(out), _stream()),
          "m3_combine_fwd")
    return out


def combine_bwd(dout, y, score, dy, dscore):
    """dy (may be None: d score only) [T*k, D] = score * dout ; dscore [T, k] = <dout, y>"""
    T, k = score.shape
    D = y.shape[-1]
    check(lib().m3_combine_bwd(_p(dout), _p(y), dt_code(y.dtype), _p(score), T, k, D, _p(dy), _p(dscore), _stream()),
          "m3_combine_bwd")


def combine_gate_bwd(dxe, k, d_logits, w_gate, dh):
    """dh [T, D] (fp32 or dxe's dtype) = sum_j dxe[t*k+j] + d_logits [T, E] @ w_gate[:D] [D, E]^T  (one pass; m3_combine_gate_bwd)"""
    T, E = d_logits.shape
    D = dxe.shape[-1]
    assert w_gate.dtype == torch.float32 and d_logits.dtype == torch.float32 and w_gate.is_contiguous()
    assert tuple(w_gate.shape) == (D, E) and dxe.shape[0] == T * k and tuple(dh.shape) == (T, D) and dh.is_contiguous()
    check(lib().m3_combine_gate_bwd(_p(dxe), dt_code(dxe.dtype), T, k, D, _p(d_logits), _p(w_gate), E, _p(dh), dt_code(dh.dtype),
                                    _stream()), "m3_combine_gate_bwd")
    return dh


def gather_rows(src, idx, dst, div=1, k=1):
    """dst[i] = sum_{j<k} src[idx[i*k+j] // div]."""
    nout, D = dst.shape
    check(lib().m3_gather_rows(_p(src), dt_code(src.dtype), _p(idx), div, nout, k, D, _p(dst), _stream()), "m3_gather_rows")
    return dst


def layernorm_fwd(x, gamma, beta, y, mean, rstd, eps=1e-6):
    T, D = x.shape
    check(lib().m3_layernorm_fwd(_p(x), T, D, _p(gamma), _p(beta), float(eps), _p(y), dt_code(y.dtype), _p(mean),
                                 _p(rstd), _stream()), "m3_layernorm_fwd")


def layernorm_bwd(dy, x, mean, rstd, gamma, dx_res, dx, dgamma, dbeta, beta=0, ws=None, dx_act=None):
    """dgamma = dbeta = None: the parameter-gradient partials stay in ws (fp32 [2, ln_bwd_blocks(T), D]) for a later
    batched layernorm_bwd_reduce."""
    T, D = x.shape
    nblk = lib().m3_ln_bwd_blocks(T, D)
    if ws is None:
        ws = torch.empty(2 * nblk * D, dtype=torch.float32, device=x.device)
    check(lib().m3_layernorm_bwd(_p(dy), dt_code(dy.dtype), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx_res), T, D,
                                 _p(dx), _p(ws), _p(dgamma), _p(dbeta), beta, _p(dx_act),
                                 dt_code(dx_act.dtype) if dx_act is not None else M3_F32, _stream()), "m3_layernorm_bwd")


class LnGradTable:
    """device-resident (dgamma, dbeta) pointer pairs of a model's LayerNorms, in workspace-slot order"""

    def __init__(self, pairs, device):
        import struct
        self.keep = pairs
        for g, b in pairs:
            _req(g, torch.float32, "dgamma"); _req(b, torch.float32, "dbeta")
        raw = b"".join(struct.pack("QQ", g.data_ptr(), b.data_ptr()) for g, b in pairs)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.n = len(pairs)


def layernorm_bwd_reduce(ws, nblk, D, table: LnGradTable, first, count, beta=1):
    """dgamma / dbeta of LayerNorms first .. first+count-1 from their partial slots ws[j] (fp32 [n, 2, nblk, D])"""
    _req(ws, torch.float32, "ws")
    assert ws.dim() == 4 and ws.shape[1] == 2 and ws.shape[2] == nblk and ws.shape[3] == D and first + count <= table.n
    check(lib().m3_layernorm_bwd_reduce(_p(ws), ws.stride(0), nblk, D, _p(table.table), first, count, beta, _stream()),
          "m3_layernorm_bwd_reduce")


# ------------------------------------------------------------------------ attention
def attention_fwd(qkv, B, N, heads, dh, o, lse):
    check(lib().m3_attention_fwd(_p(qkv), dt_code(qkv.dtype), B, N, heads, dh, _p(o), _p(lse), _stream()),
          "m3_attention_fwd")


def attention_bwd(qkv, o, d_o, lse, B, N, heads, dh, dqkv, dq_ws=None):
    need = int(lib().m3_attention_bwd_ws_elems(B, N, heads, dh))
    if need and dq_ws is None:
        dq_ws = torch.empty(need, dtype=torch.float32, device=qkv.device)
    check(lib().m3_attention_bwd(_p(qkv), _p(o), _p(d_o), _p(lse), dt_code(qkv.dtype), B, N, heads, dh, _p(dqkv),
                                 _p(dq_ws) if need else None, _stream()), "m3_attention_bwd")


# ---------------------------------------------------------------------- elementwise
def cast_matrix(src, dst, transpose=False):
    """src f32 [G,R,C] or [R,C] -> dst (act dtype) same shape, or [G,C,R] when transpose."""
    G = 1 if src.dim() == 2 else src.shape[0]
    R, C = src.shape[-2], src.shape[-1]
    check(lib().m3_cast_matrix(_p(src), G, R, C, 1 if transpose else 0, _p(dst), dt_code(dst.dtype), _stream()),
          "m3_cast_matrix")
    return dst


class CastPlan:
    """Device-resident descriptor table for m3_cast_batch: every (fp32 master -> operand copies) job of a
    model, converted by ONE launch per optimizer step."""

    def __init__(self, jobs, dst_dtype):
        # jobs: list of (src fp32 [.., rows, cols], dst [.., rows, cols] or None, dst_t [.., cols, rows] or None[, flags]):
        # the plain and / or the transposed copy, both written from one read of src
        arr = (_lib.CastDesc * len(jobs))()
        t0 = 0
        self.keep = []
        for d, job in zip(arr, jobs):
            src, dst, dst_t = job[:3]
            d.flags = job[3] if len(job) > 3 else 0
            _req(src, torch.float32, "src")
            rows, cols = src.shape[-2], src.shape[-1]
            G = src.numel() // (rows * cols)
            for o in (dst, dst_t):
                if o is not None:
                    _req(o, dst_dtype, "dst")
                    assert o.numel() == src.numel()
            assert dst is not None or dst_t is not None
            if d.flags & _lib.M3_CAST_PERM32:
                assert dst is not None and cols % 32 == 0
            if d.flags & _lib.M3_CAST_PERM32_T:
                assert dst_t is not None and rows % 32 == 0
            d.src, d.G, d.rows, d.cols, d.tile_start = src.data_ptr(), G, rows, cols, t0
            d.dst = dst.data_ptr() if dst is not None else None
            d.dst_t = dst_t.data_ptr() if dst_t is not None else None
            t0 += G * ((rows + 31) // 32) * ((cols + 31) // 32)
            self.keep += [src, dst, dst_t]
        self.n, self.total, self.dtype = len(jobs), t0, dst_dtype
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(jobs[0][0].device)

    def run(self):
        check(lib().m3_cast_batch(_p(self.table), self.n, self.total, dt_code(self.dtype), _stream()), "m3_cast_batch")


def add_f32(dst, src):
    """dst += src (flat fp32 buffers)."""
    _req(dst, torch.float32, "dst"); _req(src, torch.float32, "src")
    assert dst.numel() == src.numel()
    check(lib().m3_add_f32(_p(dst), _p(src), dst.numel(), _stream()), "m3_add_f32")
    return dst


def cast_f32(src, dst):
    check(lib().m3_cast_f32(_p(src), src.numel(), _p(dst), dt_code(dst.dtype), _stream()), "m3_cast_f32")
    return dst


def scale_rows_cast(src, row_scale, div, dst):
    """dst[r, :] = row_scale[r // div] * src[r, :] (src fp32 [rows, cols], dst fp32 / f16)"""
    _req(src, torch.float32, "src"); _req(row_scale, torch.float32, "row_scale"); _req(dst, name="dst")
    rows, cols = src.shape
    assert row_scale.numel() * div >= rows
    check(lib().m3_scale_rows_cast(_p(src), rows, cols, _p(row_scale), div, _p(dst), dt_code(dst.dtype), _stream()),
          "m3_scale_rows_cast")
    return dst


def _nhwc(t, name):
    """t is a [N, C, H, W] tensor in channels-last memory format (its storage is [N, H, W, C])"""
    if t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
        raise _lib.M3Error(f"{name} must be a 4-d channels-last tensor")
    if not t.is_cuda:
        raise _lib.M3Error(f"{name} must live on the GPU (no CPU path)")
    return t


def relu_up2x_fwd(x, relu=True, out_dtype=None):
    """y = bilinear x2 (align_corners False) of relu(x), channels-last [N, C, H, W] in -> [N, C, 2H, 2W] out (m3_relu_up2x_fwd)"""
    _nhwc(x, "x")
    N, C, H, W = x.shape
    y = torch.empty((N, C, 2 * H, 2 * W), dtype=out_dtype or x.dtype, device=x.device, memory_format=torch.channels_last)
    check(lib().m3_relu_up2x_fwd(_p(x), dt_code(x.dtype), N, H, W, C, 1 if relu else 0, _p(y), dt_code(y.dtype), _stream()),
          "m3_relu_up2x_fwd")
    return y


def relu_up2x_bwd(dy, x, relu=True):
    _nhwc(dy, "dy"); _nhwc(x, "x")
    N, C, H, W = x.shape
    assert tuple(dy.shape) == (N, C, 2 * H, 2 * W)
    dx = torch.empty_like(x, memory_format=torch.channels_last)
    check(lib().m3_relu_up2x_bwd(_p(dy), dt_code(dy.dtype), _p(x), dt_code(x.dtype), N, H, W, C, 1 if relu else 0, _p(dx),
                                 _stream()), "m3_relu_up2x_bwd")
    return dx


def im2row(img, P, rows):
    B, Cin, H, W = img.shape
    check(lib().m3_im2row(_p(img), B, Cin, H, W, P, _p(rows), dt_code(rows.dtype), _stream()), "m3_im2row")
    return rows


def assemble_tokens(patch, cls, pos, B, np_, D, tokens):
    check(lib().m3_assemble_tokens(_p(patch), _p(cls), _p(pos), B, np_, D, _p(tokens), _stream()), "m3_assemble_tokens")
    return tokens


def tokens_bwd(dtok, B, np_, D, dpatch, dpos, dcls, beta=0):
    check(lib().m3_tokens_bwd(_p(dtok), B, np_, D, _p(dpatch), dt_code(dpatch.dtype) if dpatch is not None else M3_F32,
                              _p(dpos), _p(dcls), beta, _stream()), "m3_tokens_bwd")
