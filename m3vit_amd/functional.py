"""torch.autograd.Function wrappers that make the HIP kernels (m3vit_amd.ops -> C ABI)
differentiable pieces of the reference's layer API.  Each Function's forward AND backward
are HIP kernel launches; torch only carries the graph.

Activation dtype = dtype of the incoming activation tensor (float32: exact-f32 MFMA, what
train_fastmoe.py runs; float16: f16 MFMA with fp32 accumulate, what the AMP trainer
pretrain/engine/train_one_epoch.py:35 runs).  Parameters stay fp32; their operand copies
are cast per call.
"""
from __future__ import annotations

import torch

from . import ops
from .ops import M3_ACT_GELU, M3_ACT_NONE


def _wcopy(w: torch.Tensor, dtype, transpose=False):
    """Operand copy of an fp32 parameter [.., N, K] in the activation dtype (optionally [.., K, N])."""
    w = w.detach()
    if not transpose and dtype == torch.float32:
        return w.contiguous()
    shp = list(w.shape)
    if transpose:
        shp[-1], shp[-2] = shp[-2], shp[-1]
    out = torch.empty(shp, dtype=dtype, device=w.device)
    return ops.cast_matrix(w.contiguous(), out, transpose=transpose)


def _as_act(g: torch.Tensor, dtype):
    g = g.contiguous()
    if g.dtype == dtype:
        return g
    if g.dtype == torch.float32:
        return ops.cast_f32(g, torch.empty_like(g, dtype=dtype))
    return g.to(dtype)


class MlpFn(torch.autograd.Function):
    """Dense Mlp: fc2(GELU(fc1 x)), vision_transformer_moe.py:255-261 (drop = 0).  The GELU derivative is
    fused into the fc2 dgrad GEMM's epilogue."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        dt = x2.dtype
        T = x2.shape[0]
        pre = torch.empty(T, w1.shape[0], dtype=dt, device=x.device)
        u = torch.empty_like(pre)
        ops.gemm_nt(x2, _wcopy(w1, dt), u, bias=b1.detach(), act=M3_ACT_GELU, pre_out=pre)
        y = torch.empty(T, w2.shape[0], dtype=dt, device=x.device)
        ops.gemm_nt(u, _wcopy(w2, dt), y, bias=b2.detach())
        ctx.save_for_backward(x2, w1, w2, pre, u)
        ctx.shp = shp
        return y.view(*shp[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, w1, w2, pre, u = ctx.saved_tensors
        dt = x2.dtype
        g = _as_act(gy.reshape(-1, gy.shape[-1]), dt)
        dev = x2.device
        dw2 = torch.empty(w2.shape, dtype=torch.float32, device=dev)
        ops.wgrad_tn(g, u, dw2)
        db2 = ops.colsum(g, torch.empty(w2.shape[0], dtype=torch.float32, device=dev))
        dpre = torch.empty_like(pre)
        ops.gemm_nt(g, _wcopy(w2, dt, transpose=True), dpre, gelu_grad_pre=pre)
        dw1 = torch.empty(w1.shape, dtype=torch.float32, device=dev)
        ops.wgrad_tn(dpre, x2, dw1)
        db1 = ops.colsum(dpre, torch.empty(w1.shape[0], dtype=torch.float32, device=dev))
        dx = torch.empty_like(x2)
        ops.gemm_nt(dpre, _wcopy(w1, dt, transpose=True), dx)
        return dx.view(ctx.shp), dw1, db1, dw2, db2


class PlainLinearFn(torch.autograd.Function):
    """y = x W^T + b (no activation)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        dt = x2.dtype
        y = torch.empty(x2.shape[0], weight.shape[0], dtype=dt, device=x.device)
        ops.gemm_nt(x2, _wcopy(weight, dt), y, bias=bias.detach() if bias is not None else None)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.shp = shp
        return y.view(*shp[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, weight = ctx.saved_tensors
        dt = x2.dtype
        g = _as_act(gy.reshape(-1, gy.shape[-1]), dt)
        dx = torch.empty_like(x2)
        ops.gemm_nt(g, _wcopy(weight, dt, transpose=True), dx)
        dw = torch.empty(weight.shape, dtype=torch.float32, device=weight.device)
        ops.wgrad_tn(g, x2, dw)
        db = ops.colsum(g, torch.empty(weight.shape[0], dtype=torch.float32, device=weight.device)) if ctx.has_bias else None
        return dx.view(ctx.shp), dw, db


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm(D, eps): fp32 in, activation dtype out."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_dtype):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous().float()
        T, D = x2.shape
        y = torch.empty(T, D, dtype=out_dtype, device=x.device)
        mean = torch.empty(T, device=x.device)
        rstd = torch.empty(T, device=x.device)
        ops.layernorm_fwd(x2, gamma.detach(), beta.detach(), y, mean, rstd, eps)
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.shp = shp
        ctx.in_dtype = x.dtype
        return y.view(shp)

    @staticmethod
    def backward(ctx, gy):
        x2, gamma, mean, rstd = ctx.saved_tensors
        g = gy.reshape(x2.shape).contiguous()
        if g.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            g = g.float()
        dx = torch.empty_like(x2)
        dg = torch.empty_like(gamma, dtype=torch.float32)
        db = torch.empty_like(gamma, dtype=torch.float32)
        ops.layernorm_bwd(g, x2, mean, rstd, gamma.detach(), None, dx, dg, db, beta=0)
        return dx.view(ctx.shp).to(ctx.in_dtype), dg, db, None, None


class AttentionCoreFn(torch.autograd.Function):
    """softmax(q k^T dh^-0.5) v on packed qkv [B, N, 3C]; vision_transformer_moe.py:299-310."""

    @staticmethod
    def forward(ctx, qkv, heads):
        B, N, C3 = qkv.shape
        C = C3 // 3
        dh = C // heads
        q2 = qkv.reshape(B * N, C3).contiguous()
        o = torch.empty(B * N, C, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B, heads, N, device=qkv.device)
        ops.attention_fwd(q2, B, N, heads, dh, o, lse)
        ctx.save_for_backward(q2, o, lse)
        ctx.dims = (B, N, heads, dh)
        return o.view(B, N, C)

    @staticmethod
    def backward(ctx, go):
        q2, o, lse = ctx.saved_tensors
        B, N, heads, dh = ctx.dims
        g = _as_act(go.reshape(B * N, heads * dh), q2.dtype)
        dqkv = torch.empty_like(q2)
        ops.attention_bwd(q2, o, g, lse, B, N, heads, dh, dqkv)
        return dqkv.view(B, N, 3 * heads * dh), None


class GateFn(torch.autograd.Function):
    """NoisyGate_VMoE arithmetic (noisy_gate_vmoe.py:91-93,168,197-207).  Differentiable outputs:
    score [T,k], top_logits [T,k+1] (what the Normal-CDF load term of vision_transformer_moe.py:456-457
    thresholds on), clean/noisy logits and the dense gates [T,E]; the indices carry no gradient."""

    @staticmethod
    def forward(ctx, x, w_gate, k, noise, noise_std, logit_bias):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        D = x2.shape[1]
        wg = w_gate.detach()
        wg_tok = wg if wg.shape[0] == D else wg[:D].contiguous()
        g = ops.gate_fwd(x2, wg_tok, k, logit_bias=logit_bias, noise=noise, noise_std=noise_std, dense=True)
        ctx.save_for_backward(x2, w_gate, g["noisy"], g["idx"], g["idx_next"])
        ctx.k = k
        ctx.mark_non_differentiable(g["idx"], g["idx32"], g["importance"], g["load"])
        return (g["idx"], g["score"], g["clean"], g["noisy"], g["top_logits"], g["gates"], g["idx32"],
                g["importance"], g["load"])

    @staticmethod
    def backward(ctx, g_idx, g_score, g_clean, g_noisy, g_top, g_gates, g_idx32, g_imp, g_load):
        x2, w_gate, noisy, idx, idx_next = ctx.saved_tensors
        T, E = noisy.shape
        k = ctx.k
        d_score = g_score.contiguous().float() if g_score is not None else None
        if g_gates is not None:
            # gates = zeros.scatter(1, idx, score): its gradient flows back to the selected scores
            extra = g_gates.float().gather(1, idx)
            d_score = extra if d_score is None else d_score + extra
        d_top = g_top.contiguous().float() if (g_top is not None and idx_next is not None) else None
        if g_top is not None and idx_next is None:                      # k == E: top_logits is score
            d_score = g_top.contiguous().float() if d_score is None else d_score + g_top.float()
        dl = ops.gate_bwd_logits(noisy, idx, d_score, None, k, d_top=d_top, idx_next=idx_next)
        if g_clean is not None:
            dl = dl + g_clean.float()
        if g_noisy is not None:
            dl = dl + g_noisy.float()
        D = x2.shape[1]
        wg = w_gate.detach()
        dw = torch.zeros(w_gate.shape, dtype=torch.float32, device=w_gate.device)
        dx = torch.empty(T, D, dtype=torch.float32, device=x2.device)
        if wg.shape[0] == D:
            ops.gate_bwd_params(x2, wg, dl.contiguous(), d_w_gate=dw, dx=dx)
        else:
            dwt = torch.empty(D, E, dtype=torch.float32, device=x2.device)
            ops.gate_bwd_params(x2, wg[:D].contiguous(), dl.contiguous(), d_w_gate=dwt, dx=dx)
            dw[:D] = dwt
        d_bias = dl.sum(0) if ctx.needs_input_grad[5] else None
        return dx.to(x2.dtype).view(-1, D), dw, None, None, None, d_bias


class GroupedLinearFn(torch.autograd.Function):
    """FMoELinear: rows grouped by expert, y_e = x_e W_e^T + b_e (custom_moe_layer.py:32-33)."""

    @staticmethod
    def forward(ctx, inp, weight, bias, offsets, tile_starts):
        x = inp.contiguous()
        dt = x.dtype
        R = x.shape[0]
        y = torch.empty(R, weight.shape[1], dtype=dt, device=x.device)
        ops.gemm_nt(x, _wcopy(weight, dt), y, M=R, bias=bias.detach() if bias is not None else None,
                    group_offsets=offsets, tile_starts=tile_starts)
        ctx.save_for_backward(x, weight, offsets, tile_starts)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, offsets, tile_starts = ctx.saved_tensors
        dt = x.dtype
        g = _as_act(gy, dt)
        R = x.shape[0]
        dx = torch.empty_like(x)
        ops.gemm_nt(g, _wcopy(weight, dt, transpose=True), dx, M=R, group_offsets=offsets, tile_starts=tile_starts)
        dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
        ops.wgrad_tn(g, x, dw, M=R, group_offsets=offsets)
        db = None
        if ctx.has_bias:
            db = torch.empty(weight.shape[0], weight.shape[1], dtype=torch.float32, device=x.device)
            ops.colsum(g, db, M=R, group_offsets=offsets)
        return dx, dw, db, None, None


class GatherRowsFn(torch.autograd.Function):
    """out[i] = src[idx[i] // div]; backward sums the rows that read the same source through the
    inverse index (inv [n_src * kk] lists, for every source row, the kk outputs that read it)."""

    @staticmethod
    def forward(ctx, src, idx, div, inv, kk):
        s = src.contiguous()
        out = torch.empty(idx.numel(), s.shape[1], dtype=s.dtype, device=s.device)
        ops.gather_rows(s, idx, out, div=div, k=1)
        ctx.save_for_backward(inv)
        ctx.kk = kk
        ctx.n_src = s.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        inv, = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty(ctx.n_src, g.shape[1], dtype=g.dtype, device=g.device)
        ops.gather_rows(g, inv, out, div=1, k=ctx.kk)
        return out, None, None, None, None


class CombineFn(torch.autograd.Function):
    """bmm(score[T,1,k], y[T,k,D]) -> [T,D] (custom_moe_layer.py:298-305); fp32 output."""

    @staticmethod
    def forward(ctx, y, score):
        y2 = y.contiguous()
        sc = score.contiguous().float()
        T, k = sc.shape
        out = torch.empty(T, y2.shape[-1], dtype=torch.float32, device=y.device)
        ops.combine_fwd(y2.view(T * k, -1), sc, None, out)
        ctx.save_for_backward(y2, sc)
        return out

    @staticmethod
    def backward(ctx, g):
        y2, sc = ctx.saved_tensors
        T, k = sc.shape
        dy = torch.empty_like(y2)
        ds = torch.empty_like(sc)
        ops.combine_bwd(g.contiguous().float(), y2.view(T * k, -1), sc, dy.view(T * k, -1), ds)
        return dy, ds


class GroupedFFNFn(torch.autograd.Function):
    """The fused MoE MLP of the hot path: route_build -> grouped FC1 (+bias, GELU; row gather fused)
    -> grouped FC2 (+bias; scatter to token-major fused) -> combine with the gate scores.
    Equivalent to _fmoe_general_global_forward(moe_inp, idx, _Expert, E, 1) followed by the bmm
    (custom_moe_layer.py:263-305) for the _Expert of :24-44."""

    @staticmethod
    def forward(ctx, x, idx32, score, w1, b1, w2, b2):
        x2 = x.contiguous()
        dt = x2.dtype
        T, D = x2.shape
        k = idx32.shape[1]
        E, H = w1.shape[0], w1.shape[1]
        R = T * k
        r = ops.route_build(idx32.contiguous(), E)
        pre = torch.empty(R, H, dtype=dt, device=x.device)
        hid = torch.empty_like(pre)
        ops.gemm_nt(x2, _wcopy(w1, dt), hid, M=R, bias=b1.detach(), act=M3_ACT_GELU, pre_out=pre,
                    a_row_idx=r.row_of_slot, a_row_div=k, group_offsets=r.offsets, tile_starts=r.tile_starts)
        y = torch.empty(R, D, dtype=dt, device=x.device)
        ops.gemm_nt(hid, _wcopy(w2, dt), y, M=R, bias=b2.detach(), c_row_idx=r.row_of_slot,
                    group_offsets=r.offsets, tile_starts=r.tile_starts)
        sc = score.contiguous().float()
        out = torch.empty(T, D, dtype=torch.float32, device=x.device)
        ops.combine_fwd(y, sc, None, out)
        ctx.save_for_backward(x2, sc, w1, w2, pre, hid, y, r.row_of_slot, r.offsets, r.tile_starts)
        ctx.k = k
        return out

    @staticmethod
    def backward(ctx, g):
        x2, sc, w1, w2, pre, hid, y, ros, offsets, tile_starts = ctx.saved_tensors
        dt = x2.dtype
        k = ctx.k
        T, D = x2.shape
        R = T * k
        dev = x2.device
        dy = torch.empty_like(y)
        dscore = torch.empty_like(sc)
        ops.combine_bwd(g.contiguous().float(), y, sc, dy, dscore)
        dw2 = torch.empty(w2.shape, dtype=torch.float32, device=dev)
        db2 = torch.empty(w2.shape[0], w2.shape[1], dtype=torch.float32, device=dev)
        ops.wgrad_tn(dy, hid, dw2, M=R, c_row_idx=ros, group_offsets=offsets, db=db2)     # bias grad in the same pass
        dpre = torch.empty_like(pre)
        ops.gemm_nt(dy, _wcopy(w2, dt, transpose=True), dpre, M=R, gelu_grad_pre=pre, a_row_idx=ros, a_row_div=1,
                    group_offsets=offsets, tile_starts=tile_starts)
        dw1 = torch.empty(w1.shape, dtype=torch.float32, device=dev)
        db1 = torch.empty(w1.shape[0], w1.shape[1], dtype=torch.float32, device=dev)
        ops.wgrad_tn(dpre, x2, dw1, M=R, a_row_idx=ros, a_row_div=k, group_offsets=offsets, db=db1)
        dxe = torch.empty(R, D, dtype=dt, device=dev)
        ops.gemm_nt(dpre, _wcopy(w1, dt, transpose=True), dxe, M=R, c_row_idx=ros, group_offsets=offsets,
                    tile_starts=tile_starts)
        dx = torch.empty(T, D, dtype=torch.float32, device=dev)
        ops.combine_fwd(dxe, torch.ones_like(sc), None, dx)
        return dx.to(dt), None, dscore, dw1, db1, dw2, db2
