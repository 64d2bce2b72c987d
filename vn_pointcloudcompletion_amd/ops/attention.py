"""Batched GEMM, head permutes, VN attention and VNLayerNorm (/root/reference/models/transformer.py:91-104,
models/vn_layers.py:129-150)."""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from .._lib import call, ptr
from ._state import HK, SW, POOL_PATH_COUNTS, prof_arm, trace
from .core import (BN_EPS, BN_MOMENTUM, _bn_leaky_backward, _bn_leaky_forward, _bn_statistics, _chunk_pc, _dgrad, _ld, _nchunk,
                   _per_sample_colsum, _rows, _wgrad, cat_rows, gemm, gemm_pd_stats, reduce_pairs, reduce_rows)


# ------------------------------------------------------------------------------- transformer ops (configs 4, 5)
def gemm_batched(A, B, *, op_a=0, op_b=0, out=None):
    """A, B: 3-D contiguous [nbatch, ., .]; C_i = op(A_i) op(B_i)."""
    _lib.require_gpu(A, B)
    A, B = A.contiguous(), B.contiguous()
    nb = A.shape[0]
    M, K = (A.shape[1], A.shape[2]) if op_a == 0 else (A.shape[2], A.shape[1])
    N = B.shape[1] if op_b == 0 else B.shape[2]
    if out is None:
        out = torch.empty((nb, M, N), device=A.device, dtype=torch.float32)
    call("vnp_gemm_batched", op_a, op_b, M, N, K, ptr(A), A.shape[2], A.shape[1] * A.shape[2], ptr(B), B.shape[2],
         B.shape[1] * B.shape[2], ptr(out), N, M * N, nb, 0)
    return out


def _to_heads(rows, B, N, H):
    C = rows.shape[1]
    heads = torch.empty((B * H, N, (C // H) * 3), device=rows.device, dtype=torch.float32)
    call("vnp_rows_to_heads", ptr(rows), _ld(rows), B, N, C, H, ptr(heads))
    return heads


def _from_heads(heads, B, N, C, H, out=None):
    """heads [B*H, N, (C/H)*3] -> rows [B*N*3, C]; ``out``: a [B*N*3, C] view (any row stride) to fill instead of a new tensor."""
    rows = torch.empty((B * N * 3, C), device=heads.device, dtype=torch.float32) if out is None else out
    call("vnp_heads_to_rows", ptr(heads), B, N, C, H, ptr(rows), _ld(rows))
    return rows






def _attn_token_major(B, N, C, H):
    """The rounds 1-2 layout (head dimension on N / K, bounds-checked kernels where it is not a multiple of 128 / 32) for small attention
    problems — VN-PoinTr's 128 tokens: 1.2 GFLOP per product, measured 29.1 (this layout) vs 30.2 ms per step — and the padded
    token-major / feature-major layout for big ones (Attention_VN_FoldingNet's 1024 tokens: 39 GFLOP per product, 32.8 -> 31.8 ms)."""
    return SW.ATTN_TOKEN_MAJOR or 2.0 * B * H * N * N * (C // H) * 3 < SW.ATTN_FM_MIN_FLOP


def _attn_dims(C, H):
    """(Dh, ldh, fr): the head dimension 3 C / H, its K padding (multiple of 32) and its M padding (whole 96-row tiles)."""
    Dh = (C // H) * 3
    return Dh, (Dh + 31) // 32 * 32, (Dh + 95) // 96 * 96


def _heads_tm(rows, B, N, H, ldh):
    """rows [B*N*3, C] (any row stride) -> token-major heads [B*H, N, ldh], columns beyond 3 C / H zero."""
    C = rows.shape[1]
    heads = torch.empty((B * H, N, ldh), device=rows.device, dtype=torch.float32)
    call("vnp_rows_to_heads_pad", ptr(rows), _ld(rows), B, N, C, H, ptr(heads), ldh)
    return heads


def _heads_fm(rows, B, N, H, fr):
    """rows [B*N*3, C] (any row stride) -> feature-major heads [B*H, fr, N], rows beyond 3 C / H zero."""
    C = rows.shape[1]
    heads = torch.empty((B * H, fr, N), device=rows.device, dtype=torch.float32)
    call("vnp_rows_to_heads_fm", ptr(rows), _ld(rows), B, N, C, H, ptr(heads), fr)
    return heads


def _rows_from_fm(hf, B, N, C, H, out=None):
    rows = torch.empty((B * N * 3, C), device=hf.device, dtype=torch.float32) if out is None else out
    call("vnp_heads_fm_to_rows", ptr(hf), hf.shape[1], B, N, C, H, ptr(rows), _ld(rows))
    return rows


def _attention_forward(q, k, v, B, N, H, scale):
    """softmax(q k^T scale) v per (sample, head); q, k, v rows [B*N*3, C] (column slices allowed).  Returns (out rows, P [B*H, N, N]).
    The head dimension Dh = 3 C / H (144 in Attention_VN_FoldingNet, 192 in VN-PoinTr) is zero-padded to a multiple of 32 where it is
    the K of a product (q k^T) and laid out feature-major, padded to whole row tiles, where it would have been the N (out^T = v^T P^T):
    every product then runs on the aligned kernels (round 3; before: the bounds-checked ones, 5.4 ms of config 4's step at 70-100 TF/s)."""
    C = q.shape[1]
    _, ldh, fr = _attn_dims(C, H)
    P = gemm_batched(_heads_tm(q, B, N, H, ldh), _heads_tm(k, B, N, H, ldh))      # [B*H, N(q), N(k)] scores
    call("vnp_softmax_fwd", ptr(P), B * H * N, N, float(scale))
    o_fm = gemm_batched(_heads_fm(v, B, N, H, fr), P)                               # [B*H, fr, N(q)] = v^T P^T
    return _rows_from_fm(o_fm, B, N, C, H), P


def _attention_backward(go, q, k, v, P, B, N, H, scale, outs):
    """Fills outs = (dq, dk, dv) rows views [B*N*3, C] (any row stride).  P is consumed (it stays P; dP becomes dS in place)."""
    C = q.shape[1]
    _, ldh, fr = _attn_dims(C, H)
    dP = gemm_batched(_heads_tm(go, B, N, H, ldh), _heads_tm(v, B, N, H, ldh))    # dout v^T: [B*H, N(q), N(k)]
    dv_fm = gemm_batched(_heads_fm(go, B, N, H, fr), P, op_b=1)                     # dout^T P: [B*H, fr, N(k)]
    _rows_from_fm(dv_fm, B, N, C, H, out=outs[2])
    del dv_fm
    call("vnp_softmax_bwd", ptr(P), ptr(dP), B * H * N, N, scale)                  # dP <- dS
    dq_fm = gemm_batched(_heads_fm(k, B, N, H, fr), dP)                             # k^T dS^T: [B*H, fr, N(q)]
    _rows_from_fm(dq_fm, B, N, C, H, out=outs[0])
    del dq_fm
    dk_fm = gemm_batched(_heads_fm(q, B, N, H, fr), dP, op_b=1)                     # q^T dS: [B*H, fr, N(k)]
    _rows_from_fm(dk_fm, B, N, C, H, out=outs[1])


class VNAttentionFn(Function):
    """softmax(q k^T * scale) v over tokens whose features are the flattened (C/H, 3) vectors of each head
    (models/transformer.py:91-104).  q, k, v: rows [B*N*3, C].  The matrix products are batched vnp_gemm GEMMs (fp32 results; split-bf16 or
    fp32-input MFMA, see gemm.hip) on padded token-major / feature-major head tensors (_attention_forward)."""

    @staticmethod
    def forward(ctx, q, k, v, B, N, H, scale):
        q, k, v = _rows(q), _rows(k), _rows(v)
        C = q.shape[1]
        ctx.cfg = (B, N, C, H, float(scale))
        ctx.token_major = _attn_token_major(B, N, C, H)
        if ctx.token_major:
            qh, kh, vh = _to_heads(q, B, N, H), _to_heads(k, B, N, H), _to_heads(v, B, N, H)
            P = gemm_batched(qh, kh)                                      # [B*H, N, N] scores
            call("vnp_softmax_fwd", ptr(P), B * H * N, N, float(scale))
            oh = gemm_batched(P, vh, op_b=1)                               # [B*H, N, Dh]
            ctx.save_for_backward(qh, kh, vh, P)
            return _from_heads(oh, B, N, C, H)
        out, P = _attention_forward(q, k, v, B, N, H, scale)
        ctx.save_for_backward(q, k, v, P)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        B, N, C, H, scale = ctx.cfg
        if ctx.token_major:
            qh, kh, vh, P = ctx.saved_tensors
            goh = _to_heads(_rows(go), B, N, H)
            dP = gemm_batched(goh, vh)                                     # [B*H, N, N]
            dvh = gemm_batched(P, goh, op_a=1, op_b=1)                     # P^T goh
            call("vnp_softmax_bwd", ptr(P), ptr(dP), B * H * N, N, scale)  # dP <- dS
            dqh = gemm_batched(dP, kh, op_b=1)
            dkh = gemm_batched(dP, qh, op_a=1, op_b=1)
            return (_from_heads(dqh, B, N, C, H), _from_heads(dkh, B, N, C, H), _from_heads(dvh, B, N, C, H),
                    None, None, None, None)
        q, k, v, P = ctx.saved_tensors
        outs = tuple(torch.empty((B * N * 3, C), device=go.device, dtype=torch.float32) for _ in range(3))
        _attention_backward(_rows(go), q, k, v, P, B, N, H, scale, outs)
        return outs + (None, None, None, None)


class VNAttentionPackedFn(Function):
    """VNAttentionFn on q | k | v side by side in one matrix qkv [B*N*3, 3C] — the output of ONE projection GEMM over [Wq; Wk; Wv] instead
    of three (a third of the launches, three times the tiles per launch: the 12288-row projections of VN-PoinTr's blocks fill a quarter
    of the chip each) — with the gradient returned as one [B*N*3, 3C] matrix whose column thirds are filled in place."""

    @staticmethod
    def forward(ctx, qkv, B, N, H, scale):
        qkv = _rows(qkv)
        C = qkv.shape[1] // 3
        ctx.cfg = (B, N, C, H, float(scale))
        ctx.token_major = _attn_token_major(B, N, C, H)
        if ctx.token_major:
            if 3 * C <= 5000:                                             # q | k | v -> three head tensors in one launch
                qkvh = torch.empty((3, B * H, N, (C // H) * 3), device=qkv.device, dtype=torch.float32)
                call("vnp_rows_to_heads_grouped", ptr(qkv), _ld(qkv), B, N, C, H, 3, ptr(qkvh))
                qh, kh, vh = qkvh[0], qkvh[1], qkvh[2]
            else:
                qh, kh, vh = (_to_heads(qkv[:, i * C:(i + 1) * C], B, N, H) for i in range(3))
            P = gemm_batched(qh, kh)                                      # [B*H, N, N] scores
            call("vnp_softmax_fwd", ptr(P), B * H * N, N, float(scale))
            oh = gemm_batched(P, vh, op_b=1)                               # [B*H, N, Dh]
            ctx.save_for_backward(qh, kh, vh, P)
            return _from_heads(oh, B, N, C, H)
        out, P = _attention_forward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, N, H, scale)
        ctx.save_for_backward(qkv, P)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        B, N, C, H, scale = ctx.cfg
        g = torch.empty((B * N * 3, 3 * C), device=go.device, dtype=torch.float32)
        if ctx.token_major:
            qh, kh, vh, P = ctx.saved_tensors
            goh = _to_heads(_rows(go), B, N, H)
            dP = gemm_batched(goh, vh)                                     # [B*H, N, N]
            if 3 * C <= 5000:                                             # dq | dk | dv land in one [3, B*H, N, Dh] buffer: one permute back
                dh = torch.empty((3, B * H, N, (C // H) * 3), device=go.device, dtype=torch.float32)
                dvh = gemm_batched(P, goh, op_a=1, op_b=1, out=dh[2])      # P^T goh
                call("vnp_softmax_bwd", ptr(P), ptr(dP), B * H * N, N, scale)  # dP <- dS
                gemm_batched(dP, kh, op_b=1, out=dh[0])
                gemm_batched(dP, qh, op_a=1, op_b=1, out=dh[1])
                call("vnp_heads_to_rows_grouped", ptr(dh), B, N, C, H, 3, ptr(g), _ld(g))
                return g, None, None, None, None
            dvh = gemm_batched(P, goh, op_a=1, op_b=1)                     # P^T goh
            call("vnp_softmax_bwd", ptr(P), ptr(dP), B * H * N, N, scale)  # dP <- dS
            dqh = gemm_batched(dP, kh, op_b=1)
            dkh = gemm_batched(dP, qh, op_a=1, op_b=1)
            for i, h in enumerate((dqh, dkh, dvh)):
                _from_heads(h, B, N, C, H, out=g[:, i * C:(i + 1) * C])
            return g, None, None, None, None
        qkv, P = ctx.saved_tensors
        _attention_backward(_rows(go), qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], P, B, N, H, scale,
                            (g[:, :C], g[:, C:2 * C], g[:, 2 * C:]))
        return g, None, None, None, None


class VNLayerNormFn(Function):
    """VNLayerNorm (models/vn_layers.py:129-150) on the rows layout."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        x = _rows(x)
        R, C = x.shape
        P = R // 3
        out = torch.empty((R, C), device=x.device, dtype=torch.float32)
        stats = torch.empty((P, 2), device=x.device, dtype=torch.float32)
        call("vnp_vn_layernorm_fwd", ptr(x), _ld(x), P, C, ptr(gamma), ptr(beta), ptr(out), C, ptr(stats))
        ctx.save_for_backward(x, gamma, beta, stats)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        x, gamma, beta, stats = ctx.saved_tensors
        go = _rows(go)
        R, C = x.shape
        P = R // 3
        nblk = (P + 3) // 4
        nblk_pad = (nblk + 255) // 256 * 256 if nblk > 1024 else nblk
        gx = torch.empty((R, C), device=x.device, dtype=torch.float32)
        part = (torch.zeros if nblk_pad != nblk else torch.empty)((nblk_pad, 2 * C), device=x.device, dtype=torch.float32)
        call("vnp_vn_layernorm_bwd", ptr(go), _ld(go), ptr(x), _ld(x), P, C, ptr(gamma), ptr(beta), ptr(stats), ptr(gx), C, ptr(part))
        gb = reduce_rows(part, 1, nblk_pad, 2 * C)[0]
        return gx, gb[:C].contiguous(), gb[C:].contiguous()


class AddLayerNormFn(Function):
    """(s, y) with s = x + a1 (+ a2) and y = VNLayerNorm(s) — the head of a transformer block: its input is the previous block's
    residual stream plus that block's MLP branch plus the position embedding (models/pointr/vn_pointr.py:700, 401-410), it is
    normalised for the attention AND carried on as the skip connection.  One pass forward (vnp_vn_add_layernorm_fwd) instead of two adds
    and a LayerNorm; backward, the gradient of the skip connection (the cotangent of s) is added inside the LayerNorm's backward pass
    (vnp_vn_layernorm_bwd_add) instead of by autograd's accumulation, and goes to x, a1 and a2 unchanged.  ``a1 = a2 = None``: s IS x
    (returned as it came): only the backward fusion."""

    @staticmethod
    def forward(ctx, x, a1, a2, gamma, beta):
        x = _rows(x)
        R, C = x.shape
        P = R // 3
        out = torch.empty((R, C), device=x.device, dtype=torch.float32)
        stats = torch.empty((P, 2), device=x.device, dtype=torch.float32)
        if a1 is None and a2 is not None:
            a1, a2 = a2, None
        if a1 is None:
            s = x
            call("vnp_vn_layernorm_fwd", ptr(x), _ld(x), P, C, ptr(gamma), ptr(beta), ptr(out), C, ptr(stats))
        else:
            a1 = _rows(a1)
            a2 = _rows(a2) if a2 is not None else None
            if a1.shape != x.shape or (a2 is not None and a2.shape != x.shape):
                raise _lib.VnpError("AddLayerNormFn: addends must have the shape of x")
            s = torch.empty((R, C), device=x.device, dtype=torch.float32)
            call("vnp_vn_add_layernorm_fwd", ptr(x), _ld(x), ptr(a1), _ld(a1), ptr(a2), _ld(a2) if a2 is not None else 0, P, C,
                 ptr(gamma), ptr(beta), ptr(s), C, ptr(out), C, ptr(stats))
        ctx.save_for_backward(s, gamma, beta, stats)
        ctx.set_materialize_grads(False)                 # an unused output arrives as None, not as a zero-filled [R, C] tensor
        return s, out

    @staticmethod
    @once_differentiable
    def backward(ctx, gs, go):
        s, gamma, beta, stats = ctx.saved_tensors
        R, C = s.shape
        P = R // 3
        need = ctx.needs_input_grad
        if go is None:                                   # the normalised output was not used: only the skip connection carries a gradient
            return tuple(gs if n else None for n in need[:3]) + (None, None)
        go = _rows(go)
        nblk = (P + 3) // 4
        nblk_pad = (nblk + 255) // 256 * 256 if nblk > 1024 else nblk
        gx = torch.empty((R, C), device=s.device, dtype=torch.float32)
        part = (torch.zeros if nblk_pad != nblk else torch.empty)((nblk_pad, 2 * C), device=s.device, dtype=torch.float32)
        if gs is None:
            call("vnp_vn_layernorm_bwd", ptr(go), _ld(go), ptr(s), _ld(s), P, C, ptr(gamma), ptr(beta), ptr(stats), ptr(gx), C, ptr(part))
        else:
            gs = _rows(gs)
            call("vnp_vn_layernorm_bwd_add", ptr(go), _ld(go), ptr(s), _ld(s), P, C, ptr(gamma), ptr(beta), ptr(stats), ptr(gs), _ld(gs),
                 ptr(gx), C, ptr(part))
        gb = reduce_rows(part, 1, nblk_pad, 2 * C)[0]
        return (gx if need[0] else None), (gx if need[1] else None), (gx if need[2] else None), gb[:C], gb[C:]
