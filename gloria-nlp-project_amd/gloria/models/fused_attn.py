"""BertSelfAttention's softmax(Q K^T / sqrt(d) + mask) -> dropout -> . V for short captions as ONE HIP kernel per direction
(include/glr.h: glr_attn_fwd / glr_attn_bwd; reference: transformers' BertSelfAttention inside the BertModel of
/root/reference/gloria/models/text_model.py:18-20).

`self_attention(q, k, v, key_mask, n_heads, p, training)` takes the [B, L, H] outputs of the query / key / value Linears
as they are and returns the context in the same layout.  The kernels cover what the training step runs: GPU, bf16
(autocast), head size 64, L <= 128 tokens; every other case is torch's scaled_dot_product_attention on the same
tensors.  Dropout bits: a counter-based hash of the score's position (two rounds of a 32-bit multiply-xorshift mixer,
16 bits per score: csrc/glr_attn.hip) KEYED by the CUDA generator's (seed, offset) drawn the same way the sub-layer
epilogues draw their Philox keys (fused_ln._philox_args) - reproducible under torch.manual_seed, not the same stream as
torch's own dropout.  The epilogues use Philox4x32-10 itself; here it cost 40 quarter-rate integer multiplies per 8
scores in a kernel that is VALU-bound, so the per-score draw is the cheaper mixer and only the KEY comes from Philox's
counter.  `GLR_FUSED_ATTN=0` switches the kernels off."""

import math
import os

import torch
import torch.nn.functional as F

from .. import _native as N
from .rng import philox_args

ENABLED = os.environ.get("GLR_FUSED_ATTN", "1") != "0"
_MAX_L = None


def _max_tokens():
    global _MAX_L
    if _MAX_L is None:
        _MAX_L = int(N.lib().glr_attn_max_tokens(1))
    return _MAX_L


class _SelfAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, key_mask, nh, p, scale):
        L_ = N.lib()
        B, L, H = q.shape
        dev = q.device
        o = torch.empty_like(q)
        lse = torch.empty(B * nh, 128, dtype=torch.float32, device=dev)
        keep = torch.empty(B * nh, 128, 4, dtype=torch.int32, device=dev) if p > 0 else None
        seed, off, cell = philox_args(dev) if p > 0 else (0, 0, None)
        N.check(L_.glr_attn_fwd(N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(key_mask), B, nh, L, H, H, float(scale), float(p), seed, off,
                                cell, N.ptr(o), N.ptr(lse), N.ptr(keep), N.stream()), "glr_attn_fwd")
        ctx.save_for_backward(q, k, v, o, lse, keep, key_mask)
        ctx.meta = (nh, float(p), float(scale))
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse, keep, key_mask = ctx.saved_tensors
        nh, p, scale = ctx.meta
        L_ = N.lib()
        B, L, H = q.shape
        d_o = d_o.to(torch.bfloat16).contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        N.check(L_.glr_attn_bwd(N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(o), N.ptr(d_o), N.ptr(key_mask), N.ptr(lse), N.ptr(keep),
                                B, nh, L, H, H, scale, p, N.ptr(dq), N.ptr(dk), N.ptr(dv), N.stream()), "glr_attn_bwd")
        return dq, dk, dv, None, None, None, None


class _SelfAttnPacked(torch.autograd.Function):
    """the same kernels on ONE [B, L, 3H] tensor (query | key | value columns): the output of a single fused Linear.
    The backward writes dq | dk | dv straight into the gradient of that tensor (row stride 3H on both sides)."""

    @staticmethod
    def forward(ctx, qkv, key_mask, nh, p, scale):
        L_ = N.lib()
        B, L, H3 = qkv.shape
        H = H3 // 3
        dev = qkv.device
        o = torch.empty(B, L, H, dtype=qkv.dtype, device=dev)
        lse = torch.empty(B * nh, 128, dtype=torch.float32, device=dev)
        keep = torch.empty(B * nh, 128, 4, dtype=torch.int32, device=dev) if p > 0 else None
        seed, off, cell = philox_args(dev) if p > 0 else (0, 0, None)
        base = qkv.data_ptr()
        N.check(L_.glr_attn_fwd(N.c_void_p(base), N.c_void_p(base + 2 * H), N.c_void_p(base + 4 * H), N.ptr(key_mask), B, nh, L, H3, H,
                                float(scale), float(p), seed, off, cell, N.ptr(o), N.ptr(lse), N.ptr(keep), N.stream()),
                "glr_attn_fwd")
        ctx.save_for_backward(qkv, o, lse, keep, key_mask)
        ctx.meta = (nh, float(p), float(scale))
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o, lse, keep, key_mask = ctx.saved_tensors
        nh, p, scale = ctx.meta
        L_ = N.lib()
        B, L, H3 = qkv.shape
        H = H3 // 3
        d_o = d_o.to(torch.bfloat16).contiguous()
        dqkv = torch.empty_like(qkv)
        base, dbase = qkv.data_ptr(), dqkv.data_ptr()
        N.check(L_.glr_attn_bwd(N.c_void_p(base), N.c_void_p(base + 2 * H), N.c_void_p(base + 4 * H), N.ptr(o), N.ptr(d_o),
                                N.ptr(key_mask), N.ptr(lse), N.ptr(keep), B, nh, L, H3, H, scale, p, N.c_void_p(dbase),
                                N.c_void_p(dbase + 2 * H), N.c_void_p(dbase + 4 * H), N.stream()), "glr_attn_bwd")
        return dqkv, None, None, None, None


def _fusable(q, k, v, key_mask, nh):
    B, L, H = q.shape
    return (ENABLED and q.is_cuda and q.dtype == k.dtype == v.dtype == torch.bfloat16 and H == nh * 64 and L <= _max_tokens()
            and q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
            and (key_mask is None or (key_mask.dtype == torch.bool and tuple(key_mask.shape) == (B, L) and key_mask.is_contiguous())))


def packed_fusable(x, nh, hidden, L, key_mask):
    """whether BertSelfAttention may run ONE query|key|value Linear and the packed kernels on input x"""
    return (ENABLED and x.is_cuda and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and hidden == nh * 64 and L <= _max_tokens()
            and (key_mask is None or (key_mask.dtype == torch.bool and key_mask.is_contiguous())))


def self_attention_packed(qkv, key_mask, nh, p, training):
    """qkv: bf16 [B, L, 3H] (query | key | value columns) -> context [B, L, H]"""
    return _SelfAttnPacked.apply(qkv.contiguous(), key_mask, nh, p if training else 0.0, 1.0 / 8.0)


def self_attention(q, k, v, key_mask, nh, p, training):
    """q, k, v: [B, L, H]; key_mask: bool [B, L] (True = attend) or None -> context [B, L, H]."""
    B, L, H = q.shape
    hd = H // nh
    if _fusable(q, k, v, key_mask, nh):
        return _SelfAttn.apply(q, k, v, key_mask, nh, p if training else 0.0, 1.0 / math.sqrt(hd))
    bias = None if key_mask is None else key_mask[:, None, None, :]
    qh, kh, vh = (t.view(B, L, nh, hd).transpose(1, 2) for t in (q, k, v))
    o = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=bias, dropout_p=p if training else 0.0)
    return o.transpose(1, 2).reshape(B, L, H)
