"""Gradients of BertEmbeddings' lookup tables without torch's sort-and-scatter (include/glr.h: glr_embedding_bwd /
glr_type_embedding_bwd; reference: transformers' BertEmbeddings inside the BertModel of
/root/reference/gloria/models/text_model.py:18-20).

`embedding_dense_backward` sorts the token ids on the device and runs ~17 launches per table; its `sum_and_scatter` takes
0.4 - 0.6 ms per table at 24 832 tokens (one id - [PAD], or the single token type - owns most rows): 1.4 ms and 35 launches
of the 256-pair step for two tables.  The forward stays torch's gather.  Word table: the trainer already keeps the caption
ids on the host (`ids._glr_host`, for the word-piece slotting), so the segments come from a stable numpy argsort and one
launch sums every token's rows in that fixed order (the padding id's rows are left out, like `padding_idx` does).
Token-type table (two rows): two masked column sums.  Everything else (CPU tensors, other dtypes, ids without a host copy,
`GLR_FUSED_EMBED=0`) is `F.embedding`."""

import os

import numpy as np
import torch
import torch.nn.functional as F

from .. import _native as N

ENABLED = os.environ.get("GLR_FUSED_EMBED", "1") != "0"


def embedding_plan(host_ids, padding_idx, device):
    """host ids -> (order, seg_lo, seg_hi, seg_tok, n_seg) on the device: rows of `order[lo:hi]` carry token `tok`"""
    flat = np.ascontiguousarray(host_ids).reshape(-1)
    order = np.argsort(flat, kind="stable")
    s = flat[order]
    starts = np.flatnonzero(np.r_[True, s[1:] != s[:-1]])
    ends = np.r_[starts[1:], len(flat)]
    toks = s[starts]
    if padding_idx is not None:
        keep = toks != padding_idx
        starts, ends, toks = starts[keep], ends[keep], toks[keep]
    n, m = len(flat), len(starts)
    dev = N.upload(np.concatenate([order, starts, ends, toks]).astype(np.int32), device)
    return dev[:n], dev[n:n + m], dev[n + m:n + 2 * m], dev[n + 2 * m:], m


class _WordEmbedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, weight, plan):
        ctx.plan, ctx.wshape = plan, weight.shape
        return F.embedding(ids, weight)

    @staticmethod
    def backward(ctx, dy):
        order, lo, hi, tok, m = ctx.plan
        V, D = ctx.wshape
        dy2 = dy.float().contiguous().view(-1, D)
        dw = torch.zeros(V, D, dtype=torch.float32, device=dy.device)
        N.check(N.lib().glr_embedding_bwd(N.ptr(dy2), N.ptr(order), N.ptr(lo), N.ptr(hi), N.ptr(tok), m, D, N.ptr(dw), N.stream()),
                "glr_embedding_bwd")
        return None, dw, None


class _TypeEmbedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tt, weight):
        ctx.save_for_backward(tt)
        ctx.D = weight.shape[1]
        return F.embedding(tt, weight)

    @staticmethod
    def backward(ctx, dy):
        (tt,) = ctx.saved_tensors
        D = ctx.D
        dy2 = dy.float().contiguous().view(-1, D)
        R = dy2.shape[0]
        L = N.lib()
        ws = torch.empty(L.glr_type_embedding_workspace_floats(R, D), dtype=torch.float32, device=dy.device)
        dw = torch.empty(2, D, dtype=torch.float32, device=dy.device)
        ttc = tt.contiguous().view(-1)
        N.check(L.glr_type_embedding_bwd(N.ptr(dy2), N.ptr(ttc), R, D, N.ptr(ws), N.ptr(dw), N.stream()), "glr_type_embedding_bwd")
        return None, dw


def _common(idx, weight):
    return (ENABLED and idx.is_cuda and idx.dtype == torch.int64 and weight.dtype == torch.float32 and weight.requires_grad
            and torch.is_grad_enabled())


def word_embedding(module, ids):
    """module(ids) for an nn.Embedding, its weight gradient through glr_embedding_bwd when `ids` carries its host copy"""
    host = getattr(ids, "_glr_host", None)
    w = module.weight
    D = w.shape[1]
    if (_common(ids, w) and host is not None and tuple(np.shape(host)) == tuple(ids.shape) and D % 4 == 0 and D <= 1024
            and module.max_norm is None and not module.scale_grad_by_freq and not module.sparse):
        return _WordEmbedding.apply(ids, w, embedding_plan(host, module.padding_idx, ids.device))
    return module(ids)


def type_embedding(module, token_type):
    w = module.weight
    if (_common(token_type, w) and w.shape[0] == 2 and w.shape[1] % 256 == 0 and module.padding_idx is None
            and module.max_norm is None and not module.scale_grad_by_freq and not module.sparse):
        return _TypeEmbedding.apply(token_type, w)
    return module(token_type)
