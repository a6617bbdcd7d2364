"""Differentiable GPU restatement of the attention maps of the B DIAGONAL pairs, used only by the backward of
`attention_fn`'s own outputs (weighted context + maps, K1 in pair mode): B small pairs, differentiated by torch
autograd on the GPU.  Everything else - local_loss, the attention-supervision loss, the regularisers - goes through
the hand-written K1 backward kernel.  It refuses CPU tensors (no CPU fallback of the product path)."""

import math

import torch


def diag_attention(V, words, cap_lens, word_start, temp1):
    """Attention maps of the diagonal pairs only: V [B, D, S_eff], words [B, D, L] -> a2 [B, Lmax, S_eff]
    (rows >= cap_lens[b] are garbage and must be ignored by the caller)."""
    if not V.is_cuda:
        raise RuntimeError("recompute path is GPU-only")
    B, D, _ = V.shape
    lmax = int(max(cap_lens))
    w = words[:, :, word_start:word_start + lmax]
    if w.shape[2] < lmax:
        w = torch.nn.functional.pad(w, (0, lmax - w.shape[2]))
    s = torch.einsum("bdr,bdw->brw", V, w)
    lens = torch.as_tensor(cap_lens, device=V.device).view(B, 1, 1)
    mask = torch.arange(lmax, device=V.device).view(1, 1, lmax) < lens
    a1 = torch.softmax(s.masked_fill(~mask, -math.inf), dim=2)
    a2 = torch.softmax(a1 * temp1, dim=1)               # over regions
    return a2.transpose(1, 2)                            # [B, Lmax, S_eff]
