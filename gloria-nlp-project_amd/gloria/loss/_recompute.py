"""Differentiable GPU restatement of the local attention similarity, used ONLY by the backward
pass of the fused HIP forward kernel (recompute-in-backward, image chunk by image chunk, so the
forward never keeps per-sentence copies of the image features like the reference does at
gloria_loss.py:35).

INTERIM (round 1): the gradient maths is expressed with torch ops on the GPU (library GEMMs +
elementwise kernels).  The forward values of the training step always come from the HIP kernel;
this module only differentiates the same function.  It is slated to be replaced by the
hand-written backward kernel described in DESIGN.md ("K1 backward").  It refuses CPU tensors.
"""

import math

import torch


def _segment_softmax(x, seg, nseg):
    """softmax over runs of equal `seg` id along the last axis.  x [..., N], seg [N] (long)."""
    idx = seg.expand(x.shape)
    shape = x.shape[:-1] + (nseg,)
    mx = torch.full(shape, -math.inf, dtype=x.dtype, device=x.device).scatter_reduce(-1, idx, x.detach(), "amax")
    e = torch.exp(x - mx.gather(-1, idx))
    sm = torch.zeros(shape, dtype=x.dtype, device=x.device).scatter_add(-1, idx, e)
    return e / sm.gather(-1, idx)


def local_sim_packed(V, T, seg, cap_lens_f, temp1, temp2, temp3, agg, eps):
    """V [c, D, S_eff] region features (no-attention column already prepended), T [N, D] packed
    words, seg [N] sentence id of every packed word.  Returns sim [c, n_sent], a2 [c, N, S_eff].
    Same maths as /root/reference/gloria/loss/gloria_loss.py:40-59, 150-164."""
    if not V.is_cuda:
        raise RuntimeError("recompute path is GPU-only")
    nsent = cap_lens_f.shape[0]
    s = torch.einsum("cdr,nd->crn", V, T)
    a1 = _segment_softmax(s, seg, nsent)
    a2 = torch.softmax(a1 * temp1, dim=1)
    ctx = torch.einsum("cdr,crn->cnd", V, a2)
    dot = (ctx * T.unsqueeze(0)).sum(-1)
    den = (ctx.norm(2, dim=-1) * T.norm(2, dim=-1).unsqueeze(0)).clamp(min=eps)
    ex = torch.exp(dot / den * temp2)
    if agg == "max":
        acc = torch.zeros(ex.shape[0], nsent, dtype=ex.dtype, device=ex.device).scatter_reduce(
            1, seg.expand(ex.shape), ex, "amax", include_self=False)
    else:
        acc = torch.zeros(ex.shape[0], nsent, dtype=ex.dtype, device=ex.device).scatter_add(
            1, seg.expand(ex.shape), ex)
        if agg == "mean":
            acc = acc / cap_lens_f
    return torch.log(acc) * temp3, a2.transpose(1, 2)


def diag_attention(V, words, cap_lens, word_start, temp1):
    """Attention maps of the diagonal pairs only: V [B, D, S_eff], words [B, D, L] -> a2 [B, Lmax, S_eff]
    (rows >= cap_lens[b] are garbage and must be ignored by the caller)."""
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
