"""Exact selection on device scores (include/glr.h: glr_kth_value / glr_topk_desc / glr_threshold_counts):
the index / threshold picks of the reference's retrieval ranking (gloria/models/retrival_model.py:118) and
localization metrics (gloria/lightning/callbacks.py:56-61), bit exact (radix select, no tolerance)."""

import torch

from . import _native as N


def _rows(x):
    N.require_cuda(x)
    x = x.detach().float().contiguous()
    return x.view(1, -1) if x.dim() == 1 else x.view(-1, x.shape[-1])


def kth_value(x, k):
    """k-th smallest value (k 1-indexed) along the last dim == torch.topk(x, k, largest=False).values.max()."""
    r = _rows(x)
    out = torch.empty(r.shape[0], dtype=torch.float32, device=r.device)
    N.check(N.lib().glr_kth_value(N.ptr(r), r.shape[0], r.shape[1], int(k), N.ptr(out), N.stream()), "glr_kth_value")
    return out.view(x.shape[:-1])


def topk_desc(x, k):
    """(indices int64, values) of the k largest along the last dim, in the order of np.argsort(x)[::-1][:k]
    (ties: larger index first)."""
    r = _rows(x)
    idx = torch.empty(r.shape[0], int(k), dtype=torch.int64, device=r.device)
    val = torch.empty(r.shape[0], int(k), dtype=torch.float32, device=r.device)
    N.check(N.lib().glr_topk_desc(N.ptr(r), r.shape[0], r.shape[1], int(k), N.ptr(idx), N.ptr(val), N.stream()),
            "glr_topk_desc")
    shape = tuple(x.shape[:-1]) + (int(k),)
    return idx.view(shape), val.view(shape)


def threshold_counts(pred, target, thr):
    """per row: #(pred > thr & target), #(pred > thr), #(target), #(pred > thr | target)  -> int64 [rows, 4]."""
    r = _rows(pred)
    t = target.detach().to(torch.uint8).contiguous().view(r.shape)
    N.require_cuda(t)
    th = thr.detach().float().contiguous().view(-1)
    out = torch.empty(r.shape[0], 4, dtype=torch.int64, device=r.device)
    N.check(N.lib().glr_threshold_counts(N.ptr(r), N.ptr(t), N.ptr(th), r.shape[0], r.shape[1], N.ptr(out), N.stream()),
            "glr_threshold_counts")
    return out


def cell_counts(labels, ih, iw):
    """labels [B, Hl, Wl] (bool / uint8) -> (cnt, npix) int64 [B, ih * iw]: label pixels / pixels of the
    nearest-upsampled Hl x Wl overlay that copy each cell of an ih x iw map (glr_cell_counts)."""
    N.require_cuda(labels)
    lab = labels.detach().to(torch.uint8).contiguous()
    B, Hl, Wl = lab.shape
    cnt = torch.empty(B, ih * iw, dtype=torch.int32, device=lab.device)
    npix = torch.empty(B, ih * iw, dtype=torch.int32, device=lab.device)
    N.check(N.lib().glr_cell_counts(N.ptr(lab), B, Hl, Wl, int(ih), int(iw), N.ptr(cnt), N.ptr(npix), N.stream()),
            "glr_cell_counts")
    return cnt.long(), npix.long()
