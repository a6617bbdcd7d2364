"""Linear layers of the BERT text encoder with the bias gradient taken off aten's generic reduction
(reference: transformers' BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput `nn.Linear`s inside the
BertModel of /root/reference/gloria/models/text_model.py:18-20).

Forward is torch's own `F.linear` (hipBLASLt, bias in the GEMM epilogue) and the two gradient GEMMs are torch's own
`mm`s - plain library GEMMs by design.  What changes is the third output of the backward, `db = sum over tokens of dy`:
autograd runs it as `at::native::reduce_kernel` (48 launches, 2.7 - 2.8 ms of the 256-pair step, 0.7 - 2.7 TB/s on these
[24832, 768 | 2304 | 3072] bf16 matrices); here it is `glr_colsum_bf16` (include/glr.h, HBM-bound), or - for the two
Linears whose output feeds the fused dropout + add + LayerNorm - nothing at all: that backward kernel sums d_h as it
writes it (fused_ln.dense_drop_add_ln).

Only the training configuration takes this path (GPU, bf16 operands AND bf16 parameters = the flat optimizer's shadow
weights); everything else is `F.linear`.  `GLR_FUSED_LINEAR=0` switches it off (A/B measurements)."""

import os

import torch
import torch.nn.functional as F

from .. import _native as N

ENABLED = os.environ.get("GLR_FUSED_LINEAR", "1") != "0"
_WS = {}            # (device index, stream) -> fp32 workspace: launches of one stream only (see fused_bn)
_WS_FLOATS = {}


def _workspace(dev, R, C):
    n = _WS_FLOATS.get((R, C))
    if n is None:
        n = _WS_FLOATS[(R, C)] = int(N.lib().glr_colsum_workspace_floats(R, C))
    if n == 0:
        return None
    if torch.cuda.is_current_stream_capturing():        # a graph keeps its own: the shared one may be replaced later
        return torch.empty(n, dtype=torch.float32, device=dev)
    key = (dev.index, N.stream())
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=dev)
        _WS[key] = ws
    return ws


def colsum(dy2):
    """sum over rows of a contiguous bf16 [R, C] matrix -> bf16 [C] (fp32 accumulation, fixed order)"""
    R, C = dy2.shape
    ws = _workspace(dy2.device, R, C)
    if ws is None:                       # C not a multiple of 256: torch's reduction
        return dy2.sum(0)
    out = torch.empty(C, dtype=torch.bfloat16, device=dy2.device)
    N.check(N.lib().glr_colsum_bf16(N.ptr(dy2), R, C, N.ptr(ws), N.ptr(out), 1, N.stream()), "glr_colsum_bf16")
    return out


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, ext_bias):
        ctx.save_for_backward(x, w)
        ctx.ext_bias = ext_bias
        with torch.autocast("cuda", enabled=False):
            return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = dw = db = None
        with torch.autocast("cuda", enabled=False):
            if ctx.needs_input_grad[0]:
                dx = (dy2 @ w).view(x.shape)
            if ctx.needs_input_grad[1]:
                dw = dy2.t() @ x.reshape(-1, x.shape[-1])
        if ctx.needs_input_grad[2] and not ctx.ext_bias:
            db = colsum(dy2)
        return dx, dw, db, None


def linear_fusable(x, w, b):
    return (ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and b is not None
            and b.dtype == torch.bfloat16 and x.is_contiguous() and torch.is_grad_enabled()
            and (w.requires_grad or x.requires_grad))


def linear(x, w, b, bias_grad_from_epilogue=False):
    """F.linear(x, w, b).  bias_grad_from_epilogue: the caller's next op returns b's gradient (fused_ln) - only legal
    when linear_fusable(x, w, b)."""
    if (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.bfloat16 and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        x = x.to(torch.bfloat16)         # the cast autocast would put in front of F.linear (first layer: fp32 embeddings)
    if linear_fusable(x, w, b):
        return _Linear.apply(x, w, b, bool(bias_grad_from_epilogue))
    if bias_grad_from_epilogue:
        raise RuntimeError("linear: bias_grad_from_epilogue on the unfused path")
    return F.linear(x, w, b)
