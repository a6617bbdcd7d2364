"""y = LayerNorm(dropout(h) + inp) of the BERT sub-layer outputs as ONE HBM pass per direction
(include/glr.h: glr_drop_add_ln_fwd / _bwd; reference: transformers' BertSelfOutput / BertOutput inside the BertModel
of /root/reference/gloria/models/text_model.py:18-20 under native AMP).

`drop_add_ln(h, inp, ln, p, training)` returns (y fp32, y bf16 or None).  The kernels cover what the training step
runs: GPU, bf16 autocast (h bf16 from the dense Linear, fp32 residual stream, fp32 LayerNorm parameters), hidden sizes
256 / 512 / 768 / 1024.  Every other case (fp32 parity mode, CPU tensors of the host-logic tests) is torch's own
dropout + add + layer_norm - the same operators from the library - and returns (y, None).  The second output is y
rounded to bf16, the operand of the next Linear (autocast would make that copy itself, in a pass of its own).
Dropout bits: Philox4x32-10 keyed by the CUDA generator's seed and offset (the offset is advanced), so runs are
reproducible under torch.manual_seed and resume with a restored RNG state; not the same stream as torch's own dropout.
`GLR_FUSED_LN=0` switches the kernels off (A/B measurements)."""

import os

import torch
import torch.nn.functional as F

from .. import _native as N
from .rng import philox_args

ENABLED = os.environ.get("GLR_FUSED_LN", "1") != "0"
_WS = {}            # (device index, stream) -> fp32 workspace: launches of one stream only (see fused_bn)


_WS_FLOATS = {}


def _workspace(dev, R, H):
    n = _WS_FLOATS.get((R, H))
    if n is None:
        n = _WS_FLOATS[(R, H)] = int(N.lib().glr_ln_workspace_floats(R, H))
    if torch.cuda.is_current_stream_capturing():        # a graph keeps its own: the shared one may be replaced later
        return torch.empty(n, dtype=torch.float32, device=dev)
    key = (dev.index, N.stream())
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(max(n, 1 << 21), dtype=torch.float32, device=dev)
        _WS[key] = ws
    return ws


class _DropAddLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, inp, weight, bias, eps, p, h_bias=None):
        # h_bias: the bias of the Linear that produced h, handed in only to RECEIVE its gradient (column sums of d_h,
        # accumulated by the backward kernel on its way) - see fused_linear.linear(..., bias_grad_from_epilogue=True)
        L = N.lib()
        H = h.shape[-1]
        R = h.numel() // H
        dev = h.device
        out32 = torch.empty(h.shape, dtype=torch.float32, device=dev)
        out16 = torch.empty(h.shape, dtype=torch.bfloat16, device=dev)
        stats = torch.empty(R, 2, dtype=torch.float32, device=dev)
        mask = torch.empty(R, H // 64, dtype=torch.int64, device=dev) if p > 0 else None
        seed, off, cell = philox_args(dev) if p > 0 else (0, 0, None)
        N.check(L.glr_drop_add_ln_fwd(N.ptr(h), N.ptr(inp), N.ptr(weight), N.ptr(bias), R, H, float(eps), float(p), seed, off,
                                      cell, N.ptr(out32), N.ptr(out16), N.ptr(stats), N.ptr(mask), N.stream()), "glr_drop_add_ln_fwd")
        ctx.save_for_backward(h, inp, weight, stats, mask)
        ctx.p = float(p)
        ctx.hb = None if h_bias is None else (h_bias.dtype, h_bias.shape)
        ctx.set_materialize_grads(False)
        return out32, out16

    @staticmethod
    def backward(ctx, d32, d16):
        h, inp, weight, stats, mask = ctx.saved_tensors
        L = N.lib()
        H = h.shape[-1]
        R = h.numel() // H
        dev = h.device
        if d32 is None and d16 is None:
            return None, None, None, None, None, None, None
        if d32 is not None:
            d32 = d32.float().contiguous()
        if d16 is not None:
            d16 = d16.to(torch.bfloat16).contiguous()
        d_inp = torch.empty(h.shape, dtype=torch.float32, device=dev)
        d_h = torch.empty(h.shape, dtype=torch.bfloat16, device=dev)
        dgb = torch.empty(2, H, dtype=torch.float32, device=dev)
        ws = _workspace(dev, R, H)
        dhb = None
        if ctx.hb is not None and ctx.needs_input_grad[6]:
            if ctx.hb[0] not in (torch.bfloat16, torch.float32) or tuple(ctx.hb[1]) != (H,):
                raise RuntimeError("drop_add_ln: h_bias must be a bf16 / fp32 vector of the hidden size")
            dhb = torch.empty(H, dtype=ctx.hb[0], device=dev)
        N.check(L.glr_drop_add_ln_bwd(N.ptr(d32), N.ptr(d16), N.ptr(h), N.ptr(inp), N.ptr(weight), N.ptr(stats), N.ptr(mask), R, H,
                                      ctx.p, N.ptr(d_inp), N.ptr(d_h), N.ptr(ws), N.ptr(dgb), c_off(dgb, H), N.ptr(dhb),
                                      int(dhb is not None and dhb.dtype == torch.bfloat16), N.stream()),
                "glr_drop_add_ln_bwd")
        return d_h, d_inp, dgb[0], dgb[1], None, None, dhb


def c_off(t, n_elems):
    """pointer to element n_elems of a contiguous fp32 tensor"""
    return N.c_void_p(t.data_ptr() + 4 * n_elems)


def _fusable(h, inp, ln):
    H = h.shape[-1]
    return (ENABLED and h.is_cuda and h.dtype == torch.bfloat16 and inp.dtype == torch.float32 and h.shape == inp.shape
            and H % 256 == 0 and 256 <= H <= 1024 and ln.elementwise_affine and ln.weight.dtype == torch.float32
            and ln.bias is not None and ln.bias.dtype == torch.float32 and tuple(ln.normalized_shape) == (H,)
            and h.is_contiguous() and inp.is_contiguous())


def drop_add_ln(h, inp, ln, p, training, h_bias=None):
    """(LayerNorm(dropout(h) + inp) in fp32, the same rounded to bf16 or None).  h_bias: see _DropAddLN.forward - only
    pass it when `fusable(h, inp, ln)` (the unfused path has no gradient to give it)."""
    if _fusable(h, inp, ln):
        return _DropAddLN.apply(h, inp, ln.weight, ln.bias, ln.eps, p if training else 0.0, h_bias)
    if h_bias is not None:
        raise RuntimeError("drop_add_ln: h_bias given on the unfused path")
    return ln(F.dropout(h, p, training) + inp), None


def dense_drop_add_ln(dense, x, inp, ln, p, training):
    """drop_add_ln(dense(x), inp, ...) of a BERT sub-layer output.  On the fused path under the bf16 flat optimizer the
    Linear's backward skips its bias gradient and the LayerNorm backward kernel supplies it (one pass over d_h less)."""
    from .fused_linear import linear, linear_fusable
    H = dense.out_features
    if (ENABLED and x.is_cuda and linear_fusable(x, dense.weight, dense.bias) and inp.dtype == torch.float32
            and H % 256 == 0 and 256 <= H <= 1024 and ln.elementwise_affine and ln.weight.dtype == torch.float32
            and ln.bias is not None and ln.bias.dtype == torch.float32 and tuple(ln.normalized_shape) == (H,)
            and inp.is_contiguous() and inp.shape[:-1] == x.shape[:-1] and inp.shape[-1] == H):
        h = linear(x, dense.weight, dense.bias, bias_grad_from_epilogue=True)
        return _DropAddLN.apply(h, inp, ln.weight, ln.bias, ln.eps, p if training else 0.0, dense.bias)
    return drop_add_ln(dense(x), inp, ln, p, training)
