"""BatchNorm2d (+ residual add) (+ ReLU) as ONE pair of HBM passes per direction on channels-last bf16 (or fp32) activations
(include/glr.h: glr_bn_act_fwd / glr_bn_act_bwd) for the 53 normalisation sites of the ResNet-50 image encoder
(reference: torchvision's Bottleneck through /root/reference/gloria/models/cnn_backbones.py:31-35).

Same parameters, buffers and state_dict keys as nn.BatchNorm2d: `fused_bn_act` is called WITH the nn.BatchNorm2d
module.  The kernels cover what the training step runs (GPU, channels-last, training mode, power-of-two channel counts;
bf16 under autocast and - since the end of round 3 - fp32, the parity configuration of BASELINE config 1); every other
case (eval mode, NCHW, CPU tensors of the host-logic tests) is torch's own BatchNorm + add + relu - the same operator
from the library, not a CPU fallback of the loss path.
`GLR_FUSED_BN=0` switches the kernels off (A/B measurements)."""

import os

import torch
import torch.nn.functional as F

from .. import _native as N

ENABLED = os.environ.get("GLR_FUSED_BN", "1") != "0"

# (device index, stream) -> fp32 workspace for the partial sums.  A workspace is only ever touched by launches of ONE
# stream, so stream order alone makes its reuse safe - also when the two encoders run on two streams.
_WS = {}


_WS_FLOATS = {}     # (rows, channels) -> glr_bn_workspace_floats: one ctypes call per shape, not per launch (host time)


def _workspace(dev, R, c):
    n = _WS_FLOATS.get((R, c))
    if n is None:
        n = _WS_FLOATS[(R, c)] = int(N.lib().glr_bn_workspace_floats(R, c))
    key = (dev.index, N.stream())
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=dev)
        _WS[key] = ws
    return ws


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, run_mean, run_var, nbt, eps, momentum, relu, fork):
        L = N.lib()
        n, c, h, w = x.shape
        R = n * h * w
        dev = x.device
        y = torch.empty_like(x)                              # channels_last like x
        stats = torch.empty(2, c, dtype=torch.float32, device=dev)
        ws = _workspace(dev, R, c)
        N.check(L.glr_bn_act_fwd(N.ptr(x), N.ptr(residual), N.ptr(weight), N.ptr(bias), R, c, float(eps), float(momentum),
                                 1 if relu else 0, N.ptr(run_mean), N.ptr(run_var), N.ptr(nbt), stats.data_ptr(), stats.data_ptr() + 4 * c,
                                 N.ptr(ws), N.ptr(y), N.dtype_code(x.dtype), N.stream()), "glr_bn_act_fwd")
        ctx.save_for_backward(x, y if residual is not None else None, weight, bias, stats)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        if fork:
            # a second handle on the same storage for the skip consumer of the next block: the two gradients then
            # arrive separately and are added while the backward kernel reads them (autograd would run an add kernel)
            ctx.set_materialize_grads(False)
            return y, y.detach()
        return y

    @staticmethod
    def backward(ctx, dy, dy2=None):
        x, y, weight, bias, stats = ctx.saved_tensors
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:
            return (None,) * 11
        L = N.lib()
        n, c, h, w = x.shape
        R = n * h * w
        dev = x.device
        if dy.dtype != x.dtype or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        if dy2 is not None and (dy2.dtype != x.dtype or not dy2.is_contiguous(memory_format=torch.channels_last)):
            dy2 = dy2.to(x.dtype).contiguous(memory_format=torch.channels_last)
        if dy2 is not None and not ctx.has_res:
            dy, dy2 = dy + dy2, None
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        out = torch.empty(4, c, dtype=torch.float32, device=dev)
        ws = _workspace(dev, R, c)
        N.check(L.glr_bn_act_bwd(N.ptr(x), N.ptr(dy), N.ptr(dy2), N.ptr(y), N.ptr(weight), N.ptr(bias), stats.data_ptr(), stats.data_ptr() + 4 * c,
                                 R, c, 1 if ctx.relu else 0, 1 if ctx.has_res else 0, N.ptr(ws), N.ptr(out), N.ptr(dx),
                                 N.ptr(dres), N.dtype_code(x.dtype), N.stream()), "glr_bn_act_bwd")
        return dx, dres, out[0], out[1], None, None, None, None, None, None, None


def _fusable(bn, x, residual):
    c = x.shape[1] if x.dim() == 4 else 0
    # exactly nn.BatchNorm2d: SyncBatchNorm (Trainer(sync_bn=True)) has the same attributes but its statistics span the
    # process group - the per-rank kernels would silently turn the global-batch parity mode into per-rank BN
    return (type(bn) is torch.nn.BatchNorm2d and ENABLED and x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.dim() == 4
            and bn.training and bn.affine
            and bn.track_running_stats and bn.momentum is not None and 8 <= c <= 2048 and (c & (c - 1)) == 0
            and bn.weight.dtype == torch.float32 and bn.bias.dtype == torch.float32
            and x.is_contiguous(memory_format=torch.channels_last)
            and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape
                                      and residual.is_contiguous(memory_format=torch.channels_last))))


class SkipPair:
    """(main, skip): two handles on ONE block output - the next block's convolution reads `main`, its skip connection
    `skip` - so that the two gradients reach the producing kernel separately (no autograd add in between)."""
    __slots__ = ("main", "skip")

    def __init__(self, main, skip):
        self.main, self.skip = main, skip


def fused_bn_act(bn, x, residual=None, relu=True, fork=False):
    """relu?(bn(x) (+ residual)) with nn.BatchNorm2d `bn`'s parameters and running statistics.  fork=True (fused path,
    with a residual): returns a SkipPair instead of a tensor."""
    if _fusable(bn, x, residual):
        fork = bool(fork and residual is not None and torch.is_grad_enabled())
        out = _BNAct.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps,
                           bn.momentum, relu, fork)
        return SkipPair(*out) if fork else out
    out = bn(x)
    if residual is not None:
        out = out + residual
    return F.relu(out) if relu else out


class _MaxPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        n, c, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device)
        N.check(N.lib().glr_maxpool3s2_fwd(N.ptr(x), n, h, w, c, N.ptr(y), N.ptr(idx), N.stream()), "glr_maxpool3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, c, h, w = ctx.shape
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dx = torch.empty((n, c, h, w), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
        N.check(N.lib().glr_maxpool3s2_bwd(N.ptr(dy), N.ptr(idx), n, h, w, c, N.ptr(dx), N.stream()), "glr_maxpool3s2_bwd")
        return dx


def fused_maxpool(pool, x):
    """nn.MaxPool2d(3, stride=2, padding=1) `pool` on x; channels-last bf16 on the GPU runs the one-byte-argmax kernels."""
    def _is(v, k):
        return v == k or v == (k, k)
    if (ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and _is(pool.kernel_size, 3) and _is(pool.stride, 2)
            and _is(pool.padding, 1) and _is(pool.dilation, 1) and not pool.ceil_mode and not pool.return_indices):
        return _MaxPool3s2.apply(x)
    return pool(x)
