"""BatchNorm2d (+ residual add) (+ ReLU) as ONE pair of HBM passes per direction on channels-last bf16
activations (include/glr.h: glr_bn_act_fwd / glr_bn_act_bwd) for the ResNet-50 bottlenecks of the image
encoder.  Same parameters, buffers and state_dict keys as nn.BatchNorm2d - `fused_bn_act` is called WITH the
nn.BatchNorm2d module and falls back to `bn(x)` + add + relu whenever the fused kernels do not apply (CPU,
fp32 parity mode, eval mode, NCHW memory): torch's own BatchNorm is the reference implementation of this op,
not a CPU fallback of the loss path."""

import torch
import torch.nn.functional as F

from .. import _native as N

import os

# Opt-in (GLR_FUSED_BN=1).  Measured on MI355X at batch 256 (tools/bench_bn.py, profiles/r01_fused_bn_microbench.txt):
# the fused passes run at ~3.3 TB/s and are within +-15 % of MIOpen's BatchNorm + aten add / relu per layer; the full
# training step is 127.3 ms with them against 124.3 ms without, so torch's path stays the default this round.
ENABLED = os.environ.get("GLR_FUSED_BN", "0") == "1"


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, run_mean, run_var, eps, momentum, relu):
        L = N.lib()
        n, c, h, w = x.shape
        R = n * h * w
        dev = x.device
        y = torch.empty_like(x)                              # channels_last like x
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        invstd = torch.empty(c, dtype=torch.float32, device=dev)
        ws = torch.empty(L.glr_bn_workspace_floats(R, c), dtype=torch.float32, device=dev)
        wf, bf = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        N.check(L.glr_bn_act_fwd(N.ptr(x), N.ptr(residual), N.ptr(wf), N.ptr(bf), R, c, float(eps), float(momentum),
                                 1 if relu else 0, N.ptr(run_mean), N.ptr(run_var), N.ptr(mean), N.ptr(invstd), N.ptr(ws),
                                 N.ptr(y), N.stream()), "glr_bn_act_fwd")
        ctx.save_for_backward(x, y if residual is not None else None, wf, bf, mean, invstd)
        ctx.relu, ctx.has_res, ctx.wdtype = bool(relu), residual is not None, (weight.dtype, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, wf, bf, mean, invstd = ctx.saved_tensors
        L = N.lib()
        n, c, h, w = x.shape
        R = n * h * w
        dev = x.device
        dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dgamma = torch.empty(c, dtype=torch.float32, device=dev)
        dbeta = torch.empty(c, dtype=torch.float32, device=dev)
        tmp = torch.empty(2 * c, dtype=torch.float32, device=dev)
        ws = torch.empty(L.glr_bn_workspace_floats(R, c), dtype=torch.float32, device=dev)
        N.check(L.glr_bn_act_bwd(N.ptr(x), N.ptr(dy), N.ptr(y), N.ptr(wf), N.ptr(bf), N.ptr(mean), N.ptr(invstd), R, c,
                                 1 if ctx.relu else 0, 1 if ctx.has_res else 0, N.ptr(ws), N.ptr(dgamma), N.ptr(dbeta),
                                 N.ptr(tmp), N.ptr(dx), N.ptr(dres), N.stream()), "glr_bn_act_bwd")
        return dx, dres, dgamma.to(ctx.wdtype[0]), dbeta.to(ctx.wdtype[1]), None, None, None, None, None


def _fusable(bn, x, residual):
    return (ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and bn.training and bn.affine
            and bn.track_running_stats and bn.momentum is not None and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last)
            and (residual is None or (residual.dtype == torch.bfloat16 and residual.shape == x.shape
                                      and residual.is_contiguous(memory_format=torch.channels_last))))


def fused_bn_act(bn, x, residual=None, relu=True):
    """relu?(bn(x) (+ residual)) with nn.BatchNorm2d `bn`'s parameters and running statistics."""
    if _fusable(bn, x, residual):
        bn.num_batches_tracked.add_(1)
        return _BNAct.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu)
    out = bn(x)
    if residual is not None:
        out = out + residual
    return F.relu(out) if relu else out
