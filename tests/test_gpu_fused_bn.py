"""Fused BatchNorm2d (+ residual) (+ ReLU) kernels against torch's own BatchNorm2d + add + relu on the same
channels-last bf16 tensors (the op sequence of the reference's torchvision bottleneck): outputs, input / skip /
affine gradients, running statistics.  Reference computed in fp32 from the same bf16 inputs; tolerances are the
bf16 output rounding (2^-8 relative) on O(1) values."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,c,h,w", [(4, 64, 38, 38), (3, 256, 19, 19), (2, 2048, 10, 10), (5, 8, 7, 5)])
@pytest.mark.parametrize("residual,relu", [(False, True), (True, True), (False, False)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fused_bn_matches_torch(n, c, h, w, residual, relu, dtype, monkeypatch):
    """bf16 (the training configuration: bf16-level bands) and fp32 (BASELINE config 1: the same kernels on 32-byte channel
    groups, bands 100x tighter) against nn.BatchNorm2d (+ add) (+ ReLU) in fp32"""
    from gloria.models import fused_bn as FB
    monkeypatch.setattr(FB, "ENABLED", True)
    g = torch.Generator().manual_seed(n * 1000 + c)
    k = 1.0 if dtype == torch.bfloat16 else 1e-2             # tolerance scale
    x = (torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    r = torch.randn(n, c, h, w, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last) if residual else None
    dy = torch.randn(n, c, h, w, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(c).to(DEV).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, generator=g) * 0.2)
    ref = torch.nn.BatchNorm2d(c).to(DEV).train()
    ref.load_state_dict(bn.state_dict())

    xa = x.clone().requires_grad_(True)
    ra = None if r is None else r.clone().requires_grad_(True)
    ya = FB.fused_bn_act(bn, xa, ra, relu)
    assert ya.dtype == dtype and ya.is_contiguous(memory_format=torch.channels_last)
    assert type(ya.grad_fn).__name__ == "_BNActBackward"       # the fused kernels, not the torch fallback
    ya.backward(dy)

    xb = x.float().requires_grad_(True)
    rb = None if r is None else r.float().requires_grad_(True)
    z = ref(xb)
    if rb is not None:
        z = z + rb
    yb = torch.relu(z) if relu else z
    yb.backward(dy.float())

    np.testing.assert_allclose(ya.detach().float().cpu().numpy(), yb.detach().cpu().numpy(), rtol=1e-2 * k, atol=1e-2 * k)
    np.testing.assert_allclose(xa.grad.float().cpu().numpy(), xb.grad.cpu().numpy(), rtol=2e-2 * k, atol=2e-2 * k)
    if ra is not None:
        np.testing.assert_allclose(ra.grad.float().cpu().numpy(), rb.grad.cpu().numpy(), rtol=1e-2 * k, atol=1e-2 * k)
    scale = float(ref.weight.grad.abs().max()) + 1e-6
    np.testing.assert_allclose(bn.weight.grad.cpu().numpy() / scale, ref.weight.grad.cpu().numpy() / scale, atol=2e-2 * k)
    scale = float(ref.bias.grad.abs().max()) + 1e-6
    np.testing.assert_allclose(bn.bias.grad.cpu().numpy() / scale, ref.bias.grad.cpu().numpy() / scale, atol=2e-2 * k)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), ref.running_var.cpu().numpy(), rtol=1e-3, atol=1e-5)
    assert int(bn.num_batches_tracked) == 1


def test_fused_bn_is_deterministic_and_falls_back(monkeypatch):
    from gloria.models import fused_bn as FB
    monkeypatch.setattr(FB, "ENABLED", True)
    x = torch.randn(8, 128, 19, 19, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(128).to(DEV).train()
    a = FB.fused_bn_act(bn, x)
    b = FB.fused_bn_act(bn, x)
    assert torch.equal(a, b)
    bn.eval()                                           # eval mode: torch's BatchNorm with the running statistics
    e = FB.fused_bn_act(bn, x)
    np.testing.assert_allclose(e.detach().float().cpu().numpy(), torch.relu(bn(x)).detach().float().cpu().numpy())
    xf = x.float()                                      # fp32 parity mode: torch's BatchNorm
    bn.train()
    f = FB.fused_bn_act(bn, xf)
    assert f.dtype == torch.float32


@pytest.mark.parametrize("channels_last", [False, True])
def test_input_resize_matches_interpolate(channels_last):
    """glr_upsample_bilinear_cl (resize + channels-last + bf16 in one pass) against F.interpolate(align_corners=True)
    followed by the cast autocast applies in front of conv1."""
    from gloria.models import vision_model as VM
    x = torch.rand(5, 3, 224, 224, device=DEV) * 2 - 1
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = VM._resize_299(x)
    assert y.dtype == torch.bfloat16 and y.shape == (5, 3, 299, 299) and y.is_contiguous(memory_format=torch.channels_last)
    ref = torch.nn.functional.interpolate(x, size=(299, 299), mode="bilinear", align_corners=True)
    # bf16 rounding of values in [-1, 1]: one ulp = 2^-8 relative where the two fp32 results straddle a rounding boundary
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.bfloat16().float().cpu().numpy(), atol=2 ** -8, rtol=0)
    assert (y.float() - ref).abs().max().item() < 2 ** -8
    z = VM._resize_299(x)                                # outside autocast: torch's operator
    assert z.dtype == torch.float32


@pytest.mark.parametrize("n,c,h,w", [(3, 64, 150, 150), (2, 8, 7, 9), (1, 16, 2, 2), (2, 32, 33, 17)])
def test_maxpool_matches_torch_including_ties(n, c, h, w):
    """One-byte-argmax max pooling (3, stride 2, padding 1) against torch's on ReLU-ed input (most windows hold tied
    zeros: the gradient must go to the FIRST maximum in scan order, as torch's does)."""
    from gloria.models import fused_bn as FB
    g = torch.Generator().manual_seed(n * 10 + c)
    x = torch.relu(torch.randn(n, c, h, w, generator=g)).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    pool = torch.nn.MaxPool2d(3, stride=2, padding=1)
    xa = x.clone().requires_grad_(True)
    ya = FB.fused_maxpool(pool, xa)
    xb = x.clone().requires_grad_(True)
    yb = pool(xb)
    assert ya.shape == yb.shape and ya.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(ya, yb)
    dy = torch.randn(yb.shape, generator=g).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    ya.backward(dy)
    yb.backward(dy)
    np.testing.assert_allclose(xa.grad.float().cpu().numpy(), xb.grad.float().cpu().numpy(), rtol=1e-2, atol=1e-2)
    assert torch.equal(xa.grad != 0, xb.grad != 0)          # identical routing


def test_forked_output_gradients_are_summed_in_the_kernel():
    """fused_bn_act(..., fork=True) returns two handles on one output; the backward kernel adds their two gradients
    while reading them.  Against the single-output op fed with the (fp32) sum."""
    from gloria.models import fused_bn as FB
    g = torch.Generator().manual_seed(5)
    n, c, h, w = 6, 256, 19, 19
    cl = lambda t: t.to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    x, r = cl(torch.randn(n, c, h, w, generator=g)), cl(torch.randn(n, c, h, w, generator=g))
    ga, gb = cl(torch.randn(n, c, h, w, generator=g)), cl(torch.randn(n, c, h, w, generator=g))
    bn = torch.nn.BatchNorm2d(c).to(DEV).train()
    res = {}
    for fork in (True, False):
        xa, ra = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
        bn.zero_grad()
        out = FB.fused_bn_act(bn, xa, ra, True, fork=fork)
        if fork:
            assert isinstance(out, FB.SkipPair) and out.main.data_ptr() == out.skip.data_ptr()
            ((out.main.float() * ga.float()).sum() + (out.skip.float() * gb.float()).sum()).backward()
            y = out.main
        else:
            (out.float() * (ga.float() + gb.float())).sum().backward()
            y = out
        res[fork] = (y.detach().clone(), xa.grad.float(), ra.grad.float(), bn.weight.grad.clone(), bn.bias.grad.clone())
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1:], res[False][1:]):
        rel = float((a - b).norm() / b.norm())
        assert rel < 1e-2, rel
    # only the skip handle used: the other gradient arrives as None
    xa, ra = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    out = FB.fused_bn_act(bn, xa, ra, True, fork=True)
    (out.skip.float() * gb.float()).sum().backward()
    assert torch.isfinite(xa.grad.float()).all() and float(xa.grad.float().abs().sum()) > 0


def test_resnet_with_forked_block_outputs_trains(monkeypatch):
    """ResNet-50 in training mode under bf16 autocast, fused kernels with block outputs handed on as (main, skip) pairs
    against the same kernels with autograd's own add in between: same parameters receive gradients, all finite, and
    the two agree to the level at which two identical runs of the library convolutions agree with each other."""
    from gloria.models import cnn_backbones as CB
    torch.manual_seed(0)
    model, _, _ = CB.resnet_50(pretrained=False)
    model = model.to(DEV).to(memory_format=torch.channels_last).train()
    x = torch.randn(16, 3, 128, 128, device=DEV).contiguous(memory_format=torch.channels_last)
    proj = torch.randn(16, 2048, device=DEV)

    def run():
        model.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = model(x)
        (y.float() * proj).sum().backward()
        return y.float().detach(), {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None}
    run()                                   # settles MIOpen's algorithm choice (the first call searches)
    ya, ga = run()
    y2, g2 = run()
    noise = max(float((ga[n] - g2[n]).norm() / g2[n].norm()) for n in g2 if float(g2[n].norm()) > 1e-4)

    def plain(self, t, fork_out=False):
        for blk in self:
            t = blk(t, fork=False)
        return t
    monkeypatch.setattr(CB._Stage, "forward", plain)
    yb, gb = run()
    assert set(ga) == set(gb) and all(torch.isfinite(v).all() for v in ga.values())
    worst = max(float((ga[n] - gb[n]).norm() / gb[n].norm()) for n in gb if float(gb[n].norm()) > 1e-4)
    print(f"[resnet forked vs plain] worst parameter-gradient relative difference {worst:.4f}; run-to-run {noise:.4f}")
    assert worst < max(3 * noise, 5e-2), (worst, noise)


@pytest.mark.gpu
def test_sync_batchnorm_never_takes_the_per_rank_kernels():
    """nn.SyncBatchNorm has BatchNorm2d's attributes; the fused per-rank kernels must leave it to torch
    (its statistics span the process group), exactly nn.BatchNorm2d is fused."""
    from gloria.models import fused_bn
    x = torch.randn(4, 64, 8, 8, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(64).cuda().train()
    sbn = torch.nn.SyncBatchNorm.convert_sync_batchnorm(torch.nn.BatchNorm2d(64)).cuda().train()
    assert isinstance(sbn, torch.nn.SyncBatchNorm)
    assert fused_bn._fusable(bn, x, None)
    assert not fused_bn._fusable(sbn, x, None)
