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
def test_fused_bn_matches_torch(n, c, h, w, residual, relu, monkeypatch):
    from gloria.models import fused_bn as FB
    monkeypatch.setattr(FB, "ENABLED", True)
    g = torch.Generator().manual_seed(n * 1000 + c)
    x = (torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    r = torch.randn(n, c, h, w, generator=g).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last) if residual else None
    dy = torch.randn(n, c, h, w, generator=g).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(c).to(DEV).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, generator=g) * 0.2)
    ref = torch.nn.BatchNorm2d(c).to(DEV).train()
    ref.load_state_dict(bn.state_dict())

    xa = x.clone().requires_grad_(True)
    ra = None if r is None else r.clone().requires_grad_(True)
    ya = FB.fused_bn_act(bn, xa, ra, relu)
    assert ya.dtype == torch.bfloat16 and ya.is_contiguous(memory_format=torch.channels_last)
    ya.backward(dy)

    xb = x.float().requires_grad_(True)
    rb = None if r is None else r.float().requires_grad_(True)
    z = ref(xb)
    if rb is not None:
        z = z + rb
    yb = torch.relu(z) if relu else z
    yb.backward(dy.float())

    np.testing.assert_allclose(ya.detach().float().cpu().numpy(), yb.detach().cpu().numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(xa.grad.float().cpu().numpy(), xb.grad.cpu().numpy(), rtol=2e-2, atol=2e-2)
    if ra is not None:
        np.testing.assert_allclose(ra.grad.float().cpu().numpy(), rb.grad.cpu().numpy(), rtol=1e-2, atol=1e-2)
    scale = float(ref.weight.grad.abs().max()) + 1e-6
    np.testing.assert_allclose(bn.weight.grad.cpu().numpy() / scale, ref.weight.grad.cpu().numpy() / scale, atol=2e-2)
    scale = float(ref.bias.grad.abs().max()) + 1e-6
    np.testing.assert_allclose(bn.bias.grad.cpu().numpy() / scale, ref.bias.grad.cpu().numpy() / scale, atol=2e-2)
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
