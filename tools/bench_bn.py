"""Micro-benchmark of the fused BN(+add)+ReLU kernels against torch's BatchNorm2d + add + relu (MIOpen) on the
activation shapes of ResNet-50 at 299x299, batch 256, channels-last bf16."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gloria-nlp-project_amd"))
import torch
from gloria.models import fused_bn as FB

def run(n, c, h, w, residual, fused):
    FB.ENABLED = fused
    dev = "cuda:0"
    x = torch.randn(n, c, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn(n, c, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True) if residual else None
    dy = torch.randn(n, c, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(c).to(dev).train()
    def step():
        y = FB.fused_bn_act(bn, x, r, True)
        y.backward(dy)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3

for shape in [(256, 64, 150, 150), (256, 64, 75, 75), (256, 256, 75, 75), (256, 128, 38, 38), (256, 512, 38, 38), (256, 1024, 19, 19), (256, 2048, 10, 10)]:
    for res in (False, True):
        a, b = run(*shape, res, True), run(*shape, res, False)
        gb = shape[0] * shape[1] * shape[2] * shape[3] * 2 / 1e9
        print(f"{shape} residual={res}: fused {a:.3f} ms  torch {b:.3f} ms  ({gb:.2f} GB per tensor)", flush=True)
