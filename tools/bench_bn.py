"""Micro-benchmark of the fused BN(+add)+ReLU kernels against torch's BatchNorm2d + add + relu (MIOpen + aten) on the
activation shapes of ResNet-50 at 299x299, batch 256, channels-last bf16.  Forward and backward are timed
separately (HIP events on the current stream); the effective rate counts the passes of the FUSED scheme
(fwd 3E / 4E with a skip connection, bwd 5E / 7E).
usage: bench_bn.py [--batch 256] [--only-fused] [--shapes big]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gloria-nlp-project_amd"))
import torch
from gloria.models import fused_bn as FB

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--only-fused", action="store_true")
ap.add_argument("--shapes", default="all")
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
DEV = "cuda:0"


def cl(t):
    return t.bfloat16().contiguous(memory_format=torch.channels_last)


def run(n, c, h, w, residual, fused):
    FB.ENABLED = fused
    x = cl(torch.randn(n, c, h, w, device=DEV)).requires_grad_(True)
    r = cl(torch.randn(n, c, h, w, device=DEV)).requires_grad_(True) if residual else None
    dy = cl(torch.randn(n, c, h, w, device=DEV))
    bn = torch.nn.BatchNorm2d(c).to(DEV).train()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for it in range(3 + args.iters):
        x.grad = None
        if r is not None:
            r.grad = None
        ev[0].record()
        y = FB.fused_bn_act(bn, x, r, True)
        ev[1].record()
        y.backward(dy)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 3:
            tf += ev[0].elapsed_time(ev[1])
            tb += ev[1].elapsed_time(ev[2])
    return tf / args.iters, tb / args.iters


shapes = [(64, 150, 150), (64, 75, 75), (256, 75, 75), (128, 75, 75), (128, 38, 38), (512, 38, 38), (256, 38, 38),
          (256, 19, 19), (1024, 19, 19), (512, 19, 19), (512, 10, 10), (2048, 10, 10)]
if args.shapes == "big":
    shapes = [(64, 150, 150), (256, 75, 75)]
for c, h, w in shapes:
    for res in (False, True):
        gb = args.batch * c * h * w * 2 / 1e9
        ff, fb = run(args.batch, c, h, w, res, True)
        line = (f"({args.batch},{c},{h},{w}) skip={int(res)} E={gb:.3f} GB  fused fwd {ff:.3f} ms ({(4 if res else 3) * gb / ff:.2f} TB/s)"
                f" bwd {fb:.3f} ms ({(7 if res else 5) * gb / fb:.2f} TB/s)")
        if not args.only_fused:
            tf, tb = run(args.batch, c, h, w, res, False)
            line += f" | torch fwd {tf:.3f} bwd {tb:.3f} ms | fused/torch {(ff + fb) / (tf + tb):.2f}"
        print(line, flush=True)
