"""Micro-benchmark of the local-attention forward (pack + K1) at the bench shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gloria-nlp-project_amd"))
import numpy as np, torch
from gloria.loss import gloria_loss as gl
from gloria import _native as N

def run(B, dtype, lens_mode="mix", iters=10):
    dev = "cuda:0"
    g = torch.Generator(dev).manual_seed(1234)
    img = (torch.randn(B, 768, 19, 19, device=dev, generator=g) * 0.5).to(dtype)
    words = (torch.randn(B, 768, 97, device=dev, generator=g) * 0.5).to(dtype)
    if lens_mode == "mix":
        lens = sorted((int(x) for x in np.random.default_rng(1).integers(5, 41, size=B)), reverse=True)
    else:
        lens = [96] * B
    Nw = sum(lens)
    flops = (4 * 361 * 768 + 6 * 768) * B * Nw
    plan = N.TilePlan(lens, dev)
    for _ in range(3):
        sim, _, _ = gl.local_similarity(img, words, lens, want_attn=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        sim, _, _ = gl.local_similarity(img, words, lens, want_attn=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"B={B} {dtype} {lens_mode}: N={Nw} tiles={plan.n_tiles} fill={Nw/plan.n_slots:.3f} "
          f"{dt*1e3:.3f} ms/fwd  {flops/dt/1e12:.1f} TFLOP/s algorithmic", flush=True)

if __name__ == "__main__":
    for dt in (torch.bfloat16, torch.float32):
        run(64, dt); run(256, dt)
    run(256, torch.bfloat16, "max")
