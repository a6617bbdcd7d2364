"""Run the local-similarity forward (+ optionally backward) a few times at the bench shape, for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gloria-nlp-project_amd"))
import numpy as np, torch
from gloria.loss import gloria_loss as gl

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bwd = len(sys.argv) > 2 and sys.argv[2] == "bwd"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = "cuda:0"
g = torch.Generator(dev).manual_seed(1234)
img = (torch.randn(B, 768, 19, 19, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(bwd)
words = (torch.randn(B, 768, 97, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(bwd)
lens = sorted((int(x) for x in np.random.default_rng(1).integers(5, 41, size=B)), reverse=True)
for _ in range(iters):
    sim, _, _ = gl.local_similarity(img, words, lens, want_attn=False)
    if bwd:
        l0, l1 = gl.dual_cross_entropy(sim)
        (l0 + l1).backward()
torch.cuda.synchronize()
print("ok", float(sim.float().mean()))
