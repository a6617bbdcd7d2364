"""Diagnostic: true cost of each phase of the K1 backward pair kernel, from the launch time of builds that SKIP phases
(libglr_ablate.so, GLR_K1_DBG bit mask; results are garbage, only the time matters).  The launch is bracketed by the
loss module's own "k1_bwd" event range (pairs + single tiles).  bits: 1 P1 stream, 4 P2, 8 P3 stream, 32 pass A,
64 pass B, 128 all three copy-outs, 256 plain instead of non-temporal output stores."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gloria-nlp-project_amd"))
import numpy as np, torch
from gloria import _native as N
N.LIB_PATH = N.LIB_PATH.replace("libglr.so", "libglr_ablate.so")
from gloria.loss import gloria_loss as gl

B, dev = 256, "cuda:0"
g = torch.Generator(dev).manual_seed(1234)
img = (torch.randn(B, 768, 19, 19, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(True)
words = (torch.randn(B, 768, 97, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(True)
lens = sorted((int(x) for x in np.random.default_rng(1).integers(5, 41, size=B)), reverse=True)
masks = [0, 256, 128, 1, 4, 8, 32, 64, 4 | 32 | 64, 1 | 8, 1 | 4 | 8 | 32 | 64 | 128]
names = {0: "full (nt stores)", 256: "plain stores", 128: "-copy-outs", 1: "-P1", 4: "-P2", 8: "-P3", 32: "-pass A", 64: "-pass B",
         100: "-all VALU phases", 9: "-both streams", 237: "-everything"}
res = {m: [] for m in masks}
for rnd in range(3):
    for m in masks:
        os.environ["GLR_K1_DBG"] = "0"
        gl.PROFILE = None
        sim, _, _ = gl.local_similarity(img, words, lens, want_attn=False)
        os.environ["GLR_K1_DBG"] = str(m)
        gl.PROFILE = {}
        for _ in range(4):
            img.grad = words.grad = None
            sim.sum().backward(retain_graph=True)
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in gl.PROFILE["k1_bwd"]][1:]
        res[m].append(min(ts))
full = min(res[0])
for m in masks:
    print(f"{names[m]:20s} min {min(res[m]):.3f} ms   saves {full - min(res[m]):.3f} ms")
