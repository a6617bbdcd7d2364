"""Diagnostic: true cost of each phase of the K1 forward pair kernel, from the run time of builds that SKIP phases
(libglr_ablate.so, GLR_K1_DBG bit mask; results are garbage, only wall time matters).  Interleaved rounds in one
process.  bits: 1 P1 stream, 2 statistics passes, 4 P2, 8 P3 stream, 16 P4."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gloria-nlp-project_amd"))
import numpy as np, torch
from gloria import _native as N
N.LIB_PATH = N.LIB_PATH.replace("libglr.so", "libglr_ablate.so")
from gloria.loss import gloria_loss as gl

B, dev = 256, "cuda:0"
g = torch.Generator(dev).manual_seed(1234)
img = (torch.randn(B, 768, 19, 19, device=dev, generator=g) * 0.5).bfloat16()
words = (torch.randn(B, 768, 97, device=dev, generator=g) * 0.5).bfloat16()
lens = sorted((int(x) for x in np.random.default_rng(1).integers(5, 41, size=B)), reverse=True)
masks = [0, 1, 2, 4, 8, 16, 2 | 4 | 16, 1 | 8, 31]
names = {0: "full", 1: "-P1", 2: "-stats", 4: "-P2", 8: "-P3", 16: "-P4", 22: "-all VALU phases", 9: "-both streams", 31: "-everything"}
res = {m: [] for m in masks}
for rnd in range(4):
    for m in masks:
        os.environ["GLR_K1_DBG"] = str(m)
        for _ in range(2):
            gl.local_similarity(img, words, lens, want_attn=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            gl.local_similarity(img, words, lens, want_attn=False)
        torch.cuda.synchronize()
        res[m].append((time.perf_counter() - t0) / 10 * 1e3)
full = min(res[0])
for m in masks:
    print(f"{names[m]:20s} min {min(res[m]):.3f} ms   saves {full - min(res[m]):.3f} ms")
