"""Diagnostic: cost of each phase of the 4-wave single-tile K1 forward (glr_local_attn_t1.hip) from the KERNEL time (HIP
events around the launch) of builds that SKIP phases (libglr_ablate.so, GLR_K1_DBG bit mask; results are garbage, only
time matters).  Interleaved rounds in one process.  Bits: 1 P1 stream, 2 statistics passes, 4 P2, 8 P3 stream, 16 P4,
32 the streams WITHOUT their B loads (what the L2 -> register path costs), 64 P1 without its A staging, 128 every
workgroup returns at once (what dispatching the grid costs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gloria-nlp-project_amd"))
import numpy as np, torch
from gloria import _native as N
N.LIB_PATH = N.LIB_PATH.replace("libglr.so", "libglr_ablate.so")
from gloria.loss import gloria_loss as gl

B, dev = 256, "cuda:0"
g = torch.Generator(dev).manual_seed(1234)
img = (torch.randn(B, 768, 19, 19, device=dev, generator=g) * 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
words = (torch.randn(B, 768, 97, device=dev, generator=g) * 0.5).bfloat16()
lens = sorted((int(x) + 1 for x in np.random.default_rng(1234).integers(4, 40, size=B)), reverse=True)
masks = [0, 1, 2, 4, 8, 16, 22, 9, 31, 128, 32, 96, 32 | 22]
names = {0: "full", 1: "-P1", 2: "-stats", 4: "-P2", 8: "-P3", 16: "-P4", 22: "-all VALU phases", 9: "-both streams",
         31: "-everything", 128: "empty workgroups", 32: "-B loads", 96: "-B loads -A staging", 54: "-B loads -VALU phases"}
res = {m: [] for m in masks}
for rnd in range(3):
    for m in masks:
        os.environ["GLR_K1_DBG"] = str(m)
        for _ in range(2):
            gl.local_similarity(img, words, lens, want_attn=False)
        torch.cuda.synchronize()
        gl.PROFILE = {}
        for _ in range(6):
            gl.local_similarity(img, words, lens, want_attn=False)
        torch.cuda.synchronize()
        prof, gl.PROFILE = gl.PROFILE, None
        ts = sorted(a.elapsed_time(b) for a, b in prof["k1_fwd"])
        res[m].append(ts[len(ts) // 2])
full = min(res[0])
for m in masks:
    print(f"{names[m]:24s} kernel min {min(res[m]):.3f} ms   saves {full - min(res[m]):.3f} ms", flush=True)
