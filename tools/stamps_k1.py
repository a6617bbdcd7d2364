"""Diagnostic: phase shares of K1 from in-kernel s_memtime stamps (libglr_stamps.so build).
Never part of the product or bench; read SHARES, not absolute length."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gloria-nlp-project_amd"))
import numpy as np, torch
from gloria import _native as N
N.LIB_PATH = N.LIB_PATH.replace("libglr.so", "libglr_stamps.so")
from gloria.loss import gloria_loss as gl

B = 256
bwd = len(sys.argv) > 1 and sys.argv[1] == "bwd"
dev = "cuda:0"
g = torch.Generator(dev).manual_seed(1234)
img = (torch.randn(B, 768, 19, 19, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(bwd)
words = (torch.randn(B, 768, 97, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(bwd)
lens = sorted((int(x) for x in np.random.default_rng(1).integers(5, 41, size=B)), reverse=True)
L = N.lib()
L.glr_debug_set_stamps.argtypes = [ctypes.c_void_p]
grid = 8 * 32 * 100
buf = torch.zeros(grid * 12, dtype=torch.int64, device=dev)
def run():
    sim, _, _ = gl.local_similarity(img, words, lens, want_attn=False)
    if bwd:
        l0, l1 = gl.dual_cross_entropy(sim); (l0 + l1).backward()
run(); torch.cuda.synchronize()
buf2 = torch.zeros(grid * 12, dtype=torch.int64, device=dev)
L.glr_debug_set_stamps_pair.argtypes = [ctypes.c_void_p]
L.glr_debug_set_stamps(ctypes.c_void_p(buf.data_ptr()))
L.glr_debug_set_stamps_pair(ctypes.c_void_p(buf2.data_ptr()))
run(); torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == "waves":
    # per-wave view of the P1 stream of the pair kernel (first 4096 workgroups)
    wbuf = torch.zeros(4096 * 8 * 32 * 3, dtype=torch.int64, device=dev)
    L.glr_debug_set_wave_stamps.argtypes = [ctypes.c_void_p]
    L.glr_debug_set_wave_stamps(ctypes.c_void_p(wbuf.data_ptr()))
    run(); torch.cuda.synchronize()
    w = wbuf.cpu().numpy().reshape(4096, 8, 32, 3).astype(np.float64)
    ok = w[:, 0, 5, 0] > 0
    w = w[ok][:, :, :24]                      # 24 chunks
    dma = w[:, :, 4:20, 1] - w[:, :, 4:20, 0]
    comp = w[:, :, 4:20, 2] - w[:, :, 4:20, 1]
    wait = w[:, :, 5:21, 0] - w[:, :, 4:20, 2]
    period = w[:, :, 5:21, 0] - w[:, :, 4:20, 0]
    print(f"workgroups {ok.sum()}; per chunk, median over workgroups and chunks 4..19 (cycles)")
    print("wave   DMA-issue  reads+MFMA  wait(vmcnt+barrier)  period")
    for wv in range(8):
        print(f"  {wv}   {np.median(dma[:, wv]):9.0f}  {np.median(comp[:, wv]):10.0f}  {np.median(wait[:, wv]):19.0f}  {np.median(period[:, wv]):6.0f}")
    sys.exit(0)
if not bwd:
    s2 = buf2.cpu().numpy().reshape(grid, 12)
    s2 = s2[s2[:, 0] > 0]
    if len(s2):
        n2 = ["setup (plan loads, tables)", "P1 gemm (128 words)", "table init + pass 1 (run max)", "pass 2 (run sum)",
              "P2 (lse, a1, e2, image, dot)", "P3 gemm (128 words)", "P4 Z + region sums", "P4 cosine/aggregate/maps"]
        d2 = np.diff(s2[:, :9].astype(np.float64), axis=1)
        t2 = (s2[:, 8] - s2[:, 0]).astype(np.float64)
        print(f"fwd PAIR kernel: workgroups {len(s2)}, median cycles per pair {np.median(t2):.0f}")
        for i, n in enumerate(n2):
            print(f"  {n:28s} median {np.median(d2[:, i]):9.0f}  share {np.median(d2[:, i]) / np.median(t2) * 100:5.1f}%")
if bwd:
    s2 = buf2.cpu().numpy().reshape(grid, 12)
    s2 = s2[(s2[:, 0] > 0) & (s2[:, 9] > 0)]
    if len(s2):
        n2 = ["setup (descriptor, statistics, lse rows)", "P1 gemm (128 words)", "P2 (a1, a2, image, -alpha s)", "P3 gemm (128 words)",
              "beta a2 copy out", "pass A (da1, run sums, a2 image)", "a2 copy out + pass B (X in registers)", "X -> image", "X copy out"]
        d2 = np.diff(s2[:, :10].astype(np.float64), axis=1)
        t2 = (s2[:, 9] - s2[:, 0]).astype(np.float64)
        print(f"bwd PAIR kernel: workgroups {len(s2)}, median ticks per pair {np.median(t2):.0f}")
        for i, n in enumerate(n2):
            print(f"  {n:40s} median {np.median(d2[:, i]):9.0f}  share {np.median(d2[:, i]) / np.median(t2) * 100:5.1f}%")
st = buf.cpu().numpy().reshape(grid, 12)
st = st[st[:, 0] > 0]
if bwd:
    cols = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]
    names = ["P1 gemm(T.V^T)", "-", "P2 elementwise+image", "P3 gemm(E.G^T)", "rho zero", "P4 da1 + LDS atomics",
             "X: compute+stage", "X: copy out", "a2: compute+stage", "a2: copy out", "tail"]
else:
    cols = [0, 1, 2, 3, 4, 11]
    names = ["P1 gemm(T.V^T)", "scores->LDS + walk", "P2 elementwise+image", "P3 gemm(E.G^T)", "P4 epilogue"]
d = np.diff(st[:, cols].astype(np.float64), axis=1)
tot = (st[:, 11] - st[:, 0]).astype(np.float64)
print(f"{'bwd' if bwd else 'fwd'}: workgroups {len(st)}, median cycles per tile {np.median(tot):.0f} (s_memtime ticks)")
for i, n in enumerate(names):
    print(f"  {n:26s} median {np.median(d[:, i]):9.0f}  share {np.median(d[:, i]) / np.median(tot) * 100:5.1f}%")
