"""Per-step GPU time by kernel family from a rocprofv3 --kernel-trace CSV of bench.py.
Steps are delimited by the K1 forward launches; only the last `--steps` steps are summarised
(warm-up and MIOpen find activity excluded).

    python tools/step_breakdown.py gpurun_out/prof/<host>/<pid>_kernel_trace.csv --steps 4
"""
import argparse
import csv
import re
import sys
from collections import defaultdict

FAMILIES = [
    ("K1 fwd (k_local_attn)", r"k_local_attn(_pair)?<.*(false|OpBF16)>|k_local_attn_pair"),
    ("K1 bwd (k_local_attn)", r"k_local_attn<.*true>"),
    ("glr fused BN+ReLU (k_bn_*)", r"\(anonymous namespace\)::k_bn_"),
    ("glr other (pack/ce/global/segsum/adam)", r"\(anonymous namespace\)::k_|glr_"),
    ("conv (MIOpen igemm/ck/naive)", r"igemm|conv|Conv|gridwise_gemm|GridwiseGemm|naive"),
    ("batchnorm", r"BatchNorm|batch_norm|bn_"),
    ("GEMM (hipBLASLt/Tensile)", r"Cijk_|Custom_Cijk"),
    ("softmax/layernorm/dropout/gelu", r"softmax|layer_norm|LayerNorm|dropout|Gelu|gelu"),
    ("optimizer/foreach", r"multi_tensor|foreach|adam|Adam"),
    ("pooling/upsample", r"pool|upsample|interp"),
    ("copy/cast/fill", r"copy|Copy|fill|Fill|cat|Cat"),
    ("elementwise/reduce (aten)", r"elementwise|reduce_kernel|index"),
]


def family(name):
    if "k_local_attn" in name:
        bwd = "_bwd" in name or re.search(r"k_local_attn<[^>]*, true,", name)
        return "K1 bwd (k_local_attn)" if bwd else "K1 fwd (k_local_attn)"
    for fam, pat in FAMILIES[2:]:
        if re.search(pat, name):
            return fam
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    rows = []
    if a.trace.endswith(".db"):                      # rocprofv3's default rocpd (sqlite) output
        import sqlite3
        con = sqlite3.connect(a.trace)
        rows = [(int(s), int(e), n) for s, e, n in con.execute("select start, end, name from kernels")]
    else:
        with open(a.trace) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a step starts at the first conv/elementwise kernel after the previous optimizer; use K1 fwd as the marker and
    # cut between steps at the midpoint kernel index: simpler - cut at each K1-fwd launch (a step = [fwd_i, fwd_i+1))
    marks = [i for i, r in enumerate(rows) if family(r[2]).startswith("K1 fwd")]
    # the pair + single launches of one forward are adjacent: merge marks closer than 50 kernels
    merged = []
    for m in marks:
        if not merged or m - merged[-1] > 50:
            merged.append(m)
    if len(merged) < a.steps + 1:
        sys.exit(f"only {len(merged)} K1 forward launches in the trace")
    lo, hi = merged[-a.steps - 1], merged[-1]
    sel = rows[lo:hi]
    wall = (rows[hi][0] - rows[lo][0]) / a.steps / 1e6
    fam_t, ker_t, ker_n = defaultdict(float), defaultdict(float), defaultdict(int)
    busy = 0.0
    for s, e, n in sel:
        d = (e - s) / 1e6
        fam_t[family(n)] += d
        ker_t[n] += d
        ker_n[n] += 1
        busy += d
    print(f"steps summarised: {a.steps}; wall per step {wall:.2f} ms; summed kernel time per step {busy / a.steps:.2f} ms "
          f"({len(sel) // a.steps} launches per step)")
    for fam, t in sorted(fam_t.items(), key=lambda kv: -kv[1]):
        print(f"  {t / a.steps:8.2f} ms  {100 * t / busy:5.1f}%  {fam}")
    print("top kernels (ms per step, launches per step):")
    for n, t in sorted(ker_t.items(), key=lambda kv: -kv[1])[:a.top]:
        print(f"  {t / a.steps:8.3f}  {ker_n[n] / a.steps:7.1f}  {n[:130]}")


if __name__ == "__main__":
    main()
