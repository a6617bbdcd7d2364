#!/usr/bin/env python3
"""bench.py - GLoRIA pretraining throughput on MI355X (image-text pairs / sec, whole job).

    python bench.py --gpus N --steps K --warmup W
    N > 1 from a bare shell: bench.py starts its own N rank processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, rank 0's JSON line relayed, non-zero exit if any rank fails) BEFORE anything touches the GPU;
    under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it runs as the rank it is given.

A "step" is one full optimisation step of imagenome_pretrain (BASELINE.json configs: ResNet-50 +
BERT-base encoders, local + global contrastive loss through the HIP kernels, backward, gradient
all-reduce, clip 0.25, Adam) on a synthetic batch of the GLOBAL size 256 (224x224 images, 97 tokens,
random-init weights).  The global batch is fixed, so N ranks each take 256 / N pairs ("strong" scaling);
text embeddings are all-gathered so every rank sees all 256 negatives.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant hand-written kernel, K1
(glr_local_attn_fwd: fused region x word attention similarity): achieved = algorithmic FLOPs per launch
((4*S*D + 6*D) * B_img * sum(cap_lens)) / mean launch time measured with HIP events on the
launch stream inside the timed region; peak = 2.5 PFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md).  Two
brackets are reported: `launch_ms` / `frac` around the K1 launches alone and `op_ms` / `frac_op` around the
whole forward op (operand packing + Gram GEMM + K-tiling + K1).  `loss_path` gives the K1 backward launch,
the backward op (launch + the three gradient GEMMs + scatter) and the share of library GEMMs in it.
`cpu_baseline` times the reference-structured CPU restatement (oracle/, "port") on the host cores, rank 0,
N = 1 only: the full training step at B = 16 (BASELINE.json configs[0]) as `value`, plus the loss alone
(forward + backward at B = 16 / 64, forward at B = 256) - medians, thread count = all host cores.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "gloria-nlp-project_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime starts: gloria/hipgraph.py
torch = None                    # imported by main() AFTER the launcher decision: the parent of a multi-rank run stays off the GPU

GLOBAL_BATCH = 256
PEAK_BF16_MFMA = 2.5e15        # dense, FLOP/s (MI355X_MICROARCH.md: ~2.5 PF dense bf16)
PEAK_F32_MFMA = 157.3e12


def build(cfg_batch, precision, device, dist_ctx, bert_layers=12, miopen_benchmark=False, train_flags=False):
    from gloria import builder
    from gloria.config import pretrain_config
    from gloria.trainer import Trainer
    cfg = pretrain_config("imagenome", batch_size=cfg_batch)
    if train_flags:
        # the flags the reference's training job actually passes (/root/reference/submit_job.sh:15):
        # --no_attn_vec --attention_entropy_loss_weight 1.0 --attention_divergence_loss_weight .1
        cfg.model.gloria.merge({"no_attn_vec": True, "attention_entropy_loss_weight": 1.0,
                                "attention_divergence_loss_weight": 0.1})
    if bert_layers != 12:
        cfg.set_path("model.text.bert_config", dict(num_hidden_layers=bert_layers))
    import warnings
    warnings.filterwarnings("ignore", message="resnet_50")
    torch.manual_seed(1234)                      # identical initial weights on every rank
    dm = builder.build_data_module(cfg)
    model = builder.build_lightning_model(cfg, dm)
    trainer = Trainer(cfg, device=device, precision=precision, dist_ctx=dist_ctx, miopen_benchmark=miopen_benchmark)
    trainer.setup(model)
    model.train()
    return cfg, model, trainer


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def _say(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(sample_batch=16, full=False):
    """Reference-structured CPU path: the oracle's per-sentence loss loop + the same torch encoders
    on the host cores (torch's default thread count = the cores available to the process).  A BOUNDED sample (every leg stops adding repetitions once its
    time budget is spent; medians): full training steps at B = 16 (the metric's unit -> `value`), and the loss alone
    on synthetic embeddings: forward + backward at B = 16 and B = 64; with --cpu-full also the loss forward at the
    bench size B = 256 (minutes of CPU time: not part of the default run)."""
    from gloria import builder
    from gloria.config import pretrain_config
    from gloria.datasets.synthetic import make_batch
    from oracle import gloria_oracle as orc
    import numpy as np
    # torch's own default = the cores this process may use (OpenMP honours the affinity mask / cgroup share of the
    # box; forcing os.cpu_count() over-subscribes a shared host); GLR_CPU_THREADS overrides
    if os.environ.get("GLR_CPU_THREADS"):
        torch.set_num_threads(int(os.environ["GLR_CPU_THREADS"]))
    cores = torch.get_num_threads()

    def sample(fn, max_reps, budget_s, warm=0):
        ts = []
        for i in range(warm + max_reps):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            if i >= warm:
                ts.append(dt)
            if i >= warm and sum(ts) > budget_s:
                break
        return _median(ts), len(ts)

    cfg = pretrain_config("imagenome", batch_size=sample_batch)
    torch.manual_seed(1234)
    model = builder.build_gloria_model(cfg)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=5e-5, weight_decay=1e-6, betas=(0.5, 0.999))
    batch = make_batch(sample_batch, seed=1234, lengths="words")

    def step():
        il, ig, tl, tg, sents = model(batch)
        loss, _ = orc.calc_loss(il, ig, tl, tg, sents, temp1=4.0, temp2=5.0, temp3=10.0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 0.25)
        opt.step()

    _say(f"cpu_baseline: full training steps at B={sample_batch} on {cores} threads ...")
    step_s, n_step = sample(step, 5, 25.0, warm=1)
    del model, opt, params

    def loss_fn(B, backward):
        g = torch.Generator().manual_seed(1234 + B)
        img = (torch.randn(B, 768, 19, 19, generator=g) * 0.5).requires_grad_(backward)
        words = (torch.randn(B, 768, 97, generator=g) * 0.5).requires_grad_(backward)
        ig = (torch.randn(B, 768, generator=g) * 0.5).requires_grad_(backward)
        tg = (torch.randn(B, 768, generator=g) * 0.5).requires_grad_(backward)
        lens = sorted((int(x) + 1 for x in np.random.default_rng(1234).integers(4, 40, size=B)), reverse=True)

        def run():
            with torch.set_grad_enabled(backward):
                l = orc.local_loss(img, words, lens)
                gl_ = orc.global_loss(ig, tg)
                tot = l[0] + l[1] + gl_[0] + gl_[1]
            if backward:
                tot.backward()
                img.grad = words.grad = ig.grad = tg.grad = None
        return run

    loss = {}
    for B, bwd, reps, budget in ((16, True, 5, 6.0), (64, True, 5, 12.0)) + (((256, False, 1, 0.0),) if full else ()):
        _say(f"cpu_baseline: loss {'fwd+bwd' if bwd else 'fwd'} at B={B} ...")
        t, n = sample(loss_fn(B, bwd), reps, budget)
        loss[f"B{B}_{'fwd_bwd' if bwd else 'fwd'}_s"] = t
        loss[f"B{B}_reps"] = n
    return {"value": sample_batch / step_s, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"median of {n_step} full training step(s) (ResNet-50 + BERT-base fwd/bwd + reference-structured "
                      f"per-sentence local/global loss loop + clip + Adam) at B={sample_batch}, fp32, torch CPU ops, "
                      f"after 1 warm-up step; loss_only = the loss loop alone on synthetic embeddings (medians)",
            "step_s": step_s, "loss_only": loss}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, out_fd):
    """`bench.py --gpus N` from a bare shell: start N rank processes of this script (one per GPU), relay rank 0's
    JSON line, fail if any rank fails.  The parent imports neither torch nor anything that touches the GPU, and never
    replaces itself: the ranks are plain child processes."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno(), stderr=None))
    line, rc = b"", 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if r == 0:
                    line = procs[0].stdout.read()
                if code != 0:
                    _say(f"rank {r} exited with code {code}: stopping the other ranks")
                    rc = code if code > 0 else 1
                    pending.clear()
                    break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except Exception:      # noqa: BLE001
                p.kill()
    if rc == 0:
        out = [ln for ln in line.decode().splitlines() if ln.startswith("{")]
        if len(out) != 1:
            _say(f"rank 0 printed {len(out)} JSON lines (expected 1)")
            rc = 1
        else:
            os.write(out_fd, (out[0] + "\n").encode())
    return rc


def selftest_step_loop(args, out_fd):
    """CPU rehearsal of the launcher + rank plumbing (tests/test_bench_launcher.py): gloo, a tiny model, the bench's own
    barrier / max-over-ranks timing and JSON line.  Not a measurement."""
    import torch.distributed as dist
    from gloria import dist as gdist
    dctx = gdist.init_from_env("gloo")
    world = dctx.world_size if dctx else 1
    rank = dctx.rank if dctx else 0
    if args.selftest_fail_rank is not None and rank == args.selftest_fail_rank:
        raise SystemExit(3)
    torch.manual_seed(7)
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 4))
    params = list(model.parameters())
    opt = torch.optim.Adam(params, lr=1e-2)
    GB = args.global_batch if args.global_batch % world == 0 else 8 * world
    x = torch.randn(GB, 16, generator=torch.Generator().manual_seed(1))[rank::world]
    y = torch.randn(GB, 4, generator=torch.Generator().manual_seed(2))[rank::world]

    def step():
        opt.zero_grad(set_to_none=True)
        loss = ((model(x) - y) ** 2).sum() / GB
        loss.backward()
        if dctx:
            dctx.allreduce_grads(params)
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    if dctx:
        dctx.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if dctx:
        dctx.barrier()
    elapsed = time.perf_counter() - t0
    if dctx:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = dctx.all_reduce_scalar(loss)
    else:
        tot = loss.detach()
    if rank == 0:
        rec = {"metric": "selftest (launcher rehearsal, not a measurement)", "value": GB * args.steps / elapsed,
               "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "selftest", "global_batch": GB, "per_gpu_batch": GB // world,
                          "world_size_seen": dist.get_world_size() if dctx else 1, "backend": "gloo",
                          "final_loss": float(tot)}}
        os.write(out_fd, (json.dumps(rec) + "\n").encode())
    if dctx:
        dctx.barrier()
        dist.destroy_process_group()


def main():
    # the contract is ONE JSON line on stdout: libraries that print to fd 1 (RCCL's version banner at communicator
    # creation) are sent to stderr, the line itself goes to the saved descriptor
    out_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--lengths", default="words", choices=["words", "mix", "max"],
                    help="words: caption WORD counts ~ U{4..39} (SURVEY.md 8d, the metric's workload); mix: round 1's "
                         "word-PIECE counts ~ U{4..39}; max: every caption at the 97-token limit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="cpu_baseline also runs the loss forward at B = 256 (minutes)")
    ap.add_argument("--bert-layers", type=int, default=12)
    ap.add_argument("--global-batch", type=int, default=GLOBAL_BATCH,
                    help="experiments only: the metric is defined at the default 256")
    ap.add_argument("--train-flags", action="store_true",
                    help="the reference's real training flags (submit_job.sh:15): no_attn_vec + attention entropy 1.0 + "
                         "divergence 0.1 regularisers (experiments: the metric is defined without them)")
    ap.add_argument("--selftest", action="store_true", help="CPU / gloo rehearsal of the launcher (tests only)")
    ap.add_argument("--selftest-fail-rank", type=int, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `bench.py --gpus N`: become the launcher (nothing below this line runs in the parent)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], out_fd))
    global torch
    import torch as _torch
    torch = _torch
    if args.selftest:
        return selftest_step_loop(args, out_fd)

    from gloria import dist as gdist
    from gloria import miopen_env
    from gloria.datasets.synthetic import make_batch
    from gloria.loss import gloria_loss as GL
    from gloria.models import gloria_model as GM

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py bare (it launches its own ranks) or "
                         f"under torch.distributed.run with --nproc-per-node {args.gpus}")
    dctx = gdist.init_from_env("nccl") if (world > 1 or os.environ.get("GLR_FORCE_DIST") == "1") else None
    rank = dctx.rank if dctx else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    GB = args.global_batch
    assert GB % world == 0
    per_rank = GB // world

    use_find = miopen_env.activate()   # before the first convolution: find mode without the naive reference solvers
    cfg, model, trainer = build(per_rank, args.precision, device, dctx, args.bert_layers, use_find, args.train_flags)

    # synthetic global batch, identical on every rank; rank r takes rows r::world (length-balanced)
    full = make_batch(GB, seed=1234, lengths=args.lengths)
    idx = torch.arange(rank, GB, world)
    batch = {k: v[idx] for k, v in full.items()}
    batch = trainer.to_device(batch)

    def sync():
        if dctx:
            dctx.barrier()
        torch.cuda.synchronize()

    _say(f"model built; {args.warmup} warm-up + {args.steps} timed steps at per-GPU batch {per_rank} ...")
    first_step_s = None
    for i in range(args.warmup):
        t_w = time.perf_counter()
        trainer.training_step(model, batch)
        if i == 0:
            # warm-up step 1 holds every first-use cost (MIOpen solver lookup / search, kernel loads, allocator growth):
            # with the find-db resolving every convolution it takes seconds; tens of seconds mean MIOpen searched
            torch.cuda.synchronize()
            first_step_s = time.perf_counter() - t_w
            _say(f"warm-up step 1 took {first_step_s:.1f} s")
    sync()
    GL.PROFILE = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.training_step(model, batch)
    sync()
    elapsed = time.perf_counter() - t0
    prof, GL.PROFILE = GL.PROFILE, None
    if dctx:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    def mean_ms(key):
        ev = prof.get(key, [])
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None

    k1_ms, k1_op_ms = mean_ms("k1_fwd"), mean_ms("k1_fwd_op")
    k1_bwd_ms, k1_bwd_op_ms = mean_ms("k1_bwd"), mean_ms("k1_bwd_op")
    flops = prof.get("k1_flops", [])
    mean_flops = sum(flops) / len(flops) if flops else 0.0
    peak = (PEAK_BF16_MFMA if args.precision == "bf16" else PEAK_F32_MFMA) / 1e12
    achieved = mean_flops / (k1_ms * 1e-3) / 1e12 if k1_ms else 0.0
    achieved_op = mean_flops / (k1_op_ms * 1e-3) / 1e12 if k1_op_ms else 0.0

    # kernel launches of one step (outside the timed region; the step floor at small per-GPU batches is launch bound)
    launches = None
    if rank == 0:
        try:
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CUDA]) as tp:
                trainer.training_step(model, batch)
                torch.cuda.synchronize()
            launches = sum(1 for e in tp.events() if e.device_type == torch.autograd.DeviceType.CUDA)
        except Exception:      # noqa: BLE001 - the count is informational
            launches = None
    # sum of cap_lens (words + [CLS]) of the global batch, recovered from the algorithmic FLOP count of one launch
    s_eff = 361 + (1 if cfg.model.gloria.no_attn_vec else 0)
    cap_lens_sum = round(mean_flops / ((4.0 * s_eff * 768 + 6.0 * 768) * per_rank)) if mean_flops else None

    if rank == 0:
        rec = {
            "metric": "image-text pairs/sec (whole node), imagenome_pretrain bs=256",
            "value": GB * args.steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32",
            "data": "synthetic",
            "config": {"workload": "imagenome_pretrain_config.yaml: ResNet-50 + BERT-base(12L) + local+global "
                                   "contrastive loss, full training step (fwd+bwd+clip+Adam)",
                       "global_batch": GB, "per_gpu_batch": per_rank, "image": "224x224 -> 299x299",
                       "tokens": 97, "caption_lengths": args.lengths, "miopen_find_mode": bool(use_find),
                       "first_step_s": first_step_s, "train_flags": bool(args.train_flags),
                       "world_size_seen": torch.distributed.get_world_size() if dctx else 1,
                       "encoder_streams": 2 if GM.ENCODER_STREAMS else 1,
                       "image_encoder_hipgraph": bool(model.gloria._img_graph is not None),
                       "text_encoder_hipgraph": bool(getattr(model.gloria.text_encoder, "_graph", None) is not None),
                       "kernel_launches_per_step": launches, "sum_cap_lens": cap_lens_sum,
                       "parallelism": f"dp{world}" + (" (text-embedding all-gather + grad all-reduce, RCCL)" if world > 1 else ""),
                       "final_loss": float(loss)},
            "roofline": {"bound": "mfma", "kernel": "glr_local_attn_fwd (K1: k_local_attn_t1, one 64-slot tile per 4-wave "
                                                    "workgroup; long sentences / odd tiles: pair and single-tile kernels)",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak if peak else None,
                         "launch_ms": k1_ms, "launches": len(prof.get("k1_fwd", [])),
                         "algorithmic_flops_per_launch": mean_flops,
                         # whole forward op: operand packing + Gram GEMM + K-tiling + K1 (same algorithmic FLOPs)
                         "op_ms": k1_op_ms, "achieved_op": achieved_op, "frac_op": achieved_op / peak if peak else None,
                         # HBM-side bytes per launch: NOT measured by this run (PMC needs its own rocprofv3 passes); for the
                         # metric's workload the committed pass profiles/r03_k1_pmc_counters_fwd.txt gives FETCH_SIZE
                         # 743 172 KB x 2 (gfx950 correction) + WRITE_SIZE 1.4 MB = 1.49 GB per launch against 160 MB of
                         # operands + outputs (235 MB with the Gram matrices)
                         "traffic": 1.49e9 if (GB == GLOBAL_BATCH and world == 1 and args.lengths == "words"
                                               and args.precision == "bf16" and not args.train_flags) else None,
                         "traffic_source": "profiles/r03_k1_pmc_counters_fwd.txt (separate rocprofv3 --pmc passes of "
                                           "tools/prof_k1.py at the same shape; FETCH_SIZE x 2 + WRITE_SIZE)"},
            "loss_path": {"k1_bwd_launch_ms": k1_bwd_ms, "k1_bwd_op_ms": k1_bwd_op_ms,
                          "k1_bwd_library_gemm_share": (1.0 - k1_bwd_ms / k1_bwd_op_ms) if (k1_bwd_ms and k1_bwd_op_ms) else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            # free the GPU-side model first; the CPU leg builds its own copy
            rec["cpu_baseline"] = cpu_baseline(full=args.cpu_full)
        os.write(out_fd, (json.dumps(rec) + "\n").encode())
    if dctx:
        dctx.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
