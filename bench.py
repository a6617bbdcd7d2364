#!/usr/bin/env python3
"""bench.py - GLoRIA pretraining throughput on MI355X (image-text pairs / sec, whole job).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one full optimisation step of imagenome_pretrain (BASELINE.json configs: ResNet-50 +
BERT-base encoders, local + global contrastive loss through the HIP kernels, backward, gradient
all-reduce, clip 0.25, Adam) on a synthetic batch of the GLOBAL size 256 (224x224 images, 97 tokens,
random-init weights).  The global batch is fixed, so N ranks each take 256 / N pairs ("strong" scaling);
text embeddings are all-gathered so every rank sees all 256 negatives.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant hand-written kernel, K1
(k_local_attn_fwd: fused region x word attention similarity): achieved = algorithmic FLOPs per launch
((4*S*D + 6*D) * B_img * sum(cap_lens)) / mean launch time measured with events on the
launch stream inside the timed region; peak = 2.5 PFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md).
`cpu_baseline` times the reference-structured CPU restatement (oracle/, "port") of the same training
step on the host cores at B = 16 (BASELINE.json configs[0] shape), rank 0, N = 1 only.
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

import torch  # noqa: E402

GLOBAL_BATCH = 256
PEAK_BF16_MFMA = 2.5e15        # dense, FLOP/s (MI355X_MICROARCH.md: ~2.5 PF dense bf16)
PEAK_F32_MFMA = 157.3e12


def build(cfg_batch, precision, device, dist_ctx, bert_layers=12, miopen_benchmark=False):
    from gloria import builder
    from gloria.config import pretrain_config
    from gloria.trainer import Trainer
    cfg = pretrain_config("imagenome", batch_size=cfg_batch)
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


def cpu_baseline(sample_batch=16, steps=1):
    """Reference-structured CPU training step (oracle loss + the same torch encoders on CPU)."""
    from gloria import builder
    from gloria.config import pretrain_config
    from gloria.datasets.synthetic import make_batch
    from oracle import gloria_oracle as orc
    cfg = pretrain_config("imagenome", batch_size=sample_batch)
    torch.manual_seed(1234)
    model = builder.build_gloria_model(cfg)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=5e-5, weight_decay=1e-6, betas=(0.5, 0.999))
    batch = make_batch(sample_batch, seed=1234)
    cores = torch.get_num_threads()
    times = []
    for it in range(steps + 1):                  # first iteration = warm-up (allocator, MKL init)
        t0 = time.perf_counter()
        il, ig, tl, tg, sents = model(batch)
        loss, _ = orc.calc_loss(il, ig, tl, tg, sents, temp1=4.0, temp2=5.0, temp3=10.0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 0.25)
        opt.step()
        times.append(time.perf_counter() - t0)
    best = min(times[1:]) if len(times) > 1 else times[0]
    return {"value": sample_batch / best, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{steps} full training step(s) (ResNet-50 + BERT-base fwd/bwd + reference-structured "
                      f"local/global loss loop + clip + Adam) at B={sample_batch}, fp32, torch CPU ops, after 1 warm-up step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--lengths", default="mix", choices=["mix", "max"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bert-layers", type=int, default=12)
    ap.add_argument("--global-batch", type=int, default=GLOBAL_BATCH,
                    help="experiments only: the metric is defined at the default 256")
    args = ap.parse_args()

    from gloria import dist as gdist
    from gloria import miopen_db
    from gloria.datasets.synthetic import make_batch
    from gloria.loss import gloria_loss as GL

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dctx = gdist.init_from_env("nccl") if (world > 1 or os.environ.get("GLR_FORCE_DIST") == "1") else None
    rank = dctx.rank if dctx else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    GB = args.global_batch
    assert GB % world == 0
    per_rank = GB // world

    use_find = miopen_db.activate(per_rank)   # before the first convolution
    cfg, model, trainer = build(per_rank, args.precision, device, dctx, args.bert_layers, use_find)

    # synthetic global batch, identical on every rank; rank r takes rows r::world (length-balanced)
    full = make_batch(GB, seed=1234, lengths=args.lengths)
    idx = torch.arange(rank, GB, world)
    batch = {k: v[idx] for k, v in full.items()}
    batch = trainer.to_device(batch)

    def sync():
        if dctx:
            dctx.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.training_step(model, batch)
    sync()
    GL.K1_EVENTS = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.training_step(model, batch)
    sync()
    elapsed = time.perf_counter() - t0
    events, GL.K1_EVENTS = GL.K1_EVENTS, None
    if dctx:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    k1_ms = [a.elapsed_time(b) for a, b, _ in events]
    k1_flops = [f for _, _, f in events]
    k1_mean_s = sum(k1_ms) / max(len(k1_ms), 1) / 1e3
    achieved = (sum(k1_flops) / max(len(k1_flops), 1)) / k1_mean_s / 1e12 if k1_ms else 0.0
    peak = (PEAK_BF16_MFMA if args.precision == "bf16" else PEAK_F32_MFMA) / 1e12

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
                       "tokens": 97, "caption_lengths": args.lengths, "miopen_find_db": bool(use_find),
                       "parallelism": f"dp{world}" + (" (text-embedding all-gather + grad all-reduce, RCCL)" if world > 1 else ""),
                       "final_loss": float(loss)},
            "roofline": {"bound": "mfma", "kernel": "k_local_attn_fwd (K1)", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak if peak else None,
                         # L2-side fabric bytes per launch from rocprofv3 PMC passes of this kernel at this shape
                         # (FETCH_SIZE 977,733 KB x2 gfx950 read correction + WRITE_SIZE 23,976 KB;
                         # profiles/r01_k1_pair_pmc_counters_v6.txt).  Infinity-Cache hits are counted: the 235 MB of
                         # operands fit the 256 MiB cache, the excess is word tiles re-read per image group
                         "traffic": 1.98e9 if (world == 1 and args.precision == "bf16" and args.lengths == "mix") else None,
                         "launch_ms": k1_mean_s * 1e3, "launches": len(k1_ms)},
        }
        if world == 1 and not args.no_cpu_baseline:
            # free the GPU-side model first; the CPU leg builds its own copy
            rec["cpu_baseline"] = cpu_baseline()
        print(json.dumps(rec), flush=True)
    if dctx:
        dctx.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
