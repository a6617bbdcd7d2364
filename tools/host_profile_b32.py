"""Host-side profile (cProfile) of the 32-pair training step on the data-parallel code path (single-rank RCCL group):
where the host spends a step once the GPU is no longer the limit.  GLR_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 ... python tools/host_profile_b32.py"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gloria-nlp-project_amd"))
import torch
import bench
bench.torch = torch
from gloria import dist as gdist, miopen_env
from gloria.datasets.synthetic import make_batch

dctx = gdist.init_from_env("nccl") if os.environ.get("GLR_FORCE_DIST") == "1" else None
torch.cuda.set_device(0)
B = int(os.environ.get("B", "32"))
use_find = miopen_env.activate()
cfg, model, trainer = bench.build(B, "bf16", torch.device("cuda", 0), dctx, 12, use_find, False)
batch = trainer.to_device(make_batch(B, seed=1234, lengths="words"))
for _ in range(5):
    trainer.training_step(model, batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    trainer.training_step(model, batch)
torch.cuda.synchronize()
print(f"step {1e3 * (time.perf_counter() - t0) / 10:.2f} ms (dist={dctx is not None})")
# time spent inside the reducer's hooks (they run on autograd's device thread, which cProfile does not see)
if trainer.reducer is not None:
    from gloria import dist as D, optim as OPT
    acc = {}
    def timed(obj, name):
        f = getattr(obj, name)
        def w(*a, **k):
            t = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
                acc[name + "_n"] = acc.get(name + "_n", 0) + 1
        setattr(obj, name, w)
    timed(trainer.reducer, "_on_ready"); timed(trainer.reducer, "_reduce_bucket"); timed(trainer.reducer, "_join_streams")
    timed(trainer.reducer, "finish"); timed(trainer.reducer, "zero_grad")
    for g in trainer.optimizer.groups:
        timed(g, "gather"); timed(g, "stage_grad_pointers")
    import torch.distributed as tdist
    timed(tdist, "all_reduce"); timed(tdist, "reduce_scatter_tensor"); timed(tdist, "all_gather_into_tensor")
    # hooks were registered with the bound method: re-point them through the instance attribute
    for _ in range(10):
        trainer.training_step(model, batch)
    torch.cuda.synchronize()
    print("reducer host time per step (ms):", {k: (round(v / 10 * 1e3, 3) if not k.endswith("_n") else v // 10) for k, v in sorted(acc.items())})
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    trainer.training_step(model, batch)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
if dctx:
    torch.distributed.destroy_process_group()
