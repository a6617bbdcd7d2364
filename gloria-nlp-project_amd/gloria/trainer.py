"""Minimal training driver standing in for the Lightning Trainer the reference configures in
run.py:172-207 (precision 16 -> bf16 autocast here, gradient_clip_val 0.25, Adam(0.5, 0.999),
ReduceLROnPlateau, ModelCheckpoint-style save/resume with the reference's checkpoint layout).
One process per GPU; with a DistContext the loss is the global-batch loss and parameter gradients
are SUM-all-reduced before clipping (so the clip uses the global norm)."""

import json
import os
import time

import torch


class Trainer:
    def __init__(self, cfg, device="cuda", precision=None, dist_ctx=None, log_path=None, miopen_benchmark=False,
                 sync_bn=False, flat_optimizer=None, graph_image_encoder=None, graph_text_encoder=None):
        self.cfg = cfg
        self.device = torch.device(device)
        prec = precision if precision is not None else cfg.lightning.trainer.precision
        self.autocast_dtype = torch.bfloat16 if str(prec) in ("16", "bf16") else None
        self.clip = cfg.lightning.trainer.gradient_clip_val
        self.max_epochs = cfg.lightning.trainer.max_epochs or 1
        self.dist = dist_ctx
        self.log_path = log_path
        self.global_step = 0
        self.optimizer = self.scheduler = None
        # MIOpen "benchmark" (find) mode picks the fastest solver per conv shape (-8 % step time at B = 256); its
        # per-process search is kept short by gloria.miopen_env (naive reference solvers off)
        self.miopen_benchmark = miopen_benchmark
        # data-parallel parity mode: BatchNorm statistics over the GLOBAL batch (torch SyncBatchNorm: one small
        # per-channel all-reduce per layer) - the reference oracle is single-device full-batch BN (SURVEY.md
        # section 7); the default keeps per-rank statistics (speed mode)
        self.sync_bn = bool(sync_bn)
        # hipGraph capture of both encoders' forward + backward at the first training step (GLoRIA.enable_image_graph: the
        # whole image encoder; BertEncoder.enable_graph: the 12 BERT layers): small per-rank batches are bound by the HOST
        # time of ~1100 launches.  Measured with the graph-safe runtime flag of gloria/hipgraph.py, data-parallel path on
        # one GPU (single-rank RCCL), ms per step: 32 pairs 25.3 eager -> 17.9 image graph -> 16.0 both; 64: 24.8; 128: 42.3;
        # at 256 pairs the GPU is the limit and the replays cost 1.4 ms (78.6 vs 77.3 eager).  Default (None): captured
        # when the per-rank batch is <= GRAPH_MAX_BATCH; GLR_GRAPH_IMG / GLR_GRAPH_TXT = 0 / 1 override.  The image graph
        # is off with SyncBatchNorm (a collective inside the capture), the text graph needs the flat bf16 optimizer.
        self.graph_image_encoder = self._tristate(graph_image_encoder, "GLR_GRAPH_IMG")
        self.graph_text_encoder = self._tristate(graph_text_encoder, "GLR_GRAPH_TXT")
        if self.sync_bn:
            self.graph_image_encoder = False
        self._graph_tried = False
        # bucketed all-reduce overlapped with backward (hooks) or one gather + all-reduce per group after it
        # (GradReducer.from_flat); GLR_REDUCER_OVERLAP=0/1 overrides the default
        # Default: NO overlap.  Measured on the single-rank RCCL rehearsal (one GPU, image-encoder graph on) the
        # hook-driven reducer costs ~5 ms per step at every per-rank batch - 20.3 vs 16.1 ms at 32 pairs, 29.3 vs 24.6 at
        # 64, 47.9 vs 42.2 at 128 (~360 Python hook calls and 5 bucket launches from autograd's device thread, and the
        # accumulate-grad nodes it keeps alive across steps) - against the ~1.4 ms a 268-MB bf16 all-reduce takes on
        # 8 GPUs when nothing hides it.
        env = os.environ.get("GLR_REDUCER_OVERLAP")
        self.reducer_overlap = (env != "0") if env is not None else False
        # bf16 runs on the GPU keep fp32 master weights + bf16 shadows in flat buffers and do clip + Adam in three
        # launches (gloria/optim.py); GLR_FLAT_OPTIMIZER=0 or flat_optimizer=False keeps torch's fused Adam + autocast casts
        if flat_optimizer is None:
            flat_optimizer = os.environ.get("GLR_FLAT_OPTIMIZER", "1") != "0"
        self.flat_optimizer = bool(flat_optimizer) and self.device.type == "cuda" and self.autocast_dtype is not None \
            and cfg.train.optimizer.name == "Adam"

    GRAPH_MAX_BATCH = 128

    @staticmethod
    def _tristate(arg, env):
        if arg is not None:
            return bool(arg)
        v = os.environ.get(env)
        return None if v is None else v != "0"

    # ------------------------------------------------------------------ setup
    def setup(self, model):
        model.to(self.device)
        if self.device.type == "cuda":
            torch.backends.cudnn.benchmark = bool(self.miopen_benchmark)
            model.gloria.img_encoder.to(memory_format=torch.channels_last)
        model.gloria.dist = self.dist
        if self.sync_bn and self.dist is not None and self.dist.world_size > 1:
            model.gloria.img_encoder = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model.gloria.img_encoder,
                                                                                      self.dist.group)
        # how build_optimizer is to build the optimizer for THIS run, written explicitly in both cases: the cfg also
        # travels inside checkpoints (hyper_parameters), and a stale `flat_bf16: True` from a bf16 run must not hand an
        # fp32 / GLR_FLAT_OPTIMIZER=0 run bf16 parameters.
        # data-parallel ranks gather every bucket's gradients into a flat buffer (all-reduced slice by slice during
        # backward); a single process reads the gradients where autograd leaves them (pointer table)
        self.cfg.set_path("train.optimizer.flat_bf16", bool(self.flat_optimizer))
        self.cfg.set_path("train.optimizer.flat_clip", self.clip if self.flat_optimizer else None)
        self.cfg.set_path("train.optimizer.flat_grads",
                          bool(self.flat_optimizer and self.dist is not None and self.dist.active))
        opt = model.configure_optimizers()
        self.optimizer, self.scheduler = opt["optimizer"], opt["lr_scheduler"]
        self.params = [p for g in self.optimizer.param_groups for p in g["params"]]
        from .optim import ShadowAdam
        self.flat = isinstance(self.optimizer, ShadowAdam)
        self.reducer = None
        if self.dist is not None and self.dist.active:
            from .dist import GradReducer
            if self.flat:       # the optimizer's flat gradient buffers are the all-reduce buckets
                self.reducer = GradReducer.from_flat(self.optimizer.groups, self.dist, overlap=self.reducer_overlap)
            else:
                self.reducer = GradReducer(self.params, self.dist)      # grads become views of flat buckets
        return model

    def to_device(self, batch):
        out = {}
        for k, v in batch.items():
            if torch.is_tensor(v):
                host = v.numpy() if (k == "caption_ids" and v.device.type == "cpu") else getattr(v, "_glr_host", None)
                v = v.to(self.device, non_blocking=True)
                if k == "imgs" and self.device.type == "cuda":
                    v = v.contiguous(memory_format=torch.channels_last)
                if host is not None:
                    v._glr_host = host           # the word-piece slotting runs on the host: no D2H sync per step
            out[k] = v
        return out

    # ------------------------------------------------------------------ one optimisation step
    def training_step(self, model, batch, batch_idx=0):
        batch = self.to_device(batch)
        if not self._graph_tried and self.device.type == "cuda" and self.autocast_dtype is not None:
            self._graph_tried = True          # once: the batch shape is static in training (drop_last loaders)
            auto = int(batch["imgs"].shape[0]) <= self.GRAPH_MAX_BATCH
            if auto if self.graph_image_encoder is None else self.graph_image_encoder:
                model.gloria.enable_image_graph(batch["imgs"], self.autocast_dtype)
            if (auto if self.graph_text_encoder is None else self.graph_text_encoder) and self.flat_optimizer:
                model.gloria.enable_text_graph(batch["caption_ids"], batch["attention_mask"], batch["token_type_ids"],
                                               self.autocast_dtype)
        ctx = torch.autocast(self.device.type, dtype=self.autocast_dtype) if self.autocast_dtype else _Null()
        with ctx:
            out = model.training_step(batch, batch_idx)
        loss = out["loss"]
        if self.reducer is not None:
            self.reducer.zero_grad()
            loss.backward()                   # bucket all-reduces start as soon as a bucket is complete
            self.reducer.finish()
        else:
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
        if self.clip and not self.flat:       # the flat optimizer clips inside its step (post-reduce global norm)
            torch.nn.utils.clip_grad_norm_(self.params, self.clip)
        self.optimizer.step()
        self.global_step += 1
        return self._global_loss(model, loss)

    def _global_loss(self, model, loss):
        """what a single process would report: under data parallelism a rank's own loss holds only its share of
        the segmentation / regulariser terms (GLoRIA.global_batch_loss adds the other ranks' shares)"""
        if self.dist is not None and self.dist.active and hasattr(model.gloria, "global_batch_loss"):
            return model.gloria.global_batch_loss().detach()
        return loss.detach()

    @torch.no_grad()
    def evaluate(self, model, loader, split="val"):
        model.eval()
        tot, n = 0.0, 0
        for i, batch in enumerate(loader):
            batch = self.to_device(batch)
            ctx = torch.autocast(self.device.type, dtype=self.autocast_dtype) if self.autocast_dtype else _Null()
            with ctx:
                out = (model.validation_step if split == "val" else model.test_step)(batch, i)
            tot += float(self._global_loss(model, out["loss"]))      # identical on every rank
            n += 1
        model.train()
        return tot / max(n, 1)

    # ------------------------------------------------------------------ fit loop
    def fit(self, model, dm, max_steps=None, ckpt_dir=None):
        self.setup(model)
        model.train()
        history = []
        for epoch in range(self.max_epochs):
            model.current_epoch = epoch
            t0 = time.time()
            for i, batch in enumerate(dm.train_dataloader()):
                loss = self.training_step(model, batch, i)
                history.append(float(loss))
                self._log({"step": self.global_step, "epoch": epoch, "train_loss": history[-1]})
                if max_steps is not None and self.global_step >= max_steps:
                    break
            val = self.evaluate(model, dm.val_dataloader())
            self._log({"epoch": epoch, "val_loss": val, "epoch_s": time.time() - t0})
            sch = self.scheduler["scheduler"] if self.scheduler else None
            if sch is not None:
                sch.step(val) if isinstance(sch, torch.optim.lr_scheduler.ReduceLROnPlateau) else sch.step()
            if ckpt_dir and (self.dist is None or self.dist.rank == 0):
                self.save_checkpoint(model, os.path.join(ckpt_dir, "last.ckpt"))
            if max_steps is not None and self.global_step >= max_steps:
                break
        return history

    def _log(self, rec):
        if self.log_path and (self.dist is None or self.dist.rank == 0):
            with open(self.log_path, "a") as f:
                f.write(json.dumps(rec) + "\n")

    # ------------------------------------------------------------------ checkpoints (reference layout + resume state)
    def save_checkpoint(self, model, path):
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        ckpt = model.checkpoint()
        if self.flat:        # the reference layout holds fp32 weights: write the masters, not the bf16 shadows
            for name, p in model.named_parameters():
                m = self.optimizer.master_of(p)
                if m is not None:
                    ckpt["state_dict"][name] = m.detach().clone()
        ckpt["optimizer_states"] = [self.optimizer.state_dict()]
        ckpt["global_step"] = self.global_step
        ckpt["epoch"] = model.current_epoch
        torch.save(ckpt, path)

    def resume(self, model, path):
        from .builder import clean_state_dict
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        sd = clean_state_dict(ckpt["state_dict"])
        model.load_state_dict(sd)
        if getattr(self, "flat", False):
            self.optimizer.load_masters((p, sd[name]) for name, p in model.named_parameters() if name in sd)
        if self.optimizer is not None and "optimizer_states" in ckpt:
            self.optimizer.load_state_dict(ckpt["optimizer_states"][0])
        self.global_step = ckpt.get("global_step", 0)
        model.current_epoch = ckpt.get("epoch", 0)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
