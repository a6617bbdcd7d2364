"""End-to-end check of the training step on the GPU against the CPU restatement: the same model
(own ResNet-50 + a 2-layer BERT, dropout off, fp32, no autocast) takes 3 optimisation steps on the
GPU (HIP loss path, gloria.trainer.Trainer) and on the CPU (oracle calc_loss); the loss curves must
agree.  This is the "loss curve matching reference" check at a size the CPU finishes in seconds."""

import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(B):
    from gloria.config import pretrain_config
    cfg = pretrain_config("imagenome", batch_size=B)
    cfg.set_path("model.text.bert_config", dict(vocab_size=28996, num_hidden_layers=2, hidden_dropout_prob=0.0,
                                                attention_probs_dropout_prob=0.0))
    cfg.set_path("lightning.trainer.precision", 32)
    return cfg


def test_loss_curve_gpu_vs_cpu_oracle():
    from gloria import builder
    from gloria.datasets.synthetic import make_batch
    from gloria.trainer import Trainer
    from oracle import gloria_oracle as orc

    B, steps = 8, 3
    cfg = _cfg(B)
    torch.manual_seed(7)
    model = builder.build_lightning_model(cfg, builder.build_data_module(cfg))
    cpu_model = copy.deepcopy(model.gloria)
    batches = [make_batch(B, seed=100 + i) for i in range(steps)]

    # CPU: reference-structured loss from the oracle
    cpu_model.train()
    params = [p for p in cpu_model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=5e-5, weight_decay=1e-6, betas=(0.5, 0.999))
    cpu_losses = []
    for b in batches:
        il, ig, tl, tg, sents = cpu_model(b)
        loss, _ = orc.calc_loss(il, ig, tl, tg, sents)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 0.25)
        opt.step()
        cpu_losses.append(float(loss))

    trainer = Trainer(cfg, device="cuda:0", precision=32)
    trainer.setup(model)
    model.train()
    gpu_losses = [float(trainer.training_step(model, b, i)) for i, b in enumerate(batches)]
    # step 1 is a pure forward comparison; later steps also carry two Adam updates whose
    # sign-like normalisation amplifies 1e-6 gradient differences between MIOpen and CPU convolutions
    np.testing.assert_allclose(gpu_losses[0], cpu_losses[0], rtol=1e-4)
    np.testing.assert_allclose(gpu_losses, cpu_losses, rtol=1e-2)
    assert "train_loss" in model.logged


def test_bf16_autocast_step_runs_and_learns():
    from gloria import builder
    from gloria.datasets.synthetic import make_batch
    from gloria.trainer import Trainer
    B = 16
    cfg = _cfg(B)
    cfg.set_path("lightning.trainer.precision", 16)
    cfg.set_path("lightning.trainer.lr", 2e-4)
    torch.manual_seed(3)
    model = builder.build_lightning_model(cfg, builder.build_data_module(cfg))
    trainer = Trainer(cfg, device="cuda:0")
    trainer.setup(model)
    model.train()
    batch = make_batch(B, seed=5)
    losses = [float(trainer.training_step(model, batch, i)) for i in range(8)]
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0]          # same batch repeated: the contrastive loss must go down


def test_attention_finetune_config_runs():
    """configs/imagenome_attn_finetune_config.yaml: local = global = 0, segmentation_loss_weight = 1;
    the gradient flows through the diagonal attention maps only."""
    from gloria import builder
    from gloria.config import pretrain_config
    from gloria.datasets.synthetic import make_batch
    from gloria.trainer import Trainer
    B = 8
    cfg = pretrain_config("imagenome_attn_finetune", batch_size=B)
    cfg.set_path("model.text.bert_config", dict(num_hidden_layers=2, hidden_dropout_prob=0.0,
                                                attention_probs_dropout_prob=0.0))
    cfg.set_path("lightning.trainer.precision", 32)
    torch.manual_seed(5)
    model = builder.build_lightning_model(cfg, builder.build_data_module(cfg))
    trainer = Trainer(cfg, device="cuda:0", precision=32)
    trainer.setup(model)
    batch = make_batch(B, seed=9, segmentation=True)
    l0 = float(trainer.training_step(model, batch, 0))
    assert np.isfinite(l0)
    g = [p.grad for p in model.gloria.img_encoder.local_embedder.parameters()]
    assert g[0] is not None and float(g[0].abs().sum()) > 0


def test_data_parallel_path_single_rank_rehearsal(monkeypatch):
    """The code path the 2/4/8-GPU runs take (text all-gather, block-row similarity, bucketed gradient reducer with
    gradients as views of flat buckets, global-norm clip, fused Adam on channels-last parameters), rehearsed on
    one GPU with a single-rank RCCL group: two bf16 steps must run and agree with the plain single-GPU step."""
    import socket
    import torch.distributed as dist
    from gloria import builder, dist as gdist
    from gloria.config import pretrain_config
    from gloria.datasets.synthetic import make_batch
    from gloria.trainer import Trainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    for k, v in dict(GLR_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                     MASTER_PORT=str(port)).items():
        monkeypatch.setenv(k, v)
    B = 8
    cfg = pretrain_config("imagenome", batch_size=B)
    cfg.set_path("model.text.bert_config", dict(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
    batch = make_batch(B, seed=3)
    losses = {}
    try:
        for mode in ("plain", "dist"):
            torch.manual_seed(11)
            model = builder.build_lightning_model(cfg, builder.build_data_module(cfg))
            dctx = gdist.init_from_env("nccl") if mode == "dist" else None
            tr = Trainer(cfg, device="cuda:0", precision="bf16", dist_ctx=dctx)
            tr.setup(model)
            model.train()
            losses[mode] = [float(tr.training_step(model, batch, i)) for i in range(2)]
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    assert all(np.isfinite(losses["dist"]))
    np.testing.assert_allclose(losses["dist"], losses["plain"], rtol=2e-2)
