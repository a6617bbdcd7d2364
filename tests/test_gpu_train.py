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


def _toy(seed=0):
    torch.manual_seed(seed)
    m = torch.nn.Sequential()
    m.emb = torch.nn.Embedding(50, 24, padding_idx=0)
    m.conv = torch.nn.Conv2d(3, 8, 3, padding=1, bias=False)
    m.bn = torch.nn.BatchNorm2d(8)
    m.fc1 = torch.nn.Linear(24, 40)
    m.ln = torch.nn.LayerNorm(40)
    m.fc2 = torch.nn.Linear(40, 7)
    m.free = torch.nn.Parameter(torch.randn(13))
    return m


@pytest.mark.parametrize("flat_grads", [True, False])
def test_flat_adam_matches_torch_adam_with_clipping(flat_grads):
    """gloria.optim.ShadowAdam (flat fp32 masters, bf16 shadows, clip folded in: three launches) against
    torch.optim.Adam + clip_grad_norm_ fed the SAME gradients, 4 steps: masters, moments and the clip norm agree;
    shadows are bf16(master); parameters of normalisation layers stay fp32."""
    from gloria.optim import ShadowAdam, shadow_parameter_ids
    dev = "cuda:0"
    a, b = _toy().to(dev), _toy().to(dev)
    a.conv.to(memory_format=torch.channels_last)
    b.conv.to(memory_format=torch.channels_last)
    pa, pb = list(a.parameters()), list(b.parameters())
    opt = ShadowAdam(pa, lr=1e-2, betas=(0.5, 0.999), weight_decay=1e-3, max_grad_norm=0.25,
                     shadow_ids=shadow_parameter_ids(a), flat_grads=flat_grads)
    ref = torch.optim.Adam(pb, lr=1e-2, betas=(0.5, 0.999), weight_decay=1e-3)
    assert a.fc1.weight.dtype == torch.bfloat16 and a.fc1.bias.dtype == torch.bfloat16 and a.emb.weight.dtype == torch.float32
    assert a.ln.weight.dtype == torch.float32 and a.bn.weight.dtype == torch.float32 and a.free.dtype == torch.float32
    g = torch.Generator(dev).manual_seed(1)
    for step in range(4):
        opt.zero_grad()
        for p, q in zip(pa, pb):
            grad = torch.randn(q.shape, device=dev, generator=g) * (0.3 if step % 2 else 3.0)
            grad = grad.to(p.dtype)                     # what autograd hands a bf16 / fp32 parameter
            p.grad = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=dev).copy_(grad)
            q.grad = grad.float().contiguous(memory_format=torch.channels_last) if q.dim() == 4 else grad.float()
        if flat_grads:                                  # data-parallel mode: buckets gathered into the flat buffer
            for grp in opt.groups:
                half = len(grp.params) // 2
                grp.gather(0, half)
                grp.gather(half, len(grp.params))
        norm = torch.nn.utils.clip_grad_norm_(pb, 0.25)
        ref.step()
        opt.step()
        np.testing.assert_allclose(float(opt.clip_state[0]), float(norm), rtol=1e-5)
        for p, q in zip(pa, pb):
            m = opt.master_of(p)
            np.testing.assert_allclose(m.cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-5, atol=1e-7)
            if p.dtype == torch.bfloat16:
                assert torch.equal(p.detach(), m.to(torch.bfloat16))
            np.testing.assert_allclose(opt.state[p]["exp_avg"].cpu().numpy(), ref.state[q]["exp_avg"].cpu().numpy(),
                                       rtol=2e-5, atol=1e-8)
            np.testing.assert_allclose(opt.state[p]["exp_avg_sq"].cpu().numpy(),
                                       ref.state[q]["exp_avg_sq"].cpu().numpy(), rtol=2e-5, atol=1e-10)


def test_flat_optimizer_step_equals_stock_amp_step_and_resumes(tmp_path):
    """The bf16 training step with the flat optimizer against the stock AMP recipe (autocast casts, torch fused Adam,
    clip_grad_norm_): same losses over 3 steps; a checkpoint holds fp32 masters in the reference layout and resumes
    to identical masters, shadows and moments."""
    from gloria import builder
    from gloria.config import pretrain_config
    from gloria.datasets.synthetic import make_batch
    from gloria.trainer import Trainer
    B = 8
    cfg = pretrain_config("imagenome", batch_size=B)
    cfg.set_path("model.text.bert_config", dict(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
    batches = [make_batch(B, seed=40 + i) for i in range(3)]
    losses, trainers, models = {}, {}, {}
    for mode in (False, True):
        torch.manual_seed(21)
        c = pretrain_config("imagenome", batch_size=B)
        c.set_path("model.text.bert_config", dict(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
        model = builder.build_lightning_model(c, builder.build_data_module(c))
        tr = Trainer(c, device="cuda:0", precision="bf16", flat_optimizer=mode)
        tr.setup(model)
        assert tr.flat == mode
        model.train()
        losses[mode] = [float(tr.training_step(model, b, i)) for i, b in enumerate(batches)]
        trainers[mode], models[mode] = tr, model
    # step 1 differs only by rounding paths; later steps carry Adam updates whose sign-like normalisation amplifies
    # bf16-level gradient differences (the same band as the data-parallel rehearsal above)
    np.testing.assert_allclose(losses[True][0], losses[False][0], rtol=1e-3)
    np.testing.assert_allclose(losses[True], losses[False], rtol=4e-2)
    tr, model = trainers[True], models[True]
    w = model.gloria.text_encoder.model.encoder.layer[0].attention.self.query.weight
    assert w.dtype == torch.bfloat16
    ck = tmp_path / "flat.ckpt"
    tr.save_checkpoint(model, str(ck))
    saved = torch.load(ck, map_location="cpu", weights_only=True)
    key = "gloria.text_encoder.model.encoder.layer.0.attention.self.query.weight"
    assert saved["state_dict"][key].dtype == torch.float32            # masters, the reference's layout
    torch.manual_seed(99)
    m2 = builder.build_lightning_model(cfg, builder.build_data_module(cfg))
    tr2 = Trainer(cfg, device="cuda:0", precision="bf16", flat_optimizer=True)
    tr2.setup(m2)
    tr2.resume(m2, str(ck))
    assert tr2.global_step == tr.global_step
    for (n1, p1), (n2, p2) in zip(model.named_parameters(), m2.named_parameters()):
        a, b = tr.optimizer.master_of(p1), tr2.optimizer.master_of(p2)
        if a is None:
            continue
        assert torch.equal(a, b) and torch.equal(p1.detach(), p2.detach()), n1
        assert torch.equal(tr.optimizer.state[p1]["exp_avg"], tr2.optimizer.state[p2]["exp_avg"]), n1
    l1 = float(tr.training_step(model, batches[0], 3))
    l2 = float(tr2.training_step(m2, batches[0], 3))
    # identical weights, moments and buffers (asserted above); the encoders' library GEMMs / convolutions (stream-K,
    # split reductions) are not bitwise reproducible between two model instances, hence a bf16-level band
    np.testing.assert_allclose(l1, l2, rtol=2e-3)


def test_optimizer_state_crosses_between_stock_and_flat_adam_by_parameter(tmp_path):
    """the flat optimizer leaves the (never-reached) BERT pooler out, stock Adam keeps it; text_encoder is
    registered before img_encoder, so a positional mapping would land every later moment two parameters late.
    Stock -> flat and flat -> stock resumes must map the moments BY PARAMETER; a state dict of a foreign size raises."""
    from gloria import builder
    from gloria.config import pretrain_config
    from gloria.datasets.synthetic import make_batch
    from gloria.trainer import Trainer
    B = 8

    def make(flat, seed):
        torch.manual_seed(seed)
        c = pretrain_config("imagenome", batch_size=B)
        c.set_path("model.text.bert_config", dict(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
        model = builder.build_lightning_model(c, builder.build_data_module(c))
        tr = Trainer(c, device="cuda:0", precision="bf16", flat_optimizer=flat)
        tr.setup(model)
        model.train()
        return tr, model

    def moments(tr, model):
        out = {}
        for n, p in model.named_parameters():
            st = tr.optimizer.state.get(p)
            if st and "exp_avg" in st and "pooler" not in n:
                out[n] = (st["exp_avg"].float().cpu().clone(), st["exp_avg_sq"].float().cpu().clone())
        return out

    batch = make_batch(B, seed=77)
    for src_flat in (False, True):
        tr, model = make(src_flat, 5)
        for i in range(2):
            tr.training_step(model, batch, i)
        ck = tmp_path / f"src_{int(src_flat)}.ckpt"
        tr.save_checkpoint(model, str(ck))
        n_saved = len(torch.load(ck, map_location="cpu", weights_only=True)["optimizer_states"][0]["param_groups"][0]["params"])
        assert n_saved == len([p for p in model.parameters() if p.requires_grad])      # the reference layout: every parameter
        want = moments(tr, model)
        tr2, m2 = make(not src_flat, 6)
        tr2.resume(m2, str(ck))
        got = moments(tr2, m2)
        # stock Adam holds state only for parameters that have received a gradient, the flat optimizer for all of its
        # parameters from the start (zeros): compare what both hold, and what only one holds must be all zeros
        common = set(want) & set(got)
        assert len(common) > 150, (sorted(set(want) ^ set(got))[:10], len(want), len(got))
        for n in set(want) ^ set(got):
            m = (want.get(n) or got.get(n))
            assert float(m[0].abs().max()) == 0.0 and float(m[1].abs().max()) == 0.0, n
        for n in common:
            np.testing.assert_allclose(got[n][0].numpy(), want[n][0].numpy(), rtol=0, atol=0, err_msg=n)
            np.testing.assert_allclose(got[n][1].numpy(), want[n][1].numpy(), rtol=0, atol=0, err_msg=n)
        if not src_flat:
            assert tr2.optimizer.t == 2
    # a state dict that is neither over the model's parameters nor over the optimizer's own is refused
    tr, model = make(True, 5)
    sd = tr.optimizer.state_dict()
    sd["param_groups"][0]["params"] = sd["param_groups"][0]["params"][:-3]
    with pytest.raises(ValueError):
        tr.optimizer.load_state_dict(sd)
