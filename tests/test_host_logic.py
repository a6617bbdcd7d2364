"""CPU tests of the host-side mirror: config semantics, word-piece slotting (against the reference's
outputs in tests/golden/text.npz), builder/optimizer wiring, checkpoint key layout."""

import os

import numpy as np
import pytest
import torch

import golden_inputs as gi


def tiny_cfg(batch_size=4, **kw):
    from gloria.config import pretrain_config
    cfg = pretrain_config("imagenome", batch_size=batch_size, **kw)
    cfg.set_path("model.text.bert_config", dict(vocab_size=2000, hidden_size=768, num_hidden_layers=1,
                                                num_attention_heads=12, intermediate_size=256))
    return cfg


def test_config_missing_keys_read_as_none():
    from gloria.config import Config
    cfg = Config({"model": {"gloria": {"temp1": 4.0}}})
    assert cfg.model.gloria.temp1 == 4.0
    assert cfg.model.norm is None and cfg.model.gloria.no_attn_loss_weight is None
    assert "image_transformer" not in cfg.model.keys()
    cfg.set_path("train.scheduler.interval", "step")
    assert cfg.train.scheduler.interval == "step"


def test_reference_yaml_loads(tmp_path):
    from gloria.config import load_config
    p = tmp_path / "c.yaml"
    p.write_text("phase: 'pretrain'\nmodel:\n  gloria:\n    temp1: 4.0\n    no_attn_vec: false\ntrain:\n  batch_size: 48\n")
    cfg = load_config(str(p), {"train.batch_size": 8})
    assert cfg.phase == "pretrain" and cfg.train.batch_size == 8 and cfg.model.gloria.no_attn_vec is False


def test_wordpiece_slots_match_reference_outputs(golden):
    """aggregate_tokens restated as host slotting + segment-sum == reference BertEncoder.forward"""
    from gloria.models import text_model as tm
    g = golden("text")
    ids, hidden, vocab = gi.text_inputs()
    enc = tm.BertEncoder.__new__(tm.BertEncoder)
    torch.nn.Module.__init__(enc)
    enc.vocab = tm.Vocab.from_dict(vocab)
    summed = torch.stack([torch.from_numpy(h) for h in hidden[-4:]]).sum(0)
    words, sents = enc.aggregate_tokens(summed, torch.from_numpy(ids))
    np.testing.assert_allclose(words.permute(0, 2, 1).numpy(), g["text/word_emb"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(words.mean(1).numpy(), g["text/sent_emb"], rtol=1e-5, atol=1e-5)
    assert ["\t".join(s) for s in sents] == list(g["text/sents"])
    want = [sum(1 for w in s.split("\t") if not w.startswith("[")) + 1 for s in g["text/sents"]]
    assert sents.cap_lens == want


def test_wordpiece_without_sep_drops_last_word():
    from gloria.models import text_model as tm
    v = tm.Vocab(["[PAD]", "[CLS]", "[SEP]", "a", "##b", "c"])
    ids = np.array([[1, 3, 4, 5, 5, 4]])            # no [SEP]: last open word ("c##b") is never flushed
    dst, starts, n = tm.wordpiece_slots(ids, v)
    assert n.tolist() == [3] and dst.tolist() == [[0, 1, 1, 2, -1, -1]]


def test_builder_and_checkpoint_keys(tmp_path):
    from gloria import builder
    cfg = tiny_cfg()
    dm = builder.build_data_module(cfg)
    model = builder.build_lightning_model(cfg, dm)
    keys = list(model.state_dict())
    assert "gloria.img_encoder.local_embedder.weight" in keys
    assert "gloria.img_encoder.global_embedder.bias" in keys
    assert any(k.startswith("gloria.img_encoder.model.layer3.5.conv3") for k in keys)
    assert "gloria.text_encoder.model.embeddings.word_embeddings.weight" in keys
    assert "gloria.text_encoder.model.encoder.layer.0.attention.self.query.weight" in keys
    opt = model.configure_optimizers()
    assert opt["optimizer"].defaults["betas"] == (0.5, 0.999)
    assert opt["lr_scheduler"]["monitor"] == "val_loss"
    path = tmp_path / "last.ckpt"
    torch.save(model.checkpoint(), path)
    again = builder.build_lightning_model(cfg, dm, ckpt=str(path))
    for (k1, v1), (k2, v2) in zip(model.state_dict().items(), again.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_no_attn_vec_parameter_name():
    from gloria import builder
    cfg = tiny_cfg(**{"model.gloria.no_attn_vec": True})
    g = builder.build_gloria_model(cfg)
    assert "no_attn_vec" in dict(g.named_parameters())


def test_synthetic_batch_contract():
    from gloria.datasets.synthetic import make_batch
    b = make_batch(6, segmentation=True)
    assert b["imgs"].shape == (6, 3, 224, 224) and b["imgs"].dtype == torch.float32
    assert float(b["imgs"].min()) >= -1 and float(b["imgs"].max()) <= 1
    assert b["caption_ids"].shape == (6, 97) and b["caption_ids"].dtype == torch.int64
    lens = b["cap_lens"].tolist()
    assert lens == sorted(lens, reverse=True)
    assert (b["caption_ids"][:, 0] == 101).all()
    frac = b["segmentation_labels"].float().mean((1, 2))
    assert (frac > 0.03).all() and (frac < 0.5).all()


def test_loss_refuses_cpu_tensors():
    """No CPU / eager fallback: the product path raises on CPU tensors (with or without the regularisers)."""
    from gloria.loss import gloria_loss as GL
    with pytest.raises(RuntimeError):
        GL.global_loss(torch.zeros(2, 64), torch.zeros(2, 64))
    with pytest.raises(RuntimeError):
        GL.local_loss(torch.zeros(2, 64, 3, 3), torch.zeros(2, 64, 5), [2, 2])
    with pytest.raises(RuntimeError):
        GL.local_loss(torch.zeros(2, 64, 3, 3), torch.zeros(2, 64, 5), [2, 2], no_attn_loss_weight=1.0)


def test_reference_shaped_checkpoint_loads_and_resumes(tmp_path):
    """A checkpoint as the REFERENCE writes it (builder.py:35-50): `gloria.*` keys, BatchNorm
    `num_batches_tracked`, and transformers==4.2.1's persistent `embeddings.position_ids` buffer - through
    build_gloria_from_ckpt, load_from_checkpoint(cfg=cfg) and Trainer.save_checkpoint -> resume (weights-only
    loader): parameters, optimizer state and global_step survive."""
    from gloria import builder
    from gloria.trainer import Trainer
    cfg = tiny_cfg()
    dm = builder.build_data_module(cfg)
    model = builder.build_lightning_model(cfg, dm)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    assert "gloria.img_encoder.model.bn1.num_batches_tracked" in sd
    sd["gloria.text_encoder.model.embeddings.position_ids"] = torch.arange(512).unsqueeze(0)
    path = tmp_path / "ref_like.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": cfg.to_dict(), "epoch": 3, "global_step": 77}, path)
    g = builder.build_gloria_from_ckpt(str(path))
    for k, v in g.state_dict().items():
        assert torch.equal(v, sd["gloria." + k]), k
    again = builder.build_lightning_model(cfg, dm, ckpt=str(path))
    assert all(torch.equal(v, sd[k]) for k, v in again.state_dict().items())

    # resume: one optimisation step on a stand-in loss, save, load into a fresh trainer + model
    tr = Trainer(cfg, device="cpu", precision=32)
    tr.setup(model)
    loss = sum((p.float() ** 2).sum() for p in list(model.parameters())[:6])
    loss.backward()
    tr.optimizer.step()
    tr.global_step, model.current_epoch = 5, 2
    ck = tmp_path / "last.ckpt"
    tr.save_checkpoint(model, str(ck))
    m2 = builder.build_lightning_model(cfg, dm)
    tr2 = Trainer(cfg, device="cpu", precision=32)
    tr2.setup(m2)
    tr2.resume(m2, str(ck))
    assert tr2.global_step == 5 and m2.current_epoch == 2
    for (k1, v1), (k2, v2) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    s1, s2 = tr.optimizer.state_dict()["state"], tr2.optimizer.state_dict()["state"]
    assert s1.keys() == s2.keys() and len(s1) > 0
    for k in s1:
        assert torch.equal(s1[k]["exp_avg"], s2[k]["exp_avg"]) and torch.equal(s1[k]["exp_avg_sq"], s2[k]["exp_avg_sq"])


def test_unloadable_hyper_parameters_are_reported(tmp_path):
    """hyper_parameters pickled as arbitrary objects (the reference stores OmegaConf) are refused by the
    weights-only loader: the error says what to do, nothing from the file is executed"""
    from gloria import builder
    import fractions
    path = tmp_path / "omegaconf_like.ckpt"
    torch.save({"state_dict": {}, "hyper_parameters": fractions.Fraction(1, 3)}, path)   # any non-allow-listed class
    with pytest.raises(RuntimeError, match="weights-only loader refused"):
        builder.build_gloria_from_ckpt(str(path))


def test_plan_rowflags_match_python_restatement():
    """glr_plan_rowflags: run starts / ends / owners per tile and lane half against a direct Python walk of the
    slots (row k of half h = slots with ((w >> 2) & 1) == h in word order)"""
    from gloria import _native as N
    rng = np.random.default_rng(5)
    checked = 0
    for trial in range(20):
        n = int(rng.integers(1, 40))
        lens = [int(x) for x in rng.integers(1, 129 if trial % 3 == 0 else 41, size=n)]
        plan = N.TilePlan(lens, "cpu")
        if not plan.n_pair:
            continue
        flags = plan.rowflags_host
        slot0 = plan.sent_slot0_host
        want = np.zeros_like(flags)
        row_of = lambda w: 16 * (w >> 5) + 4 * ((w & 31) >> 3) + (w & 3)   # noqa: E731
        for i, ln in enumerate(lens):
            k = (ln + 63) // 64
            for sub in range(k):
                t = slot0[i] // 64 + sub
                a = slot0[i] % 64 if k == 1 else 0
                e = a + ln if k == 1 else min(64, ln - sub * 64)
                for hh in range(2):
                    ws = [w for w in range(a, e) if ((w >> 2) & 1) == hh]
                    if not ws:
                        continue
                    want[t, hh] |= 1 << row_of(ws[0])
                    want[t, 2 + hh] |= 1 << row_of(ws[-1])
                    if sub == 0 and ws[0] == a:
                        want[t, 4 + hh] |= 1 << row_of(ws[0])
        assert np.array_equal(flags, want), (lens, flags, want)
        owners = sum(bin(int(x)).count("1") for x in flags[:, 4:6].reshape(-1))
        assert owners == n                      # exactly one owner run per sentence
        checked += 1
    assert checked >= 5


def test_planner_caps_sentences_per_tile_when_that_saves_work_items():
    """glr_plan_tiles with pairing in view: every sentence keeps cap_lens consecutive slots inside one tile, pairs hold
    at most 8 sentences, ordinary tiles are contiguous behind the multi-tile runs, and at the bench's caption lengths
    no tile is left unpaired (plain first fit leaves 6 .. 7 crowded tiles that cannot pair)."""
    from gloria import _native as N
    for seed in (1234, 1, 2):
        rng = np.random.default_rng(seed)
        lens = (np.sort(rng.integers(4, 40, size=256))[::-1] + 1).astype(int)
        p = N.TilePlan(lens, "cpu")
        assert p.n_single <= 1 and p.n_pair >= 46
        slot0 = p.sent_slot0.numpy() if hasattr(p.sent_slot0, "numpy") else np.asarray(p.sent_slot0)
        used = np.zeros(p.n_tiles * 64, dtype=int)
        for i, n in enumerate(lens):
            assert slot0[i] // 64 == (slot0[i] + n - 1) // 64           # inside one tile
            used[slot0[i]:slot0[i] + n] += 1
        assert used.max() == 1 and used.sum() == lens.sum()
        tf, pairs = p.tile_first.numpy(), p.pair_tile.numpy()
        for t in pairs[:p.n_pair]:
            assert tf[t + 2] - tf[t] <= N.MAX_PAIR_SEG
        # without pairing (the fp32 mode's 32-word tiles) the planner is plain first fit: no more tiles than needed + slack
        q = N.TilePlan(lens, "cpu", capacity=32, allow_pairs=False)
        assert q.n_pair == 0 and q.n_tiles * 32 >= lens.sum()
    mixed = np.random.default_rng(5).integers(1, 300, size=40)
    p = N.TilePlan(mixed, "cpu")
    nsub = p.tile_nsub.numpy()
    first_ordinary = int(np.argmax(nsub == 0)) if (nsub == 0).any() else len(nsub)
    assert (nsub[first_ordinary:] == 0).all()                            # ordinary tiles are contiguous at the end


def test_chexpert_pretrain_config_builds_and_steps_on_cpu():
    """BASELINE.json configs[0]: chexpert_pretrain_config.yaml semantics (no no_attn_vec, weights 1 / 1, temps 4 / 5 / 10,
    sentence-level captions) instantiated through the builder; one CPU forward of the encoders + the oracle loss (the
    reference-structured CPU path: plumbing only, no GPU)."""
    import torch
    from gloria import builder
    from gloria.config import pretrain_config
    from gloria.datasets.synthetic import make_batch
    from oracle import gloria_oracle as orc
    cfg = pretrain_config("chexpert", batch_size=4)
    assert cfg.data.dataset == "chexpert" and cfg.data.text.full_report is False
    g = cfg.model.gloria
    assert (g.local_loss_weight, g.global_loss_weight, g.temp1, g.temp2, g.temp3) == (1.0, 1.0, 4.0, 5.0, 10.0)
    assert not g.no_attn_vec and g.no_attn_loss_weight is None and g.segmentation_loss_weight is None
    cfg.set_path("model.text.bert_config", dict(num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
    torch.manual_seed(0)
    model = builder.build_lightning_model(cfg, builder.build_data_module(cfg))
    assert model.gloria.no_attn_vec is None
    batch = make_batch(4, seed=2)
    il, ig, tl, tg, sents = model.gloria(batch)
    assert il.shape == (4, 768, 19, 19) and ig.shape == (4, 768) and tl.shape == (4, 768, 97) and tg.shape == (4, 768)
    loss, maps = orc.calc_loss(il, ig, tl, tg, sents, temp1=g.temp1, temp2=g.temp2, temp3=g.temp3)
    assert torch.isfinite(loss) and len(maps) == 4
    opt = model.configure_optimizers()["optimizer"]
    assert opt.param_groups[0]["betas"] == (0.5, 0.999) and abs(opt.param_groups[0]["lr"] - 5e-5) < 1e-12


def test_hipgraph_gate_and_consistency_check(monkeypatch):
    """gloria/hipgraph.py on the CPU: the runtime flag is put in the environment (and graphs are refused when the user
    forces packet capture on), and the replay check flags NaN / drifted replays"""
    import importlib
    import warnings
    import torch
    from gloria import hipgraph
    assert os.environ.get(hipgraph.ENV) == "0" and hipgraph.SAFE          # conftest / package import put it there
    monkeypatch.setenv(hipgraph.ENV, "1")
    assert importlib.reload(hipgraph).SAFE is False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert hipgraph.usable("image encoder") is False and "runs eagerly" in str(w[-1].message)
    monkeypatch.delenv(hipgraph.ENV)
    assert importlib.reload(hipgraph).SAFE is True and os.environ[hipgraph.ENV] == "0"      # CPU box: torch.cuda never initialised
    ref = [torch.arange(6.0).reshape(2, 3), torch.ones(4)]
    good = [t.clone() for t in ref]
    near = [t * 1.01 for t in ref]
    bad = [ref[0].clone(), torch.tensor([1.0, float("nan"), 1.0, 1.0])]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert hipgraph.consistent("x", ref, [good, near])
        assert not hipgraph.consistent("x", ref, [good, bad]) and "differs from the eager pass" in str(w[-1].message)
        assert not hipgraph.consistent("x", ref, [[t * 2 for t in ref]])
        assert not hipgraph.consistent("x", [torch.zeros(3)], [[torch.zeros(3)]])       # no signal to compare against
