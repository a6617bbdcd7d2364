"""Exact selection: the HIP radix-select kernels must pick the SAME elements as the
reference's CPU code - np.argsort(x)[::-1][:k] (retrival_model.py:118) and
torch.topk(x, n - k, largest=False).values.max() (callbacks.py:56).  Bit exact: indices and values are compared
with array_equal, not allclose."""

import numpy as np
import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def sel():
    from gloria import select
    return select


@pytest.mark.parametrize("n,k", [(1, 1), (7, 3), (1000, 5), (4097, 1), (50176, 1024), (50176, 50), (300001, 17)])
def test_topk_matches_argsort(n, k):
    x = gi.normal(100 + n, 3, n, std=3.0).astype(np.float32)
    idx, val = sel().topk_desc(torch.from_numpy(x).to(DEV), k)
    want = np.stack([np.argsort(r, kind="stable")[::-1][:k] for r in x])
    assert np.array_equal(idx.cpu().numpy(), want)
    assert np.array_equal(val.cpu().numpy(), np.take_along_axis(x, want, 1))


def test_topk_ties_zero_signs_and_extremes():
    x = np.array([[1.0, 5.0, 5.0, -0.0, 0.0, 5.0, -np.inf, np.inf, -3.0, 1.0, 1e-45, -1e-45]], dtype=np.float32)
    idx, _ = sel().topk_desc(torch.from_numpy(x).to(DEV), x.shape[1])
    want = np.argsort(x[0], kind="stable")[::-1]          # equal values: larger index first; -0.0 == 0.0
    assert np.array_equal(idx.cpu().numpy()[0], want)


@pytest.mark.parametrize("n", [10, 50176, 224 * 224 * 3 + 5])
def test_kth_value_matches_torch_topk(n):
    x = torch.from_numpy(gi.normal(7 + n, 4, n, std=1.0).astype(np.float32))
    x[0, : n // 3] = x[0, 0]                               # heavy duplicates
    for p in (0.05, 0.1, 0.2, 0.3, 0.999):
        k = n - int(n * p)
        if k < 1:
            continue
        want = torch.topk(x, k, largest=False).values.max(-1).values
        got = sel().kth_value(x.to(DEV), k).cpu()
        assert torch.equal(got, want), (n, p)


def test_localization_metrics_match_cpu_restatement():
    """`Metrics` of callbacks.py:26-70 restated on the CPU with the overlay MATERIALISED (nn.Upsample nearest, :319)
    against the product's cell-based GPU evaluation that never forms it.  Threshold, counts-based metrics: exact;
    AUROC / AP against scikit-learn on the 50 176 pixels (the definitions torchmetrics ported); entropy against
    torch.distributions.Categorical exactly as callbacks.py:16-19.  Maps 3 and 4 hold tied cell values."""
    from sklearn.metrics import average_precision_score, roc_auc_score
    from torch.distributions.categorical import Categorical
    from gloria.lightning.callbacks import Metrics
    g = torch.Generator().manual_seed(5)
    maps = torch.rand(6, 19, 19, generator=g)
    maps = maps / maps.sum((1, 2), keepdim=True) * torch.tensor([1.0, 0.9, 0.7, 1.0, 0.5, 1.0]).view(6, 1, 1)
    maps[3] = (maps[3] * 2000).round() / 2000                 # heavy ties between cells
    maps[4, 5:9] = maps[4, 5, 0]
    H, W = 224, 224
    overlay = torch.nn.Upsample(size=(H, W))(maps[:, None])[:, 0]
    label = torch.zeros(6, H, W, dtype=torch.bool)
    for i in range(5):                                        # last label stays empty
        label[i, 20 * i:20 * i + 60, 30:30 + 25 * (i + 1)] = True
    out = Metrics()(maps.to(DEV), label.to(DEV), curves=True)
    total = H * W
    for i in range(6):
        preds, tg = overlay[i].reshape(-1), label[i].reshape(-1)
        flat = maps[i].reshape(-1)
        ent = Categorical(torch.cat([(1 - flat.sum(-1)).unsqueeze(0), flat], 0)).entropy()
        np.testing.assert_allclose(float(out["attn_entropy"][i]), float(ent), rtol=1e-6)
        np.testing.assert_allclose(float(out["no_attn_weight"][i]), float(1 - flat.sum(-1)), atol=1e-6)
        if tg.sum() == 0:
            for k, v in out.items():
                if k.startswith(("precision", "recall", "f1", "iou", "auroc", "avg")):
                    assert torch.isnan(v[i]), k
            continue
        np.testing.assert_allclose(float(out["auroc"][i]), roc_auc_score(tg.numpy(), preds.numpy()), rtol=1e-9)
        np.testing.assert_allclose(float(out["avg_precision"][i]), average_precision_score(tg.numpy(), preds.numpy()),
                                   rtol=1e-9)
        fpr, tpr, _ = out["roc_curve"][i]
        assert float(fpr[0]) == 0 and float(tpr[0]) == 0 and float(fpr[-1]) == 1 and float(tpr[-1]) == 1
        for p in (.05, .1, .2, .3):
            top_k = int(total * p)
            thr = torch.topk(preds, total - top_k, largest=False).values.max()
            assert float(out["threshold_at_%f" % p][i]) == float(thr)             # the same float
            ge, gt = preds >= thr, preds > thr                                     # torchmetrics binarises with >=
            tp = float((ge & tg).sum())
            pr, re = tp / float(ge.sum()), tp / float(tg.sum())
            np.testing.assert_allclose(float(out["precision_at_%f" % p][i]), pr, rtol=1e-12)
            np.testing.assert_allclose(float(out["recall_at_%f" % p][i]), re, rtol=1e-12)
            np.testing.assert_allclose(float(out["f1_at_%f" % p][i]), 2 * pr * re / (pr + re), rtol=1e-12)
            iou = float((gt & tg).sum()) / float((gt | tg).sum())                  # callbacks.py:59-60, strict mask
            np.testing.assert_allclose(float(out["iou_at_%f" % p][i]), iou, rtol=1e-12)


@pytest.mark.parametrize("shape", [(224, 224, 19, 19), (300, 257, 19, 19), (97, 131, 7, 5), (19, 19, 19, 19)])
def test_cell_counts_match_materialised_upsample(shape):
    """glr_cell_counts against counting on the explicitly upsampled index map (torch's nearest rule)"""
    H, W, ih, iw = shape
    rng = np.random.default_rng(H * 1000 + W)
    lab = torch.from_numpy(rng.random((3, H, W)) < 0.3)
    cnt, npix = sel().cell_counts(lab.to(DEV), ih, iw)
    cell = torch.arange(ih * iw, dtype=torch.float32).view(1, 1, ih, iw)
    idx = torch.nn.Upsample(size=(H, W))(cell)[0, 0].long().reshape(-1)
    for b in range(3):
        want_n = torch.bincount(idx, minlength=ih * iw)
        want_c = torch.bincount(idx, weights=lab[b].reshape(-1).double(), minlength=ih * iw).long()
        assert torch.equal(npix[b].cpu(), want_n) and torch.equal(cnt[b].cpu(), want_c)


def test_retriever_ranking_matches_reference_structured_cpu():
    """Retriver.retrieve on fixed embeddings: local similarity (words 1..n of the [CLS]-stripped embeddings,
    retrival_model.py:127-166), global cosine (:101-104), z-normalised mix (:110-115), argsort ranking (:118)."""
    from gloria.models.retrival_model import Retriver
    from oracle import gloria_oracle as orc
    N_T, D, L = 40, 768, 30
    img_l, img_g = gi.normal(901, 1, D, 19, 19), gi.normal(902, 1, D)
    words, txt_g = gi.normal(903, N_T, D, L), gi.normal(904, N_T, D)
    cap = [int(c) for c in np.random.default_rng(9).integers(1, L - 2, size=N_T)]

    r = Retriver.__new__(Retriver)
    r.device, r.top_k, r.targets_classes = torch.device(DEV), 7, np.arange(N_T) % 5

    class G:
        temp1, temp2, temp3 = 4.0, 5.0, 10.0
    r.gloria = G()
    t = lambda a: torch.from_numpy(a).to(DEV)
    r.targets = {"global_embeddings": t(txt_g), "local_embeddings": t(words)}
    r.cap_lens = cap
    src = {"global_embeddings": t(img_g), "local_embeddings": t(img_l)}

    # CPU restatement of the reference loop with the oracle's attention / cosine
    loc = []
    ti, tw = torch.from_numpy(img_l), torch.from_numpy(words)
    for i in range(N_T):
        word = tw[i, :, 1:cap[i] + 1].unsqueeze(0)
        wc, _ = orc.attention_fn(word, ti, 4.0)
        row = orc.cosine_similarity(word.transpose(1, 2).reshape(cap[i], D), wc.transpose(1, 2).reshape(cap[i], D))
        loc.append(float(torch.log(torch.exp(row.reshape(-1) * 5.0).sum())) * 10.0)
    loc = np.array(loc)
    a, b = img_g / np.linalg.norm(img_g, axis=1, keepdims=True), txt_g / np.linalg.norm(txt_g, axis=1, keepdims=True)
    glob = (a @ b.T)[0]
    norm = lambda x: (x - x.mean(axis=0)) / x.std(axis=0)
    for kind, want_s in (("local", loc), ("global", glob), ("both", np.stack([norm(loc), norm(glob)]).mean(0))):
        idx, cls = r.retrieve(src, kind)
        want = np.argsort(want_s)[::-1][:7]
        np.testing.assert_allclose(r.similarities(kind).cpu().numpy(), want_s, rtol=2e-4, atol=2e-4)
        assert np.array_equal(idx, want), kind             # same ranking as the CPU path
        assert np.array_equal(cls, r.targets_classes[want])


def test_get_similarities_and_zero_shot_vs_oracle():
    """gloria.gloria.get_similarities / zero_shot_classification (ref gloria/gloria.py:184-275) on a small
    random model: local = max-over-words variant (gloria_model.py:171-207), global = cosine, both = mean."""
    from gloria import builder, gloria as api
    from gloria.config import pretrain_config
    from gloria.datasets.synthetic import make_batch
    from oracle import gloria_oracle as orc
    cfg = pretrain_config("imagenome", batch_size=4)
    cfg.set_path("model.text.bert_config", dict(vocab_size=28996, num_hidden_layers=1, hidden_dropout_prob=0.0,
                                                attention_probs_dropout_prob=0.0))
    torch.manual_seed(3)
    model = builder.build_gloria_model(cfg).to(DEV).eval()
    b = make_batch(4, seed=21)
    imgs = b["imgs"].to(DEV)
    with torch.no_grad():
        _, _, sents = model.text_encoder_forward(b["caption_ids"].to(DEV), b["attention_mask"].to(DEV),
                                                 b["token_type_ids"].to(DEV))
    cap = [c - 1 for c in model._cap_lens(sents)]               # words without [CLS] (what process_text stores)
    txts = {k: b[k].to(DEV) for k in ("caption_ids", "attention_mask", "token_type_ids")}
    txts["cap_lens"] = cap
    with torch.no_grad():
        il, ig = model.image_encoder_forward(imgs)
        tl, tg, _ = model.text_encoder_forward(txts["caption_ids"], txts["attention_mask"], txts["token_type_ids"])
    want_l = orc.local_similarities_inference(il.float().cpu(), tl.float().cpu(), cap).numpy()
    a, t = ig.float().cpu().numpy(), tg.float().cpu().numpy()
    want_g = (a / np.linalg.norm(a, axis=1, keepdims=True)) @ (t / np.linalg.norm(t, axis=1, keepdims=True)).T
    np.testing.assert_allclose(api.get_similarities(model, imgs, txts, "local"), want_l, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(api.get_similarities(model, imgs, txts, "global"), want_g, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(api.get_similarities(model, imgs, txts, "both"), (want_l + want_g) / 2, rtol=1e-4, atol=1e-4)
    sims, names = api.zero_shot_classification(model, imgs, {"c0": txts, "c1": txts})
    both = ((want_l + want_g) / 2).max(1)
    ref = np.stack([both, both], 1)
    ref = (ref - ref.mean(0)) / ref.std(0)
    assert names == ["c0", "c1"]
    np.testing.assert_allclose(sims, ref, rtol=2e-3, atol=2e-3)
    with pytest.raises(RuntimeError):
        api.get_similarities(model, imgs, ["raw text"])
