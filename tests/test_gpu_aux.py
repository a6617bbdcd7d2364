"""GPU parity of the auxiliary kernels: standalone cosine (a-2), word-piece segment-sum K5 (a-6, against
the reference's BertEncoder outputs in tests/golden/text.npz) and the attention-supervision loss K4 (a-5)."""

import numpy as np
import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_cosine_matches_oracle_with_grads():
    from gloria.loss import gloria_loss as GL
    from oracle import gloria_oracle as orc
    x1 = gi.normal(5, 300, 768)
    x2 = gi.normal(6, 300, 768)
    x1[7] = 0.0                                    # eps clamp on the product of the norms
    a, b = torch.from_numpy(x1).to(DEV).requires_grad_(True), torch.from_numpy(x2).to(DEV).requires_grad_(True)
    out = GL.cosine_similarity(a, b)
    ra, rb = torch.from_numpy(x1).requires_grad_(True), torch.from_numpy(x2).requires_grad_(True)
    ref = orc.cosine_similarity(ra, rb)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    w = torch.from_numpy(gi.normal(7, 300))
    (out * w.to(DEV)).sum().backward()
    (ref * w).sum().backward()
    keep = np.arange(300) != 7
    np.testing.assert_allclose(a.grad.cpu().numpy()[keep], ra.grad.numpy()[keep], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(b.grad.cpu().numpy()[keep], rb.grad.numpy()[keep], rtol=1e-4, atol=1e-7)
    assert torch.isfinite(a.grad).all()


def _k5(ids, hidden, vocab, grad=False):
    """K5 through the C ABI: token -> slot indices from the host ids, last four hidden states on the GPU"""
    from gloria.models import text_model as tm
    v = tm.Vocab.from_dict(vocab)
    dst, starts, n_words = tm.wordpiece_slots(ids, v)
    layers = [torch.from_numpy(h).to(DEV).requires_grad_(grad) for h in hidden[-4:]]
    dst_d = torch.from_numpy(dst.astype(np.int32)).to(DEV)
    word, sent = tm.WordpieceSegSumFn.apply(dst_d, False, *layers)
    return word, sent, layers


def test_wordpiece_segsum_golden_d64_with_grads(golden):
    """K5 DIRECTLY against the reference's BertEncoder.forward (tests/golden/text_wide.npz, D = 64): outputs
    and the gradients of the four hidden states"""
    g = golden("text_wide")
    ids, hidden, vocab = gi.text_inputs(D=64)
    word, sent, layers = _k5(ids, hidden, vocab, grad=True)
    np.testing.assert_allclose(word.detach().cpu().numpy(), g["text64/word_emb"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sent.detach().cpu().numpy(), g["text64/sent_emb"], rtol=1e-5, atol=1e-6)
    gw = torch.from_numpy(gi.normal(9, *word.shape)).to(DEV)
    gs = torch.from_numpy(gi.normal(10, *sent.shape)).to(DEV)
    ((word * gw).sum() + (sent * gs).sum()).backward()
    for k in range(1, 5):
        np.testing.assert_allclose(layers[-k].grad.cpu().numpy(), g[f"text64/grad_hidden_m{k}"], rtol=1e-5, atol=1e-6)


def test_wordpiece_segsum_golden_d768_and_d48(golden):
    """K5 against the reference at BERT-base width (sub-sampled fixture) and at the original D = 48 fixture
    (a width that is not a multiple of the kernel's 64-feature tile)"""
    g = golden("text_wide")
    ids, hidden, vocab = gi.text_inputs(D=768)
    word, sent, _ = _k5(ids, hidden, vocab)
    w = word.cpu().numpy()
    assert tuple(g["text768/word_emb.shape"]) == w.shape
    np.testing.assert_allclose(gi.subsample(w), g["text768/word_emb.sample"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sent.cpu().numpy(), g["text768/sent_emb"], rtol=1e-5, atol=1e-6)
    g48 = golden("text")
    ids, hidden, vocab = gi.text_inputs()
    word, sent, _ = _k5(ids, hidden, vocab)
    np.testing.assert_allclose(word.cpu().numpy(), g48["text/word_emb"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sent.cpu().numpy(), g48["text/sent_emb"], rtol=1e-5, atol=1e-6)


def test_text_encoder_gpu_always_takes_k5(golden):
    """BertEncoder.forward on GPU tensors has ONE aggregation path (K5), whatever the width: the encoder's own
    post-processing on the D = 48 fixture equals the reference's outputs"""
    from gloria.models import text_model as tm
    g48 = golden("text")
    ids, hidden, vocab = gi.text_inputs()
    enc = tm.BertEncoder.__new__(tm.BertEncoder)
    torch.nn.Module.__init__(enc)
    enc.vocab = tm.Vocab.from_dict(vocab)
    enc.last_n_layers, enc.aggregate_method, enc.norm, enc.agg_tokens = 4, "sum", False, True
    enc.embedding_dim, enc.emb_local, enc.emb_global = 48, None, None
    hs = tuple(torch.from_numpy(h).to(DEV) for h in hidden)
    enc.model = lambda i, m, tt: (None, None, hs)
    calls = []
    orig = tm.WordpieceSegSumFn.apply
    tm.WordpieceSegSumFn.apply = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        word, sent, sents = enc.forward(torch.from_numpy(ids).to(DEV), None, None)
    finally:
        tm.WordpieceSegSumFn.apply = orig
    assert calls, "the GPU call did not go through K5"
    np.testing.assert_allclose(word.cpu().numpy(), g48["text/word_emb"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sent.cpu().numpy(), g48["text/sent_emb"], rtol=1e-5, atol=1e-6)
    assert ["\t".join(s) for s in sents] == list(g48["text/sents"])


def test_attention_supervision_matches_oracle():
    from gloria.datasets.synthetic import make_batch
    from gloria.loss import gloria_loss as GL
    from oracle import gloria_oracle as orc
    B, ih, iw = 6, 19, 19
    cap_lens = [17, 9, 9, 4, 2, 1]
    rng = np.random.default_rng(3)
    maps_ref = []
    flat = []
    for n in cap_lens:
        m = rng.random((1, n, ih, iw), dtype=np.float32)
        m /= m.sum((2, 3), keepdims=True)
        maps_ref.append(torch.from_numpy(m).requires_grad_(True))
        flat.append(m.reshape(-1))
    labels = make_batch(B, seed=11, segmentation=True)["segmentation_labels"]
    ref = orc.attention_supervision_loss(maps_ref, labels, 1.0)
    ref.backward()
    f = torch.from_numpy(np.concatenate(flat)).to(DEV).requires_grad_(True)
    maps = GL.split_attention_maps(f, cap_lens, ih, iw)
    out = GL.attention_supervision_loss(maps, labels.to(DEV))
    np.testing.assert_allclose(float(out), float(ref), rtol=1e-5)
    out.backward()
    want = np.concatenate([m.grad.numpy().reshape(-1) for m in maps_ref])
    np.testing.assert_allclose(f.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("dtype,no_attn", [(torch.float32, False), (torch.float32, True), (torch.bfloat16, True)])
def test_attention_supervision_gradient_through_k1(dtype, no_attn):
    """The gradient of the attention-supervision loss (gloria_model.py:143-147) reaches the embeddings through the
    diagonal attention maps: K4 emits d(loss)/d(map), K1 backward takes it as `dattn`.  Checked against autograd
    through the oracle, alone and mixed with the contrastive terms; one sentence spans several word tiles."""
    from gloria.datasets.synthetic import make_batch
    from gloria.loss import gloria_loss as GL
    from oracle import gloria_oracle as orc
    import golden_inputs as gi
    B, D, H, W, L = 6, 768, 19, 19, 97
    cap_lens = [70, 33, 17, 9, 4, 1]
    img = torch.from_numpy(gi.normal(611, B, D, H, W)).to(dtype)
    words = torch.from_numpy(gi.normal(612, B, D, L)).to(dtype)
    na = torch.from_numpy(gi.normal(613, D, std=1.0)).to(dtype) if no_attn else None
    labels = make_batch(B, seed=17, segmentation=True)["segmentation_labels"]
    for w_contrastive in (0.0, 0.3):
        ti, tw = img.detach().to(DEV).requires_grad_(True), words.detach().to(DEV).requires_grad_(True)
        tn = None if na is None else na.detach().to(DEV).requires_grad_(True)
        l0, l1, _, _, _, maps = GL.local_loss(ti, tw, cap_lens, no_attn_vec=tn)
        seg = GL.attention_supervision_loss(maps, labels.to(DEV))
        (seg + w_contrastive * (l0 + l1)).backward()
        ri, rw = img.detach().float().clone().requires_grad_(True), words.detach().float().clone().requires_grad_(True)
        rn = None if na is None else na.detach().float().clone().requires_grad_(True)
        r = orc.local_loss(ri, rw, cap_lens, no_attn_vec=rn)
        segr = orc.attention_supervision_loss(r[5], labels, 1.0)
        (segr + w_contrastive * (r[0] + r[1])).backward()
        f32 = dtype == torch.float32
        np.testing.assert_allclose(float(seg.detach()), float(segr), rtol=1e-4 if f32 else 2e-2)
        pairs = [(ti.grad, ri.grad), (tw.grad, rw.grad)] + ([] if tn is None else [(tn.grad, rn.grad)])
        for a, b in pairs:
            a, b = a.float().cpu().numpy(), b.numpy()
            if f32:
                np.testing.assert_allclose(a, b, rtol=2e-3, atol=1e-6)
            else:
                assert np.linalg.norm(a - b) / np.linalg.norm(b) < 0.05
