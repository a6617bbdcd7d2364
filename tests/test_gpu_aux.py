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


def test_wordpiece_segsum_matches_reference(golden):
    from gloria.models import text_model as tm
    g = golden("text")
    ids, hidden, vocab = gi.text_inputs(D=64)      # kernel needs D % 64 == 0: separate golden below
    v = tm.Vocab.from_dict(vocab)
    dst, starts, n_words = tm.wordpiece_slots(ids, v)
    layers = [torch.from_numpy(h).to(DEV).requires_grad_(True) for h in hidden[-4:]]
    dst_d = torch.from_numpy(dst.astype(np.int32)).to(DEV)
    word, sent = tm.WordpieceSegSumFn.apply(dst_d, False, *layers)
    # torch restatement (validated against the reference's outputs in tests/test_host_logic.py)
    enc = tm.BertEncoder.__new__(tm.BertEncoder)
    torch.nn.Module.__init__(enc)
    enc.vocab = v
    cl = [torch.from_numpy(h).requires_grad_(True) for h in hidden[-4:]]
    ref_words, _ = enc.aggregate_tokens(torch.stack(cl).sum(0), torch.from_numpy(ids))
    np.testing.assert_allclose(word.detach().cpu().numpy(), ref_words.permute(0, 2, 1).detach().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sent.detach().cpu().numpy(), ref_words.mean(1).detach().numpy(), rtol=1e-5, atol=1e-6)
    gw = torch.from_numpy(gi.normal(9, *word.shape))
    gs = torch.from_numpy(gi.normal(10, *sent.shape))
    ((word * gw.to(DEV)).sum() + (sent * gs.to(DEV)).sum()).backward()
    ((ref_words.permute(0, 2, 1) * gw).sum() + (ref_words.mean(1) * gs).sum()).backward()
    for a, b in zip(layers, cl):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-5, atol=1e-6)


def test_wordpiece_segsum_golden_shape(golden):
    """the exact fixture of the reference run (D = 48) through the encoder's torch path on the GPU"""
    from gloria.models import text_model as tm
    g = golden("text")
    ids, hidden, vocab = gi.text_inputs()
    enc = tm.BertEncoder.__new__(tm.BertEncoder)
    torch.nn.Module.__init__(enc)
    enc.vocab = tm.Vocab.from_dict(vocab)
    summed = torch.stack([torch.from_numpy(h).to(DEV) for h in hidden[-4:]]).sum(0)
    words, sents = enc.aggregate_tokens(summed, torch.from_numpy(ids))
    np.testing.assert_allclose(words.permute(0, 2, 1).cpu().numpy(), g["text/word_emb"], rtol=1e-5, atol=1e-5)


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
