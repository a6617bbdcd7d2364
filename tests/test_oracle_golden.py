"""Pins the CPU oracle (oracle/gloria_oracle.py) to outputs of the REAL reference.

The fixtures under tests/golden were produced by oracle/gen_golden.py, which imports
/root/reference/gloria/loss/gloria_loss.py and .../models/text_model.py in the build
container.  Inputs are regenerated here from the same seeded streams (golden_inputs.py).
The reference holds no tests of its own for this path.
"""

import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import gloria_oracle as orc

TOL = dict(rtol=2e-5, atol=2e-6)


def check(golden, key, arr, rtol=2e-5, atol=2e-6):
    arr = arr.detach().numpy() if torch.is_tensor(arr) else np.asarray(arr)
    assert tuple(golden[key + ".shape"]) == tuple(arr.shape), key
    np.testing.assert_allclose(gi.subsample(arr), golden[key + ".sample"], rtol=rtol, atol=atol, err_msg=key)
    np.testing.assert_allclose(gi.checksums(arr), golden[key + ".sums"], rtol=1e-4, atol=1e-4, err_msg=key)


def t(a, grad=False):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return x.requires_grad_(True) if grad else x


@pytest.mark.parametrize("name", list(gi.ATTN_CASES))
def test_attention_fn(golden, name):
    g = golden("attention")
    q, ctx, temp1, na = gi.attn_inputs(name)
    wc, attn = orc.attention_fn(t(q), t(ctx), temp1, no_attn_vec=None if na is None else t(na))
    check(g, f"attn/{name}/weighted", wc)
    check(g, f"attn/{name}/map", attn)


@pytest.mark.parametrize("name", list(gi.ATTN_CASES))
def test_attention_fn_output_grads(golden, name):
    """gradients of attention_fn's OWN outputs (weighted context and maps) by the reference's autograd"""
    g = golden("attention_grad")
    q, ctx, temp1, na = gi.attn_inputs(name)
    tq, tc = t(q, True), t(ctx, True)
    tna = None if na is None else t(na, True)
    wc, attn = orc.attention_fn(tq, tc, temp1, no_attn_vec=tna)
    gw, ga = gi.attn_upstream(name, tuple(wc.shape), tuple(attn.shape))
    ((wc * t(gw)).sum() + (attn * t(ga)).sum()).backward()
    check(g, f"attn_grad/{name}/grad_query", tq.grad, rtol=2e-4, atol=2e-6)
    check(g, f"attn_grad/{name}/grad_context", tc.grad, rtol=2e-4, atol=2e-6)
    if tna is not None:
        check(g, f"attn_grad/{name}/grad_no_attn", tna.grad, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("name", list(gi.LOCAL_CASES))
def test_local_loss_and_grads(golden, name):
    g = golden("local")
    cfg = gi.LOCAL_CASES[name]
    img, words, cap_lens, na = gi.local_inputs(name)
    assert list(g[f"local/{name}/cap_lens"]) == cap_lens
    timg, twords = t(img, True), t(words, True)
    tna = None if na is None else t(na, True)
    aux = cfg["aux"] or (None, None, None)
    l0, l1, nal, kl, ent, maps, sim = orc.local_loss(
        timg, twords, cap_lens, agg=cfg["agg"], no_attn_vec=tna, no_attn_loss_weight=aux[0],
        attention_divergence_loss_weight=aux[1], attention_entropy_loss_weight=aux[2], return_sim=True)
    got = np.array([float(x) for x in (l0, l1, nal, kl, ent)])
    np.testing.assert_allclose(got, g[f"local/{name}/losses"], rtol=2e-5, atol=2e-6)
    check(g, f"local/{name}/maps", torch.cat([m.reshape(-1) for m in maps]))
    check(g, f"local/{name}/sim", sim, rtol=2e-5, atol=2e-5)
    (l0 + l1 + nal + kl + ent).backward()
    check(g, f"local/{name}/grad_img", timg.grad, rtol=2e-4, atol=2e-7)
    check(g, f"local/{name}/grad_words", twords.grad, rtol=2e-4, atol=2e-7)
    if tna is not None:
        check(g, f"local/{name}/grad_no_attn", tna.grad, rtol=2e-4, atol=2e-7)


@pytest.mark.parametrize("name", list(gi.LOCAL_CASES))
def test_batched_formulation_matches(golden, name):
    """the masked/batched restatement gives the reference's similarity matrix too"""
    g = golden("local")
    cfg = gi.LOCAL_CASES[name]
    img, words, cap_lens, na = gi.local_inputs(name)
    sim = orc.local_similarity_matrix(t(img), t(words), cap_lens, agg=cfg["agg"],
                                      no_attn_vec=None if na is None else t(na))
    check(g, f"local/{name}/sim", sim, rtol=2e-5, atol=2e-5)


def test_sim_matrix_b64_and_sharded(golden):
    g = golden("sim")
    name = "s1_b64_mix"
    img, words, cap_lens, _ = gi.local_inputs(name)
    assert list(g[f"sim/{name}/cap_lens"]) == cap_lens
    ref = g[f"sim/{name}/sim"]
    sim = orc.local_similarity_matrix(t(img), t(words), cap_lens)
    np.testing.assert_allclose(sim.numpy(), ref, rtol=2e-5, atol=5e-5)
    l0, l1 = orc.dual_ce(sim)
    np.testing.assert_allclose([float(l0), float(l1)], g[f"sim/{name}/losses"], rtol=1e-5)
    # 4-way sharded restatement (fake process group = list of shards) gives the same matrix
    sh = orc.sharded_local_similarity(t(img), t(words), cap_lens, world=4)
    np.testing.assert_allclose(sh.numpy(), ref, rtol=2e-5, atol=5e-5)


@pytest.mark.parametrize("name", list(gi.GLOBAL_CASES))
def test_global_loss(golden, name):
    g = golden("global")
    img, txt = gi.global_inputs(name)
    ti, tt = t(img, True), t(txt, True)
    l0, l1 = orc.global_loss(ti, tt)
    np.testing.assert_allclose([float(l0), float(l1)], g[f"global/{name}/losses"], rtol=1e-5)
    (l0 + l1).backward()
    zero_row = gi.GLOBAL_CASES[name][2]
    if zero_row is not None:
        # the eps clamp turns the zero row's gradient into (temp3/eps) * sum_i dsim[b,i] * T_i with
        # sum_i dsim ~ 0: a 1e9-scaled cancellation, only meaningful relative to its own scale
        got, ref = ti.grad.numpy(), g[f"global/{name}/grad_img.sample"].reshape(ti.shape)
        scale = np.abs(ref[zero_row]).max()
        np.testing.assert_allclose(got[zero_row] / scale, ref[zero_row] / scale, atol=5e-3)
        keep = np.arange(ti.shape[0]) != zero_row
        np.testing.assert_allclose(got[keep], ref[keep], rtol=1e-4, atol=1e-7)
    else:
        check(g, f"global/{name}/grad_img", ti.grad, rtol=1e-4, atol=1e-7)
    check(g, f"global/{name}/grad_txt", tt.grad, rtol=1e-4, atol=1e-7)


def test_text_postprocess(golden):
    g = golden("text")
    ids, hidden, vocab = gi.text_inputs()
    word, sent, sents = orc.text_postprocess([t(h) for h in hidden], t(ids), vocab)
    np.testing.assert_allclose(word.numpy(), g["text/word_emb"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sent.numpy(), g["text/sent_emb"], rtol=1e-5, atol=1e-6)
    assert ["\t".join(s) for s in sents] == list(g["text/sents"])
    # cap_lens rule (gloria_model.py:107-109): bracket tokens are not counted, +1
    lens = orc.cap_lens_from_sents(sents)
    assert lens == [sum(1 for w in s if not w.startswith("[")) + 1 for s in sents]


def test_text_postprocess_wide(golden):
    """the same post-processing at the widths of the HIP kernel's tile (64) and of BERT-base (768)"""
    g = golden("text_wide")
    ids, hidden, vocab = gi.text_inputs(D=64)
    hs = [t(h, True) for h in hidden]
    word, sent, _ = orc.text_postprocess(hs, t(ids), vocab)
    np.testing.assert_allclose(word.detach().numpy(), g["text64/word_emb"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sent.detach().numpy(), g["text64/sent_emb"], rtol=1e-5, atol=1e-6)
    gw, gs = gi.normal(9, *word.shape), gi.normal(10, *sent.shape)
    ((word * t(gw)).sum() + (sent * t(gs)).sum()).backward()
    for k in range(1, 5):
        np.testing.assert_allclose(hs[-k].grad.numpy(), g[f"text64/grad_hidden_m{k}"], rtol=1e-5, atol=1e-6)
    ids, hidden, vocab = gi.text_inputs(D=768)
    word, sent, _ = orc.text_postprocess([t(h) for h in hidden], t(ids), vocab)
    check(g, "text768/word_emb", word, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sent.numpy(), g["text768/sent_emb"], rtol=1e-5, atol=1e-6)


def test_float64_agrees_with_float32():
    """the oracle in float64 stays within 1e-4 of its float32 run (headroom for the 1e-4 bar)"""
    img, words, cap_lens, _ = gi.local_inputs("l1_tiny")
    a = orc.local_loss(t(img), t(words), cap_lens, return_sim=True)
    b = orc.local_loss(t(img).double(), t(words).double(), cap_lens, return_sim=True)
    np.testing.assert_allclose(a[-1].numpy(), b[-1].numpy(), rtol=1e-5, atol=1e-4)
