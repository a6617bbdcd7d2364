"""Fused dropout + residual add + LayerNorm kernels (glr_drop_add_ln_fwd / _bwd) against torch's own ops on the same
bf16 / fp32 tensors: outputs and all four gradients, with the kernel's OWN dropout mask decoded from its bit layout
(element 4 (l + 64 i) + c of a row = bit l of word 4 i + c), plus the Bernoulli rate and reproducibility of the mask."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _decode_mask(mask, R, H):
    """uint64 words [R, H/64] -> bool [R, H]"""
    w = mask.cpu().numpy().view(np.uint64).reshape(R, H // 64)
    lanes = np.arange(64, dtype=np.uint64)
    bits = ((w[:, :, None] >> lanes[None, None, :]) & np.uint64(1)).astype(bool)       # [R, word, lane]
    out = np.zeros((R, H), dtype=bool)
    for word in range(H // 64):
        i, c = word // 4, word % 4
        out[:, 4 * (np.arange(64) + 64 * i) + c] = bits[:, word, :]
    return torch.from_numpy(out)


@pytest.mark.parametrize("R,H,p", [(97, 768, 0.1), (1000, 768, 0.0), (33, 256, 0.25), (130, 1024, 0.1), (5, 512, 0.5)])
def test_drop_add_ln_matches_torch(R, H, p):
    from gloria import _native as N
    from gloria.models import fused_ln as FL
    g = torch.Generator().manual_seed(R * 7 + H)
    h = (torch.randn(R, H, generator=g) * 1.3).to(DEV).bfloat16()
    inp = (torch.randn(R, H, generator=g) * 0.8 + 0.1).to(DEV)
    w = (torch.rand(H, generator=g) + 0.5).to(DEV)
    b = (torch.randn(H, generator=g) * 0.2).to(DEV)
    d32 = torch.randn(R, H, generator=g).to(DEV)
    d16 = torch.randn(R, H, generator=g).to(DEV).bfloat16()
    L = N.lib()
    out32 = torch.empty(R, H, device=DEV)
    out16 = torch.empty(R, H, device=DEV, dtype=torch.bfloat16)
    stats = torch.empty(R, 2, device=DEV)
    mask = torch.zeros(R, H // 64, dtype=torch.int64, device=DEV) if p > 0 else None
    N.check(L.glr_drop_add_ln_fwd(N.ptr(h), N.ptr(inp), N.ptr(w), N.ptr(b), R, H, 1e-12, p, 1234, 8, None, N.ptr(out32), N.ptr(out16),
                                  N.ptr(stats), N.ptr(mask), N.stream()), "fwd")
    keep = _decode_mask(mask, R, H).to(DEV) if p > 0 else torch.ones(R, H, dtype=torch.bool, device=DEV)
    if p > 0:
        frac = keep.float().mean().item()
        sigma = (p * (1 - p) / (R * H)) ** 0.5
        assert abs(frac - (1 - p)) < 5 * sigma + 1e-3, (frac, 1 - p)
    hr, ir = h.float().requires_grad_(True), inp.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    z = hr * keep / (1 - p) + ir
    y = torch.nn.functional.layer_norm(z, (H,), wr, br, 1e-12)
    np.testing.assert_allclose(out32.cpu().numpy(), y.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(out16.float().cpu().numpy(), y.detach().cpu().numpy(), rtol=1e-2, atol=1e-2)
    (y * (d32 + d16.float())).sum().backward()
    d_inp = torch.empty(R, H, device=DEV)
    d_h = torch.empty(R, H, device=DEV, dtype=torch.bfloat16)
    dgb = torch.empty(2, H, device=DEV)
    ws = torch.empty(L.glr_ln_workspace_floats(R, H), device=DEV)
    dhs = torch.empty(H, device=DEV)
    dhs16 = torch.empty(H, device=DEV, dtype=torch.bfloat16)
    N.check(L.glr_drop_add_ln_bwd(N.ptr(d32), N.ptr(d16), N.ptr(h), N.ptr(inp), N.ptr(w), N.ptr(stats), N.ptr(mask), R, H, p,
                                  N.ptr(d_inp), N.ptr(d_h), N.ptr(ws), N.ptr(dgb), FL.c_off(dgb, H), N.ptr(dhs16), 1, N.stream()), "bwd")
    N.check(L.glr_drop_add_ln_bwd(N.ptr(d32), N.ptr(d16), N.ptr(h), N.ptr(inp), N.ptr(w), N.ptr(stats), N.ptr(mask), R, H, p,
                                  N.ptr(d_inp), N.ptr(d_h), N.ptr(ws), N.ptr(dgb), FL.c_off(dgb, H), N.ptr(dhs), 0, N.stream()), "bwd")
    # column sums of d_h = the bias gradient of the Linear that produced h
    ref_hs = hr.grad.sum(0)
    scale = float(ref_hs.abs().max())
    np.testing.assert_allclose(dhs.cpu().numpy() / scale, ref_hs.cpu().numpy() / scale, atol=2e-3)
    np.testing.assert_allclose(dhs16.float().cpu().numpy() / scale, ref_hs.cpu().numpy() / scale, atol=1e-2)
    np.testing.assert_allclose(d_inp.cpu().numpy(), ir.grad.cpu().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(d_h.float().cpu().numpy(), hr.grad.cpu().numpy(), rtol=1e-2, atol=1e-2)
    scale = float(wr.grad.abs().max())
    np.testing.assert_allclose(dgb[0].cpu().numpy() / scale, wr.grad.cpu().numpy() / scale, atol=1e-4)
    scale = float(br.grad.abs().max())
    np.testing.assert_allclose(dgb[1].cpu().numpy() / scale, br.grad.cpu().numpy() / scale, atol=1e-4)
    # only one of the two gradient streams present
    N.check(L.glr_drop_add_ln_bwd(None, N.ptr(d16), N.ptr(h), N.ptr(inp), N.ptr(w), N.ptr(stats), N.ptr(mask), R, H, p,
                                  N.ptr(d_inp), N.ptr(d_h), N.ptr(ws), N.ptr(dgb), FL.c_off(dgb, H), None, 0, N.stream()), "bwd16")
    ir.grad = None
    (torch.nn.functional.layer_norm(hr.detach() * keep / (1 - p) + ir, (H,), w, b, 1e-12) * d16.float()).sum().backward()
    np.testing.assert_allclose(d_inp.cpu().numpy(), ir.grad.cpu().numpy(), rtol=2e-4, atol=2e-4)


def test_dropout_mask_is_a_function_of_seed_and_offset():
    from gloria import _native as N
    L = N.lib()
    R, H = 64, 768
    h = torch.randn(R, H, device=DEV).bfloat16()
    inp = torch.randn(R, H, device=DEV)
    w, b = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)

    def run(seed, off, cell=None):
        o32 = torch.empty(R, H, device=DEV); o16 = torch.empty(R, H, device=DEV, dtype=torch.bfloat16)
        st = torch.empty(R, 2, device=DEV); m = torch.zeros(R, H // 64, dtype=torch.int64, device=DEV)
        N.check(L.glr_drop_add_ln_fwd(N.ptr(h), N.ptr(inp), N.ptr(w), N.ptr(b), R, H, 1e-12, 0.1, seed, off, N.ptr(cell), N.ptr(o32), N.ptr(o16),
                                      N.ptr(st), N.ptr(m), N.stream()), "fwd")
        return o32, m
    a, ma = run(7, 0)
    b2, mb = run(7, 0)
    assert torch.equal(a, b2) and torch.equal(ma, mb)
    _, mc = run(7, 4)
    _, md = run(8, 0)
    assert not torch.equal(ma, mc) and not torch.equal(ma, md)
    # key from a device cell {seed, offset base} (+ the site offset passed by value): the same masks
    cell = torch.tensor([7, 0], dtype=torch.int64, device=DEV)
    _, me = run(123, 0, cell)
    cell.copy_(torch.tensor([7, 4], dtype=torch.int64))
    _, mf = run(0, 0, cell)
    cell.copy_(torch.tensor([7, 0], dtype=torch.int64))
    _, mg = run(0, 4, cell)
    assert torch.equal(me, ma) and torch.equal(mf, mc) and torch.equal(mg, mc)


def test_bert_fused_sublayers_match_unfused_under_autocast(monkeypatch):
    """BertModel in eval mode (no dropout) under bf16 autocast: fused sub-layer epilogues vs torch's ops, forward and the
    gradient of a scalar of the last hidden states; training mode is reproducible under torch.manual_seed."""
    from gloria.models import bert as B
    from gloria.models import fused_ln as FL
    torch.manual_seed(0)
    cfg = B.BertConfig(vocab_size=1000, hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=512)
    model = B.BertModel(cfg).to(DEV).eval()
    ids = torch.randint(5, 1000, (6, 40), device=DEV)
    am = torch.ones_like(ids); am[:, 30:] = 0
    proj = torch.randn(2, 6, 40, 256, device=DEV)

    def run(enabled):
        monkeypatch.setattr(FL, "ENABLED", enabled)
        model.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            last, pooled, hidden = model(ids, am)
        loss = (hidden[-1].float() * proj[0]).sum() + (hidden[-2].float() * proj[1]).sum()   # (mean / mean-square of a LayerNorm output have ~zero gradients)
        loss.backward()
        return last.float(), model.encoder.layer[0].attention.self.query.weight.grad.clone(), \
            model.encoder.layer[1].output.LayerNorm.weight.grad.clone()
    a, ga, la = run(True)
    b, gb, lb = run(False)
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=3e-2, atol=3e-2)
    rel = (ga - gb).norm() / gb.norm()
    assert rel < 5e-2, rel
    rel = (la - lb).norm() / lb.norm()
    assert rel < 5e-2, rel
    monkeypatch.setattr(FL, "ENABLED", True)
    model.train()
    outs = []
    for _ in range(2):
        torch.manual_seed(3)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            outs.append(model(ids, am)[0].float())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("R,C", [(1, 256), (37, 768), (24832, 768), (4099, 2304), (1000, 3072)])
def test_colsum_matches_torch(R, C):
    """glr_colsum_bf16 (bias gradient of the text encoder's Linears) against an fp64 sum of the same bf16 values"""
    from gloria import _native as N
    L = N.lib()
    torch.manual_seed(R + C)
    x = (torch.randn(R, C, device=DEV) + 0.1).bfloat16()
    ws = torch.empty(L.glr_colsum_workspace_floats(R, C), device=DEV)
    out = torch.empty(C, device=DEV)
    out16 = torch.empty(C, device=DEV, dtype=torch.bfloat16)
    N.check(L.glr_colsum_bf16(N.ptr(x), R, C, N.ptr(ws), N.ptr(out), 0, N.stream()), "colsum")
    N.check(L.glr_colsum_bf16(N.ptr(x), R, C, N.ptr(ws), N.ptr(out16), 1, N.stream()), "colsum16")
    ref = x.double().sum(0)
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-3 * R ** 0.5)
    assert torch.equal(out16, out.bfloat16())
    assert L.glr_colsum_workspace_floats(10, 100) == 0           # unsupported width: the host falls back to torch
    assert L.glr_colsum_bf16(N.ptr(x), R, 100, N.ptr(ws), N.ptr(out), 0, N.stream()) != 0


def test_bert_bf16_parameter_linears_match_autograd(monkeypatch):
    """The training configuration of the flat optimizer (bf16 shadow weights of the Linears, fp32 LayerNorms and
    embeddings): every parameter gradient with the colsum / LayerNorm-epilogue bias gradients against torch's own
    linear backward (GLR_FUSED_LINEAR=0 path) on the same dropout-free model."""
    from gloria.models import bert as B
    from gloria.models import fused_linear as FLIN
    torch.manual_seed(0)
    cfg = B.BertConfig(vocab_size=1000, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = B.BertModel(cfg).to(DEV).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Linear):
            m.to(torch.bfloat16)
    ids = torch.randint(5, 1000, (6, 40), device=DEV)
    am = torch.ones_like(ids); am[:, 30:] = 0
    proj = torch.randn(6, 40, 256, device=DEV)
    calls = {"n": 0}
    real = FLIN._Linear.apply

    def run(enabled):
        monkeypatch.setattr(FLIN, "ENABLED", enabled)
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            hs = model(ids, am)[2]
        (hs[-1].float() * proj).sum().backward()
        return {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None}

    ref = run(False)
    got = run(True)
    assert set(ref) == set(got)
    n_bias = 0
    for n in ref:
        scale = float(ref[n].abs().max()) + 1e-6
        tol = 2e-2 if n.endswith("bias") else 1e-2
        np.testing.assert_allclose(got[n].cpu().numpy() / scale, ref[n].cpu().numpy() / scale, atol=tol, err_msg=n)
        n_bias += n.endswith("dense.bias") or n.endswith("query.bias")
    assert n_bias >= 8


def test_embedding_table_gradients_match_torch():
    """glr_embedding_bwd (host-sorted segments, padding rows dropped) and glr_type_embedding_bwd (two masked column sums)
    against torch's embedding backward on BertEmbeddings: the SAME fp32 values summed in a different order"""
    from gloria.models import bert as B
    from gloria.models import fused_embed as FE
    torch.manual_seed(0)
    cfg = B.BertConfig(vocab_size=500, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=256,
                       hidden_dropout_prob=0.0)
    emb = B.BertEmbeddings(cfg).to(DEV).train()
    Bn, L = 64, 97
    ids = torch.randint(1, 500, (Bn, L))
    ids[:, 0] = 2                                    # [CLS]-like: one long segment
    ids[:, 40:] = 0                                  # padding id: most rows, no gradient
    ids[5, 3] = 499
    tt = torch.zeros(Bn, L, dtype=torch.int64)
    tt[::3, 10:20] = 1
    proj = torch.randn(Bn, L, 768, device=DEV)

    def run(fused, host):
        FE.ENABLED = fused
        d_ids = ids.to(DEV)
        if host:
            d_ids._glr_host = ids.numpy()
        emb.zero_grad(set_to_none=True)
        (emb(d_ids, tt.to(DEV)) * proj).sum().backward()
        return {n: p.grad.clone() for n, p in emb.named_parameters()}
    try:
        ref = run(False, False)
        got = run(True, True)
        nohost = run(True, False)                    # word table falls back to torch, the type table stays fused
    finally:
        FE.ENABLED = True
    assert set(ref) == set(got)
    for n in ref:
        scale = float(ref[n].abs().max())
        np.testing.assert_allclose(got[n].cpu().numpy() / scale, ref[n].cpu().numpy() / scale, atol=2e-6, err_msg=n)
        np.testing.assert_allclose(nohost[n].cpu().numpy() / scale, ref[n].cpu().numpy() / scale, atol=2e-6, err_msg=n)
    assert float(got["word_embeddings.weight"][0].abs().max()) == 0.0           # padding_idx row
    assert float(got["word_embeddings.weight"][499].abs().max()) > 0.0
    again = run(True, True)
    assert all(torch.equal(again[n], got[n]) for n in got)                       # fixed summation order: bitwise repeatable
