"""Short-sequence self-attention kernels (glr_attn_fwd / glr_attn_bwd) against a plain fp32 torch restatement on the same
bf16 tensors: context, and the gradients of Q, K, V - without dropout and with the kernel's OWN dropout mask decoded
from its keep bits (key 32 j + i of query row r = bit i of word (r, j)); ragged key masks; the Bernoulli rate."""

import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(q, k, v, key_mask, nh, keep, p):
    B, L, H = q.shape
    hd = H // nh
    qh, kh, vh = (t.float().view(B, L, nh, hd).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / math.sqrt(hd)
    if key_mask is not None:
        s = s.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    pr = torch.softmax(s, dim=-1)
    if keep is not None:
        pr = pr * keep / (1 - p)
    return (pr @ vh).transpose(1, 2).reshape(B, L, H)


def _decode_keep(keep, B, nh, L):
    w = keep.cpu().numpy().view(np.uint32).reshape(B, nh, 128, 4)
    bits = ((w[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)          # [B, nh, 128, 4, 32]
    return torch.from_numpy(bits.reshape(B, nh, 128, 128)[:, :, :L, :L])


@pytest.mark.parametrize("B,nh,L,p,ragged", [(3, 12, 97, 0.0, True), (2, 12, 97, 0.1, True), (2, 4, 40, 0.3, False),
                                              (2, 2, 112, 0.1, True), (2, 2, 128, 0.1, True), (3, 1, 16, 0.0, False), (2, 3, 1, 0.0, False),
                                              (2, 2, 33, 0.2, True)])
def test_attention_matches_torch(B, nh, L, p, ragged):
    from gloria import _native as N
    H = nh * 64
    g = torch.Generator().manual_seed(B * 100 + L)
    q, k, v, d_o = ((torch.randn(B, L, H, generator=g) * s).to(DEV).bfloat16() for s in (1.5, 1.5, 1.0, 1.0))
    key_mask = None
    if ragged:
        lens = torch.randint(max(1, L // 3), L + 1, (B,), generator=g)
        key_mask = (torch.arange(L)[None, :] < lens[:, None]).to(DEV)
    Lb = N.lib()
    o = torch.empty_like(q)
    lse = torch.empty(B * nh, 128, device=DEV)
    keep = torch.zeros(B * nh, 128, 4, dtype=torch.int32, device=DEV) if p > 0 else None
    scale = 1.0 / math.sqrt(64)
    N.check(Lb.glr_attn_fwd(N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(key_mask), B, nh, L, H, H, scale, p, 99, 4, None, N.ptr(o), N.ptr(lse),
                            N.ptr(keep), N.stream()), "fwd")
    if p > 0:          # the key read from a device cell {seed, offset base}: the same bits
        cell = torch.tensor([99, 0], dtype=torch.int64, device=DEV)
        o2, keep2 = torch.empty_like(q), torch.zeros_like(keep)
        N.check(Lb.glr_attn_fwd(N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(key_mask), B, nh, L, H, H, scale, p, 0, 4, N.ptr(cell), N.ptr(o2),
                                N.ptr(lse), N.ptr(keep2), N.stream()), "fwd cell")
        assert torch.equal(keep, keep2) and torch.equal(o, o2)
    km = None
    if p > 0:
        km = _decode_keep(keep, B, nh, L).to(DEV)
        valid = torch.ones(B, nh, L, L, dtype=torch.bool, device=DEV) if key_mask is None else key_mask[:, None, None, :].expand(B, nh, L, L)
        frac = km[valid].float().mean().item()
        n = int(valid.sum())
        assert abs(frac - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5 + 2e-3, (frac, 1 - p)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = _reference(qr, kr, vr, key_mask, nh, km, p)
    scale_o = float(ref.abs().max())
    np.testing.assert_allclose(o.float().cpu().numpy() / scale_o, ref.detach().cpu().numpy() / scale_o, atol=1.5e-2)
    (ref * d_o.float()).sum().backward()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    N.check(Lb.glr_attn_bwd(N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(o), N.ptr(d_o), N.ptr(key_mask), N.ptr(lse), N.ptr(keep), B, nh, L, H, H,
                            scale, p, N.ptr(dq), N.ptr(dk), N.ptr(dv), N.stream()), "bwd")
    for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        assert torch.isfinite(got.float()).all()
        if float(want.norm()) < 1e-6:                     # a single key: the softmax is constant, dq = dk = 0
            assert float(got.float().norm()) < 1e-3, name
            continue
        rel = float((got.float() - want).norm() / want.norm())
        print(f"[attn B{B} nh{nh} L{L} p{p}] {name} relative Frobenius error {rel:.4f}")
        assert rel < 3e-2, (name, rel)


def test_attention_op_in_bert_matches_sdpa(monkeypatch):
    """BertModel (eval mode, bf16 autocast) with the fused attention + sub-layer epilogues against torch's own ops."""
    from gloria.models import bert as B
    from gloria.models import fused_attn as FA
    from gloria.models import fused_ln as FL
    torch.manual_seed(0)
    cfg = B.BertConfig(vocab_size=1000, hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=512)
    model = B.BertModel(cfg).to(DEV).eval()
    ids = torch.randint(5, 1000, (6, 40), device=DEV)
    am = torch.ones_like(ids); am[:, 30:] = 0; am[2, 11:] = 0
    proj = torch.randn(6, 40, 256, device=DEV) * am[:, :, None]

    def run(enabled):
        monkeypatch.setattr(FA, "ENABLED", enabled)
        monkeypatch.setattr(FL, "ENABLED", enabled)
        model.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            last, pooled, hidden = model(ids, am)
        (last.float() * proj).sum().backward()
        return last.float() * am[:, :, None], {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    a, ga = run(True)
    b, gb = run(False)
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=4e-2, atol=4e-2)
    for n in gb:
        # (the key bias shifts every score of a query row alike: its true gradient is zero, what is left is rounding)
        if gb[n].norm() > 1e-6 and not n.endswith("key.bias"):
            rel = float((ga[n] - gb[n]).norm() / gb[n].norm())
            assert rel < 6e-2, (n, rel)


def test_dropout_hash_statistics():
    """The attention kernel draws its keep bits from a keyed counter hash (two rounds of a 32-bit multiply-xorshift
    mixer, 16 bits per score), not from Philox like the LayerNorm epilogue.  Beyond the Bernoulli rate: at p = 0.1 / 0.5 /
    0.9 the rate is within 5 sigma, and at p = 0.5 (every bit a fair coin) neighbouring bits are uncorrelated along keys,
    along queries, across heads, across sentences and across consecutive generator offsets (= consecutive training steps and
    sites), the per-row and per-column keep counts have binomial spread, and no two rows of a head repeat."""
    from gloria import _native as N
    B, nh, L, H = 8, 12, 128, 768
    g = torch.Generator().manual_seed(5)
    q, k, v = ((torch.randn(B, L, H, generator=g)).to(DEV).bfloat16() for _ in range(3))
    Lb = N.lib()

    def bits(p, seed, off):
        o = torch.empty_like(q)
        lse = torch.empty(B * nh, 128, device=DEV)
        keep = torch.zeros(B * nh, 128, 4, dtype=torch.int32, device=DEV)
        N.check(Lb.glr_attn_fwd(N.ptr(q), N.ptr(k), N.ptr(v), None, B, nh, L, H, H, 0.125, p, seed, off, None, N.ptr(o), N.ptr(lse),
                                N.ptr(keep), N.stream()), "fwd")
        return _decode_keep(keep, B, nh, L).numpy()                         # [B, nh, L, L] bool

    n = B * nh * L * L
    for p in (0.1, 0.5, 0.9):
        rate = bits(p, 1234, 8).mean()
        assert abs(rate - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5, (p, rate)
    a = bits(0.5, 1234, 8).astype(np.float64) * 2 - 1                      # +-1 coins
    tol = 5 / np.sqrt(n)

    def corr(x, y):
        return float((x * y).mean())
    assert abs(corr(a[..., :-1], a[..., 1:])) < tol                         # neighbouring keys
    assert abs(corr(a[:, :, :-1, :], a[:, :, 1:, :])) < tol                 # neighbouring queries
    assert abs(corr(a[:, :-1], a[:, 1:])) < tol                             # neighbouring heads
    assert abs(corr(a[:-1], a[1:])) < tol                                   # neighbouring sentences
    assert abs(corr(a[..., :-32], a[..., 32:])) < tol                       # the 32-key stride of the kernel's lane layout
    for off in (12, 16, 8 + 4 * 36):                                        # next site, the one after, the next step
        assert abs(corr(a, bits(0.5, 1234, off).astype(np.float64) * 2 - 1)) < tol, off
    assert abs(corr(a, bits(0.5, 1235, 8).astype(np.float64) * 2 - 1)) < tol          # another seed
    # keep counts per query row / per key column: binomial(128, 0.5) spread (variance 32)
    for axis in (-1, -2):
        cnt = (a > 0).sum(axis).astype(np.float64)
        assert abs(cnt.var() / 32.0 - 1.0) < 0.05, (axis, cnt.var())
    rows = (a > 0).reshape(B * nh, L, L)
    packed = np.packbits(rows, axis=-1)
    for h in range(0, B * nh, 7):
        assert len({r.tobytes() for r in packed[h]}) == L                   # no repeated mask row inside a head
