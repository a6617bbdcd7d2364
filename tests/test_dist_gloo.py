"""world_size-2 tests of the data-parallel wiring on CPU (gloo): the sharded loss path of
GLoRIA.calc_loss (text all-gather -> block-row similarity -> row all-gather -> dual CE, gradients
reduce-scattered back) must reproduce the single-process full-batch result.  The HIP similarity
functions are replaced by the CPU oracle here (tests may use the oracle; the product never does), so
what is under test is the collective plumbing, offsets and gradient flow - not the kernels."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_inputs as gi

WORLD = 2
B, D, H, W, L = 4, 64, 3, 3, 9
CAP = [7, 5, 3, 2]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    img = torch.from_numpy(gi.normal(11, B, D, H, W))
    words = torch.from_numpy(gi.normal(12, B, D, L))
    ig = torch.from_numpy(gi.normal(13, B, D))
    tg = torch.from_numpy(gi.normal(14, B, D))
    return img, words, ig, tg


def _seg_labels():
    rng = np.random.default_rng(21)
    lab = torch.from_numpy(rng.random((B, 12, 12)) < 0.3)
    lab[:, 0, 0] = True                   # never empty
    return lab


def _patch_with_oracle():
    """CPU stand-ins with the product functions' signatures, built on the oracle."""
    from gloria.loss import gloria_loss as GL
    from oracle import gloria_oracle as orc

    def local_similarity(img, words, cap_lens, temp1=4.0, temp2=5.0, temp3=10.0, agg="sum", no_attn_vec=None,
                         eps=1e-8, want_attn=True, img_offset=0, word_start=0, want_amean=False):
        sim, a2, seg = orc.local_similarity_matrix(img, words, cap_lens, temp1, temp2, temp3, agg, no_attn_vec,
                                                   return_attn=True)
        woff = np.concatenate([[0], np.cumsum(cap_lens)])
        S = img.shape[2] * img.shape[3]
        shift = a2.shape[-1] - S                     # 1 with a no-attention column
        flat = torch.zeros(int(woff[-1]) * S)
        parts = []
        for b in range(img.shape[0]):
            i = img_offset + b
            parts.append((int(woff[i]) * S, a2[b, woff[i]:woff[i + 1], shift:].reshape(-1)))
        for off, p in parts:
            flat = torch.cat([flat[:off], p, flat[off + p.numel():]])
        amean = None
        if want_amean:        # word-mean attention rows of all pairs, padded like the kernel's output
            rows = torch.stack([a2[:, woff[i]:woff[i + 1]].mean(1) for i in range(len(cap_lens))], 1)
            amean = torch.nn.functional.pad(rows, (0, 64 - rows.shape[-1] % 64))
        return sim, flat, amean

    def attention_regularisers(amean, s_eff, shift, img_offset, w_na, w_kl, w_ent):
        """torch restatement of K6 + the reference's means (this rank's share of the global-batch means)."""
        A = amean[:, :, :s_eff]
        n = amean.shape[1]
        P = torch.cat([1 - A[:, :, 1:].sum(-1, keepdim=True), A[:, :, 1:]], -1) if shift else A
        idx = torch.arange(A.shape[0])
        Pd = P[idx, img_offset + idx].unsqueeze(1)
        ent = -(P * P.log()).sum(-1)
        kl = 0.5 * ((Pd - P) * (Pd.log() - P.log())).sum(-1)
        na = torch.log(1 - A[idx, img_offset + idx][:, shift:].sum(-1))
        return ((w_na * na.sum() / n) if w_na is not None else 0,
                (w_kl * -(kl.sum() / (n * (n - 1)))) if w_kl is not None else 0,
                (ent.sum() / (n * n)) if w_ent is not None else 0)

    def dual_cross_entropy(sim_rows, sim_full=None, row0=0):
        if sim_full is None:
            return orc.dual_ce(sim_rows)
        full = torch.cat([sim_full[:row0], sim_rows, sim_full[row0 + sim_rows.shape[0]:]], 0)
        return orc.dual_ce(full)

    GL.local_similarity = local_similarity
    GL.attention_regularisers = attention_regularisers
    GL.dual_cross_entropy = dual_cross_entropy
    GL.global_similarity = lambda a, t, eps=1e-8, temp3=10.0: orc.global_similarity_matrix(a, t, eps, temp3)


def _bare_gloria(dctx, aux=False):
    from gloria.models.gloria_model import GLoRIA
    g = GLoRIA.__new__(GLoRIA)
    torch.nn.Module.__init__(g)
    g.dist = dctx
    g.temp1, g.temp2, g.temp3 = 4.0, 5.0, 10.0
    g.no_attn_vec = None
    g.local_loss_weight = g.global_loss_weight = 1.0
    g.no_attn_loss_weight = g.attention_divergence_loss_weight = g.attention_entropy_loss_weight = None
    g.segmentation_loss_weight = None
    if aux:          # the flags of the reference's training command (submit_job.sh:15)
        g.no_attn_vec = torch.nn.Parameter(torch.from_numpy(gi.normal(15, D)))
        g.no_attn_loss_weight, g.attention_divergence_loss_weight, g.attention_entropy_loss_weight = 1.0, 0.1, 1.0
    return g


def _worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from gloria.dist import DistContext
        dctx = DistContext()
        _patch_with_oracle()
        img, words, ig, tg = _inputs()
        per = B // WORLD
        sl = slice(rank * per, (rank + 1) * per)
        li, lw = img[sl].clone().requires_grad_(True), words[sl].clone().requires_grad_(True)
        lig, ltg = ig[sl].clone().requires_grad_(True), tg[sl].clone().requires_grad_(True)

        class Sents(list):
            cap_lens = CAP[sl]
        g = _bare_gloria(dctx)
        loss, maps = g.calc_loss(li, lig, lw, ltg, Sents())
        loss.backward()
        # bucketed gradient all-reduce: sum over ranks
        p = [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(3, 2))]
        p[0].grad = torch.full((5,), float(rank + 1))
        p[1].grad = torch.full((3, 2), 10.0 * (rank + 1))
        dctx.allreduce_grads(p, bucket_bytes=16)
        ints = dctx.all_gather_ints([rank, rank + 10])
        # overlapped bucket reducer: grads are views of flat buckets, hooks fire during backward
        from gloria.dist import GradReducer
        torch.manual_seed(0)
        lin = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3), torch.nn.Linear(3, 1))
        unused = torch.nn.Parameter(torch.ones(4))                 # never receives a gradient
        plist = list(lin.parameters()) + [unused]
        red = GradReducer(plist, dctx, bucket_bytes=64)
        red.zero_grad()
        x = torch.full((2, 6), float(rank + 1))
        lin(x).sum().backward()
        red.finish()
        rgrads = [None if p.grad is None else p.grad.clone() for p in plist]
        # a parameter nobody used is skipped by the optimizer exactly as in a single process (no weight decay on it)
        opt = torch.optim.Adam(plist, lr=0.1, weight_decay=1e-2)
        opt.step()
        unused_after = unused.detach().clone()
        # the next cycle re-attaches the bucket views; a second backward inside one cycle is refused
        red.zero_grad()
        assert all(p.grad is not None for p in plist)
        lin(x).sum().backward()
        double = False
        try:
            lin(x).sum().backward()
        except RuntimeError as e:
            double = "one backward per" in str(e)
        red.finish()
        # attention-finetune configuration (imagenome_attn_finetune_config.yaml:51-53): contrastive weights 0,
        # segmentation weight 1 -> a rank's own loss is only its SHARE; global_batch_loss() is the full value
        gf = _bare_gloria(dctx)
        gf.local_loss_weight = gf.global_loss_weight = 0
        gf.segmentation_loss_weight = 1.0
        labels = _seg_labels()[sl]
        fl, _ = gf.calc_loss(img[sl].clone(), ig[sl].clone(), words[sl].clone(), tg[sl].clone(), Sents(), labels)
        val = float(gf.global_batch_loss())
        sch_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        sch = torch.optim.lr_scheduler.ReduceLROnPlateau(sch_opt, factor=0.5, patience=0)
        sch.step(1.0)
        sch.step(val + 1.0)            # worse than the best value on every rank alike -> the LR halves everywhere
        fin = {"own": float(fl), "val": val, "lr": sch_opt.param_groups[0]["lr"]}
        g2 = _bare_gloria(dctx)
        l2, _ = g2.calc_loss(li.detach(), lig.detach(), lw.detach(), ltg.detach(), Sents())
        fin["plain_val"] = float(g2.global_batch_loss())
        # same step with the no-attention vector and the three attention regularisers switched on
        ga = _bare_gloria(dctx, aux=True)
        ai, aw = img[sl].clone().requires_grad_(True), words[sl].clone().requires_grad_(True)
        l0, l1, na, kl, ent, _ = ga._calc_local_loss(ai, aw, Sents())
        (l0 + l1 + na + kl + ent).backward()      # l0 / l1: full value, gradient of this rank's rows only; aux terms: shares
        aux = {"l": [float(l0), float(l1)], "shares": [float(na), float(kl), float(ent)], "gi": ai.grad, "gw": aw.grad,
               "gna": ga.no_attn_vec.grad}
        torch.save({"loss": loss.detach(), "gi": li.grad, "gw": lw.grad, "gig": lig.grad, "gtg": ltg.grad, "aux": aux,
                    "maps": [m.detach() for m in maps], "p0": p[0].grad, "p1": p[1].grad, "ints": ints, "rgrads": rgrads,
                    "unused_after": unused_after, "double": double, "fin": fin},
                   os.path.join(out, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_loss_equals_full_batch(tmp_path):
    from oracle import gloria_oracle as orc
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt", weights_only=False) for r in range(WORLD)]

    img, words, ig, tg = _inputs()
    fi, fw = img.clone().requires_grad_(True), words.clone().requires_grad_(True)
    fig, ftg = ig.clone().requires_grad_(True), tg.clone().requires_grad_(True)
    l = orc.local_loss(fi, fw, CAP)
    gl_ = orc.global_loss(fig, ftg)
    full = l[0] + l[1] + gl_[0] + gl_[1]
    full.backward()
    per = B // WORLD
    for r in range(WORLD):
        sl = slice(r * per, (r + 1) * per)
        np.testing.assert_allclose(float(res[r]["loss"]), float(full), rtol=1e-5)
        np.testing.assert_allclose(res[r]["gi"].numpy(), fi.grad[sl].numpy(), rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(res[r]["gw"].numpy(), fw.grad[sl].numpy(), rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(res[r]["gig"].numpy(), fig.grad[sl].numpy(), rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(res[r]["gtg"].numpy(), ftg.grad[sl].numpy(), rtol=1e-4, atol=2e-6)
        for k in range(per):
            np.testing.assert_allclose(res[r]["maps"][k].numpy(), l[5][r * per + k].detach().numpy(), rtol=1e-5, atol=1e-7)
        assert torch.equal(res[r]["p0"], torch.full((5,), 3.0)) and torch.equal(res[r]["p1"], torch.full((3, 2), 30.0))
        assert res[r]["ints"] == [0, 10, 1, 11]
    # attention regularisers: the rank shares add up to the single-process values; gradients likewise
    ai, aw = img.clone().requires_grad_(True), words.clone().requires_grad_(True)
    nav = torch.from_numpy(gi.normal(15, D)).requires_grad_(True)
    ref = orc.local_loss(ai, aw, CAP, no_attn_vec=nav, no_attn_loss_weight=1.0, attention_divergence_loss_weight=0.1,
                         attention_entropy_loss_weight=1.0)
    (ref[0] + ref[1] + ref[2] + ref[3] + ref[4]).backward()
    shares = np.sum([res[r]["aux"]["shares"] for r in range(WORLD)], 0)
    np.testing.assert_allclose(shares, [float(ref[2]), float(ref[3]), float(ref[4])], rtol=1e-5, atol=1e-6)
    gna = sum(res[r]["aux"]["gna"] for r in range(WORLD))
    np.testing.assert_allclose(gna.numpy(), nav.grad.numpy(), rtol=1e-4, atol=2e-6)
    for r in range(WORLD):
        sl = slice(r * per, (r + 1) * per)
        np.testing.assert_allclose(res[r]["aux"]["l"], [float(ref[0]), float(ref[1])], rtol=1e-5)
        np.testing.assert_allclose(res[r]["aux"]["gi"].numpy(), ai.grad[sl].numpy(), rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(res[r]["aux"]["gw"].numpy(), aw.grad[sl].numpy(), rtol=1e-4, atol=2e-6)
    # reducer: every rank ends with the SUM over ranks of the single-process gradients
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3), torch.nn.Linear(3, 1))
    want = [torch.zeros_like(p) for p in lin.parameters()]
    for r in range(WORLD):
        lin.zero_grad()
        lin(torch.full((2, 6), float(r + 1))).sum().backward()
        want = [w + p.grad for w, p in zip(want, lin.parameters())]
    for r in range(WORLD):
        for a, b in zip(res[r]["rgrads"][:-1], want):
            np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5, atol=1e-6)
        assert res[r]["rgrads"][-1] is None                       # unused parameter: hidden from the optimizer
        assert torch.equal(res[r]["unused_after"], torch.ones(4))  # ... so weight decay did not move it
        assert res[r]["double"], "a second backward inside one reducer cycle must raise"
    # attention-finetune configuration: validation value and plateau LR identical on every rank and equal to the
    # single-process loss (the rank's own loss is only its share)
    maps = orc.local_loss(img, words, CAP)[5]
    seg_full = float(orc.attention_supervision_loss(maps, _seg_labels(), 1.0))
    for r in range(WORLD):
        f = res[r]["fin"]
        np.testing.assert_allclose(f["val"], seg_full, rtol=1e-5)
        assert f["val"] == res[0]["fin"]["val"] and f["lr"] == res[0]["fin"]["lr"] == 0.5
        assert abs(f["own"] - seg_full) > 1e-3 * abs(seg_full)     # the un-reduced value differs: this was the bug
        np.testing.assert_allclose(f["plain_val"], float(full), rtol=1e-5)
    np.testing.assert_allclose(sum(res[r]["fin"]["own"] for r in range(WORLD)), seg_full, rtol=1e-5)
