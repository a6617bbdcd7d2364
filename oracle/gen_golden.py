#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference on synthetic inputs.

Runs only in the build container (needs /root/reference); the reference never travels to
the GPU box - only the arrays written here do.  The reference modules are loaded by file
path (the `gloria` package itself is not importable here):

  /root/reference/gloria/loss/gloria_loss.py    -> attention_fn, cosine_similarity,
                                                   global_loss, local_loss
  /root/reference/gloria/models/text_model.py   -> BertEncoder.forward / aggregate_tokens
                                                   (object built with __new__, stub BERT)

Inputs come from tests/golden_inputs.py (seeded numpy streams), so fixtures hold outputs
only.  Large outputs are stored as a strided sample + (sum, sum of squares).

usage:  python oracle/gen_golden.py [--skip-b256]
"""

import argparse
import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_inputs as gi  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def load_ref(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pack(d, key, arr):
    arr = np.asarray(arr.detach().cpu().numpy() if torch.is_tensor(arr) else arr)
    d[key + ".shape"] = np.array(arr.shape, dtype=np.int64)
    d[key + ".sample"] = gi.subsample(arr).astype(np.float32)
    d[key + ".sums"] = gi.checksums(arr)


def t(a, grad=False):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return x.requires_grad_(True) if grad else x


def gen_attention(ref, out):
    for name in gi.ATTN_CASES:
        q, ctx, temp1, na = gi.attn_inputs(name)
        with torch.no_grad():
            wc, attn = ref.attention_fn(t(q), t(ctx), temp1, no_attn_vec=None if na is None else t(na))
        pack(out, f"attn/{name}/weighted", wc)
        pack(out, f"attn/{name}/map", attn)
        print("attention", name, tuple(wc.shape), tuple(attn.shape))


def gen_attention_grads(ref, out):
    """attention_fn OUTPUT gradients (G1 with grads): d/d(query, context, no_attn_vec) of
    sum(weightedContext * gw) + sum(attn * ga) by the reference's own autograd."""
    for name in gi.ATTN_CASES:
        q, ctx, temp1, na = gi.attn_inputs(name)
        tq, tc = t(q, True), t(ctx, True)
        tna = None if na is None else t(na, True)
        wc, attn = ref.attention_fn(tq, tc, temp1, no_attn_vec=tna)
        gw, ga = gi.attn_upstream(name, tuple(wc.shape), tuple(attn.shape))
        ((wc * t(gw)).sum() + (attn * t(ga)).sum()).backward()
        pack(out, f"attn_grad/{name}/grad_query", tq.grad)
        pack(out, f"attn_grad/{name}/grad_context", tc.grad)
        if tna is not None:
            pack(out, f"attn_grad/{name}/grad_no_attn", tna.grad)
        print("attention grads", name, float(tq.grad.abs().max()), float(tc.grad.abs().max()))


def ref_sim_matrix(ref, img, words, cap_lens, no_attn, temp1=4.0, temp2=5.0, temp3=10.0, agg="sum"):
    """B x B similarity matrix assembled from the reference's own attention_fn and
    cosine_similarity, sentence by sentence (the reference's local_loss does not return it)."""
    B = img.shape[0]
    cols = []
    with torch.no_grad():
        for i in range(B):
            n = cap_lens[i]
            word = words[i, :, :n].unsqueeze(0).contiguous().repeat(B, 1, 1)
            wc, _ = ref.attention_fn(word, img, temp1, no_attn_vec=no_attn)
            row = ref.cosine_similarity(word.transpose(1, 2).reshape(B * n, -1),
                                        wc.transpose(1, 2).reshape(B * n, -1)).view(B, n)
            row = (row * temp2).exp()
            row = row.sum(1, keepdim=True) if agg == "sum" else row.mean(1, keepdim=True)
            cols.append(torch.log(row))
    return torch.cat(cols, 1) * temp3


def gen_local(ref, out):
    for name, cfg in gi.LOCAL_CASES.items():
        img, words, cap_lens, na = gi.local_inputs(name)
        timg, twords = t(img, True), t(words, True)
        tna = None if na is None else t(na, True)
        aux = cfg["aux"] or (None, None, None)
        res = ref.local_loss(timg, twords, cap_lens, temp1=4.0, temp2=5.0, temp3=10.0, agg=cfg["agg"],
                             no_attn_vec=tna, no_attn_loss_weight=aux[0],
                             attention_divergence_loss_weight=aux[1], attention_entropy_loss_weight=aux[2])
        l0, l1, nal, kl, ent, maps = res
        scal = [float(l0), float(l1), float(nal), float(kl), float(ent)]
        out[f"local/{name}/losses"] = np.array(scal, dtype=np.float64)
        out[f"local/{name}/cap_lens"] = np.array(cap_lens, dtype=np.int64)
        pack(out, f"local/{name}/maps", torch.cat([m.reshape(-1) for m in maps]))
        total = l0 + l1 + nal + kl + ent
        total.backward()
        pack(out, f"local/{name}/grad_img", timg.grad)
        pack(out, f"local/{name}/grad_words", twords.grad)
        if tna is not None:
            pack(out, f"local/{name}/grad_no_attn", tna.grad)
        sim = ref_sim_matrix(ref, t(img), t(words), cap_lens, None if na is None else t(na), agg=cfg["agg"])
        pack(out, f"local/{name}/sim", sim)
        print("local", name, scal)


def gen_sim(ref, out, skip_b256):
    for name, cfg in gi.SIM_CASES.items():
        if skip_b256 and cfg["B"] >= 256:
            continue
        img, words, cap_lens, na = gi.local_inputs(name)
        t0 = time.time()
        sim = ref_sim_matrix(ref, t(img), t(words), cap_lens, None)
        labels = torch.arange(cfg["B"])
        l0 = torch.nn.CrossEntropyLoss()(sim, labels)
        l1 = torch.nn.CrossEntropyLoss()(sim.t(), labels)
        out[f"sim/{name}/cap_lens"] = np.array(cap_lens, dtype=np.int64)
        out[f"sim/{name}/sim"] = sim.numpy().astype(np.float32)
        out[f"sim/{name}/losses"] = np.array([float(l0), float(l1)], dtype=np.float64)
        print("sim", name, float(l0), float(l1), "%.1fs" % (time.time() - t0))


def gen_global(ref, out):
    for name in gi.GLOBAL_CASES:
        img, txt = gi.global_inputs(name)
        ti, tt = t(img, True), t(txt, True)
        l0, l1 = ref.global_loss(ti, tt, temp3=10.0)
        (l0 + l1).backward()
        out[f"global/{name}/losses"] = np.array([float(l0), float(l1)], dtype=np.float64)
        pack(out, f"global/{name}/grad_img", ti.grad)
        pack(out, f"global/{name}/grad_txt", tt.grad)
        print("global", name, float(l0), float(l1))


def gen_text(out):
    tm = load_ref("ref_text_model", "gloria/models/text_model.py")
    ids, hidden, vocab = gi.text_inputs()
    enc = tm.BertEncoder.__new__(tm.BertEncoder)
    torch.nn.Module.__init__(enc)
    enc.last_n_layers, enc.aggregate_method, enc.norm = 4, "sum", False
    enc.embedding_dim, enc.agg_tokens = hidden[0].shape[-1], True
    enc.emb_local = enc.emb_global = None
    enc.idxtoword = vocab
    hs = tuple(t(h) for h in hidden)
    enc.model = lambda i, m, tt: (None, None, hs)          # stub BERT: returns the synthetic hidden states
    with torch.no_grad():
        word, sent, sents = enc.forward(t(ids), None, None)
    out["text/word_emb"] = word.numpy().astype(np.float32)
    out["text/sent_emb"] = sent.numpy().astype(np.float32)
    out["text/sents"] = np.array(["\t".join(s) for s in sents])
    print("text", tuple(word.shape), tuple(sent.shape))


def _ref_text_encoder(tm, hidden, vocab, grad=False):
    enc = tm.BertEncoder.__new__(tm.BertEncoder)
    torch.nn.Module.__init__(enc)
    enc.last_n_layers, enc.aggregate_method, enc.norm = 4, "sum", False
    enc.embedding_dim, enc.agg_tokens = hidden[0].shape[-1], True
    enc.emb_local = enc.emb_global = None
    enc.idxtoword = vocab
    hs = tuple(t(h, grad) for h in hidden)
    enc.model = lambda i, m, tt: (None, None, hs)
    return enc, hs


def gen_text_wide(out):
    """word-piece aggregation of the real BertEncoder.forward at D = 64 (all outputs + gradients of the four
    hidden states used) and D = 768 (sub-sampled outputs): the widths of the HIP kernel's tile and of BERT-base."""
    tm = load_ref("ref_text_model", "gloria/models/text_model.py")
    for D in (64, 768):
        ids, hidden, vocab = gi.text_inputs(D=D)
        enc, hs = _ref_text_encoder(tm, hidden, vocab, grad=(D == 64))
        if D == 64:
            word, sent, sents = enc.forward(t(ids), None, None)
            gw, gs = gi.normal(9, *word.shape), gi.normal(10, *sent.shape)
            ((word * t(gw)).sum() + (sent * t(gs)).sum()).backward()
            out["text64/word_emb"] = word.detach().numpy().astype(np.float32)
            out["text64/sent_emb"] = sent.detach().numpy().astype(np.float32)
            for k in range(1, 5):
                out[f"text64/grad_hidden_m{k}"] = hs[-k].grad.numpy().astype(np.float32)
            assert hs[0].grad is None          # only the last four layers take part
        else:
            with torch.no_grad():
                word, sent, sents = enc.forward(t(ids), None, None)
            pack(out, "text768/word_emb", word)
            out["text768/sent_emb"] = sent.numpy().astype(np.float32)
        print("text wide", D, tuple(word.shape), tuple(sent.shape))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-b256", action="store_true")
    ap.add_argument("--only", default=None, help="comma-separated fixture files to (re)generate, e.g. attention_grad,text_wide")
    args = ap.parse_args()
    torch.manual_seed(0)
    ref = load_ref("ref_gloria_loss", "gloria/loss/gloria_loss.py")
    os.makedirs(OUT, exist_ok=True)
    for fname, fn in (("attention.npz", lambda o: gen_attention(ref, o)),
                      ("local.npz", lambda o: gen_local(ref, o)),
                      ("global.npz", lambda o: gen_global(ref, o)),
                      ("text.npz", gen_text),
                      ("attention_grad.npz", lambda o: gen_attention_grads(ref, o)),
                      ("text_wide.npz", gen_text_wide),
                      ("sim.npz", lambda o: gen_sim(ref, o, args.skip_b256))):
        if args.only is not None and fname[:-4] not in args.only.split(","):
            continue
        out = {}
        fn(out)
        np.savez_compressed(os.path.join(OUT, fname), **out)
        print("wrote", fname, "%.1f KB" % (os.path.getsize(os.path.join(OUT, fname)) / 1024))


if __name__ == "__main__":
    main()
