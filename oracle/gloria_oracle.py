"""CPU oracle for the GLoRIA global+local contrastive hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker / the timed CPU baseline.  The product path
(``gloria-nlp-project_amd/gloria``) never imports this module and raises when the
HIP library is missing.

What it restates (all citations relative to /root/reference):

* ``cosine_similarity``    gloria/loss/gloria_loss.py:11-16
* ``attention_fn``         gloria/loss/gloria_loss.py:19-63
* ``global_loss``          gloria/loss/gloria_loss.py:66-88
* ``local_loss``           gloria/loss/gloria_loss.py:99-201  (per-sentence loop, same op order)
* ``calc_loss``            gloria/models/gloria_model.py:105-150 (cap_lens rule, weighting,
                           attention-supervision term)
* ``local_similarities_inference``  gloria/models/gloria_model.py:171-207
* ``aggregate_wordpieces`` gloria/models/text_model.py:32-90, 96-131

Pinning: ``oracle/gen_golden.py`` imports the *real* reference loss module by file
path in the build container and stores its inputs/outputs under ``tests/golden``;
``tests/test_oracle_golden.py`` checks every function here against those vectors.
The reference itself holds no tests/fixtures for this path.

Two formulations are kept on purpose:

* ``local_loss`` walks sentences one by one exactly like the reference (this is the
  one timed as the CPU baseline);
* ``local_similarity_matrix`` is a masked, batched formulation of the same maths used
  to check the image x sentence similarity matrix at sizes where the loop is slow and
  to restate the sharded (multi-rank) computation.

Everything is plain torch on CPU; dtype follows the inputs (tests use float32 and
float64).
"""

from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------------------

def cosine_similarity(x1: torch.Tensor, x2: torch.Tensor, dim: int = 1, eps: float = 1e-8):
    """<x1,x2> / max(|x1|*|x2|, eps), squeezed.  Ref gloria_loss.py:11-16 (note the clamp is
    on the PRODUCT of the norms)."""
    num = (x1 * x2).sum(dim)
    den = (x1.norm(2, dim) * x2.norm(2, dim)).clamp(min=eps)
    return (num / den).squeeze()


def _ce_diag(logits: torch.Tensor) -> torch.Tensor:
    """Cross entropy with labels = arange (mean over rows).  Ref gloria_loss.py:86-87,169-170."""
    lse = torch.logsumexp(logits, dim=1)
    return (lse - logits.diagonal()).mean()


def dual_ce(sim: torch.Tensor):
    """(CE over rows, CE over columns) of a square similarity matrix."""
    return _ce_diag(sim), _ce_diag(sim.t())


def attention_fn(query: torch.Tensor, context: torch.Tensor, temp1: float,
                 no_attn_vec: Optional[torch.Tensor] = None):
    """Region x word attention.  Ref gloria_loss.py:19-63.

    query   [B, D, n]      words of ONE sentence repeated B times
    context [B, D, H, W]   image region features
    returns weightedContext [B, D, n], attn [B, n, H, W] (no-attn column stripped)
    """
    B, D, n = query.shape
    H, W = context.shape[2], context.shape[3]
    ctx = context.reshape(B, D, H * W)
    if no_attn_vec is not None:                                   # :31-34
        col = no_attn_vec.reshape(1, D, 1).expand(B, D, 1)
        ctx = torch.cat([col, ctx], dim=2)
    # scores[b, r, w] = sum_d ctx[b, d, r] * query[b, d, w]        (:40)
    scores = torch.einsum("bdr,bdw->brw", ctx, query)
    a1 = torch.softmax(scores, dim=2)                             # over words   (:42-43)
    a2 = torch.softmax(a1.transpose(1, 2) * temp1, dim=2)         # over regions (:48-52) [B, n, S]
    weighted = torch.einsum("bdr,bwr->bdw", ctx, a2)              # (:59)
    amap = a2[:, :, 1:] if no_attn_vec is not None else a2        # (:60-61)
    return weighted, amap.reshape(B, n, H, W)


def global_loss(cnn_code: torch.Tensor, rnn_code: torch.Tensor, eps: float = 1e-8, temp3: float = 10.0):
    """Global InfoNCE.  Ref gloria_loss.py:66-88."""
    sim = global_similarity_matrix(cnn_code, rnn_code, eps=eps, temp3=temp3)
    return dual_ce(sim)


def global_similarity_matrix(cnn_code, rnn_code, eps: float = 1e-8, temp3: float = 10.0):
    """temp3 * <I_b, T_i> / max(|I_b|*|T_i|, eps).  Ref gloria_loss.py:75-80."""
    ni = cnn_code.norm(2, dim=1, keepdim=True)
    nt = rnn_code.norm(2, dim=1, keepdim=True)
    raw = cnn_code @ rnn_code.t()
    return raw / (ni @ nt.t()).clamp(min=eps) * temp3


def kl_divergence(p, q):
    """Ref gloria_loss.py:91-92."""
    return (p * torch.log(p / q)).sum(-1)


def entropy(p):
    """Ref gloria_loss.py:95-96."""
    return -(p * torch.log(p)).sum(-1)


# --------------------------------------------------------------------------------------
# local loss: reference-structured loop
# --------------------------------------------------------------------------------------

def local_loss(img_features, words_emb, cap_lens: Sequence[int], temp1=4.0, temp2=5.0, temp3=10.0,
               agg="sum", no_attn_vec=None, no_attn_loss_weight=None,
               attention_divergence_loss_weight=None, attention_entropy_loss_weight=None,
               return_sim: bool = False):
    """Local (region x word) InfoNCE, sentence by sentence.  Ref gloria_loss.py:99-201.

    img_features [B, D, H, W], words_emb [B, D, L], cap_lens host ints.
    Returns (loss0, loss1, no_attn_loss, kl_loss, entropy_loss, att_maps) like the
    reference; with ``return_sim`` the B x B similarity matrix (already * temp3) is
    appended.
    """
    B = img_features.shape[0]
    cols, att_maps = [], []
    na_cols, ent_cols, flat_cols = [], [], []
    want_flat = attention_divergence_loss_weight is not None or attention_entropy_loss_weight is not None
    for i in range(words_emb.shape[0]):
        n = int(cap_lens[i])
        word = words_emb[i, :, :n].unsqueeze(0).expand(B, -1, -1).contiguous()   # :122-123
        wctx, attn = attention_fn(word, img_features, temp1, no_attn_vec=no_attn_vec)  # :126
        if no_attn_loss_weight is not None:                                      # :129-130
            na_cols.append(torch.log(1 - attn.sum(-1).sum(-1).mean(-1).unsqueeze(-1)))
        if want_flat:                                                            # :131-139
            flat = attn.reshape(B, n, -1).mean(1)
            if no_attn_vec is not None:
                flat = torch.cat([1 - flat.sum(-1, keepdim=True), flat], -1)
            if attention_entropy_loss_weight is not None:
                ent_cols.append(entropy(flat).unsqueeze(1))
            if attention_divergence_loss_weight is not None:
                flat_cols.append(flat.unsqueeze(1))
        att_maps.append(attn[i].unsqueeze(0).contiguous())                       # :141-143
        w2 = word.transpose(1, 2).reshape(B * n, -1)
        c2 = wctx.transpose(1, 2).reshape(B * n, -1)
        row = cosine_similarity(w2, c2).reshape(B, n)                            # :150-151
        row = torch.exp(row * temp2)                                             # :153
        row = row.sum(1, keepdim=True) if agg == "sum" else row.mean(1, keepdim=True)
        cols.append(torch.log(row))                                              # :158
    sim = torch.cat(cols, 1) * temp3                                             # :162-164
    loss0, loss1 = dual_ce(sim)                                                  # :169-170

    eye = torch.eye(B, dtype=torch.bool)
    if no_attn_loss_weight is not None:                                          # :173-177
        no_attn_loss = no_attn_loss_weight * torch.cat(na_cols, 1)[eye].mean()
    else:
        no_attn_loss = 0
    if attention_divergence_loss_weight is not None:                             # :180-192
        flats = torch.cat(flat_cols, 1)            # [B_img, B_txt, S(+1)]
        kls = []
        for b in range(B):
            cur = flats[b, b].unsqueeze(0).expand(B, -1)
            kls.append(((kl_divergence(cur, flats[b]) + kl_divergence(flats[b], cur)) / 2).unsqueeze(1))
        kls = torch.cat(kls, 1)                    # [txt, img]
        kl_loss = attention_divergence_loss_weight * (-kls[~eye].mean())
    else:
        kl_loss = 0
    if attention_entropy_loss_weight is not None:                                # :195-197 (weight unused)
        entropy_loss = torch.cat(ent_cols, 1).mean()
    else:
        entropy_loss = 0
    out = (loss0, loss1, no_attn_loss, kl_loss, entropy_loss, att_maps)
    return out + (sim,) if return_sim else out


# --------------------------------------------------------------------------------------
# local similarity: masked batched formulation (same maths, no per-sentence loop)
# --------------------------------------------------------------------------------------

def pack_words(words_emb: torch.Tensor, cap_lens: Sequence[int]):
    """Concatenate the first cap_lens[i] word columns of every sentence: T [N, D], seg [N]."""
    parts, seg = [], []
    for i, n in enumerate(cap_lens):
        parts.append(words_emb[i, :, : int(n)].t())
        seg += [i] * int(n)
    return torch.cat(parts, 0), torch.tensor(seg, dtype=torch.long)


def _segment_softmax(x: torch.Tensor, seg: torch.Tensor, nseg: int):
    """softmax over the last axis restricted to runs of equal seg id."""
    shape = x.shape[:-1] + (nseg,)
    idx = seg.expand(x.shape)
    mx = torch.full(shape, -math.inf, dtype=x.dtype).scatter_reduce(-1, idx, x, "amax")
    e = torch.exp(x - mx.gather(-1, idx))
    sm = torch.zeros(shape, dtype=x.dtype).scatter_add(-1, idx, e)
    return e / sm.gather(-1, idx)


def local_similarity_matrix(img_features, words_emb, cap_lens: Sequence[int], temp1=4.0, temp2=5.0,
                            temp3=10.0, agg="sum", no_attn_vec=None, eps=1e-8, img_chunk: int = 8,
                            return_attn: bool = False, word_start: int = 0):
    """sim[b, i] = temp3 * log(sum|mean_w exp(temp2 * cos[b, i, w])) for every image b and
    sentence i (ref gloria_loss.py:116-164).  Images are processed in
    chunks so the [chunk, S, N] score tensor stays small.  ``word_start`` = 1 gives the
    inference variant that skips [CLS] (gloria_model.py:179).

    With ``return_attn`` also returns a2 [B_img, N, S] (word-packed, no-attn column kept).
    """
    B, D = img_features.shape[:2]
    V = img_features.reshape(B, D, -1)
    if no_attn_vec is not None:
        V = torch.cat([no_attn_vec.reshape(1, D, 1).expand(B, D, 1), V], 2)
    if word_start:
        shifted = torch.zeros_like(words_emb)
        shifted[:, :, : words_emb.shape[2] - word_start] = words_emb[:, :, word_start:]
        words_emb = shifted
    T, seg = pack_words(words_emb, cap_lens)                 # [N, D], [N]
    nsent = len(cap_lens)
    tnorm = T.norm(2, dim=1)
    sims, attns = [], []
    for b0 in range(0, B, img_chunk):
        Vc = V[b0:b0 + img_chunk]                            # [c, D, S]
        s = torch.einsum("cdr,nd->crn", Vc, T)               # [c, S, N]
        a1 = _segment_softmax(s, seg, nsent)                 # softmax over words of each sentence
        a2 = torch.softmax(a1 * temp1, dim=1)                # over regions
        ctx = torch.einsum("cdr,crn->cnd", Vc, a2)           # [c, N, D]
        dot = (ctx * T.unsqueeze(0)).sum(-1)
        cos = dot / (ctx.norm(2, dim=-1) * tnorm.unsqueeze(0)).clamp(min=eps)
        ex = torch.exp(cos * temp2)                          # [c, N]
        if agg == "max":                                     # inference variant, gloria_model.py:199
            acc = torch.zeros(ex.shape[0], nsent, dtype=ex.dtype).scatter_reduce(
                1, seg.expand(ex.shape), ex, "amax", include_self=False)
        else:
            acc = torch.zeros(ex.shape[0], nsent, dtype=ex.dtype).scatter_add(1, seg.expand(ex.shape), ex)
            if agg != "sum":
                acc = acc / torch.tensor([float(n) for n in cap_lens], dtype=ex.dtype)
        sims.append(torch.log(acc) * temp3)
        if return_attn:
            attns.append(a2.transpose(1, 2))                 # [c, N, S]
    sim = torch.cat(sims, 0)
    if return_attn:
        return sim, torch.cat(attns, 0), seg
    return sim


def sharded_local_similarity(img_features, words_emb, cap_lens, world: int, **kw):
    """Restates the data-parallel computation with a list of shards as the fake process
    group: rank r owns images [r*B/P, (r+1)*B/P), "all-gathers" every sentence, computes its
    block-row of sim, and the block-rows are concatenated."""
    B = img_features.shape[0]
    assert B % world == 0
    per = B // world
    rows = [local_similarity_matrix(img_features[r * per:(r + 1) * per], words_emb, cap_lens, **kw)
            for r in range(world)]
    return torch.cat(rows, 0)


# --------------------------------------------------------------------------------------
# model-level glue (gloria_model.py)
# --------------------------------------------------------------------------------------

def cap_lens_from_sents(sents: List[List[str]]) -> List[int]:
    """1 + number of aggregated words that do not start with '['.  Ref gloria_model.py:107-109."""
    return [sum(1 for w in s if not w.startswith("[")) + 1 for s in sents]


def attention_supervision_loss(att_maps: List[torch.Tensor], segmentation_labels: torch.Tensor, weight: float):
    """-log sum(label * normalised nearest-upsampled mean attention).  Ref gloria_model.py:143-147."""
    mean_maps = torch.cat([m.mean(1) for m in att_maps], 0)                      # [B, H, W]
    up = F.interpolate(mean_maps.unsqueeze(1), size=segmentation_labels.shape[1:]).squeeze(1)
    up = up / up.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return -torch.log((segmentation_labels * up).sum(-1).sum(-1)).mean() * weight


def calc_loss(img_emb_l, img_emb_g, text_emb_l, text_emb_g, sents, *, local_loss_weight=1.0,
              global_loss_weight=1.0, temp1=4.0, temp2=5.0, temp3=10.0, no_attn_vec=None,
              no_attn_loss_weight=None, attention_divergence_loss_weight=None,
              attention_entropy_loss_weight=None, segmentation_labels=None, segmentation_loss_weight=None):
    """Weighted total.  Ref gloria_model.py:132-150 (the local loss always runs, :135-139)."""
    cap_lens = cap_lens_from_sents(sents)
    l0, l1, na, kl, ent, maps = local_loss(
        img_emb_l, text_emb_l, cap_lens, temp1=temp1, temp2=temp2, temp3=temp3, no_attn_vec=no_attn_vec,
        no_attn_loss_weight=no_attn_loss_weight,
        attention_divergence_loss_weight=attention_divergence_loss_weight,
        attention_entropy_loss_weight=attention_entropy_loss_weight)
    total = 0
    if local_loss_weight != 0:
        total = total + (l0 + l1) * local_loss_weight
    if global_loss_weight != 0:
        g0, g1 = global_loss(img_emb_g, text_emb_g, temp3=temp3)
        total = total + (g0 + g1) * global_loss_weight
    if segmentation_labels is not None and segmentation_loss_weight:
        total = total + attention_supervision_loss(maps, segmentation_labels, segmentation_loss_weight)
    total = total + na + kl + ent
    return total, maps


def local_similarities_inference(img_emb_l, text_emb_l, cap_lens, no_attn_vec=None):
    """Inference variant: words 1..n (skips [CLS]), temps 4/5 hard-coded, MAX over words,
    no temp3.  Ref gloria_model.py:171-207."""
    B = img_emb_l.shape[0]
    cols = []
    for i in range(len(text_emb_l)):
        n = int(cap_lens[i])
        word = text_emb_l[i, :, 1:n + 1].unsqueeze(0).expand(B, -1, -1).contiguous()
        wctx, _ = attention_fn(word, img_emb_l, 4.0, no_attn_vec=no_attn_vec)
        row = cosine_similarity(word.transpose(1, 2).reshape(B * n, -1),
                                wctx.transpose(1, 2).reshape(B * n, -1)).reshape(B, n)
        cols.append(torch.log(torch.exp(row * 5.0).max(1, keepdim=True).values))
    return torch.cat(cols, 1)


# --------------------------------------------------------------------------------------
# text post-processing (text_model.py)
# --------------------------------------------------------------------------------------

def aggregate_wordpieces(hidden: torch.Tensor, caption_ids: torch.Tensor, idxtoword: dict):
    """Word-piece -> word aggregation.  Ref text_model.py:32-90.

    hidden [B, layers, L, D] (already permuted like :105), caption_ids [B, L].
    Scans tokens left to right; '##' pieces join the open word (sum); any other token closes
    it; '[SEP]' closes the open word, is appended as its own word and stops the scan; the
    rest is zero / '[PAD]'.  Returns (agg [B, layers, L, D], sentences).
    """
    B, nl, L, D = hidden.shape
    out = torch.zeros_like(hidden)
    sentences = []
    for b in range(B):
        words, k = [], 0
        bank_vec, bank_str = None, []
        for t in range(L):
            tok = idxtoword[int(caption_ids[b, t])]
            vec = hidden[b, :, t]
            if tok == "[SEP]":
                out[b, :, k] = bank_vec
                words.append("".join(bank_str)); k += 1
                out[b, :, k] = vec
                words.append(tok); k += 1
                break
            if tok.startswith("##"):
                bank_vec = vec.clone() if bank_vec is None else bank_vec + vec
                bank_str.append(tok[2:])
            else:
                if bank_str:
                    out[b, :, k] = bank_vec
                    words.append("".join(bank_str)); k += 1
                bank_vec, bank_str = vec.clone(), [tok]
        sentences.append(words + ["[PAD]"] * (L - k))
    return out, sentences


def text_postprocess(hidden_states: Sequence[torch.Tensor], caption_ids, idxtoword, last_n_layers=4,
                     aggregate_method="sum"):
    """Stack last layers, merge word pieces, sum over layers, mean over ALL L slots for the
    sentence embedding.  Ref text_model.py:96-131.  Returns (word_emb [B, D, L], sent_emb [B, D], sents)."""
    emb = torch.stack(list(hidden_states[-last_n_layers:])).permute(1, 0, 2, 3)   # [B, layers, L, D]
    emb, sents = aggregate_wordpieces(emb, caption_ids, idxtoword)
    sent = emb.mean(2)
    if aggregate_method == "sum":
        word, sent = emb.sum(1), sent.sum(1)
    else:
        word, sent = emb.mean(1), sent.mean(1)
    return word.permute(0, 2, 1), sent, sents
