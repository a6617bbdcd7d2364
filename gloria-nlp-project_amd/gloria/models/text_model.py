"""BertEncoder with the reference's interface (/root/reference/gloria/models/text_model.py:6-144):
forward(ids, attn_mask, token_type) -> (word_embeddings [B, D, L], sent_embeddings [B, D], sents).

What changes underneath:
  * the BERT stack is gloria.models.bert.BertModel (HF-compatible parameter names);
  * `aggregate_tokens` (ref :32-90: a Python double loop with one `.item()` device sync per token)
    becomes a segment-sum: word-piece -> word slot indices are computed on the host from the caption
    ids with vectorised numpy (one small copy per step, no per-token sync) and the pieces are summed
    into their word slot on the GPU.  Because merging pieces is a sum, the 4-layer sum (ref :112)
    is taken first and merged once.
  * `sents` is a SentenceBatch: a list of B lists of L word strings like the reference returns, built
    lazily, that also carries the cap_lens the loss needs (ref gloria_model.py:107-109) so the hot
    path never touches Python strings.
"""

import os

import numpy as np
import torch
import torch.nn as nn

from .. import _native as N
from .bert import BertConfig, BertModel

PAD, UNK, CLS, SEP, MASK = 0, 100, 101, 102, 103


class Vocab:
    """id -> token properties needed by the aggregation (and, optionally, the strings)."""

    def __init__(self, tokens):
        self.tokens = list(tokens)
        self.is_cont = np.array([t.startswith("##") for t in self.tokens], dtype=bool)
        self.is_bracket = np.array([t.startswith("[") for t in self.tokens], dtype=bool)
        self.sep_id = self.tokens.index("[SEP]")

    def __len__(self):
        return len(self.tokens)

    @classmethod
    def synthetic(cls, size=28996, seed=99, cont_frac=0.2):
        """Offline stand-in for the Bio_ClinicalBERT word-piece vocabulary:
        ids 0/100/101/102/103 = PAD/UNK/CLS/SEP/MASK, `cont_frac` of the rest are '##' pieces."""
        rng = np.random.default_rng(seed)
        cont = rng.random(size) < cont_frac
        toks = [("##p%d" % i) if cont[i] else ("w%d" % i) for i in range(size)]
        for i, t in ((PAD, "[PAD]"), (UNK, "[UNK]"), (CLS, "[CLS]"), (SEP, "[SEP]"), (MASK, "[MASK]")):
            toks[i] = t
        return cls(toks)

    @classmethod
    def from_file(cls, path):
        with open(path, encoding="utf-8") as f:
            return cls([line.rstrip("\n") for line in f])

    @classmethod
    def from_dict(cls, idxtoword):
        return cls([idxtoword[i] for i in range(len(idxtoword))])


class SentenceBatch(list):
    """list of B word lists (materialised on first element access) + host cap_lens."""

    def __init__(self, ids, dst, starts, n_words, vocab, L):
        super().__init__()
        self._ids, self._dst, self._n_words, self._vocab, self._L = ids, dst, n_words, vocab, L
        first = np.full((ids.shape[0], L), -1, dtype=np.int64)      # first token id of every word slot
        b, t = np.nonzero((dst >= 0) & starts)                       # exactly one opening token per word
        first[b, dst[b, t]] = ids[b, t]
        has = first >= 0
        brk = np.zeros_like(has)
        brk[has] = vocab.is_bracket[first[has]]
        # cap_len = 1 + #words not starting with "[" ; padding slots are "[PAD]" (ref gloria_model.py:107-109)
        self.cap_lens = (1 + (has & ~brk).sum(1)).astype(np.int64).tolist()
        self._built = False

    def _build(self):
        if self._built:
            return
        toks = self._vocab.tokens
        out = []
        for b in range(self._ids.shape[0]):
            words = [""] * int(self._n_words[b])
            for t in range(self._ids.shape[1]):
                k = self._dst[b, t]
                if k >= 0:
                    s = toks[int(self._ids[b, t])]
                    words[k] += s[2:] if s.startswith("##") else s
            out.append(words + ["[PAD]"] * (self._L - len(words)))
        list.extend(self, out)
        self._built = True

    def __iter__(self):
        self._build()
        return list.__iter__(self)

    def __getitem__(self, i):
        self._build()
        return list.__getitem__(self, i)

    def __len__(self):
        return self._ids.shape[0]


def wordpiece_slots(ids: np.ndarray, vocab: Vocab):
    """Vectorised restatement of the scan of text_model.py:48-76.
    ids [B, L] -> dst [B, L] (word slot of every token, -1 = dropped), starts [B, L], n_words [B].
    A token that is not a '##' piece opens a new word; '[SEP]' is its own word and ends the scan;
    without a '[SEP]' the last open word is never flushed (dropped), exactly like the reference."""
    B, L = ids.shape
    starts = ~vocab.is_cont[ids]
    starts[:, 0] = True
    widx = np.cumsum(starts, axis=1) - 1
    is_sep = ids == vocab.sep_id
    has_sep = is_sep.any(1)
    sep_pos = np.where(has_sep, is_sep.argmax(1), L)
    t = np.arange(L)[None, :]
    keep = t <= sep_pos[:, None]
    last = widx[:, -1]
    keep &= has_sep[:, None] | (widx < last[:, None])
    dst = np.where(keep, widx, -1)
    n_words = np.where(has_sep, widx[np.arange(B), np.minimum(sep_pos, L - 1)] + 1, last)
    return dst, starts, n_words


class WordpieceSegSumFn(torch.autograd.Function):
    """K5: (hidden states of the last layers, token -> word-slot index) -> word_emb [B, D, L], sent_emb [B, D]."""

    @staticmethod
    def forward(ctx, dst, mean_layers, *layers):
        from .. import _native as N
        import ctypes
        hs = [h.detach().contiguous() for h in layers]
        if hs[0].dtype not in (torch.float32, torch.bfloat16):
            hs = [h.float() for h in hs]
        B, L, D = hs[0].shape
        dev = hs[0].device
        word = torch.empty(B, D, L, dtype=torch.float32, device=dev)
        sent = torch.empty(B, D, dtype=torch.float32, device=dev)
        ptrs = (ctypes.c_void_p * len(hs))(*[h.data_ptr() for h in hs])
        N.check(N.lib().glr_wordpiece_segsum_fwd(ptrs, len(hs), N.dtype_code(hs[0].dtype), N.ptr(dst), N.ptr(word),
                                                 N.ptr(sent), B, L, D, 1 if mean_layers else 0, N.stream()),
                "glr_wordpiece_segsum_fwd")
        ctx.save_for_backward(dst)
        ctx.meta = (B, L, D, len(hs), mean_layers, [h.dtype for h in layers])
        return word, sent

    @staticmethod
    def backward(ctx, d_word, d_sent):
        from .. import _native as N
        (dst,) = ctx.saved_tensors
        B, L, D, nl, mean_layers, dts = ctx.meta
        dw = None if d_word is None else d_word.float().contiguous()
        ds = None if d_sent is None else d_sent.float().contiguous()
        out_dt = torch.bfloat16 if dts[0] == torch.bfloat16 else torch.float32
        dh = torch.empty(B, L, D, dtype=out_dt, device=dst.device)
        N.check(N.lib().glr_wordpiece_segsum_bwd(N.ptr(dw), N.ptr(ds), N.ptr(dst), N.ptr(dh), N.dtype_code(out_dt),
                                                 B, L, D, nl, 1 if mean_layers else 0, N.stream()),
                "glr_wordpiece_segsum_bwd")
        return (None, None) + tuple(dh.to(dt) for dt in dts)


def host_ids(ids):
    """caption ids as a host array WITHOUT a device sync when the trainer kept the host copy (Trainer.to_device
    attaches it as `_glr_host`); a plain `.cpu()` here would drain the stream every step (the image encoder is
    already queued) and expose the launch latency of everything behind it."""
    host = getattr(ids, "_glr_host", None)
    if host is not None:
        return host
    return ids.detach().cpu().numpy() if torch.is_tensor(ids) else np.asarray(ids)


class BertEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.bert_type = cfg.model.text.bert_type
        self.last_n_layers = cfg.model.text.last_n_layers
        self.aggregate_method = cfg.model.text.aggregate_method
        self.norm = cfg.model.text.norm
        self.embedding_dim = cfg.model.text.embedding_dim
        self.freeze_bert = cfg.model.text.freeze_bert
        self.agg_tokens = cfg.model.text.agg_tokens

        # no network: a local directory with config/weights/vocab may be given as bert_type,
        # otherwise BERT-base geometry with random init and the synthetic vocabulary.
        bcfg = BertConfig(hidden_size=self.embedding_dim) if self.embedding_dim != 768 else BertConfig()
        if cfg.model.text.bert_config is not None:
            bcfg = BertConfig(**cfg.model.text.bert_config)
        self.model = BertModel(bcfg)
        vocab_file = os.path.join(str(self.bert_type), "vocab.txt")
        if os.path.exists(vocab_file):
            self.vocab = Vocab.from_file(vocab_file)
            weights = os.path.join(str(self.bert_type), "model.safetensors")
            if os.path.exists(weights):
                from safetensors.torch import load_file
                sd = {k.replace("bert.", "", 1) if k.startswith("bert.") else k: v for k, v in load_file(weights).items()}
                self.model.load_state_dict(sd, strict=False)
        else:
            self.vocab = Vocab.synthetic(bcfg.vocab_size)
        self.idxtoword = None          # built on demand (reference attribute, text_model.py:23)

        # set by enable_graph(); out of the module tree (the graphed callable wraps self.model.encoder)
        object.__setattr__(self, "_graph", None)
        self._graph_key = None
        self._graph_rng = None

        self.emb_global, self.emb_local = None, None
        if self.freeze_bert is True:
            print("Freezing BERT model")
            for param in self.model.parameters():
                param.requires_grad = False

    def enable_graph(self, ids, attn_mask, token_type, autocast_dtype=None, warmup=3):
        """Capture the encoder layers' forward AND backward (12 layers: ~600 of the step's launches, static shapes) into
        two hipGraphs (torch.cuda.make_graphed_callables on bert.BertStackPath).  Small per-rank batches are bound by the
        HOST time of exactly these launches.  Dropout keys of the fused kernels come from a device cell rewritten before
        every replay (models/rng.py), so a replay draws fresh masks from torch's generator like the eager launches."""
        from .. import hipgraph
        from .bert import BertStackPath
        from .rng import GraphRng, capturing
        n_layers = len(self.model.encoder.layer)
        if (not ids.is_cuda or attn_mask is None or self.freeze_bert is True or self.last_n_layers <= 1
                or self.last_n_layers > n_layers or not hipgraph.usable("text encoder")):
            return False
        path = BertStackPath(self.model.encoder, self.last_n_layers)
        path.train(self.training)
        ctx = torch.autocast("cuda", dtype=autocast_dtype, cache_enabled=False) if autocast_dtype is not None \
            else torch.autocast("cuda", enabled=False)
        rng = GraphRng(ids.device)
        with ctx:
            with torch.no_grad():
                x = self.model.embeddings(ids, token_type)
            x = x.detach().clone().requires_grad_(True)
            mask4 = (attn_mask != 0)[:, None, None, :].clone()
            with capturing(rng):
                graphed = torch.cuda.make_graphed_callables(path, (x, mask4), num_warmup_iters=warmup)
            if not hipgraph.verify_capture("text encoder", path, graphed, (x, mask4), before_replay=rng.refresh):
                return False
        object.__setattr__(self, "_graph", graphed)
        self._graph_key = (tuple(ids.shape), autocast_dtype)
        self._graph_rng = rng
        return True

    def _encode(self, ids, attn_mask, token_type):
        """hidden states of the last `last_n_layers` layers, through the captured graphs when they fit this call"""
        if (getattr(self, "_graph", None) is not None and self.training and torch.is_grad_enabled() and attn_mask is not None
                and self._graph_key == (tuple(ids.shape), torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None)):
            x = self.model.embeddings(ids, token_type)
            self._graph_rng.refresh()
            return self._graph(x, (attn_mask != 0)[:, None, None, :])
        return None

    def aggregate_tokens(self, embeddings, caption_ids):
        """embeddings [B, L, D] (already reduced over layers) -> word slots [B, L, D], sents."""
        ids = host_ids(caption_ids)
        B, L = ids.shape
        dst, starts, n_words = wordpiece_slots(ids, self.vocab)
        sents = SentenceBatch(ids, dst, starts, n_words, self.vocab, L)
        flat = dst + (np.arange(B) * L)[:, None]
        src = np.nonzero(dst.reshape(-1) >= 0)[0]
        idx = torch.from_numpy(np.stack([src, flat.reshape(-1)[src]])).to(embeddings.device, non_blocking=True)
        x = embeddings.reshape(B * L, -1)
        out = torch.zeros_like(x).index_add_(0, idx[1], x.index_select(0, idx[0]))
        return out.view(B, L, -1), sents

    def forward(self, ids, attn_mask, token_type):
        layers = self._encode(ids, attn_mask, token_type) if self.last_n_layers > 1 else None
        outputs = self.model(ids, attn_mask, token_type) if layers is None else None
        fused = None
        if self.last_n_layers > 1:
            if layers is None:
                layers = outputs[2][-self.last_n_layers:]
            if self.aggregate_method not in ("sum", "mean"):
                print(self.aggregate_method)
                raise Exception("Aggregation method not implemented")
            if self.agg_tokens and layers[0].is_cuda:
                # K5 (every GPU call, any D / layer count): segment-sum fused with the layer reduction and the
                # L-mean, [B, D, L] written directly.  The torch branch below only ever sees CPU tensors (host-logic
                # tests and the CPU baseline of bench.py).
                host = host_ids(ids)
                B, L = host.shape
                dst, starts, n_words = wordpiece_slots(host, self.vocab)
                sents = SentenceBatch(host, dst, starts, n_words, self.vocab, L)
                dst_d = N.upload(dst.astype(np.int32), layers[0].device)
                fused = WordpieceSegSumFn.apply(dst_d, self.aggregate_method == "mean", *layers)
                word_embeddings = fused[0].permute(0, 2, 1)        # [B, L, D] view; permuted back below
                sent_embeddings = fused[1]
            else:
                embeddings = torch.stack(layers).sum(0) if self.aggregate_method == "sum" else torch.stack(layers).mean(0)
                if self.agg_tokens:
                    word_embeddings, sents = self.aggregate_tokens(embeddings, ids)
                else:
                    word_embeddings = embeddings
                    host = host_ids(ids)
                    sents = [[self.vocab.tokens[int(w)] for w in sent] for sent in host]
                sent_embeddings = word_embeddings.mean(dim=1)          # over ALL L slots (ref :110)
        else:
            word_embeddings, sent_embeddings = outputs[0], outputs[1]
            host = host_ids(ids)
            sents = [[self.vocab.tokens[int(w)] for w in sent] for sent in host]

        batch_dim, num_words, feat_dim = word_embeddings.shape
        if self.emb_local is not None:
            word_embeddings = self.emb_local(word_embeddings.reshape(batch_dim * num_words, feat_dim))
            word_embeddings = word_embeddings.view(batch_dim, num_words, self.embedding_dim)
        word_embeddings = word_embeddings.permute(0, 2, 1)
        if self.emb_global is not None:
            sent_embeddings = self.emb_global(sent_embeddings)
        if self.norm is True:
            word_embeddings = word_embeddings / torch.norm(word_embeddings, 2, dim=1, keepdim=True).expand_as(
                word_embeddings)
            sent_embeddings = sent_embeddings / torch.norm(sent_embeddings, 2, dim=1, keepdim=True).expand_as(
                sent_embeddings)
        return word_embeddings, sent_embeddings, sents
