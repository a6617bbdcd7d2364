"""Synthetic batches with the layout the reference's collate functions emit
(/root/reference/gloria/datasets/mimic_for_gloria.py:86-108, pretraining_dataset.py:250-282):
imgs float32 [B, 3, 224, 224] in [-1, 1]; caption_ids / attention_mask / token_type_ids int64 [B, 97];
cap_lens int64 [B] sorted descending with every field permuted alike; optional segmentation_labels
bool [B, 224, 224] (random axis-aligned boxes covering 5-40 % of the image).  Values per SURVEY.md 8d."""

import numpy as np
import torch

PAD, CLS, SEP = 0, 101, 102

_CONT = {}


def _cont_table(vocab_size, seed=99, cont_frac=0.2):
    """'##' flags of gloria.models.text_model.Vocab.synthetic (same stream), without building the strings"""
    key = (vocab_size, seed, cont_frac)
    if key not in _CONT:
        cont = np.random.default_rng(seed).random(vocab_size) < cont_frac
        cont[:104] = False
        _CONT[key] = (np.nonzero(~cont)[0][np.nonzero(~cont)[0] >= 104], np.nonzero(cont)[0])
    return _CONT[key]


def _word_bodies(rng, n_words, vocab_size, max_tokens):
    """token ids of `n_words` words: every word opens with an ordinary piece; extra '##' pieces (a quarter of a
    piece per word on average, i.e. 20 % of all tokens) follow randomly chosen words"""
    heads, conts = _cont_table(vocab_size)
    n_extra = min(int(rng.binomial(n_words, 0.25)), max_tokens - n_words)
    after = np.sort(rng.integers(0, n_words, size=n_extra))
    body = []
    k = 0
    for w in range(n_words):
        body.append(int(heads[rng.integers(0, len(heads))]))
        while k < n_extra and after[k] == w:
            body.append(int(conts[rng.integers(0, len(conts))]))
            k += 1
    return body


def make_batch(batch_size, seed=1234, word_num=97, vocab_size=28996, lengths="mix", imsize=224,
               segmentation=False, device=None):
    rng = np.random.default_rng(seed)
    bodies = None
    if lengths == "words":
        # sentence length ~ U{4..39} WORDS (cap_lens = words + 1, mean 22.5); the pieces follow
        n_words = np.sort(rng.integers(4, 40, size=batch_size))[::-1]
        bodies = [_word_bodies(rng, int(n), vocab_size, word_num - 2) for n in n_words]
        n_tok = np.array([len(b) for b in bodies])
    elif lengths == "mix":
        n_tok = rng.integers(4, 40, size=batch_size)          # word PIECES between [CLS] and [SEP] (20 % are '##')
    elif lengths == "max":
        n_tok = np.full(batch_size, word_num - 2)
    else:
        n_tok = np.asarray(lengths)
    if bodies is None:
        n_tok = np.sort(n_tok)[::-1].copy()
    ids = np.zeros((batch_size, word_num), dtype=np.int64)
    mask = np.zeros_like(ids)
    for b in range(batch_size):
        n = int(n_tok[b])
        body = bodies[b] if bodies is not None else rng.integers(104, vocab_size, size=n)
        ids[b, 0], ids[b, 1:1 + n], ids[b, 1 + n] = CLS, body, SEP
        mask[b, :n + 2] = 1
    g = torch.Generator().manual_seed(seed)
    batch = {
        "imgs": torch.rand(batch_size, 3, imsize, imsize, generator=g) * 2 - 1,
        "caption_ids": torch.from_numpy(ids),
        "attention_mask": torch.from_numpy(mask),
        "token_type_ids": torch.zeros(batch_size, word_num, dtype=torch.int64),
        "cap_lens": torch.from_numpy((np.asarray(n_tok) + 2).astype(np.int64)),
    }
    if segmentation:
        lab = torch.zeros(batch_size, imsize, imsize, dtype=torch.bool)
        for b in range(batch_size):
            frac = rng.uniform(0.05, 0.40)
            h = int(np.clip(np.sqrt(frac) * imsize * rng.uniform(0.7, 1.4), 8, imsize))
            w = int(np.clip(frac * imsize * imsize / h, 8, imsize))
            y0, x0 = rng.integers(0, imsize - h + 1), rng.integers(0, imsize - w + 1)
            lab[b, y0:y0 + h, x0:x0 + w] = True
        batch["segmentation_labels"] = lab
    if device is not None:
        batch = {k: v.to(device) for k, v in batch.items()}
    return batch


class _Loader:
    def __init__(self, n_batches, batch_size, seed, **kw):
        self.n_batches, self.batch_size, self.seed, self.kw = n_batches, batch_size, seed, kw
        self.dataset = range(n_batches * batch_size)

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        for i in range(self.n_batches):
            yield make_batch(self.batch_size, seed=self.seed + i, **self.kw)


class SyntheticPretrainDataModule:
    def __init__(self, cfg):
        self.cfg = cfg
        self.batch_size = cfg.train.batch_size
        self.n_train = cfg.data.synthetic_train_batches or 8
        self.n_val = cfg.data.synthetic_val_batches or 2
        self.kw = dict(word_num=cfg.data.text.word_num or 97,
                       segmentation=bool(cfg.model.gloria.segmentation_loss_weight))

    def train_dataloader(self):
        return _Loader(self.n_train, self.batch_size, 1234, **self.kw)

    def val_dataloader(self):
        return _Loader(self.n_val, self.batch_size, 99991, **self.kw)

    def test_dataloader(self):
        return _Loader(self.n_val, self.batch_size, 77773, **self.kw)
