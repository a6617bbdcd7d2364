"""Deterministic synthetic inputs shared by oracle/gen_golden.py (which runs the real
reference on them, in the build container only) and by the tests (which re-create the
same inputs anywhere, so the fixtures only have to hold the reference's OUTPUTS).

numpy's PCG64 stream for a given seed is platform independent, so the GPU box regenerates
bit-identical inputs.  Values follow SURVEY.md 8d: embeddings ~ N(0, 0.5^2).
"""

import numpy as np

STD = 0.5


def normal(seed, *shape, std=STD):
    return (np.random.default_rng(seed).standard_normal(shape, dtype=np.float32) * std).astype(np.float32)


def subsample(a, limit=65536):
    """Fixtures keep every element of small arrays and a prime-strided sample of large ones
    (plus sum / sum of squares, stored separately)."""
    flat = np.asarray(a).reshape(-1)
    if flat.size <= limit:
        return flat.copy()
    stride = 97 if flat.size > 97 * 4096 else 31
    return flat[::stride].copy()


def checksums(a):
    flat = np.asarray(a, dtype=np.float64).reshape(-1)
    return np.array([flat.sum(), (flat * flat).sum()], dtype=np.float64)


# ---------------------------------------------------------------------------- case tables

ATTN_CASES = {
    # name: (B, D, H, W, n, no_attn_vec, temp1)
    "a1_tiny": (2, 64, 5, 5, 3, False, 4.0),
    "a2_base": (4, 768, 19, 19, 17, False, 4.0),
    "a3_noattn_n96": (2, 768, 19, 19, 96, True, 4.0),
    "a4_stress_197x256": (2, 768, 14, 14, 256, True, 4.0),
}

LOCAL_CASES = {
    # name: dict(B, D, H, W, L, cap_lens or ("rand", lo, hi), no_attn_vec, aux (3 weights or None), agg, grads)
    "l1_tiny": dict(B=4, D=64, H=5, W=5, L=12, cap_lens=[12, 7, 2, 1], no_attn=False, aux=None, agg="sum", grads=True),
    "l2_edges": dict(B=8, D=768, H=19, W=19, L=97, cap_lens=[96, 96, 40, 23, 11, 5, 2, 1], no_attn=False,
                     aux=None, agg="sum", grads=True),
    "l3_noattn_aux": dict(B=8, D=768, H=19, W=19, L=97, cap_lens=[96, 61, 33, 32, 17, 16, 3, 1], no_attn=True,
                          aux=(1.0, 0.1, 1.0), agg="sum", grads=True),
    "l4_mean": dict(B=4, D=768, H=19, W=19, L=97, cap_lens=[31, 30, 9, 2], no_attn=False, aux=None, agg="mean",
                    grads=True),
    "l5_b16_mix": dict(B=16, D=768, H=19, W=19, L=97, cap_lens=("rand", 5, 40), no_attn=False, aux=None,
                       agg="sum", grads=True),
    "l6_noattn_plain": dict(B=4, D=768, H=19, W=19, L=97, cap_lens=[40, 22, 21, 6], no_attn=True, aux=None,
                            agg="sum", grads=True),
}

# similarity-matrix-only cases at bench sizes (reference run under no_grad)
SIM_CASES = {
    "s1_b64_mix": dict(B=64, D=768, H=19, W=19, L=97, cap_lens=("rand", 5, 40), no_attn=False),
    "s2_b256_mix": dict(B=256, D=768, H=19, W=19, L=97, cap_lens=("rand", 5, 40), no_attn=False),
}

GLOBAL_CASES = {
    # name: (B, D, zero_row)
    "g1_b8_zero_row": (8, 768, 3),
    "g2_b16": (16, 768, None),
    "g3_b256": (256, 768, None),
}


def case_seed(name):
    return 1234 + sum(ord(c) * (i + 1) for i, c in enumerate(name)) % 100000


def cap_lens_of(cfg, seed):
    cl = cfg["cap_lens"]
    if isinstance(cl, tuple):
        _, lo, hi = cl
        lens = np.random.default_rng(seed + 7).integers(lo, hi + 1, size=cfg["B"])
        return sorted((int(x) for x in lens), reverse=True)     # collate sorts by length, descending
    return list(cl)


def local_inputs(name, table=None):
    cfg = (table or {**LOCAL_CASES, **SIM_CASES})[name]
    seed = case_seed(name)
    img = normal(seed, cfg["B"], cfg["D"], cfg["H"], cfg["W"])
    words = normal(seed + 1, cfg["B"], cfg["D"], cfg["L"])
    no_attn = normal(seed + 2, cfg["D"], std=1.0) if cfg["no_attn"] else None
    return img, words, cap_lens_of(cfg, seed), no_attn


def attn_inputs(name):
    B, D, H, W, n, na, temp1 = ATTN_CASES[name]
    seed = case_seed(name)
    query1 = normal(seed, 1, D, n)
    query = np.repeat(query1, B, axis=0)
    ctx = normal(seed + 1, B, D, H, W)
    no_attn = normal(seed + 2, D, std=1.0) if na else None
    return query, ctx, temp1, no_attn


def attn_upstream(name, wc_shape, map_shape):
    """upstream gradients of attention_fn's two outputs for the attention_grad fixtures"""
    seed = case_seed(name)
    return normal(seed + 10, *wc_shape), normal(seed + 11, *map_shape)


def global_inputs(name):
    B, D, zero_row = GLOBAL_CASES[name]
    seed = case_seed(name)
    img = normal(seed, B, D)
    txt = normal(seed + 1, B, D)
    if zero_row is not None:
        img[zero_row] = 0.0
    return img, txt


# synthetic vocabulary for the word-piece aggregation fixtures
VOCAB_SIZE = 2000
PAD, UNK, CLS, SEP, MASK = 0, 100, 101, 102, 103


def synthetic_vocab():
    """id -> token string; 20 % of ordinary ids are '##' continuation pieces."""
    rng = np.random.default_rng(99)
    cont = rng.random(VOCAB_SIZE) < 0.2
    table = {}
    for i in range(VOCAB_SIZE):
        table[i] = ("##p%d" % i) if cont[i] else ("w%d" % i)
    table.update({PAD: "[PAD]", UNK: "[UNK]", CLS: "[CLS]", SEP: "[SEP]", MASK: "[MASK]"})
    return table


def text_inputs(B=6, L=97, D=48, layers=4, seed=4321):
    """caption ids ([CLS] words... [SEP] [PAD]...) + `layers` hidden-state tensors [B, L, D]."""
    rng = np.random.default_rng(seed)
    vocab = synthetic_vocab()
    ids = np.zeros((B, L), dtype=np.int64)
    lens = [L - 2, 40, 17, 5, 1, 30][:B]
    for b in range(B):
        n = lens[b]
        body = rng.integers(104, VOCAB_SIZE, size=n)
        if vocab[int(body[0])].startswith("##"):
            body[0] = 104 + int(np.argmax([not vocab[i].startswith("##") for i in range(104, VOCAB_SIZE)]))
        if b == 2:
            body[3] = MASK                      # bracket token mid-sentence
        ids[b, 0] = CLS
        ids[b, 1:1 + n] = body
        ids[b, 1 + n] = SEP
    hidden = [normal(seed + 10 + k, B, L, D) for k in range(layers + 1)]
    return ids, hidden, vocab
