"""BERT-base encoder on plain torch ops (hipBLASLt GEMMs + fused SDPA on ROCm), parameter names
identical to HuggingFace `BertModel` so `gloria.text_encoder.model.*` keys of reference checkpoints
load unchanged (the reference builds it with AutoModel.from_pretrained,
/root/reference/gloria/models/text_model.py:18-20).  transformers is not needed at run time.
"""

import torch
import torch.nn as nn
import torch.nn.functional as F

from .fused_attn import packed_fusable, self_attention, self_attention_packed
from .fused_embed import type_embedding, word_embedding
from .fused_linear import linear
from .fused_ln import dense_drop_add_ln


class BertConfig:
    def __init__(self, vocab_size=28996, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2,
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12):
        self.vocab_size, self.hidden_size = vocab_size, hidden_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.intermediate_size, self.max_position_embeddings = intermediate_size, max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.hidden_dropout_prob, self.attention_probs_dropout_prob = hidden_dropout_prob, attention_probs_dropout_prob
        self.layer_norm_eps = layer_norm_eps


class BertEmbeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)
        self.register_buffer("position_ids", torch.arange(c.max_position_embeddings).unsqueeze(0), persistent=False)

    def forward(self, ids, token_type):
        L = ids.shape[1]
        if token_type is None:
            token_type = torch.zeros_like(ids)
        # positions are arange(L) (position_ids[:, :L]): the first L rows of the table, broadcast over the batch - the
        # backward is then a plain sum over the batch instead of the sort-and-scatter of an embedding lookup (0.55 ms)
        # the two lookups keep torch's gather; their table gradients skip torch's device sort (models/fused_embed.py)
        x = word_embedding(self.word_embeddings, ids) + type_embedding(self.token_type_embeddings, token_type) \
            + self.position_embeddings.weight[:L]
        return self.dropout(self.LayerNorm(x))


class BertSelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.nh, self.hd = c.num_attention_heads, c.hidden_size // c.num_attention_heads
        self.query = nn.Linear(c.hidden_size, c.hidden_size)
        self.key = nn.Linear(c.hidden_size, c.hidden_size)
        self.value = nn.Linear(c.hidden_size, c.hidden_size)
        self.p = c.attention_probs_dropout_prob

    def forward(self, x, bias):
        # bias: bool key mask [B, 1, 1, L] (True = attend) or None.  softmax(QK^T) -> dropout -> .V on the Linear outputs
        # as they are: one HIP kernel per direction for short captions under bf16 autocast (models/fused_attn.py)
        B, L, H = x.shape
        key_mask = None if bias is None else bias.reshape(B, L)
        if packed_fusable(x, self.nh, H, L, key_mask):
            # one [H -> 3H] GEMM instead of three [H -> H] ones (at 25k tokens x 768 the library runs the small ones at
            # under 200 TFLOP/s), and the kernels read / write the packed tensor in place: no split, no gradient adds
            w = torch.cat([self.query.weight, self.key.weight, self.value.weight], 0)
            b_ = torch.cat([self.query.bias, self.key.bias, self.value.bias], 0)
            return self_attention_packed(linear(x, w, b_), key_mask, self.nh, self.p, self.training)
        return self_attention(self.query(x), self.key(x), self.value(x), key_mask, self.nh, self.p, self.training)


class BertSelfOutput(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)

    def forward(self, h, inp):
        # (y fp32, y bf16 | None): dropout + residual + LayerNorm in one HIP pass under bf16 autocast (models/fused_ln.py)
        return dense_drop_add_ln(self.dense, h, inp, self.LayerNorm, self.dropout.p, self.training)


class BertAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self = BertSelfAttention(c)
        self.output = BertSelfOutput(c)

    def forward(self, x, bias, x16=None):
        # x16: the bf16 copy of x the fused sub-layer epilogue already wrote (the operand autocast would cast x to)
        return self.output(self.self(x if x16 is None else x16, bias), x)


class BertIntermediate(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)

    def forward(self, x):
        return F.gelu(linear(x, self.dense.weight, self.dense.bias))


class BertOutput(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.intermediate_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)

    def forward(self, h, inp):
        return dense_drop_add_ln(self.dense, h, inp, self.LayerNorm, self.dropout.p, self.training)


class BertLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attention = BertAttention(c)
        self.intermediate = BertIntermediate(c)
        self.output = BertOutput(c)

    def forward(self, x, bias, x16=None):
        """(hidden state fp32, its bf16 copy or None)"""
        a, a16 = self.attention(x, bias, x16)
        return self.output(self.intermediate(a if a16 is None else a16), a)


class BertEncoderStack(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(c) for _ in range(c.num_hidden_layers)])


class BertStackPath(nn.Module):
    """The encoder layers as an (embedding output, key mask) -> (last n hidden states) module of tensors only: the
    callable BertEncoder.enable_graph hands to torch.cuda.make_graphed_callables.  The embeddings stay outside (their
    backward sorts the token ids: launch shapes that depend on the data), and so does the pooler (unused parameters)."""

    def __init__(self, encoder, n_out):
        super().__init__()
        self.encoder = encoder
        self.n_out = int(n_out)

    def forward(self, x, key_mask4):
        hidden, x16 = [], None
        for layer in self.encoder.layer:
            x, x16 = layer(x, key_mask4, x16)
            hidden.append(x)
        return tuple(hidden[-self.n_out:])


class BertPooler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)

    def forward(self, x):
        return torch.tanh(self.dense(x[:, 0]))


class BertModel(nn.Module):
    """forward(ids, attention_mask, token_type_ids) -> (last_hidden, pooled, all_hidden_states)
    like HF BertModel(output_hidden_states=True) indexed positionally (outputs[0], [1], [2])."""

    def __init__(self, config=None):
        super().__init__()
        self.config = c = config or BertConfig()
        self.embeddings = BertEmbeddings(c)
        self.encoder = BertEncoderStack(c)
        self.pooler = BertPooler(c)
        self.apply(self._init)

    @staticmethod
    def _init(m):
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, std=0.02)
            nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, std=0.02)
            if m.padding_idx is not None:
                with torch.no_grad():
                    m.weight[m.padding_idx].zero_()
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

    def forward(self, ids, attention_mask=None, token_type_ids=None):
        x = self.embeddings(ids, token_type_ids)
        bias = None
        if attention_mask is not None:          # boolean key mask (True = attend), broadcast over heads/queries
            bias = (attention_mask != 0)[:, None, None, :]
        hidden = [x]
        x16 = None
        for layer in self.encoder.layer:
            x, x16 = layer(x, bias, x16)
            hidden.append(x)
        return x, self.pooler(x), tuple(hidden)
