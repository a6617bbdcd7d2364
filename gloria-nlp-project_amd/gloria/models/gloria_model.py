"""GLoRIA nn.Module with the reference's interface
(/root/reference/gloria/models/gloria_model.py:45-211): same constructor contract, attribute names
(`text_encoder`, `img_encoder`, `no_attn_vec`, loss weights, temps), `forward`, `calc_loss`,
`_calc_local_loss`, `_calc_global_loss`, `get_global_similarities`, `get_local_similarities`,
`get_attn_maps`.  Host-side inference utilities of the reference (process_text / process_img /
plotting, :213-384) are out of scope.

Data-parallel extension: when `self.dist` is a gloria.dist.DistContext with world_size > 1, calc_loss
forms the FULL global-batch contrastive matrices: text embeddings are all-gathered over RCCL, this
rank computes its block-row of the similarity matrices with the HIP kernels, the block-rows are
all-gathered for the column-direction cross entropy, and gradients of the gathered text embeddings
are reduce-scattered back.  The reference's own 'dp' mode (per-replica negatives) is
not reproduced.
"""

import numpy as np
import os

import torch
import torch.nn as nn

from .. import builder
from .. import loss
from ..loss import gloria_loss as GL


class PositionEmbeddings(nn.Module):
    """Ref gloria_model.py:17-42 (optional, off in the pretrain configs)."""

    def __init__(self, num_positions, hidden_size, num_spatial_dims=1):
        super().__init__()
        self.num_positions, self.hidden_size, self.num_spatial_dims = num_positions, hidden_size, num_spatial_dims
        self.image_position_embeddings = nn.Embedding(num_positions, hidden_size // num_spatial_dims)

    def forward(self, spatial_shape):
        if isinstance(spatial_shape, int):
            spatial_shape = (spatial_shape,) * self.num_spatial_dims
        for d in spatial_shape:
            assert d <= self.num_positions
        device = self.image_position_embeddings.weight.device
        embs = [self.image_position_embeddings(torch.arange(d, device=device)) for d in spatial_shape]
        pos_dim = embs[0].shape[-1]
        embs = [e.reshape(*(1 if i != j else d for j, d in enumerate(spatial_shape)), pos_dim)
                 .expand(*spatial_shape, pos_dim) for i, e in enumerate(embs)]
        pad = torch.zeros(*spatial_shape, self.hidden_size - len(spatial_shape) * pos_dim, device=device)
        return torch.cat(embs + [pad], -1)


# two-stream encoders.  A plain module attribute read at every forward: tests and tools flip it
# (`gloria_model.ENCODER_STREAMS = False`) to compare the two paths in one process.
ENCODER_STREAMS = os.environ.get("GLR_ENCODER_STREAMS", "1") != "0"
_SIDE = {}


def _side_stream(dev):
    s = _SIDE.get(dev.index)
    if s is None:
        s = _SIDE[dev.index] = torch.cuda.Stream(device=dev)
    return s


class _ImagePath(nn.Module):
    """image_encoder_forward as a tensor -> (local, global) module: the callable torch.cuda.make_graphed_callables
    captures (GLoRIA.enable_image_graph)."""

    def __init__(self, img_encoder):
        super().__init__()
        self.img_encoder = img_encoder

    def forward(self, imgs):
        img_feat_g, img_emb_l = self.img_encoder(imgs, get_local=True)
        img_emb_g, img_emb_l = self.img_encoder.generate_embeddings(img_feat_g, img_emb_l)
        return img_emb_l, img_emb_g


class GLoRIA(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.text_encoder = builder.build_text_model(cfg)
        self.img_encoder = builder.build_img_model(cfg)
        self.position_embeddings = PositionEmbeddings(
            cfg.model.image_position_embeddings.num, cfg.model.text.embedding_dim, num_spatial_dims=2) \
            if cfg.model.image_position_embeddings is not None else None
        self.image_transformer = nn.TransformerEncoder(
            nn.TransformerEncoderLayer(cfg.model.text.embedding_dim, cfg.model.image_transformer.num_heads),
            cfg.model.image_transformer.num_layers) if "image_transformer" in cfg.model.keys() else None
        self.no_attn_vec = nn.Parameter(torch.randn(cfg.model.text.embedding_dim)) \
            if cfg.model.gloria.no_attn_vec else None

        self.local_loss = loss.gloria_loss.local_loss
        self.global_loss = loss.gloria_loss.global_loss
        g = cfg.model.gloria
        self.local_loss_weight = g.local_loss_weight
        self.global_loss_weight = g.global_loss_weight
        self.sparse_attn_loss_weight = g.sparse_attn_loss_weight
        self.no_attn_loss_weight = g.no_attn_loss_weight
        self.attention_divergence_loss_weight = g.attention_divergence_loss_weight
        self.attention_entropy_loss_weight = g.attention_entropy_loss_weight
        self.segmentation_loss_weight = g.segmentation_loss_weight
        self.temp1, self.temp2, self.temp3 = g.temp1, g.temp2, g.temp3
        self.batch_size = cfg.train.batch_size
        self.dist = None                        # set by the trainer for data-parallel runs
        # set by enable_image_graph(); kept OUT of the module tree (object.__setattr__): the graphed callable wraps
        # img_encoder, and a registered copy would duplicate every image-encoder key in state_dict() / checkpoints
        object.__setattr__(self, "_img_graph", None)
        self._img_graph_shape = None
        self.ixtoword = None                    # reference attribute (:79); strings live in text_encoder.vocab

    # ------------------------------------------------------------------ encoders (ref :81-103)
    def text_encoder_forward(self, caption_ids, attention_mask, token_type_ids):
        return self.text_encoder(caption_ids, attention_mask, token_type_ids)

    def enable_image_graph(self, sample_imgs, autocast_dtype=None, warmup=3):
        """Capture the image encoder's forward AND backward (static shapes, no dropout, no data-dependent allocation:
        ~500 of the step's ~1100 launches) into two hipGraphs (torch.cuda.make_graphed_callables).  Small per-rank batches
        are bound by the HOST time of these launches; a replay costs the host two.  The text encoder's layers are captured
        the same way by `enable_text_graph`.  Refused unless the HIP runtime's graph packet capture is off (gloria/hipgraph.py:
        it races with MIOpen's memset nodes), and verified against an eager pass before it is used.  BatchNorm buffers are
        restored after the capture's warm-up / verification passes, so the first real step sees the statistics an eager
        run would."""
        from .. import hipgraph
        if not sample_imgs.is_cuda or self.position_embeddings is not None or self.image_transformer is not None:
            return False
        if not hipgraph.usable("image encoder"):
            return False
        path = _ImagePath(self.img_encoder)
        path.train(self.training)
        keep = {k: v.detach().clone() for k, v in self.img_encoder.named_buffers()}
        ctx = torch.autocast("cuda", dtype=autocast_dtype, cache_enabled=False) if autocast_dtype is not None \
            else torch.autocast("cuda", enabled=False)
        sample = (sample_imgs.detach().clone(),)
        with ctx:
            graphed = torch.cuda.make_graphed_callables(path, sample, num_warmup_iters=warmup)
            # replays against an eager pass before the graph is trusted (gloria/hipgraph.py: memset nodes race on this stack)
            ok = hipgraph.verify_capture("image encoder", path, graphed, sample)
        with torch.no_grad():
            for k, v in self.img_encoder.named_buffers():
                v.copy_(keep[k])
        if not ok:
            return False
        object.__setattr__(self, "_img_graph", graphed)
        self._img_graph_shape = (tuple(sample_imgs.shape), sample_imgs.dtype)
        return True

    def enable_text_graph(self, caption_ids, attention_mask, token_type_ids, autocast_dtype=None, warmup=3):
        """hipGraph capture of the text encoder's layers, forward and backward (text_model.BertEncoder.enable_graph)"""
        if not hasattr(self.text_encoder, "enable_graph"):
            return False
        return self.text_encoder.enable_graph(caption_ids, attention_mask, token_type_ids, autocast_dtype, warmup)

    def image_encoder_forward(self, imgs):
        if (self._img_graph is not None and self.training and torch.is_grad_enabled()
                and (tuple(imgs.shape), imgs.dtype) == self._img_graph_shape):
            return self._img_graph(imgs)
        img_feat_g, img_emb_l = self.img_encoder(imgs, get_local=True)
        img_emb_g, img_emb_l = self.img_encoder.generate_embeddings(img_feat_g, img_emb_l)
        b, c, h, w = img_emb_l.shape
        if self.position_embeddings is not None:
            pos = self.position_embeddings((h, w)).permute(2, 0, 1).expand(b, c, h, w)
            img_emb_l = img_emb_l + pos
        if self.image_transformer is not None:
            flat = img_emb_l.reshape(b, c, h * w).permute(2, 0, 1)
            img_emb_l = self.image_transformer(flat).permute(1, 2, 0).reshape(b, c, h, w)
        return img_emb_l, img_emb_g

    # ------------------------------------------------------------------ losses (ref :105-150)
    @staticmethod
    def _cap_lens(sents):
        if hasattr(sents, "cap_lens"):                # SentenceBatch: same rule, computed without strings
            return list(sents.cap_lens)
        return [len([w for w in sent if not w.startswith("[")]) + 1 for sent in sents]

    def _calc_local_loss(self, img_emb_l, text_emb_l, sents):
        cap_lens = self._cap_lens(sents)
        if self.dist is not None and self.dist.active:
            return self._calc_local_loss_sharded(img_emb_l, text_emb_l, cap_lens)
        return self.local_loss(
            img_emb_l, text_emb_l, cap_lens, temp1=self.temp1, temp2=self.temp2, temp3=self.temp3,
            no_attn_vec=self.no_attn_vec, no_attn_loss_weight=self.no_attn_loss_weight,
            attention_divergence_loss_weight=self.attention_divergence_loss_weight,
            attention_entropy_loss_weight=self.attention_entropy_loss_weight)

    def _calc_local_loss_sharded(self, img_emb_l, text_emb_l, cap_lens):
        want_aux = (self.no_attn_loss_weight is not None or self.attention_divergence_loss_weight is not None
                    or self.attention_entropy_loss_weight is not None)
        d = self.dist
        b_loc = img_emb_l.shape[0]
        words_all = d.all_gather_grad(text_emb_l)                       # [B, D, L], grads reduce-scattered
        lens_all = d.all_gather_ints(cap_lens)
        sim_rows, attn, amean = GL.local_similarity(img_emb_l, words_all, lens_all, self.temp1, self.temp2,
                                                    self.temp3, "sum", self.no_attn_vec, img_offset=d.rank * b_loc,
                                                    want_amean=want_aux)
        sim_full = d.all_gather_nograd(sim_rows)
        l0, l1 = GL.dual_cross_entropy(sim_rows, sim_full, d.rank * b_loc)
        ih, iw = img_emb_l.shape[2], img_emb_l.shape[3]
        maps = GL.split_attention_maps(attn, lens_all, ih, iw, first=d.rank * b_loc, count=b_loc)
        na = kl = ent = 0
        if want_aux:       # this rank's share of the global-batch means (the shares of all ranks add up)
            shift = 0 if self.no_attn_vec is None else 1
            na, kl, ent = GL.attention_regularisers(amean, ih * iw + shift, shift, d.rank * b_loc,
                                                    self.no_attn_loss_weight, self.attention_divergence_loss_weight,
                                                    self.attention_entropy_loss_weight)
        return l0, l1, na, kl, ent, maps

    def _calc_global_loss(self, img_emb_g, text_emb_g):
        if self.dist is not None and self.dist.active:
            d = self.dist
            txt_all = d.all_gather_grad(text_emb_g)
            sim_rows = GL.global_similarity(img_emb_g, txt_all, temp3=self.temp3)
            sim_full = d.all_gather_nograd(sim_rows)
            return GL.dual_cross_entropy(sim_rows, sim_full, d.rank * img_emb_g.shape[0])
        return self.global_loss(img_emb_g, text_emb_g, temp3=self.temp3)

    def calc_loss(self, img_emb_l, img_emb_g, text_emb_l, text_emb_g, sents, segmentation_labels=None):
        # `shared`: terms every data-parallel rank evaluates on the GLOBAL batch (identical on all ranks);
        # `share`: terms of which a rank holds only its images' share (the shares of all ranks add up)
        shared, share = 0, 0
        l_loss0, l_loss1, no_attn_loss, kl_loss, entropy_loss, attn_maps = self._calc_local_loss(
            img_emb_l, text_emb_l, sents)
        if self.local_loss_weight != 0:
            shared = shared + (l_loss0 + l_loss1) * self.local_loss_weight
        if self.global_loss_weight != 0:
            g_loss0, g_loss1 = self._calc_global_loss(img_emb_g, text_emb_g)
            shared = shared + (g_loss0 + g_loss1) * self.global_loss_weight
        if segmentation_labels is not None and self.segmentation_loss_weight:
            # attention-supervision term (ref :143-147)
            if getattr(attn_maps, "flat", None) is not None and attn_maps.flat.is_cuda:
                seg = GL.attention_supervision_loss(attn_maps, segmentation_labels)          # K4
            else:
                mean_maps = torch.cat([m.mean(1) for m in attn_maps], 0)
                up = nn.functional.interpolate(mean_maps.unsqueeze(1), size=segmentation_labels.shape[1:]).squeeze(1)
                up = up / up.sum(-1, keepdim=True).sum(-2, keepdim=True)
                seg = -torch.log((segmentation_labels * up).sum(-1).sum(-1)).mean()
            if self.dist is not None and self.dist.active:
                seg = seg / self.dist.world_size          # mean over the global batch
            share = share + seg * self.segmentation_loss_weight
        share = share + no_attn_loss + kl_loss + entropy_loss
        loss_ = shared + share
        self._loss_parts = (shared.detach() if torch.is_tensor(shared) else shared,
                            share.detach() if torch.is_tensor(share) else share)
        return loss_, attn_maps

    def global_batch_loss(self):
        """The value of the last calc_loss over the GLOBAL batch, identical on every data-parallel rank:
        the global-batch terms once plus the SUM over ranks of the per-rank shares (segmentation and regulariser
        terms).  Single process: the loss itself.  One small collective, used for logging, validation and the
        plateau scheduler - the value a rank differentiates stays its own (gradients are SUM-reduced)."""
        shared, share = self._loss_parts
        if self.dist is not None and self.dist.active:
            share = self.dist.all_reduce_scalar(share, like=shared)
        return shared + share

    def forward(self, x):
        if ENCODER_STREAMS and x["imgs"].is_cuda and torch.is_grad_enabled():
            # The two encoders are independent: the text encoder runs on a side HIP stream (autograd runs its backward
            # there too), so its many short kernels fill the gaps between the image encoder's (88.6 -> 81 ms per step).
            # Data parallel too: the reducer orders its bucket launches behind every stream that produced one of the
            # bucket's gradients (gloria.dist.GradReducer._join_streams).
            cur = torch.cuda.current_stream()
            side = _side_stream(x["imgs"].device)
            side.wait_stream(cur)
            for k in ("caption_ids", "attention_mask", "token_type_ids"):
                if torch.is_tensor(x[k]):
                    x[k].record_stream(side)          # allocated on the main stream, read on the side stream
            with torch.cuda.stream(side):
                text_emb_l, text_emb_g, sents = self.text_encoder_forward(
                    x["caption_ids"], x["attention_mask"], x["token_type_ids"])
            img_emb_l, img_emb_g = self.image_encoder_forward(x["imgs"])
            cur.wait_stream(side)
            for t in (text_emb_l, text_emb_g):
                if torch.is_tensor(t):
                    t.record_stream(cur)
            return img_emb_l, img_emb_g, text_emb_l, text_emb_g, sents
        img_emb_l, img_emb_g = self.image_encoder_forward(x["imgs"])
        text_emb_l, text_emb_g, sents = self.text_encoder_forward(
            x["caption_ids"], x["attention_mask"], x["token_type_ids"])
        return img_emb_l, img_emb_g, text_emb_l, text_emb_g, sents

    # ------------------------------------------------------------------ inference helpers (ref :164-211)
    def get_global_similarities(self, img_emb_g, text_emb_g):
        with torch.no_grad():
            return GL.global_similarity(img_emb_g, text_emb_g, temp3=1.0).detach().cpu()

    def get_local_similarities(self, img_emb_l, text_emb_l, cap_lens):
        """words 1..n (skips [CLS]), temps 4/5, MAX over words, no temp3 (ref :171-207)."""
        with torch.no_grad():
            sim, _, _ = GL.local_similarity(img_emb_l, text_emb_l, [int(c) for c in cap_lens], 4.0, 5.0, 1.0,
                                            "max", self.no_attn_vec, want_attn=False, word_start=1)
        return sim.detach().cpu()

    def get_attn_maps(self, img_emb_l, text_emb_l, sents):
        _, _, _, _, _, attn_maps = self._calc_local_loss(img_emb_l, text_emb_l, sents)
        return attn_maps
