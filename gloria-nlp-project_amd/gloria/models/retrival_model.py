"""Image -> text retrieval with the reference's interface (/root/reference/gloria/models/retrival_model.py:8-166)
on the MI355X kernels: local similarities by K1 (words 1..n of the [CLS]-stripped embeddings, sum / mean
aggregate - ref :127-166), global similarities by K3 (cosine, ref :101-104), ranking by the exact top-k kernel
(`np.argsort(similarities)[::-1][:top_k]`, ref :118).

What is NOT rebuilt: `process_text` / `process_img` (tokenizer, cv2 - host preprocessing):
targets and sources are given as already-encoded embeddings or as batch dicts for the encoders."""

import numpy as np
import torch

from .. import select
from ..loss import gloria_loss as GL


class Retriver:
    def __init__(self, gloria, targets=None, target_classes=None, device=None, top_k=5):
        self.device = torch.device(device if device is not None else "cuda:0")
        self.gloria = gloria.to(self.device)
        self.top_k = top_k
        self.targets = self._process_targets(targets) if targets is not None else None
        self.targets_classes = None if target_classes is None else np.array(target_classes)

    def _process_targets(self, target):
        """target: batch dict with caption_ids / attention_mask / token_type_ids (ref :26-61)."""
        with torch.no_grad():
            text_emb_l, text_emb_g, sents = self.gloria.text_encoder_forward(
                target["caption_ids"].to(self.device), target["attention_mask"].to(self.device),
                target["token_type_ids"].to(self.device))
        # ref :52-54 - no "+1" here, unlike GLoRIA._calc_local_loss
        self.cap_lens = [c - 1 for c in self.gloria._cap_lens(sents)]
        return {"global_embeddings": text_emb_g.detach(), "local_embeddings": text_emb_l[:, :, 1:].detach(),   # [CLS] removed :57
                "processed_input": target}

    def _process_source(self, imgs):
        with torch.no_grad():
            img_emb_l, img_emb_g = self.gloria.image_encoder_forward(imgs.to(self.device))
        return {"global_embeddings": img_emb_g.detach(), "local_embeddings": img_emb_l.detach(), "processed_input": imgs}

    def _compute_local_similarity(self, img_features, words_emb, cap_lens, temp1=4.0, temp2=5.0, temp3=10.0, agg="sum"):
        """ref :127-166: one image against every target; words [1, n+1) of the given embeddings."""
        with torch.no_grad():
            sim, _, _ = GL.local_similarity(img_features[:1], words_emb, [int(c) for c in cap_lens], temp1, temp2, temp3,
                                            agg, None, want_attn=False, word_start=1)
        return sim[0]

    def similarities(self, similarity_type="both"):
        if similarity_type not in ["both", "local", "global"]:
            raise Exception("similarity_type must be one of ['both', 'local', 'global']")
        g = self.gloria
        local = self._compute_local_similarity(self.source["local_embeddings"], self.targets["local_embeddings"],
                                               self.cap_lens, g.temp1, g.temp2, g.temp3)
        with torch.no_grad():
            glob = GL.global_similarity(self.source["global_embeddings"][:1], self.targets["global_embeddings"],
                                        temp3=1.0)[0]
        if similarity_type == "local":
            return local
        if similarity_type == "global":
            return glob
        norm = lambda x: (x - x.mean()) / x.std(unbiased=False)        # numpy's population std (ref :110)
        return (norm(local) + norm(glob)) / 2

    def retrieve(self, source, similarity_type="both"):
        """Returns (indices of the top_k targets in ranking order, their classes or None) - ref :118-125."""
        self.source = self._process_source(source) if torch.is_tensor(source) else source
        sims = self.similarities(similarity_type)
        idx, _ = select.topk_desc(sims, min(self.top_k, sims.numel()))
        idx = idx.cpu().numpy()
        cls = self.targets_classes[idx] if self.targets_classes is not None else None
        return idx, cls
