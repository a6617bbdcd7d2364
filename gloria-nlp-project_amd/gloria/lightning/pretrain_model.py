"""PretrainModel with the reference LightningModule's surface
(/root/reference/gloria/lightning/pretrain_model.py:12-86): `.gloria`, `.lr`, `.dm`,
configure_optimizers(), training/validation/test_step(batch, batch_idx) -> dict, shared_step,
load_from_checkpoint(path, cfg=cfg), DummyObjectWrapper.  pytorch_lightning is not available here:
this is a plain nn.Module driven by gloria.trainer.Trainer (or by a real Lightning Trainer when
installed - only the methods Lightning calls are provided)."""

import torch
import torch.nn as nn

from .. import builder


class DummyObjectWrapper:
    def __init__(self, obj):
        self.obj = obj


class PretrainModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.hparams = cfg
        self.gloria = builder.build_gloria_model(cfg)
        self.lr = cfg.lightning.trainer.lr
        self.dm = None
        self.current_epoch = 0
        self.logged = {}            # last value of every self.log(...) key
        self._log_sink = None       # optional callable(name, value) installed by the trainer

    def configure_optimizers(self):
        optimizer = builder.build_optimizer(self.cfg, self.lr, self.gloria)
        scheduler = builder.build_scheduler(self.cfg, optimizer, self.dm)
        return {"optimizer": optimizer, "lr_scheduler": scheduler}

    def log(self, name, value, **kwargs):
        self.logged[name] = value.detach() if torch.is_tensor(value) else value
        if self._log_sink is not None:
            self._log_sink(name, self.logged[name])

    def _step(self, batch, split):
        loss, attn_maps, img_emb_l, img_emb_g, text_emb_l, text_emb_g, sents = self.shared_step(batch, split)
        return dict(loss=loss, attn_maps=attn_maps, img_emb_l=img_emb_l, img_emb_g=img_emb_g,
                    text_emb_l=text_emb_l, text_emb_g=text_emb_g, sents=sents)

    def training_step(self, batch, batch_idx):
        # (the reference also renders attention PNGs every cfg.train.update_interval batches,
        #  pretrain_model.py:31-36: host-side plotting, out of scope)
        return self._step(batch, "train")

    def validation_step(self, batch, batch_idx):
        return self._step(batch, "val")

    def test_step(self, batch, batch_idx):
        return self._step(batch, "test")

    def shared_step(self, batch, split):
        """Similar to traning step"""
        img_emb_l, img_emb_g, text_emb_l, text_emb_g, sents = self.gloria(batch)
        seg_labels = batch["segmentation_labels"] if "segmentation_labels" in batch.keys() else None
        loss, attn_maps = self.gloria.calc_loss(img_emb_l, img_emb_g, text_emb_l, text_emb_g, sents, seg_labels)
        self.log(f"{split}_loss", loss, on_epoch=True, on_step=(split == "train"), logger=True, prog_bar=True)
        return (loss, DummyObjectWrapper(attn_maps), DummyObjectWrapper(img_emb_l), DummyObjectWrapper(img_emb_g),
                DummyObjectWrapper(text_emb_l), DummyObjectWrapper(text_emb_g), sents)

    # ---- checkpoint format of the reference: {"state_dict": {"gloria.*": ...}, "hyper_parameters": cfg}
    def checkpoint(self):
        return {"state_dict": {k: v for k, v in self.state_dict().items()},
                "hyper_parameters": self.cfg.to_dict() if hasattr(self.cfg, "to_dict") else dict(self.cfg)}

    @classmethod
    def load_from_checkpoint(cls, path, cfg=None):
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        if cfg is None:
            from ..config import Config
            cfg = Config(ckpt["hyper_parameters"])
        module = cls(cfg)
        module.load_state_dict(builder.clean_state_dict(ckpt["state_dict"]))
        return module
