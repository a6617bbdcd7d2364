"""Attribute-style config with the OmegaConf-2.0 behaviour the reference relies on: reading a
missing key gives None (e.g. cfg.model.norm, cfg.model.ckpt_path, cfg.model.gloria.no_attn_loss_weight
are absent from the pretrain YAMLs).  omegaconf is not available here; PyYAML is."""

import copy

import yaml


class Config(dict):
    def __init__(self, data=None):
        super().__init__()
        for k, v in (data or {}).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, Config):
            return Config(v)
        if isinstance(v, list):
            return [Config._wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Config._wrap(v))

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return self.get(k, None)

    def __setattr__(self, k, v):
        self[k] = v

    def __getitem__(self, k):
        return self.get(k, None)

    def __deepcopy__(self, memo):
        return Config(copy.deepcopy(dict(self), memo))

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Config) else v) for k, v in self.items()}

    def merge(self, other):
        for k, v in (other or {}).items():
            if isinstance(v, dict) and isinstance(self.get(k), Config):
                self[k].merge(v)
            else:
                self[k] = v
        return self

    def set_path(self, dotted, value):
        """cfg.set_path('model.gloria.temp1', 4.0)"""
        node = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if not isinstance(node.get(p), Config):
                node[p] = Config()
            node = node[p]
        node[parts[-1]] = value
        return self


def load_config(path, overrides=None):
    with open(path) as f:
        cfg = Config(yaml.safe_load(f))
    for k, v in (overrides or {}).items():
        cfg.set_path(k, v)
    return cfg


def pretrain_config(name="imagenome", batch_size=48, **overrides):
    """The fields of configs/{chexpert,imagenome}_pretrain_config.yaml /
    imagenome_attn_finetune_config.yaml that the pretraining path reads (reference values)."""
    cfg = Config({
        "experiment_name": "gloria_pretrain",
        "phase": "pretrain",
        "random_seed": 0,
        "lightning": {"trainer": {"gpus": "0", "max_epochs": 50, "distributed_backend": "dp",
                                  "gradient_clip_val": 0.25, "lr": 0.00005, "precision": 16}},
        "model": {
            "gloria": {"local_loss_weight": 1.0, "global_loss_weight": 1.0, "temp1": 4.0, "temp2": 5.0,
                       "temp3": 10.0, "no_attn_vec": False},
            "vision": {"model_name": "resnet_50", "freeze_cnn": False, "pretrained": True},
            "text": {"bert_type": "emilyalsentzer/Bio_ClinicalBERT", "last_n_layers": 4,
                     "aggregate_method": "sum", "norm": False, "embedding_dim": 768, "freeze_bert": False,
                     "agg_tokens": True},
        },
        "data": {"dataset": "chexpert" if name == "chexpert" else "imagenome", "text": {"word_num": 97, "captions_per_image": 5,
                                            "full_report": name != "chexpert"},
                 "image": {"imsize": 256}},
        "transforms": {"norm": "half", "random_crop": {"crop_size": 224}},
        "train": {"update_interval": None, "batch_size": batch_size, "num_workers": 8,
                  "optimizer": {"name": "Adam", "weight_decay": 1e-6},
                  "scheduler": {"name": "plateau", "monitor": "val_loss", "inerval": "epoch", "frequency": 1}},
    })
    if name == "imagenome_attn_finetune":
        cfg.model.gloria.merge({"local_loss_weight": 0, "global_loss_weight": 0, "segmentation_loss_weight": 1.0})
    for k, v in overrides.items():
        cfg.set_path(k, v)
    return cfg
