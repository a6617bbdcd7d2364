"""Factory API with the reference's names and call conventions (/root/reference/gloria/builder.py:11-137).
`cfg` is an attribute-style config that returns None for missing keys (gloria.config.Config).
build_loss / build_transformation (segmentation / PIL host transforms) are out of scope."""

import torch

from . import datasets
from . import lightning
from . import models


def build_data_module(cfg):
    data_module = datasets.DATA_MODULES[cfg.data.dataset.lower()]
    return data_module(cfg)


def build_lightning_model(cfg, dm, ckpt=None):
    module = lightning.LIGHTNING_MODULES[cfg.phase.lower()]
    if ckpt is not None:
        module = module.load_from_checkpoint(ckpt, cfg=cfg)
    else:
        module = module(cfg)
    module.dm = dm
    return module


def build_gloria_model(cfg):
    return models.gloria_model.GLoRIA(cfg)


def clean_state_dict(state_dict):
    """Keys a REFERENCE checkpoint carries that this build does not register as state:
    `*.embeddings.position_ids` - transformers==4.2.1 (requirements.txt:83) keeps BertEmbeddings.position_ids as a
    persistent buffer, so every reference checkpoint holds `gloria.text_encoder.model.embeddings.position_ids`;
    it is the constant arange(512) and is rebuilt here as a non-persistent buffer."""
    return {k: v for k, v in state_dict.items() if not k.endswith("embeddings.position_ids")}


def build_gloria_from_ckpt(ckpt, cfg=None):
    """Reference checkpoint layout: {"state_dict": {"gloria.<...>": tensor}, "hyper_parameters": cfg}
    (/root/reference/gloria/builder.py:35-50).  The file is read with the weights-only loader (nothing in it is
    executed); `hyper_parameters` written by the reference are OmegaConf objects, which that loader refuses - pass
    `cfg` (the YAML the run was started from) for such files."""
    from .config import Config
    try:
        ckpt = torch.load(ckpt, map_location="cpu", weights_only=True)
    except Exception as e:              # noqa: BLE001 - the unpickler's own error types vary between torch versions
        raise RuntimeError(
            f"{ckpt}: the weights-only loader refused this checkpoint ({type(e).__name__}: {e}). Reference checkpoints "
            "pickle their OmegaConf hyper_parameters; re-save the file with only 'state_dict' (and plain-dict "
            "hyper_parameters) in the environment that wrote it - it is never unpickled here") from e
    if cfg is None:
        cfg = Config(ckpt["hyper_parameters"])
    fixed = {k.split("gloria.")[-1]: v for k, v in clean_state_dict(ckpt["state_dict"]).items()}
    gloria_model = build_gloria_model(cfg)
    gloria_model.load_state_dict(fixed)
    return gloria_model


def build_img_model(cfg):
    image_model = models.IMAGE_MODELS[cfg.phase.lower()]
    return image_model(cfg)


def build_text_model(cfg):
    return models.text_model.BertEncoder(cfg)


def build_optimizer(cfg, lr, model):
    # get params for optimization (ref :65-82)
    if cfg.model.train_last_local_image_layer or cfg.model.train_prompt:
        for p in model.parameters():
            p.requires_grad = False
        params = []
        if cfg.model.train_last_local_image_layer:
            params += model.img_encoder.model.layer3.parameters()
        if cfg.model.train_prompt:
            params += model.text_encoder.model.embeddings.parameters()
        for p in params:
            p.requires_grad = True
    params = [p for p in model.parameters() if p.requires_grad]

    if cfg.train.optimizer.name == "Adam" and cfg.train.optimizer.flat_bf16 and params and all(p.is_cuda for p in params):
        # MI355X training path (set by gloria.trainer.Trainer for bf16 runs): the same Adam on flat fp32 masters with
        # bf16 shadow weights, gradient clipping folded in (gloria/optim.py).  Parameters the graph never reaches get
        # no gradient and are skipped by torch.optim.Adam; the flat optimizer leaves them out altogether: the BERT
        # pooler when last_n_layers > 1 (BertEncoder.forward then ignores outputs[1], text_model.py:96-114).
        from .optim import ShadowAdam, shadow_parameter_ids
        skip = set()
        if (cfg.model.text.last_n_layers or 1) > 1 and hasattr(model, "text_encoder"):
            skip = {id(p) for p in model.text_encoder.model.pooler.parameters()}
        all_params = params
        params = [p for p in params if id(p) not in skip]
        return ShadowAdam(params, all_params=all_params, lr=lr, betas=(0.5, 0.999), weight_decay=float(cfg.train.optimizer.weight_decay),
                          max_grad_norm=cfg.train.optimizer.flat_clip, shadow_ids=shadow_parameter_ids(model),
                          flat_grads=bool(cfg.train.optimizer.flat_grads))

    if cfg.train.optimizer.name == "SGD":
        return torch.optim.SGD(params, lr=lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
    elif cfg.train.optimizer.name == "Adam":
        fused = bool(params) and all(p.is_cuda for p in params)      # one multi-tensor launch chain on the GPU
        return torch.optim.Adam(params, lr=lr, weight_decay=float(cfg.train.optimizer.weight_decay),
                                betas=(0.5, 0.999), fused=fused)
    elif cfg.train.optimizer.name == "AdamW":
        return torch.optim.AdamW(params, lr=lr, weight_decay=float(cfg.train.optimizer.weight_decay))


def build_scheduler(cfg, optimizer, dm=None):
    name = cfg.train.scheduler.name
    if name == "warmup":
        def lambda_lr(epoch):
            if epoch <= 3:
                return 0.001 + epoch * 0.003
            if epoch >= 22:
                return 0.01 * (1 - epoch / 200.0) ** 0.9
            return 0.01
        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda_lr)
    elif name == "cos":
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=10)
    elif name == "plateau":
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, factor=0.5, patience=5)
    elif name == "step":
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=1, gamma=0.8)
    else:
        scheduler = None

    if cfg.lightning.trainer.val_check_interval is not None:
        cfg.train.scheduler.interval = "step"
        num_iter = len(dm.train_dataloader().dataset)
        if type(cfg.lightning.trainer.val_check_interval) == float:
            cfg.train.scheduler.frequency = int(num_iter * cfg.lightning.trainer.val_check_interval)
        else:
            cfg.train.scheduler.frequency = cfg.lightning.trainer.val_check_interval

    return {"scheduler": scheduler, "monitor": cfg.train.scheduler.monitor,
            "interval": cfg.train.scheduler.interval, "frequency": cfg.train.scheduler.frequency}
