"""ImageEncoder with the reference's interface (/root/reference/gloria/models/vision_model.py:8-86):
bilinear upsample to 299x299 (align_corners=True, :70), ResNet-50 stem..layer3 -> local features
[B, 1024, 19, 19] (:72-80), layer4 + avgpool -> global [B, 2048] (:81-84), then
Linear(2048, 768) / bias-free Conv1x1(1024, 768) embedders (:20-28, :52-65).
Parameter names equal the reference's (`model.*`, `global_embedder.*`, `local_embedder.weight`).
"""

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _native as N
from . import cnn_backbones
from .fused_bn import fused_bn_act, fused_maxpool


_FUSED_RESIZE = os.environ.get("GLR_FUSED_RESIZE", "1") != "0"


def _resize_299(x):
    """F.interpolate(x, (299, 299), bilinear, align_corners=True) (vision_model.py:68).  Under bf16 autocast on the GPU
    the resize, the channels-last copy and the cast in front of conv1 are one HIP pass (glr_upsample_bilinear_cl)."""
    if (_FUSED_RESIZE and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        B, C, H, W = x.shape
        y = torch.empty((B, C, 299, 299), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        sn, sc, sh, sw = x.stride()
        N.check(N.lib().glr_upsample_bilinear_cl(N.ptr(x), sn, sc, sh, sw, B, C, H, W, 299, 299, N.ptr(y), N.stream()),
                "glr_upsample_bilinear_cl")
        return y
    return F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=True)


class ImageEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.output_dim = cfg.model.text.embedding_dim
        self.norm = cfg.model.norm
        model_function = getattr(cnn_backbones, cfg.model.vision.model_name)
        self.model, self.feature_dim, self.interm_feature_dim = model_function(
            pretrained=cfg.model.vision.pretrained)
        self.global_embedder = nn.Linear(self.feature_dim, self.output_dim)
        self.local_embedder = nn.Conv2d(self.interm_feature_dim, self.output_dim, kernel_size=1, stride=1,
                                        padding=0, bias=False)
        self.pool = nn.AdaptiveAvgPool2d(output_size=(1, 1))
        if cfg.model.vision.freeze_cnn:
            print("Freezing CNN model")
            for param in self.model.parameters():
                param.requires_grad = False

    def forward(self, x, get_local=False):
        # --> fixed-size input: batch x 3 x 299 x 299
        global_ft, local_ft = self.resnet_forward(x, extract_features=True)
        return (global_ft, local_ft) if get_local else global_ft

    def generate_embeddings(self, global_features, local_features):
        global_emb = self.global_embedder(global_features)
        local_emb = self.local_embedder(local_features)
        if self.norm is True:
            local_emb = local_emb / torch.norm(local_emb, 2, dim=1, keepdim=True).expand_as(local_emb)
            global_emb = global_emb / torch.norm(global_emb, 2, dim=1, keepdim=True).expand_as(global_emb)
        return global_emb, local_emb

    def resnet_forward(self, x, extract_features=False):
        x = _resize_299(x)
        m = self.model
        x = fused_maxpool(m.maxpool, fused_bn_act(m.bn1, m.conv1(x)))         # (B, 64, 75, 75)
        x = m.layer1(x, fork_out=True)                   # (B, 256, 75, 75); stage outputs travel as (main, skip) pairs
        x = m.layer2(x, fork_out=True)                   # (B, 512, 38, 38)
        x = m.layer3(x)                                  # (B, 1024, 19, 19)
        local_features = x
        x = m.layer4(x)                                  # (B, 2048, 10, 10)
        x = self.pool(x)
        return x.view(x.size(0), -1), local_features
