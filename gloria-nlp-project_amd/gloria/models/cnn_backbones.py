"""ResNet-50 backbone factory with the reference's contract
(/root/reference/gloria/models/cnn_backbones.py:31-35): returns (model, feature_dim, interm_dim) with
the classifier replaced by identity.  torchvision is not available here, so the bottleneck stack is
defined directly on torch ops (MIOpen convolutions on ROCm); parameter names match torchvision's
resnet50 so reference checkpoints (`gloria.img_encoder.model.layer3...`) load unchanged.
"""

import os
import warnings

import torch
import torch.nn as nn

from .fused_bn import SkipPair, fused_bn_act, fused_maxpool


class Identity(nn.Module):
    def forward(self, x):
        return x


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x, fork=False):
        # BatchNorm + ReLU (+ the skip connection) run as one fused HIP pass pair per direction on channels-last bf16
        # (gloria/models/fused_bn.py); same modules / parameters / buffers as torchvision's bottleneck.
        # x may be a SkipPair from the previous block of the stage; fork=True asks for one for the next block.
        identity = x
        if isinstance(x, SkipPair):
            x, identity = x.main, x.skip
        out = fused_bn_act(self.bn1, self.conv1(x))
        out = fused_bn_act(self.bn2, self.conv2(out))
        out = self.conv3(out)
        if self.downsample is not None:
            identity = fused_bn_act(self.downsample[1], self.downsample[0](identity), relu=False)
        return fused_bn_act(self.bn3, out, residual=identity, fork=fork)


class _Stage(nn.Sequential):
    """nn.Sequential of bottlenecks (same child names); inside the stage a block hands its output to the next as a
    SkipPair, the stage itself returns a tensor."""

    def forward(self, x, fork_out=False):
        """fork_out: also the LAST block returns a SkipPair (the caller feeds it to the next stage, whose first block
        reads `main` with its convolution and `skip` with its downsample branch)."""
        blocks = list(self)
        for i, blk in enumerate(blocks):
            x = blk(x, fork=fork_out or i + 1 < len(blocks))
        return x


class ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 3, 1)
        self.layer2 = self._make_layer(128, 4, 2)
        self.layer3 = self._make_layer(256, 6, 2)
        self.layer4 = self._make_layer(512, 3, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return _Stage(*layers)

    def forward(self, x):
        x = fused_maxpool(self.maxpool, fused_bn_act(self.bn1, self.conv1(x)))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x, fork_out=True), fork_out=True)))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet_50(pretrained=True):
    """(model, 2048, 1024).  `pretrained` may be a path to a torchvision resnet50 state_dict; with
    True and no network access the weights are the Kaiming initialisation (a warning says so)."""
    model = ResNet50()
    if isinstance(pretrained, str) and os.path.exists(pretrained):
        model.load_state_dict(torch.load(pretrained, map_location="cpu", weights_only=True))
    elif pretrained:
        warnings.warn("resnet_50(pretrained=True): no ImageNet weights available offline, using random init")
    feature_dims = model.fc.in_features
    model.fc = Identity()
    return model, feature_dims, 1024
