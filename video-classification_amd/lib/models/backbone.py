"""3D-ResNet backbone behind the reference's interface (/root/reference/lib/models/backbone.py).

Same class names, constructor arguments, attribute names (= checkpoint keys) and depth
table as the reference; every 2D op is inflated to 3D: temporal extent =
spatial extent, stem temporal stride 1, AvgPool2d(7) -> global average pool.

The modules below are parameter containers: stock nn.Conv3d / nn.BatchNorm3d objects give
the usual initialisation and state_dict keys, but their own forward is never called.
``BackBone.forward`` hands the clip batch to vc3d.R3DEngine, which runs the hand-written
gfx950 kernels (implicit-GEMM Conv3d on bf16 MFMA, fused BN statistics, channels-last bf16
activations).  There is no CPU path: a CPU tensor raises.
"""
import logging
import os
from collections import OrderedDict

import torch
import torch.nn as nn

import vc3d

BN_MOMENTUM = 0.1  # reference backbone.py:13
logger = logging.getLogger(__name__)


def _conv(cin, cout, k, stride=1, pad=0):
    return nn.Conv3d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False)


def conv3x3(in_planes, out_planes, stride=1):
    """3x3x3 convolution with padding (reference backbone.py:17-20, inflated)."""
    return _conv(in_planes, out_planes, 3, stride, 1)


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("blocks are parameter containers; call the BackBone (runs on vc3d kernels)")


class BasicBlock(_Holder):
    """conv-bn-relu-conv-bn (+residual) relu; reference backbone.py:33-62."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm3d(planes, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm3d(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(_Holder):
    """1x1 -> 3x3 (carries the stride) -> 1x1 x4; reference backbone.py:65-103."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = nn.BatchNorm3d(planes, momentum=BN_MOMENTUM)
        self.conv2 = _conv(planes, planes, 3, stride, 1)
        self.bn2 = nn.BatchNorm3d(planes, momentum=BN_MOMENTUM)
        self.conv3 = _conv(planes, planes * self.expansion, 1)
        self.bn3 = nn.BatchNorm3d(planes * self.expansion, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class BackBone(nn.Module):
    """reference backbone.py:106-157.  forward(x: N x 3 x T x H x W) -> N x (512*expansion)."""

    def __init__(self, block, layers):
        self.inplanes = 64
        super().__init__()
        self.conv1 = _conv(3, 64, 7, (1, 2, 2), 3)
        self.bn1 = nn.BatchNorm3d(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool3d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.final_layer = nn.Sequential(nn.AdaptiveAvgPool3d(1), nn.Flatten())
        self.out_features = 512 * block.expansion
        self._engine = None
        self._to_channels_last()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:  # reference backbone.py:129
            downsample = nn.Sequential(
                _conv(self.inplanes, planes * block.expansion, 1, stride),
                nn.BatchNorm3d(planes * block.expansion, momentum=BN_MOMENTUM),
            )
        seq = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def _to_channels_last(self):
        """conv weights live as [Cout][kt][kh][kw][Cin] in memory (torch channels_last_3d): the
        K-contiguous layout the implicit-GEMM kernels read and the wgrad kernel writes."""
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Conv3d):
                    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last_3d)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._to_channels_last()
        if self._engine is not None:
            self._engine.flat_grad = None  # device or dtype may have changed
        return out

    @property
    def engine(self):
        if self._engine is None:
            self._engine = vc3d.R3DEngine(self)
        return self._engine

    def forward(self, x):
        return vc3d.backbone_apply(self.engine, x, self.training)

    def init_weights(self, pretrained=''):
        """reference backbone.py:159-185: load a state_dict or {'state_dict': ...}, strip the
        DataParallel 'module.' prefix, strict=False; ValueError when the file is missing.
        2D (ImageNet) conv kernels are inflated along T (repeat / kT), the usual 3D bootstrap."""
        if not os.path.isfile(pretrained):
            logger.error('=> pretrained model %s does not exist', pretrained)
            raise ValueError('imagenet pretrained model does not exist')
        logger.info('=> loading pretrained model %s', pretrained)
        ckpt = torch.load(pretrained, map_location='cpu', weights_only=True)
        if isinstance(ckpt, dict) and 'state_dict' in ckpt:
            ckpt = ckpt['state_dict']
        if not isinstance(ckpt, dict):
            raise RuntimeError('No state_dict found in checkpoint file {}'.format(pretrained))
        own = self.state_dict()
        state = OrderedDict()
        for key, val in ckpt.items():
            key = key[7:] if key.startswith('module.') else key
            if key in own and val.dim() == 4 and own[key].dim() == 5:
                kt = own[key].shape[2]
                val = val.unsqueeze(2).repeat(1, 1, kt, 1, 1) / kt
            state[key] = val
        self.load_state_dict(state, strict=False)
        self._to_channels_last()


resnet_spec = {18: (BasicBlock, [2, 2, 2, 2]),
               34: (BasicBlock, [3, 4, 6, 3]),
               50: (Bottleneck, [3, 4, 6, 3]),
               101: (Bottleneck, [3, 4, 23, 3]),
               152: (Bottleneck, [3, 8, 36, 3])}


def getBackbone(cfg, is_train):
    """reference backbone.py:195-213."""
    block_class, layers = resnet_spec[cfg.MODEL.RESNET_TYPE]
    model = BackBone(block_class, layers)
    if is_train and cfg.MODEL.INIT_WEIGHTS:
        model.init_weights(cfg.MODEL.PRETRAINED_PATH)
    return model
