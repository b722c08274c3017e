"""ORACLE (test infrastructure, not product code).

Stock ``torch.nn`` fp32 3D-ResNet: the temporal inflation of the reference's 2D
ResNet topology.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this file.  The product path
(``video-classification_amd/``) never does.

Parity status: the reference contains no 3D op, no tests and no golden vectors, so the
3D arithmetic is "parity unpinned" by the reference.  It is pinned here to stock
``torch.nn`` CPU fp32 (torch 2.10.0).  The one piece the reference itself pins -- the
block topology -- is checked by the T=1 centre-tap test (tests/test_oracle_cpu.py)
against outputs of the reference's own ``lib/models/backbone.py`` run in the build
container (fixtures under tests/golden/, generator oracle/make_golden.py).

Topology follows, line by line:
  /root/reference/lib/models/backbone.py:17-20   conv3x3 (3x3(x3), pad 1, no bias)
  /root/reference/lib/models/backbone.py:33-62   BasicBlock order
  /root/reference/lib/models/backbone.py:65-103  Bottleneck order (stride on the 3x3)
  /root/reference/lib/models/backbone.py:106-125 stem / maxpool / 4 stages / final pool
  /root/reference/lib/models/backbone.py:127-142 downsample rule
  /root/reference/lib/models/backbone.py:188-192 depth table
  /root/reference/lib/models/backbone.py:13      BN momentum 0.1
  /root/reference/lib/models/model.py:24         classifier head Linear(F, CLFDIM)
  /root/reference/lib/core/loss.py:25-30         loss

Inflation rule: every 2D op becomes its 3D twin with temporal extent equal to the
spatial one, except that the stem's temporal stride is 1; AvgPool2d(7) becomes a global
average pool.
"""
import torch
import torch.nn as nn

BN_MOMENTUM = 0.1


def conv3x3x3(in_planes, out_planes, stride=1):
    return nn.Conv3d(in_planes, out_planes, kernel_size=3, stride=stride,
                     padding=1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm3d(planes, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3x3(planes, planes)
        self.bn2 = nn.BatchNorm3d(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        residual = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        out = out + residual
        return self.relu(out)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv3d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm3d(planes, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv3d(planes, planes, kernel_size=3, stride=stride,
                               padding=1, bias=False)
        self.bn2 = nn.BatchNorm3d(planes, momentum=BN_MOMENTUM)
        self.conv3 = nn.Conv3d(planes, planes * self.expansion, kernel_size=1,
                               bias=False)
        self.bn3 = nn.BatchNorm3d(planes * self.expansion, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        residual = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        out = out + residual
        return self.relu(out)


resnet_spec = {18: (BasicBlock, [2, 2, 2, 2]),
               34: (BasicBlock, [3, 4, 6, 3]),
               50: (Bottleneck, [3, 4, 6, 3]),
               101: (Bottleneck, [3, 4, 23, 3]),
               152: (Bottleneck, [3, 8, 36, 3])}


class BackBone3D(nn.Module):
    def __init__(self, block, layers):
        self.inplanes = 64
        super().__init__()
        self.conv1 = nn.Conv3d(3, 64, kernel_size=7, stride=(1, 2, 2), padding=3,
                               bias=False)
        self.bn1 = nn.BatchNorm3d(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool3d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.final_layer = nn.Sequential(nn.AdaptiveAvgPool3d(1), nn.Flatten())
        self.out_features = 512 * block.expansion

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv3d(self.inplanes, planes * block.expansion, kernel_size=1,
                          stride=stride, bias=False),
                nn.BatchNorm3d(planes * block.expansion, momentum=BN_MOMENTUM),
            )
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.final_layer(x)


class R3DOracle(nn.Module):
    """backbone + classifier head; ``forward`` returns logits (N x classes)."""

    def __init__(self, resnet_type=18, num_classes=101):
        super().__init__()
        block, layers = resnet_spec[resnet_type]
        self.backbone = BackBone3D(block, layers)
        self.clf_head = nn.Linear(self.backbone.out_features, num_classes)

    def forward(self, x):
        return self.clf_head(self.backbone(x))


def init_weights_(model, seed=0):
    """Kaiming-normal fan-out convs, BN gamma=1 beta=0, fc default, seeded."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.Conv3d):
            fan_out = m.out_channels * m.kernel_size[0] * m.kernel_size[1] * m.kernel_size[2]
            std = (2.0 / fan_out) ** 0.5
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
        elif isinstance(m, nn.BatchNorm3d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Linear):
            bound = 1.0 / (m.in_features ** 0.5)
            with torch.no_grad():
                m.weight.copy_((torch.rand(m.weight.shape, generator=g) * 2 - 1) * bound)
                m.bias.copy_((torch.rand(m.bias.shape, generator=g) * 2 - 1) * bound)
    return model


def loss_oracle(score, y, log_prob, advantages, r_act=1.0, r_crt=1.0):
    """/root/reference/lib/core/loss.py:25-30."""
    clf = nn.functional.cross_entropy(score, y)
    crt = advantages.pow(2).mean()
    act = -(advantages.detach() * log_prob).mean()
    return clf + r_act * act + r_crt * crt


def synthetic_batch(n, t=16, h=112, w=112, classes=101, seed=1234):
    """clips ~ N(0,1), labels uniform in [0, classes)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, t, h, w, generator=g)
    y = torch.randint(0, classes, (n,), generator=g)
    return x, y
