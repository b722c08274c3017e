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


# --------------------------------------------------------------------------------------
# bf16-storage emulation of the SAME network (still the oracle: plain torch on the CPU).
#
# The product keeps activations and activation-gradients in bf16 and accumulates in fp32.
# Against the pure-fp32 oracle that shows up as ~1.5 % noise on late activations, and --
# because ReLU's gradient mask is discontinuous -- as O(10 %) noise on gradients of a
# randomly initialised net (about 1 % of the masks flip).  To test the kernels rather than
# the number format, `forward_emulated` evaluates the oracle with a bf16 rounding at every
# point where the product stores a tensor (conv output, BN/ReLU output, every gradient
# tensor), fp32 everywhere else.  Agreement with it is expected at the 1e-3 level.
def _rbf(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _RoundGrad(torch.autograd.Function):
    """identity forward on already-bf16 values; rounds the gradient to bf16."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _rbf(g)


class _BNActEmul(torch.autograd.Function):
    """train-mode BatchNorm (+residual)(+ReLU) with the product's storage points: batch
    statistics from the fp32 conv output, normalisation of its bf16 rounding, bf16 result;
    backward with the standard batch-stat formula on the rounded incoming gradient."""

    @staticmethod
    def forward(ctx, y, gamma, beta, res, relu, bn):
        dims = (0, 2, 3, 4)
        mean = y.mean(dims)
        var = (y * y).mean(dims) - mean * mean
        var = var.clamp_min(0)
        n = y.numel() // y.shape[1]
        if bn is not None and bn.track_running_stats:
            with torch.no_grad():
                bn.running_mean.mul_(1 - bn.momentum).add_(bn.momentum * mean)
                bn.running_var.mul_(1 - bn.momentum).add_(bn.momentum * var * n / max(n - 1, 1))
        invstd = torch.rsqrt(var + (bn.eps if bn is not None else 1e-5))
        scale = gamma * invstd
        shift = beta - mean * scale
        sh = (1, -1, 1, 1, 1)
        yr = _rbf(y)
        v = yr * scale.view(sh) + shift.view(sh)
        if res is not None:
            v = v + res
        if relu:
            v = torch.relu(v)
        out = _rbf(v)
        ctx.save_for_backward(yr, mean, invstd, scale, out)
        ctx.relu = relu
        ctx.has_res = res is not None
        return out

    @staticmethod
    def backward(ctx, dz):
        yr, mean, invstd, scale, out = ctx.saved_tensors
        sh = (1, -1, 1, 1, 1)
        dims = (0, 2, 3, 4)
        g = _rbf(dz)
        if ctx.relu:
            g = g * (out > 0).to(g.dtype)
        xhat = (yr - mean.view(sh)) * invstd.view(sh)
        n = yr.numel() // yr.shape[1]
        s1 = g.sum(dims)
        s2 = (g * xhat).sum(dims)
        dy = _rbf(scale.view(sh) * (g - (s1 / n).view(sh) - xhat * (s2 / n).view(sh)))
        return dy, s2, s1, (g if ctx.has_res else None), None, None


def _conv_e(conv, x):
    return nn.functional.conv3d(x, _rbf(conv.weight), None, conv.stride, conv.padding)


def forward_emulated(model, x):
    """logits of R3DOracle `model` (train mode) with the product's bf16 storage points."""
    bb = model.backbone
    h = _BNActEmul.apply(_conv_e(bb.conv1, _rbf(x)), bb.bn1.weight, bb.bn1.bias, None, True, bb.bn1)
    h = _RoundGrad.apply(bb.maxpool(h))
    for layer in (bb.layer1, bb.layer2, bb.layer3, bb.layer4):
        for blk in layer:
            xin = h
            if isinstance(blk, Bottleneck):
                seq = [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)]
            else:
                seq = [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)]
            for conv, bn in seq[:-1]:
                h = _BNActEmul.apply(_conv_e(conv, h), bn.weight, bn.bias, None, True, bn)
            if blk.downsample is not None:
                r = _BNActEmul.apply(_conv_e(blk.downsample[0], xin), blk.downsample[1].weight,
                                     blk.downsample[1].bias, None, False, blk.downsample[1])
            else:
                r = xin
            conv, bn = seq[-1]
            h = _BNActEmul.apply(_conv_e(conv, h), bn.weight, bn.bias, r, True, bn)
    feat = h.mean(dim=(2, 3, 4))
    return model.clf_head(feat)
