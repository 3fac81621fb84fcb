"""
Capacitance CNN on the device -- the model that sits inside the
reference's step between the rendered charge-stability diagrams and the Kalman
update (src/qadapt/environment/env.py:568-581): `(C,1,R,R)` float32 images in,
`values (C,3)` and `log_vars (C,3)` out.  Here it runs on the whole env batch
`(B*C,1,R,R)` on the GPU, on the same stream as the simulation kernels, reading the
`barrier_images` tensor libqdsim wrote and handing its outputs to
`qd_update_capacitance` -- no host round trip.

The modules restate the reference's model definitions
(src/qadapt/capacitance_model/CapacitancePrediction.py):

  * `CapacitancePredictionModel`  (:114-199)  MobileNetV3 backbone, first conv taking
    one channel, classifier removed, value / confidence heads.  The reference gets the
    backbone from torchvision, which is absent in this image; `MobileNetV3Backbone`
    below is a plain-torch restatement of torchvision's `mobilenet_v3_small/large`
    module tree with the SAME state_dict key names (`features.<i>.block.<j>...`),
    so a checkpoint written by the reference (`model_state_dict`, env.py:738-747)
    loads with `load_state_dict(strict=True)`.  Pinned here only by the published
    parameter counts of the two backbones (tests/test_capacitance_cnn.py); no
    reference checkpoint ships in the repository (`eval_runs/*.pth` is git-ignored),
    so numerical parity of a trained model is "parity unpinned".
  * `IMPALACapacitanceModel`      (:68-111), `IMPALABackbone` (:29-65), `ResNetBlock` (:13-26)
  * `SeparateHeadMobileNet`       (:205-316), `SeparateHeadIMPALA` (:319-383)
  * `create_model`                (:539-560)

`DeviceCapacitanceModel` is the inference engine the env uses: eval mode, BatchNorm
folded into the preceding convolutions, channels-last activations, optional bf16,
and the image batch processed in bounded chunks (activation memory stays a few GB
however many envs the batch holds).  Training the CNN is out of scope.
"""
from __future__ import annotations

import copy
import os
from typing import Callable, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


# =============================================================================
# MobileNetV3 backbone (torchvision module tree, plain torch)
# =============================================================================

def _make_divisible(v: float, divisor: int = 8) -> int:
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class _ConvBNAct(nn.Sequential):
    """[0] conv (no bias), [1] BatchNorm2d(eps 1e-3, momentum 0.01), [2] activation."""

    def __init__(self, cin, cout, kernel, stride=1, groups=1, act: Optional[Callable] = None):
        layers: List[nn.Module] = [
            nn.Conv2d(cin, cout, kernel, stride, padding=(kernel - 1) // 2, groups=groups, bias=False),
            nn.BatchNorm2d(cout, eps=0.001, momentum=0.01)]
        if act is not None:
            layers.append(act(inplace=True))
        super().__init__(*layers)


class _SqueezeExcite(nn.Module):
    def __init__(self, channels, squeeze):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(channels, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, channels, 1)
        self.activation = nn.ReLU()
        self.scale_activation = nn.Hardsigmoid()

    def forward(self, x):
        s = self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))
        return x * s


class _InvertedResidual(nn.Module):
    def __init__(self, cin, kernel, expanded, cout, use_se, use_hs, stride):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        act = nn.Hardswish if use_hs else nn.ReLU
        layers: List[nn.Module] = []
        if expanded != cin:
            layers.append(_ConvBNAct(cin, expanded, 1, act=act))
        layers.append(_ConvBNAct(expanded, expanded, kernel, stride=stride, groups=expanded, act=act))
        if use_se:
            layers.append(_SqueezeExcite(expanded, _make_divisible(expanded // 4, 8)))
        layers.append(_ConvBNAct(expanded, cout, 1, act=None))
        self.block = nn.Sequential(*layers)
        self.out_channels = cout

    def forward(self, x):
        y = self.block(x)
        return x + y if self.use_res_connect else y


# (in, kernel, expanded, out, SE, hardswish, stride)
_V3_SMALL = [(16, 3, 16, 16, True, False, 2), (16, 3, 72, 24, False, False, 2), (24, 3, 88, 24, False, False, 1),
             (24, 5, 96, 40, True, True, 2), (40, 5, 240, 40, True, True, 1), (40, 5, 240, 40, True, True, 1),
             (40, 5, 120, 48, True, True, 1), (48, 5, 144, 48, True, True, 1), (48, 5, 288, 96, True, True, 2),
             (96, 5, 576, 96, True, True, 1), (96, 5, 576, 96, True, True, 1)]
_V3_LARGE = [(16, 3, 16, 16, False, False, 1), (16, 3, 64, 24, False, False, 2), (24, 3, 72, 24, False, False, 1),
             (24, 5, 72, 40, True, False, 2), (40, 5, 120, 40, True, False, 1), (40, 5, 120, 40, True, False, 1),
             (40, 3, 240, 80, False, True, 2), (80, 3, 200, 80, False, True, 1), (80, 3, 184, 80, False, True, 1),
             (80, 3, 184, 80, False, True, 1), (80, 3, 480, 112, True, True, 1), (112, 3, 672, 112, True, True, 1),
             (112, 5, 672, 160, True, True, 2), (160, 5, 960, 160, True, True, 1), (160, 5, 960, 160, True, True, 1)]


class MobileNetV3Backbone(nn.Module):
    """`features` -> global average pool -> flatten -> `classifier` (Identity, as the
    reference replaces it, CapacitancePrediction.py:155).  feature_dim 576 / 960."""

    def __init__(self, arch: str = "small", in_channels: int = 1):
        super().__init__()
        if arch not in ("small", "large"):
            raise ValueError(f"mobilenet must be 'small' or 'large', got {arch!r}")
        spec = _V3_SMALL if arch == "small" else _V3_LARGE
        layers: List[nn.Module] = [_ConvBNAct(in_channels, spec[0][0], 3, stride=2, act=nn.Hardswish)]
        layers += [_InvertedResidual(*row) for row in spec]
        last_in = spec[-1][3]
        self.feature_dim = 6 * last_in
        layers.append(_ConvBNAct(last_in, self.feature_dim, 1, act=nn.Hardswish))
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Identity()
        for m in self.modules():                       # torchvision's initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight); nn.init.zeros_(m.bias)

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


def _head(feature_dim, widths, out, dropouts):
    layers: List[nn.Module] = []
    prev = feature_dim
    for w, d in zip(widths, dropouts):
        layers += [nn.Linear(prev, w), nn.ReLU()]
        if d:
            layers.append(nn.Dropout(0.2))
        prev = w
    layers.append(nn.Linear(prev, out))
    return nn.Sequential(*layers)


class CapacitancePredictionModel(nn.Module):
    """CapacitancePrediction.py:114-199 (the class env.py:718 instantiates with output_size 3)."""

    def __init__(self, output_size, mobilenet="small"):
        super().__init__()
        self.backbone = MobileNetV3Backbone(mobilenet, in_channels=1)
        self.output_size = output_size
        fd = self.backbone.feature_dim
        self.value_head = _head(fd, (256, 128), output_size, (True, True))
        self.confidence_head = _head(fd, (256, 128), output_size, (True, True))

    def forward(self, x):
        f = self.backbone(x)
        return self.value_head(f), self.confidence_head(f)


# =============================================================================
# IMPALA backbone
# =============================================================================

class ResNetBlock(nn.Module):
    """x + conv2(relu(conv1(relu(x))))  (CapacitancePrediction.py:13-26)."""

    def __init__(self, channels: int):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, 3, 1, 1)
        self.conv2 = nn.Conv2d(channels, channels, 3, 1, 1)

    def forward(self, x):
        return x + self.conv2(F.relu(self.conv1(F.relu(x))))


class IMPALABackbone(nn.Module):
    """Per stage: conv3x3, maxpool 3/2, `num_res_blocks` residual blocks, ReLU; then a
    4x4 adaptive average pool (CapacitancePrediction.py:29-65)."""

    def __init__(self, in_channels: int = 1, channels: list = None, num_res_blocks: int = 2):
        super().__init__()
        channels = [16, 32, 32] if channels is None else list(channels)
        self.channels, self.num_res_blocks = channels, num_res_blocks
        layers: List[nn.Module] = []
        prev = in_channels
        for c in channels:
            layers += [nn.Conv2d(prev, c, 3, 1, 1), nn.MaxPool2d(3, 2, 1)]
            layers += [ResNetBlock(c) for _ in range(num_res_blocks)]
            layers.append(nn.ReLU())
            prev = c
        layers += [nn.AdaptiveAvgPool2d((4, 4)), nn.Flatten()]
        self.cnn = nn.Sequential(*layers)
        self.feature_dim = channels[-1] * 16

    def forward(self, x):
        return self.cnn(x)


class IMPALACapacitanceModel(nn.Module):
    """CapacitancePrediction.py:68-111."""

    def __init__(self, output_size: int, channels: list = None, num_res_blocks: int = 2):
        super().__init__()
        self.backbone = IMPALABackbone(1, channels, num_res_blocks)
        self.output_size = output_size
        fd = self.backbone.feature_dim
        self.value_head = _head(fd, (128, 64), output_size, (True, True))
        self.confidence_head = _head(fd, (128, 64), output_size, (True, True))

    def forward(self, x):
        f = self.backbone(x)
        return self.value_head(f), self.confidence_head(f)


# =============================================================================
# separate NN / NNN heads
# =============================================================================

class _SeparateHeads(nn.Module):
    def forward(self, x):
        f = self.backbone(x)
        return {"nn": (self.nn_value_head(f), self.nn_confidence_head(f)),
                "nnn": (self.nnn_value_head(f), self.nnn_confidence_head(f))}

    def forward_combined(self, x):
        """values / log_vars (batch, 3) ordered [NN, NNN_right, NNN_left]."""
        out = self.forward(x)
        return (torch.cat([out["nn"][0], out["nnn"][0]], dim=1),
                torch.cat([out["nn"][1], out["nnn"][1]], dim=1))


class SeparateHeadMobileNet(_SeparateHeads):
    """CapacitancePrediction.py:205-316."""

    def __init__(self, mobilenet="small"):
        super().__init__()
        self.backbone = MobileNetV3Backbone(mobilenet, in_channels=1)
        fd = self.backbone.feature_dim
        self.nn_value_head = _head(fd, (128, 64), 1, (True, False))
        self.nn_confidence_head = _head(fd, (128, 64), 1, (True, False))
        self.nnn_value_head = _head(fd, (128, 64), 2, (True, False))
        self.nnn_confidence_head = _head(fd, (128, 64), 2, (True, False))


class SeparateHeadIMPALA(_SeparateHeads):
    """CapacitancePrediction.py:319-383."""

    def __init__(self, channels=None, num_res_blocks=2):
        super().__init__()
        self.backbone = IMPALABackbone(1, channels, num_res_blocks)
        fd = self.backbone.feature_dim
        self.nn_value_head = _head(fd, (64,), 1, (True,))
        self.nn_confidence_head = _head(fd, (64,), 1, (True,))
        self.nnn_value_head = _head(fd, (64,), 2, (True,))
        self.nnn_confidence_head = _head(fd, (64,), 2, (True,))


def create_model(output_size, backbone="mobilenet", mobilenet="small", impala_channels=None, num_res_blocks=2,
                 separate_heads=False):
    """CapacitancePrediction.py:539-560."""
    if separate_heads:
        if backbone == "impala":
            return SeparateHeadIMPALA(channels=impala_channels, num_res_blocks=num_res_blocks)
        return SeparateHeadMobileNet(mobilenet=mobilenet)
    if backbone == "impala":
        return IMPALACapacitanceModel(output_size, channels=impala_channels, num_res_blocks=num_res_blocks)
    return CapacitancePredictionModel(output_size, mobilenet=mobilenet)


# =============================================================================
# checkpoints
# =============================================================================

def load_checkpoint(model: nn.Module, path: str, map_location="cpu") -> nn.Module:
    """env.py:724-747: the file is either a bare state_dict or a training checkpoint with a
    `model_state_dict` entry.  Loaded with `weights_only=True` (nothing in the file is
    executed); strict key matching."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"Model weights not found at: {path}")
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    state = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    model.load_state_dict(state)
    return model


# =============================================================================
# inference engine
# =============================================================================

def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> nn.Conv2d:
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    out = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                    conv.dilation, conv.groups, bias=True)
    out.weight.data = (conv.weight * scale.reshape(-1, 1, 1, 1)).detach().clone()
    b0 = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
    out.bias.data = ((b0 - bn.running_mean) * scale + bn.bias).detach().clone()
    return out


def fold_batchnorm(model: nn.Module) -> nn.Module:
    """Deep copy of `model` (eval mode) with every conv+BatchNorm pair of the MobileNet
    blocks replaced by one biased convolution (inference only)."""
    m = copy.deepcopy(model).eval()
    for mod in m.modules():
        if isinstance(mod, _ConvBNAct) and isinstance(mod[1], nn.BatchNorm2d):
            mod[0] = _fold(mod[0], mod[1])
            mod[1] = nn.Identity()
    return m


class DeviceCapacitanceModel:
    """Callable `images (n,1,R,R) float32 on the GPU -> (values (n,k), log_vars (n,k)) float32`,
    the `capacitance_model=` argument of VecQuantumDeviceEnv.

    dtype float32 reproduces the reference's arithmetic; bfloat16 is offered for throughput
    (the Kalman gate at variance 0.05 makes the update tolerant of ~1e-2 relative error, but
    it is not the reference's precision and is off by default)."""

    def __init__(self, model: nn.Module, device="cuda", dtype=torch.float32, chunk_images: int = 16384,
                 channels_last: bool = True, fold_bn: bool = True):
        m = fold_batchnorm(model) if fold_bn else copy.deepcopy(model).eval()
        self.dtype = dtype
        self.device = torch.device(device)
        self.channels_last = channels_last
        self.chunk_images = int(chunk_images)
        m = m.to(self.device, dtype=dtype)
        if channels_last:
            m = m.to(memory_format=torch.channels_last)
        for p in m.parameters():
            p.requires_grad_(False)
        self.model = m
        self._combined = hasattr(m, "forward_combined")

    @torch.no_grad()
    def __call__(self, images):
        x = images.to(self.device)
        outs_v, outs_l = [], []
        for s in range(0, x.shape[0], self.chunk_images):
            xb = x[s:s + self.chunk_images].to(self.dtype)
            if self.channels_last:
                xb = xb.contiguous(memory_format=torch.channels_last)
            v, l = self.model.forward_combined(xb) if self._combined else self.model(xb)
            outs_v.append(v.float()); outs_l.append(l.float())
        if len(outs_v) == 1:
            return outs_v[0], outs_l[0]
        return torch.cat(outs_v), torch.cat(outs_l)


def build_device_model(checkpoint: Optional[str] = None, output_size: int = 3, backbone: str = "mobilenet",
                       mobilenet: str = "small", separate_heads: bool = False, device="cuda", seed: Optional[int] = None,
                       **engine_kw) -> DeviceCapacitanceModel:
    """Model as env.py:716-749 builds it (`CapacitancePredictionModel(output_size)` + checkpoint),
    wrapped for batched device inference.  `checkpoint=None` gives seeded random weights
    (throughput runs; there is no network for pretrained weights)."""
    if seed is not None:
        torch.manual_seed(seed)
    model = create_model(output_size, backbone=backbone, mobilenet=mobilenet, separate_heads=separate_heads)
    if checkpoint:
        load_checkpoint(model, checkpoint)
    return DeviceCapacitanceModel(model, device=device, **engine_kw)
