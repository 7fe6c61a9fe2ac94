"""Import the upstream reference (read-only at /root/reference) in THIS container.

Used only by tests/golden/make_golden.py, which runs in the build container;
nothing here runs on the GPU box (/root/reference does not exist there) and no
reference source is copied: missing third-party modules are replaced by inert
stand-ins so that `models.CoPoNeRF` imports and its render path runs on CPU
(SURVEY.md §8(c)).
"""
from __future__ import annotations

import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"


def _plain_resnet34(pretrained=False):
    """Structural stand-in for torchvision.models.resnet34 (only constructed, never
    compared: the encoder is outside the render path being pinned)."""

    class Block(nn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(cout)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(cout)
            self.downsample = None
            if stride != 1 or cin != cout:
                self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False),
                                                nn.BatchNorm2d(cout))

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            y = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
            return self.relu(y + idt)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            cfg, cin, layers = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], 64, []
            for cout, n, s in cfg:
                layers.append(nn.Sequential(*[Block(cin if i == 0 else cout, cout, s if i == 0 else 1)
                                              for i in range(n)]))
                cin = cout
            self.layer1, self.layer2, self.layer3, self.layer4 = layers
            self.avgpool = nn.AdaptiveAvgPool2d(1)
            self.fc = nn.Linear(512, 1000)

    return Net()


def install():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True

    jt = types.ModuleType("jaxtyping")

    class _Ann:
        def __class_getitem__(cls, item):
            return cls

    jt.Float = jt.Int64 = jt.Bool = _Ann
    sys.modules.setdefault("jaxtyping", jt)

    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_layers = types.ModuleType("timm.models.layers")
    timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
    timm_layers.DropPath = nn.Identity
    timm.models = timm_models
    timm_models.layers = timm_layers
    sys.modules.setdefault("timm", timm)
    sys.modules.setdefault("timm.models", timm_models)
    sys.modules.setdefault("timm.models.layers", timm_layers)

    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    # data/realestate10k_dataio.py and utils_training/data_util.py import these at module level; the functions pinned by
    # make_golden_input.py (square crop, /127.5-1, intrinsics, frame / ray sampling) never call into them
    for name in ("imageio", "skimage", "h5py"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["imageio"].imread = None

    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.resnet34 = _plain_resnet34
    tv.models = tvm
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tvm)

    # the reference hard-codes .cuda() in the hot path (geometry.py:320,398)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self


def build_reference_model(render_weights, npoints=64, H=256):
    """Reference CoPoNeRF(n_view=2) with the render-path weights loaded."""
    install()
    import io
    import contextlib
    from models import CoPoNeRF as ref_mod  # noqa: E402  (reference module)

    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(0)
        model = ref_mod.CoPoNeRF(n_view=2, npoints=npoints)
    missing, unexpected = model.load_state_dict(render_weights, strict=False)
    assert not unexpected, unexpected
    model.eval()
    model.H = model.W = H
    return model
