"""``YoloPoseNet`` ("Yolo-Pose+") with the reference's Python surface, computed by HIP kernels.

Drop-in for tpm/lib/network/yolo_posenet.py:87-158 (+ ResNetBackBone :26-56, resnet34 layers
tpm/lib/network/resnet.py:27-56,134-148): same constructor, same 223 ``state_dict`` keys --
including ``model0.layer3.*``, which the reference builds and checkpoints but never executes
(yolo_posenet.py:42,54) and which is therefore held here and skipped at compile time -- and the
same ``forward(x) -> Tensor[B, A*(5+3J), H/16, W/16]`` with the per-slice sigmoid casts applied.
"""
import ctypes as C

import torch
import torch.nn as nn
from torch.nn import init

from .. import _lib
from ._hipnet import HipNetModule, bn_, conv_
from .rtpose_light3d import _ResUnit


def _res_layer(cin, cout, blocks, stride):
    return nn.Sequential(*([_ResUnit(cin, cout, stride)] + [_ResUnit(cout, cout) for _ in range(blocks - 1)]))


class _BackBone(nn.Module):
    def __init__(self, input_dim):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = bn_(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.avgpool = nn.AvgPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = _res_layer(64, 64, 3, 1)
        self.layer2 = _res_layer(64, 128, 4, 2)
        self.layer3 = _res_layer(128, 256, 6, 2)     # dead weight, kept for checkpoint compatibility


def _cbl(cin, cout, pool=False):
    mods = [conv_(cin, cout, 3, bias=False), bn_(cout), nn.LeakyReLU(0.1, inplace=True)]
    if pool:
        mods.append(nn.MaxPool2d(2, 2))
    return nn.Sequential(*mods)


class YoloPoseNet(HipNetModule):
    _kind = _lib.PN_NET_YOLO_POSENET

    def __init__(self, num_parts=15, input_dim=3, anchors=[(6., 3.), (12., 6.)]):
        super().__init__()
        self.num_parts, self.anchors, self.input_dim = num_parts, anchors, input_dim
        self.model0 = _BackBone(input_dim)
        neck = []
        for i, (cin, cout) in enumerate([(128, 256), (256, 256), (256, 256), (256, 256)]):
            neck += [conv_(cin, cout, 3, bias=True), bn_(cout), nn.LeakyReLU(0.1, inplace=True)]
        neck.append(conv_(256, 256, 3, bias=True))
        self.model1 = nn.Sequential(*neck)
        self.model2_1 = _cbl(256, 256, pool=True)
        self.model2_2 = _cbl(256, 256)
        self.model2_3 = _cbl(256, 128)
        self.model2_4 = nn.Sequential(conv_(128, len(anchors) * (5 + 3 * num_parts), 3, bias=False))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                init.normal_(m.weight, mean=0, std=0.01)

    def _net_args(self):
        return self._kind, self.num_parts, len(self.anchors), self.input_dim

    def forward(self, x):
        x = self._check_input(x)
        B, _, H, W = x.shape
        dev = x.device
        net = self._compile(dev, B, H, W)
        out = torch.empty((B, len(self.anchors) * (5 + 3 * self.num_parts), H // 16, W // 16),
                          device=dev, dtype=torch.float32)
        self._ctx.check(_lib.lib().pn_yolo_forward(net, C.c_void_p(x.data_ptr()), B, C.c_void_p(out.data_ptr()),
                                                   _lib.current_stream_ptr(dev)), "pn_yolo_forward")
        return out

    def backbone_features(self, B):
        key = self._net[1]
        return self._activation("feat", B, (128, key[2] // 8, key[3] // 8), torch.device("cuda", key[0]))
