"""``rtpose_light3d`` ("Open-Pose+") with the reference's Python surface, computed by HIP kernels.

Drop-in for tpm/lib/network/rtpose_light3d.py:249-356: same constructor, same sub-module names
(``model0``, ``model{1,2}_{1,2,3}``) and therefore the same 234 ``state_dict`` keys (SURVEY
Appendix A), same ``forward(x) -> ((paf, heat, z), saved_for_loss[6])``.  The sub-modules only
HOLD parameters; ``forward`` hands the folded/packed weights to libpopnet_hip.so
(pn_rtpose_forward) -- direct MFMA convolution with fused BN + activation, the stage-2 concat
written in place by the producers, sigmoid range casts in the last epilogue.
"""
import ctypes as C

import torch
import torch.nn as nn
from torch.nn import init

from .. import _lib
from ._hipnet import HipNetModule, bn_, conv_


class _ResUnit(nn.Module):
    """Parameter holder laid out like BasicBlock (rtpose_light3d.py:36-72)."""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1, self.bn1 = conv_(cin, cout, 3, stride), bn_(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = conv_(cout, cout, 3), bn_(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(conv_(cin, cout, 1, stride), bn_(cout))


class _Stem(nn.Module):
    """Parameter holder laid out like ResPreprocessNet(BasicBlock, [2, 1]) (rtpose_light3d.py:124-219)."""

    def __init__(self, input_dim):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = bn_(64)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer1 = nn.Sequential(_ResUnit(64, 64), _ResUnit(64, 64))
        self.avgpool1 = nn.AvgPool2d(kernel_size=3, stride=2, padding=1)
        self.layer2 = nn.Sequential(_ResUnit(64, 128))
        self.conv2, self.bn2 = conv_(128, 128, 1), bn_(128)
        self.relu2 = nn.ReLU(inplace=True)
        self.avgpool2 = nn.AvgPool2d(kernel_size=3, stride=2, padding=1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')


def _branch(spec):
    """Sequential with conv(+bias) at 0,3,6,9,12, BN at 1,4,7,10, LeakyReLU(0.1) at 2,5,8,11 --
    the index layout make_stages produces (rtpose_light3d.py:222-246)."""
    mods = []
    for cin, cout, k in spec[:-1]:
        mods += [conv_(cin, cout, k, bias=True), bn_(cout), nn.LeakyReLU(0.1, inplace=True)]
    cin, cout, k = spec[-1]
    mods.append(conv_(cin, cout, k, bias=True))
    return nn.Sequential(*mods)


class rtpose_light3d(HipNetModule):
    _kind = _lib.PN_NET_RTPOSE_LIGHT3D

    def __init__(self, num_parts=18, num_limbs=19, num_stages=2, input_dim=3):
        super().__init__()
        if num_stages != 2:
            raise ValueError("rtpose_light3d is hard-wired for 2 stages (as the reference, rtpose_light3d.py:311)")
        self.num_parts, self.num_limbs, self.num_stages, self.input_dim = num_parts, num_limbs, num_stages, input_dim
        n_paf, n_heat, n_z = 2 * num_limbs, num_parts + 1, num_limbs + 1
        self.model0 = _Stem(input_dim)
        for stage, cin in ((1, 128), (2, 128 + n_paf + n_heat + n_z)):
            setattr(self, 'model%d_1' % stage, _branch([(cin, 256, 3), (256, 256, 3), (256, 256, 3), (256, 128, 1), (128, n_paf, 1)]))
            setattr(self, 'model%d_2' % stage, _branch([(cin, 128, 3), (128, 128, 3), (128, 128, 3), (128, 128, 3), (128, n_heat, 3)]))
            setattr(self, 'model%d_3' % stage, _branch([(cin, 128, 3), (128, 64, 3), (64, 64, 3), (64, 64, 3), (64, n_z, 3)]))
        for m in self.modules():          # _initialize_weights_norm (rtpose_light3d.py:358-362)
            if isinstance(m, nn.Conv2d):
                init.normal_(m.weight, mean=0, std=0.01)

    def _net_args(self):
        return self._kind, self.num_parts, self.num_limbs, self.input_dim

    # ---- train mode: the reference's forward (rtpose_light3d.py:206-219, 326-356) on the autograd-wrapped HIP primitives ----
    def _forward_train(self, x):
        from . import _autograd as ag
        x = self._check_input(x)
        if x.shape[2] % 8 or x.shape[3] % 8:
            raise _lib.PopnetError("input size must be a multiple of 8")
        for p in self.parameters():
            if p.device != x.device:
                raise _lib.PopnetError("popnet_amd: module parameters are on %s, the input on %s -- call model.cuda() (the HIP path has no CPU fallback)" % (p.device, x.device))

        def block(u, a):                       # BasicBlock (rtpose_light3d.py:36-72)
            y = ag.bn_act(ag.conv(a, u.conv1), u.bn1, ag.ACT_RELU)
            y = ag.conv(y, u.conv2)
            idn = a if u.downsample is None else ag.bn_act(ag.conv(a, u.downsample[0]), u.downsample[1], ag.ACT_NONE)
            return ag.bn_act(y, u.bn2, ag.ACT_RELU, res=idn)

        def branch(seq, a):                    # make_stages Sequential (:222-246)
            for i in (0, 3, 6, 9):
                a = ag.bn_act(ag.conv(a, seq[i]), seq[i + 1], ag.ACT_LEAKY)
            return ag.conv(a, seq[12])

        m0 = self.model0
        a = ag.bn_act(ag.conv(x, m0.conv1), m0.bn1, ag.ACT_RELU)
        for u in m0.layer1:
            a = block(u, a)
        a = ag.avgpool(a)
        for u in m0.layer2:
            a = block(u, a)
        a = ag.bn_act(ag.conv(a, m0.conv2), m0.bn2, ag.ACT_RELU)
        out1 = ag.avgpool(a)
        # the range casts of :335-337 / :347-349 and the concat of :339 are element-wise glue: left to torch (autograd orders them)
        out1_1 = (branch(self.model1_1, out1).sigmoid() - 0.5) * 4
        out1_2 = branch(self.model1_2, out1).sigmoid()
        out1_3 = (branch(self.model1_3, out1).sigmoid() - 0.5) * 4
        out2 = torch.cat([out1_1, out1_2, out1_3, out1], 1)
        out2_1 = (branch(self.model2_1, out2).sigmoid() - 0.5) * 4
        out2_2 = branch(self.model2_2, out2).sigmoid()
        out2_3 = (branch(self.model2_3, out2).sigmoid() - 0.5) * 4
        self.invalidate()                      # the weights are about to change: the eval-mode net is re-folded on its next use
        return (out2_1, out2_2, out2_3), [out1_1, out1_2, out1_3, out2_1, out2_2, out2_3]

    def forward(self, x):
        if self.training:
            return self._forward_train(x)
        x = self._check_input(x)
        B, _, H, W = x.shape
        dev = x.device
        net = self._compile(dev, B, H, W)
        h, w = H // 8, W // 8
        n_paf, n_heat, n_z = 2 * self.num_limbs, self.num_parts + 1, self.num_limbs + 1
        paf = torch.empty((B, n_paf, h, w), device=dev, dtype=torch.float32)
        heat = torch.empty((B, n_heat, h, w), device=dev, dtype=torch.float32)
        z = torch.empty((B, n_z, h, w), device=dev, dtype=torch.float32)
        self._ctx.check(_lib.lib().pn_rtpose_forward(
            net, C.c_void_p(x.data_ptr()), B, C.c_void_p(paf.data_ptr()), C.c_void_p(heat.data_ptr()),
            C.c_void_p(z.data_ptr()), _lib.current_stream_ptr(dev)), "pn_rtpose_forward")
        saved_for_loss = [self._activation("paf1", B, (n_paf, h, w), dev),
                          self._activation("heat1", B, (n_heat, h, w), dev),
                          self._activation("z1", B, (n_z, h, w), dev), paf, heat, z]
        return (paf, heat, z), saved_for_loss

    def stem_features(self, B):
        """[B,128,H/8,W/8] output of model0 for the last forward (diagnostics / tests)."""
        key = self._net[1]
        return self._activation("feat", B, (128, key[2] // 8, key[3] // 8), torch.device("cuda", key[0]))
