"""Network forward (ORACLE; test infrastructure -- see oracle/__init__.py).

Plain PyTorch fp32 CPU restatement, driven by a reference-format ``state_dict`` (with or
without the DataParallel ``module.`` prefix), of

  rtpose_light3d.forward    third_party_methods/lib/network/rtpose_light3d.py:326-356
    ResPreprocessNet        ...:124-219   (stem, stride 8)
    BasicBlock              ...:36-72
    make_stages             ...:222-246
  YoloPoseNet.forward       third_party_methods/lib/network/yolo_posenet.py:131-158
    ResNetBackBone          ...:26-56
    resnet.BasicBlock       third_party_methods/lib/network/resnet.py:27-56

Eval-mode semantics only (BatchNorm uses running statistics, eps = 1e-5).
"""
import torch
import torch.nn.functional as F

EPS = 1e-5


def strip_module_prefix(sd):
    """evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py:136-139 strips the first dotted
    component unconditionally; here only a literal ``module.`` prefix is removed."""
    if all(k.startswith('module.') for k in sd):
        return {k[len('module.'):]: v for k, v in sd.items()}
    return dict(sd)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], False, 0.0, EPS)


def _conv(x, sd, p, stride=1, pad=0):
    return F.conv2d(x, sd[p + '.weight'], sd.get(p + '.bias'), stride, pad)


def _basic_block(x, sd, p, stride=1):
    out = F.relu(_bn(_conv(x, sd, p + '.conv1', stride, 1), sd, p + '.bn1'))
    out = _bn(_conv(out, sd, p + '.conv2', 1, 1), sd, p + '.bn2')
    if (p + '.downsample.0.weight') in sd:
        x = _bn(_conv(x, sd, p + '.downsample.0', stride, 0), sd, p + '.downsample.1')
    return F.relu(out + x)


def _stage(x, sd, p, leaky=True):
    """make_stages Sequential: conv(+bias) BN LeakyReLU(0.1) at indices 0,3,6,9; bare conv at 12."""
    for i in (0, 3, 6, 9):
        w = sd['%s.%d.weight' % (p, i)]
        x = F.conv2d(x, w, sd.get('%s.%d.bias' % (p, i)), 1, w.shape[-1] // 2)
        x = F.leaky_relu(_bn(x, sd, '%s.%d' % (p, i + 1)), 0.1)
    w = sd[p + '.12.weight']
    return F.conv2d(x, w, sd.get(p + '.12.bias'), 1, w.shape[-1] // 2)


def rtpose_stem(x, sd):
    x = F.relu(_bn(_conv(x, sd, 'model0.conv1', 2, 3), sd, 'model0.bn1'))
    x = _basic_block(x, sd, 'model0.layer1.0')
    x = _basic_block(x, sd, 'model0.layer1.1')
    x = F.avg_pool2d(x, 3, 2, 1)                      # count_include_pad=True
    x = _basic_block(x, sd, 'model0.layer2.0')
    x = F.relu(_bn(_conv(x, sd, 'model0.conv2'), sd, 'model0.bn2'))
    return F.avg_pool2d(x, 3, 2, 1)


def rtpose_light3d_forward(x, sd, return_intermediate=False):
    """x [B,1,H,W] float32 (normalised depth).  Returns (paf, heat, z) of stage 2
    (and a dict of intermediates when asked)."""
    sd = strip_module_prefix(sd)
    with torch.no_grad():
        feat = rtpose_stem(x, sd)
        l1 = (_stage(feat, sd, 'model1_1').sigmoid() - 0.5) * 4
        s1 = _stage(feat, sd, 'model1_2').sigmoid()
        d1 = (_stage(feat, sd, 'model1_3').sigmoid() - 0.5) * 4
        cat = torch.cat([l1, s1, d1, feat], 1)
        l2 = (_stage(cat, sd, 'model2_1').sigmoid() - 0.5) * 4
        s2 = _stage(cat, sd, 'model2_2').sigmoid()
        d2 = (_stage(cat, sd, 'model2_3').sigmoid() - 0.5) * 4
    if return_intermediate:
        return (l2, s2, d2), {'feat': feat, 'paf1': l1, 'heat1': s1, 'z1': d1}
    return l2, s2, d2


def yolo_posenet_forward(x, sd, num_parts=15, num_anchors=2, return_intermediate=False):
    sd = strip_module_prefix(sd)
    with torch.no_grad():
        x = F.relu(_bn(_conv(x, sd, 'model0.conv1', 2, 3), sd, 'model0.bn1'))
        x = F.max_pool2d(x, 3, 2, 1)
        for i in range(3):
            x = _basic_block(x, sd, 'model0.layer1.%d' % i)
        x = _basic_block(x, sd, 'model0.layer2.0', stride=2)
        for i in range(1, 4):
            x = _basic_block(x, sd, 'model0.layer2.%d' % i)
        feat = x
        x = _stage(x, sd, 'model1')
        x = F.leaky_relu(_bn(_conv(x, sd, 'model2_1.0', 1, 1), sd, 'model2_1.1'), 0.1)
        x = F.max_pool2d(x, 2, 2)
        x = F.leaky_relu(_bn(_conv(x, sd, 'model2_2.0', 1, 1), sd, 'model2_2.1'), 0.1)
        x = F.leaky_relu(_bn(_conv(x, sd, 'model2_3.0', 1, 1), sd, 'model2_3.1'), 0.1)
        out = _conv(x, sd, 'model2_4.0', 1, 1).clone()
        naf = 5 + 3 * num_parts
        for i in range(num_anchors):
            b = i * naf
            out[:, b:b + 2] = (out[:, b:b + 2].sigmoid() - 0.5) * 2
            out[:, b + 2:b + 4] = out[:, b + 2:b + 4].sigmoid() * 2
            out[:, b + 4] = out[:, b + 4].sigmoid()
            out[:, b + 5:b + naf] = (out[:, b + 5:b + naf].sigmoid() - 0.5) * 4
    if return_intermediate:
        return out, {'feat': feat}
    return out
