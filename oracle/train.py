"""One training step of rtpose_light3d (ORACLE; test infrastructure -- see oracle/__init__.py).

Plain PyTorch fp32 CPU restatement (functional, driven by a reference-format state_dict) of what
train_rtpose_light3d_kdh3d_mpaug.py:160-180 (CR) does per batch:
  model.train() forward            third_party_methods/lib/network/rtpose_light3d.py:326-356 (BatchNorm on batch statistics,
                                   running statistics updated with momentum 0.1)
  rtpose_light3d_loss_fgweight     third_party_methods/lib/network/losses.py:65-106
  total_loss.backward()            autograd
  SGD(lr, momentum, nesterov=True) train_rtpose_light3d_kdh3d_mpaug.py:313-316, torch.optim.SGD semantics (dampening 0)
Pinned by tests/golden/train_step.npz, produced by the reference's own module, loss function and torch.optim.SGD.
"""
import torch
import torch.nn.functional as F

from .nets import EPS, strip_module_prefix

BN_MOMENTUM = 0.1


def _is_stat(k):
    return k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")


def forward_train(x, params, stats):
    """params: name -> leaf tensor (weights, biases, BN affine); stats: name -> running statistic, updated in place.
    Returns saved_for_loss [paf1, heat1, z1, paf2, heat2, z2] (rtpose_light3d.py:340-354)."""
    def bn(x, p):
        return F.batch_norm(x, stats[p + ".running_mean"], stats[p + ".running_var"], params[p + ".weight"], params[p + ".bias"], True, BN_MOMENTUM, EPS)

    def conv(x, p, stride=1, pad=0):
        return F.conv2d(x, params[p + ".weight"], params.get(p + ".bias"), stride, pad)

    def block(x, p):
        out = F.relu(bn(conv(x, p + ".conv1", 1, 1), p + ".bn1"))
        out = bn(conv(out, p + ".conv2", 1, 1), p + ".bn2")
        if (p + ".downsample.0.weight") in params:
            x = bn(conv(x, p + ".downsample.0"), p + ".downsample.1")
        return F.relu(out + x)

    def stage(x, p):
        for i in (0, 3, 6, 9):
            w = params["%s.%d.weight" % (p, i)]
            x = F.leaky_relu(bn(F.conv2d(x, w, params["%s.%d.bias" % (p, i)], 1, w.shape[-1] // 2), "%s.%d" % (p, i + 1)), 0.1)
        w = params[p + ".12.weight"]
        return F.conv2d(x, w, params[p + ".12.bias"], 1, w.shape[-1] // 2)

    x = F.relu(bn(conv(x, "model0.conv1", 2, 3), "model0.bn1"))
    x = block(block(x, "model0.layer1.0"), "model0.layer1.1")
    x = F.avg_pool2d(x, 3, 2, 1)
    x = block(x, "model0.layer2.0")
    x = F.relu(bn(conv(x, "model0.conv2"), "model0.bn2"))
    feat = F.avg_pool2d(x, 3, 2, 1)
    l1 = (stage(feat, "model1_1").sigmoid() - 0.5) * 4
    s1 = stage(feat, "model1_2").sigmoid()
    d1 = (stage(feat, "model1_3").sigmoid() - 0.5) * 4
    cat = torch.cat([l1, s1, d1, feat], 1)
    l2 = (stage(cat, "model2_1").sigmoid() - 0.5) * 4
    s2 = stage(cat, "model2_2").sigmoid()
    d2 = (stage(cat, "model2_3").sigmoid() - 0.5) * 4
    return [l1, s1, d1, l2, s2, d2]


def loss_fgweight(saved, heat_gt, paf_gt, z_gt, fg_mask):
    """losses.py:65-90: per stage MSE(paf) + MSE(heat) + mean((z - z_gt)^2 * (0.1 + 0.9 fg)).  -> (total, [6 terms])"""
    weight = torch.ones_like(fg_mask) * 0.1 + fg_mask * 0.9
    terms = []
    for j in range(2):
        terms.append(F.mse_loss(saved[3 * j], paf_gt))
        terms.append(F.mse_loss(saved[3 * j + 1], heat_gt))
        terms.append((((saved[3 * j + 2] - z_gt) ** 2) * weight).mean())
    total = 0
    for t in terms:
        total = total + t
    return total, terms


def train_step(sd, img, heat_gt, paf_gt, z_gt, fg_mask, lr=1.0, momentum=0.9, bufs=None, apply=True, dtype=torch.float32):
    """One step from state_dict `sd` (not modified).  Returns dict(loss, terms, grads {name: tensor}, new_sd, bufs).
    dtype=torch.float64 evaluates the same graph in double precision (tests use it to separate conditioning from error)."""
    sd = strip_module_prefix(sd)
    params = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items() if not _is_stat(k) and not k.startswith("model0.layer3")}
    stats = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.detach().clone()) for k, v in sd.items() if _is_stat(k)}
    saved = forward_train(img, params, stats)
    total, terms = loss_fgweight(saved, heat_gt, paf_gt, z_gt, fg_mask)
    names = sorted(params)
    grads = dict(zip(names, torch.autograd.grad(total, [params[n] for n in names])))
    new_sd = {k: v.detach().clone() for k, v in sd.items()}
    new_sd.update(stats)
    new_bufs = {}
    if apply:
        for n in names:          # torch.optim.SGD, nesterov, dampening 0, no weight decay
            g = grads[n]
            b = g.clone() if bufs is None or n not in bufs else bufs[n] * momentum + g
            new_bufs[n] = b
            new_sd[n] = params[n].detach() - lr * (g + momentum * b)
    return {"loss": float(total.detach()), "terms": [float(t.detach()) for t in terms], "grads": grads, "new_sd": new_sd, "bufs": new_bufs,
            "saved": [s.detach() for s in saved]}
