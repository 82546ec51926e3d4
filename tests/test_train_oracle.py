"""Oracle (oracle/train.py) == the reference's own training step: tests/golden/train_step.npz holds what
rtpose_light3d(...).train(), rtpose_light3d_loss_fgweight, backward() and torch.optim.SGD(nesterov) produced for two
consecutive steps on a seeded batch (tests/golden/make_golden.py::golden_train)."""
import os

import numpy as np
import torch

from helpers import sample_indices, state_dict_from_keys, train_case_inputs
from oracle import train as otrain

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_step.npz"))


def test_two_training_steps_equal_the_reference(golden):
    sd = state_dict_from_keys(golden.keys["rtpose_light3d"], seed=0)
    batch = [torch.from_numpy(a) for a in train_case_inputs()]
    bufs = None
    for step in range(2):
        r = otrain.train_step(sd, *batch, lr=1.0, momentum=0.9, bufs=bufs)
        assert abs(r["loss"] - float(G["s%d_loss" % step])) < 1e-6 * abs(float(G["s%d_loss" % step]))
        assert np.allclose(r["terms"], G["s%d_terms" % step], rtol=1e-6, atol=0)
        ext = G["s%d_extrema" % step]
        assert abs(float(r["saved"][4][:, :-1].max()) - ext[0]) < 1e-6 and abs(float(r["saved"][3].min()) - ext[3]) < 1e-6
        assert len(r["grads"]) == 135          # 39 conv weights + 30 stage-conv biases + 33 BatchNorms x 2
        # scale of the gradient field: the largest per-entry rms of any parameter.  The biases of convs that feed a BatchNorm
        # have an analytically ZERO gradient (the batch mean is subtracted again): both sides hold rounding noise there
        # (1e-9), which only an absolute floor relative to that scale can compare.
        floor = 1e-6 * max(float(G["s%d_g_norm/%s" % (step, n)]) / np.sqrt(g.numel()) for n, g in r["grads"].items())
        for name, g in r["grads"].items():
            g = g.numpy().ravel()
            ref_norm = float(G["s%d_g_norm/%s" % (step, name)])
            idx = sample_indices(name, g.size)
            rms = ref_norm / np.sqrt(g.size)
            assert abs(np.sqrt((g.astype(np.float64) ** 2).sum()) - ref_norm) <= 2e-5 * ref_norm + floor * np.sqrt(g.size), (step, name)
            assert np.abs(g[idx] - G["s%d_g_samp/%s" % (step, name)]).max() <= 2e-4 * rms + floor, (step, name)
            p = r["new_sd"][name].numpy().ravel()
            assert np.abs(p[idx] - G["s%d_p_samp/%s" % (step, name)]).max() <= 1e-5 * max(1.0, np.abs(p[idx]).max()), (step, name)
        for k in G.files:
            if k.startswith("s%d_stat/" % step):
                assert np.allclose(r["new_sd"][k.split("/", 1)[1]].numpy(), G[k], rtol=1e-5, atol=1e-6), k
        sd, bufs = r["new_sd"], r["bufs"]
