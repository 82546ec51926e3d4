"""Train-mode vs eval-mode loss of a model trained by TrainEngine in either precision (running-statistics sanity check)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import popnet_amd  # noqa: E402,F401
from popnet_amd import synth, targets  # noqa: E402
from popnet_amd.network.rtpose_light3d import rtpose_light3d  # noqa: E402
from popnet_amd.train import TrainEngine  # noqa: E402
import synthetic_train_eval as ste  # noqa: E402
import train_mpaug  # noqa: E402
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
pool = [[t.contiguous() for t in targets.mpaug_batch(*ste.scenes(dev, 32, 1000 + i))] for i in range(40)]
held = [t.contiguous() for t in targets.mpaug_batch(*ste.scenes(dev, 32, 9000))]
for prec in ("fp32", "bf16x3"):
    eng = TrainEngine(synth.init_like_state_dict(seed=0), device=dev, lr=0.5, precision=prec)
    for k in range(steps):
        if k == steps * 2 // 3:
            eng.lr *= 0.2
        t = eng.step(*pool[k % 40])
    module = rtpose_light3d(15, 14, 2, input_dim=1)
    module.load_state_dict(eng.state_dict())
    module = module.to(dev).eval()
    module.precision = "fp32"
    ev = train_mpaug.eval_loss(module, held)
    probe = TrainEngine(eng.state_dict(), device=dev, lr=0.0)
    tr = float(probe.forward_backward(*held).sum())
    rv = torch.cat([v.flatten() for k, v in eng.state_dict().items() if k.endswith("running_var")])
    rm = torch.cat([v.flatten() for k, v in eng.state_dict().items() if k.endswith("running_mean")])
    print(prec, "last train loss %.5f | held-out: train-mode %.5f eval-mode %.5f | running_var min %.3g max %.3g, |running_mean| max %.3g, |w| max %.3g" % (
        float(t.sum()), tr, ev, float(rv.min()), float(rv.max()), float(rm.abs().max()), float(eng.flat_p.abs().max())))
