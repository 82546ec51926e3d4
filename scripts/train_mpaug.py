#!/usr/bin/env python
"""MI355X drop-in for the reference's trainer third_party_methods/train_rtpose_light3d_kdh3d_mpaug.py (CR line endings):
same data layout and the command-line arguments that matter, the whole per-batch body on the GPU --
multi-person composition + targets (popnet_amd.targets.mpaug_batch) -> train-mode forward, rtpose_light3d_loss_fgweight,
backward, Nesterov SGD (popnet_amd.train.TrainEngine) -- validation loss per epoch, ReduceLROnPlateau(0.8, patience 5,
cooldown 3) and the best checkpoint saved as `best_pose.pth` with the DataParallel `module.` prefix the evaluation scripts
expect (:300-340).  Data parallel over the GPUs of one node: `--gpus N` (per-replica BatchNorm statistics, gradients averaged).

    python scripts/train_mpaug.py --train-annotations labels_train_*.json --val-annotations labels_test_*.json \
        --image-dir depth_maps --bg-file labels_bg.json --bg-dir bg_maps --seg-dir seg_maps --output-dir out [--epochs 200]

Not reproduced: the random augmentation chain (Rotate / RenderDepth / Crop of data_augmentation_2d3d.py) -- items go through
the evaluation transform (Cvt2ndarray + Resize), which is what popnet_amd.targets mirrors bit for bit.
"""
import argparse
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Plateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor, patience, threshold (rel), cooldown) on a plain number."""

    def __init__(self, factor=0.8, patience=5, threshold=1e-4, cooldown=3):
        self.factor, self.patience, self.threshold, self.cooldown = factor, patience, threshold, cooldown
        self.best, self.bad, self.cool = float("inf"), 0, 0

    def step(self, metric, lr):
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.cool > 0:
            self.cool -= 1
            self.bad = 0
        if self.bad > self.patience:
            self.cool, self.bad = self.cooldown, 0
            return lr * self.factor
        return lr


def eval_loss(module, batch):
    """Validation loss as the reference's validate() (:213-262): eval-mode forward, the same six terms, no update."""
    img, heat, paf, z, fg = batch
    with torch.no_grad():
        _, saved = module(img)
        w = 0.1 + 0.9 * fg
        total = 0.0
        for j in range(2):
            total = total + ((saved[3 * j] - paf) ** 2).mean() + ((saved[3 * j + 1] - heat) ** 2).mean() + (((saved[3 * j + 2] - z) ** 2) * w).mean()
    return float(total)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--train-annotations", nargs="+", required=True)
    ap.add_argument("--val-annotations", nargs="+", default=None)
    ap.add_argument("--image-dir", required=True)
    ap.add_argument("--bg-file", required=True)
    ap.add_argument("--bg-dir", required=True)
    ap.add_argument("--seg-dir", required=True)
    ap.add_argument("--output-dir", default="./trained_model/rtpose_light3d_kdh3d_mpaug")
    ap.add_argument("--batch-size", type=int, default=30)
    ap.add_argument("--lr", "--learning-rate", type=float, default=1.0)
    ap.add_argument("--momentum", type=float, default=0.9)
    ap.add_argument("--weight-decay", "--wd", type=float, default=0.0)
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--square-edge", type=int, default=224)
    ap.add_argument("--z-radius", type=int, default=2)
    ap.add_argument("--print-freq", type=int, default=20)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--weight", default=None, help="start from this checkpoint instead of the module's initial state")
    ap.add_argument("--gpus", type=int, default=1)
    args = ap.parse_args(argv)

    import popnet_amd  # noqa: F401
    from popnet_amd import launch
    if args.gpus > 1 and not launch.under_torchrun():
        sys.exit(launch.relaunch(os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), args.gpus))
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev)
    if args.seed is not None:
        random.seed(args.seed)                       # every rank shuffles the id lists identically, then takes its share of each batch
        torch.manual_seed(args.seed)

    from popnet_amd import targets
    from popnet_amd.network.rtpose_light3d import rtpose_light3d
    from popnet_amd.train import LOSS_NAMES, TrainEngine

    train_set = targets.MPAugTrainSet(args.image_dir, args.train_annotations, args.bg_file, args.bg_dir, args.seg_dir, device=dev)
    val_set = targets.MPAugTrainSet(args.image_dir, args.val_annotations, args.bg_file, args.bg_dir, args.seg_dir, device=dev, shuffle=False) if args.val_annotations else None
    module = rtpose_light3d(15, 14, 2, input_dim=1)
    if args.weight:
        module.load_state_dict(torch.load(args.weight, map_location="cpu"))
    eng = TrainEngine.from_module(module, device=dev, lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay, world_size=world)
    module = module.to(dev).eval()
    module.precision = "fp32"
    plateau, best, captured_lr = Plateau(), float("inf"), None
    os.makedirs(args.output_dir, exist_ok=True)
    if args.batch_size % world:
        # DataParallel would give the first replicas one sample more; a silent `//` would instead drop batch_size % world
        # samples of every batch and change the effective batch size
        raise SystemExit("--batch-size %d is not divisible by the %d replicas" % (args.batch_size, world))
    per_rank = args.batch_size // world               # DataParallel splits the batch over the replicas
    for epoch in range(args.epochs):
        order = list(range(len(train_set)))
        random.shuffle(order)
        n_batches = len(order) // args.batch_size     # drop_last=True (:122)
        t0, run = time.time(), 0.0
        for i in range(n_batches):
            mine = order[i * args.batch_size + rank * per_rank: i * args.batch_size + (rank + 1) * per_rank]
            batch = [t.contiguous() for t in targets.mpaug_batch(*train_set.batch(mine), input_size=args.square_edge, z_radius=args.z_radius)]
            if eng.steps >= 1 and captured_lr != eng.lr:          # the step as one hipGraph (re-captured when the plateau rule moved lr)
                eng.capture(*batch, warmup_steps=0)
                captured_lr = eng.lr
            terms = eng.step(*batch)
            if i % args.print_freq == 0 and rank == 0:
                tl = terms.cpu().tolist()
                run = sum(tl)
                print("Epoch: [%d][%d/%d]\tLoss %.4f\t%s\t(%.1f frames/s)" % (epoch, i, n_batches, run, "  ".join("%s %.4f" % (n, v) for n, v in zip(LOSS_NAMES, tl)),
                                                                              (i + 1) * args.batch_size / max(time.time() - t0, 1e-9)))
        val = run
        if val_set is None and world > 1:
            # without a validation split the plateau rule runs on the last printed training loss: every rank must take the
            # SAME lr decision (it is baked into each rank's captured graph), so rank 0's figure is broadcast
            v = torch.tensor([run], device=dev, dtype=torch.float64)
            dist.broadcast(v, src=0)
            val = float(v[0])
        if val_set is not None:
            module.load_state_dict(eng.state_dict())
            module.eval()
            vals = []
            for s in range(0, len(val_set) - per_rank * world + 1, per_rank * world):
                idx = list(range(s + rank * per_rank, s + (rank + 1) * per_rank))
                vals.append(eval_loss(module, [t.contiguous() for t in targets.mpaug_batch(*val_set.batch(idx), input_size=args.square_edge, z_radius=args.z_radius)]))
            v = torch.tensor([sum(vals), float(len(vals))], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(v)
            val = float(v[0] / max(float(v[1]), 1.0))
        eng.lr = plateau.step(val, eng.lr)
        if rank == 0:
            print("Epoch %d: val loss %.5f  lr %.4g" % (epoch, val, eng.lr))
            if val < best:
                best = val
                torch.save(eng.state_dict(prefix="module."), os.path.join(args.output_dir, "best_pose.pth"))
    if world > 1:
        dist.destroy_process_group()
    return best


if __name__ == "__main__":
    main()
