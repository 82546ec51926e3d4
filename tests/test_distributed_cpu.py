"""CPU, world_size 2, gloo: the multi-GPU result path (frame sharding + ONE all-gather of fixed-size
records + de-interleave to global frame order) is correct by construction."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, n_frames, port, ret):
    sys.path.insert(0, ROOT)
    import popnet_amd  # noqa: F401
    from popnet_amd.pipeline import gather_records, shard_indices
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    item = 64
    mine = shard_indices(n_frames, rank, world)
    # a "record" whose bytes encode the global frame index it came from
    local = torch.zeros((len(mine), item), dtype=torch.uint8)
    for s, i in enumerate(mine):
        local[s] = torch.tensor([(i >> (8 * (k % 2))) & 255 for k in range(item)], dtype=torch.uint8)
    out = gather_records(local, n_frames, rank, world)
    ok = out.shape == (n_frames, item)
    for i in range(n_frames):
        ok &= int(out[i, 0]) == (i & 255) and int(out[i, 1]) == ((i >> 8) & 255)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_gather_records_world2_gloo():
    for n_frames in (7, 8, 300):
        mgr = mp.Manager()
        ret = mgr.dict()
        port = 29500 + (os.getpid() % 1000) + n_frames % 7
        mp.spawn(_worker, args=(2, n_frames, port, ret), nprocs=2, join=True)
        assert ret[0] and ret[1]


class _FakeEngine:
    """Stands in for PoseEngine on the CPU: a 'record' of a frame is 64 bytes derived from the frame's content, so the
    sharded sweep can be checked end to end (dataset listing -> shard -> batches -> gather -> global order) without a GPU."""
    def __init__(self, max_batch):
        self.device, self.max_batch = torch.device("cpu"), max_batch
        self.frames = torch.empty((max_batch, 64), dtype=torch.uint8)

    def predict(self, depth):
        tag = depth.float().reshape(depth.shape[0], -1)[:, 0].round().to(torch.int64)     # the frame stores its own index
        out = torch.zeros((depth.shape[0], 64), dtype=torch.uint8)
        out[:, 0] = (tag & 255).to(torch.uint8)
        out[:, 1] = ((tag >> 8) & 255).to(torch.uint8)
        return out


def _sweep_worker(rank, world, root_dir, n_frames, port, ret):
    sys.path.insert(0, ROOT)
    import popnet_amd  # noqa: F401
    from popnet_amd import _lib, dataset
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fr = dataset.MP3DHPFrames(root_dir, os.path.join(root_dir, "labels.json"))
    item = 64
    # run_sweep reinterprets the gathered bytes with the record dtype: patch the lookup to a 64-byte dtype for the fake
    old = _lib.YOLO_FRAME_DTYPE
    _lib.YOLO_FRAME_DTYPE = np.dtype([("b", np.uint8, (item,))])
    try:
        recs = dataset.run_sweep(_FakeEngine(3), fr, batch_size=3, rank=rank, world=world)
    finally:
        _lib.YOLO_FRAME_DTYPE = old
    ok = len(recs) == n_frames
    for i in range(n_frames):
        ok &= int(recs[i]["b"][0]) == (i & 255) and int(recs[i]["b"][1]) == (i >> 8)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world,n_frames", [(2, 11), (8, 45)])
def test_sharded_dataset_sweep_world2_gloo(tmp_path, world, n_frames):
    """dataset.run_sweep (what scripts/evaluate_mpreal.py drives) with two and with EIGHT ranks -- the node size of BASELINE configs[2]:
    every rank ends up with all records in label-file order (ragged last batches, shards of unequal length)."""
    import json
    labels = {"intrinsics": {"fx": 1, "fy": 1, "cx": 0, "cy": 0}}
    for i in range(n_frames):
        name = "z%02d.npy" % (n_frames - i)                      # names NOT in sorted order: dict order must win
        np.save(tmp_path / name, np.full((4, 4), float(i), np.float32))
        labels[name] = []
    json.dump(labels, open(tmp_path / "labels.json", "w"))
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29700 + (os.getpid() % 1000)
    mp.spawn(_sweep_worker, args=(world, str(tmp_path), n_frames, port + world, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)) and len(ret) == world


def test_bench_self_launches_ranks_from_a_bare_shell():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how the driver starts it) must start two fresh
    ranks itself before touching a GPU and relay rank 0's single JSON line; the dry-run flag swaps the GPU work for a
    gloo all-reduce so the launcher path runs on the CPU box."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--launcher-dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["launcher_dry_run"] and out["n_gpus"] == 2 and out["rank_sum"] == 3
    # bench.py's collective check (gather_records + the 22 MB gradient all-reduce) entered by BOTH ranks, as main() does
    assert out["dist"]["world_size"] == 2 and out["dist"]["gather_records_ok"] and out["dist"]["flat_gradient_allreduce_ok"], out["dist"]


def test_torchrun_command_shape():
    import popnet_amd  # noqa: F401
    from popnet_amd import launch
    cmd = launch.torchrun_command("bench.py", ["--gpus", "4"], 4, port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-3:] == ["bench.py", "--gpus", "4"]


def _train_dp_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import popnet_amd  # noqa: F401
    from popnet_amd.train import TrainEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(5525814, generator=g)                       # this replica's gradient of its half batch (flat buffer, as TrainEngine holds it)
    mine = flat.clone()
    scale = TrainEngine.reduce_flat_gradient(flat, world)          # what TrainEngine.apply does before pn_sgd_nesterov(grad_scale=scale)
    other = torch.randn(5525814, generator=torch.Generator().manual_seed(100 + (1 - rank)))
    ret[rank] = bool(scale == 0.5 and torch.allclose(flat * scale, (mine + other) / 2, rtol=0, atol=1e-6))
    dist.destroy_process_group()


def test_training_gradient_exchange_world2_gloo():
    """Data-parallel training: every replica ends up with the MEAN of the replicas' gradients -- one all-reduce of the
    flat 5 525 814-float buffer, then the 1 / world scale that pn_sgd_nesterov applies.  The function under test is the one
    TrainEngine.apply calls (RCCL on the GPUs, gloo here)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_train_dp_worker, args=(2, 29500 + (os.getpid() % 1000) + 11, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def test_bench_world8_dry_run_and_rank0_only_legs():
    """Readiness for the driver's 8-GPU SCALE pass, on the CPU: `bench.py --gpus 8` from a bare shell starts eight
    ranks, every rank enters the collectives of dist_check (the record all-gather and the 22 MB gradient all-reduce, gloo here, RCCL
    there), rank 0 prints ONE line; and the plan of secondary legs is empty for every rank of a multi-rank run -- no child process, CPU
    baseline or extra engine next to the other ranks."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--launcher-dry-run"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["launcher_dry_run"] and out["n_gpus"] == 8 and out["rank_sum"] == 36
    assert out["dist"]["world_size"] == 8 and out["dist"]["gather_records_ok"] and out["dist"]["flat_gradient_allreduce_ok"], out["dist"]
    sys.path.insert(0, ROOT)
    import bench
    for world in (2, 4, 8):
        for rank in range(world):
            assert not any(bench.plan_legs(world, rank, "rtpose", "bf16", False, False).values()), (world, rank)
    solo = bench.plan_legs(1, 0, "rtpose", "bf16", False, False)
    assert all(solo.values())
    assert not bench.plan_legs(1, 0, "rtpose", "bf16", True, True)["children"]         # --no-extras (what the child legs themselves run with)


def test_bench_refuses_ablation_switches():
    """A timed region must not be one environment variable away from skipping launches: bench.py exits before
    anything runs when a timing-only / result-changing experiment switch is set, and the shipped library has them compiled out."""
    import subprocess
    for var in ("POPNET_ABLATE_SKIP", "POPNET_X3_BF16_CONVS"):
        env = dict(os.environ, **{var: "pool"})
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launcher-dry-run"], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "refusing to run" in (r.stderr + r.stdout), (var, r.stderr[-500:])
    import popnet_amd  # noqa: F401
    from popnet_amd import _lib
    assert _lib.lib().pn_build_experiments() == 0
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"POPNET_ABLATE_SKIP" not in blob and b"POPNET_X3_BF16_CONVS" not in blob
