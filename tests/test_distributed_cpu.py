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
