"""RCCL on the hardware.

The boxes of this pool have one GPU, so the multi-GPU exchanges can only be shown there at world_size 1 -- but on the real
backend: `bench.py --force-dist` starts ONE fresh child rank under torch.distributed.run (before anything in the parent has
touched the GPU; the pytest process itself is never re-exec'ed), initialises the process group with backend "nccl" (= RCCL
on ROCm), runs the timed region with its all-gather of the records in place, and then the two exchanges of the repository on
device tensors: pipeline.gather_records (north_star's single all-gather of pose results) and
TrainEngine.reduce_flat_gradient on the 22 MB flat gradient buffer (configs[4]'s all-reduce).  The world_size-2 semantics
(interleaving, ragged tails, averaging) are covered on CPU with gloo in tests/test_distributed_cpu.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_process_group_gather_and_allreduce_on_the_gpu(gpu):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "6", "--warmup", "2", "--reps", "1",
                        "--pool", "2", "--no-extras", "--no-cpu-baseline", "--no-h2d"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    d = out["dist"]
    assert d["backend"] == "nccl" and d["world_size"] == 1
    assert d["gather_records_ok"] is True and d["flat_gradient_allreduce_ok"] is True
    assert any("rccl" in lib or "nccl" in lib for lib in d["collective_library_loaded"]), d
    # the headline figure of the forced-distributed run is a normal one-GPU figure (its region contains the all-gather)
    assert out["n_gpus"] == 1 and out["value"] > 1000 and out["frame_stats"]["same_batch_same_records_across_steps_slots_and_input_modes"]


def test_one_gpu_wire_output_equals_the_full_records_to_float32(gpu):
    """What `--gpus N` gathers is the compact pn_pose_wire record: joints, 3D positions and confidences as FLOAT32 where the
    one-GPU pn_pose_frame record (and the reference's result schema) holds float64.  Stated and bounded here: the wire values
    are exactly the float32 roundings of the full record's values -- i.e. the gathered result differs from the one-GPU result
    by at most 6e-8 relative (well inside north_star's 1e-3 m), never in assignment, visibility or person order."""
    from popnet_amd import _lib, synth
    from popnet_amd.pipeline import PoseEngine, records_to_numpy
    eng = PoseEngine(precision="fp32", device=gpu, max_batch=16)
    depth = torch.from_numpy(synth.synth_depth(16, 640, 480, seed=31)).to(gpu)
    wire_dev = torch.zeros((16, _lib.POSE_WIRE_DTYPE.itemsize), device=gpu, dtype=torch.uint8)
    full = records_to_numpy(eng.predict(depth, None, wire_dev))
    wire = wire_dev.cpu().numpy().view(_lib.POSE_WIRE_DTYPE).reshape(-1)
    assert wire["vals"].dtype == np.float32 and full["joints_3d"].dtype == np.float64
    seen = 0
    for w, f in zip(wire, full):
        n = int(f["n_persons"])
        assert int(w["n_persons"]) == n and int(w["status"]) == 0
        assert np.array_equal(w["person_joint"][:n], f["person_joint"][:n].astype(np.int16))
        ref = np.concatenate([f["joints_2d"][:n], f["joints_3d"][:n], f["part_conf"][:n][..., None]], axis=-1)
        got = w["vals"][:n].astype(np.float64)
        assert np.array_equal(w["vals"][:n], ref.astype(np.float32))
        assert np.all(np.abs(got - ref) <= 1e-6 * np.maximum(np.abs(ref), 1e-30))
        seen += n
    assert seen > 0
