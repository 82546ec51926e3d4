"""One-process-per-GPU self-launch for the entry scripts (bench.py, scripts/evaluate_mpreal.py).

The reference pins every script to one GPU (CUDA_VISIBLE_DEVICES, 26 occurrences, e.g.
tpm/evaluate/evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py:42,142); here `--gpus N` from a bare shell
starts N fresh ranks under torch.distributed.run (one per GPU, RCCL over xGMI) and relays their output.

Nothing in this module touches the GPU: the parent only spawns children and exits with their return code
(a process that has initialised HIP must never be re-exec'ed on this pool).
"""
import os
import socket
import subprocess
import sys


def under_torchrun():
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def torchrun_command(script, argv, nproc, port=None):
    port = free_port() if port is None else port
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)


def relaunch(script, argv, nproc, timeout=None):
    """Runs `script argv` as `nproc` ranks under torch.distributed.run in a CHILD process, relays stdout / stderr and
    returns the children's return code.  Call before anything initialises the GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(torchrun_command(script, argv, nproc), env=env, timeout=timeout)
    return r.returncode
