#!/usr/bin/env python
"""Headline benchmark: end-to-end depth-frames/s (480x640 f16 frames -> 3D joints) on N MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One step = one batch of 32 synthetic 480(w)x640(h) depth frames, already resident in HBM, through
the whole hot path on one GPU: pn_preprocess -> rtpose_light3d forward (bf16 MFMA) -> pose parsing
-> records copied to pinned host memory.  This is BASELINE.json configs[1].  Weights are seeded
random with the heat head calibrated to a realistic peak density (pipeline.calibrate_heads); data is
synthetic (no dataset / checkpoint ships with the reference).

Multi-GPU: weak scaling, every rank runs its own 32-frame batches (frames are independent, no
collective on the data path) and ONE all-gather of the pose records over RCCL/xGMI closes the timed
region.  Prints exactly one JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = int(os.environ.get("POPNET_BENCH_BATCH", "32"))     # 32 = BASELINE configs[1]; the override is for experiments only
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(engine, depth_host, frames_sample=32, reps=6):
    """The oracle (a port of the reference's CPU path: numpy/cv2-restatement pre-proc, torch fp32 CPU
    forward, NumPy parse) timed on a bounded sample of the same workload.  torch's intra-op pool is
    tried at 32 threads and at every host core (small convolutions do not scale to hundreds of
    threads); the faster setting is the one reported, with its thread count."""
    from oracle import nets as onets, parse_paf as oparse, preproc as opre
    cores = os.cpu_count() or 1
    sd = {k: v.detach().cpu().clone() for k, v in engine.model.state_dict().items()}
    d = depth_host[:frames_sample]
    t0 = time.time()
    x = torch.from_numpy(opre.preprocess_batch(d))
    t_pre = time.time() - t0
    best = None
    for threads in sorted({min(32, cores), cores}):
        torch.set_num_threads(threads)
        onets.rtpose_light3d_forward(x[:1], sd)                      # warm the thread pool / allocator
        t1 = time.time()
        for _ in range(reps):
            paf, heat, z = onets.rtpose_light3d_forward(x, sd)
        t_fwd = (time.time() - t1) / reps
        if best is None or t_fwd < best[0]:
            best = (t_fwd, threads, paf, heat, z)
    t_fwd, threads, paf, heat, z = best
    t2 = time.time()
    paf, heat, z = (a.numpy().transpose(0, 2, 3, 1) for a in (paf, heat, z))
    for _ in range(reps):
        for b in range(len(d)):
            oparse.frame_to_records(heat[b].copy(), paf[b].copy(), z[b].copy())
    t_parse = (time.time() - t2) / reps
    total = t_pre + t_fwd + t_parse
    return {"value": round(len(d) / total, 3), "unit": "frames/s", "cores": threads, "kind": "port", "host_cores": cores,
            "sample": "%d of the step's 32 frames (forward and parse repeated 6x, means reported), fp32: preproc %.2fs (1 thread) + torch-CPU forward %.2fs (%d threads, best of {32, all %d cores}) + "
                      "numpy parse %.2fs (1 thread)" % (len(d), t_pre, t_fwd, threads, cores, t_parse)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying one hipGraph per step")
    ap.add_argument("--net", default="rtpose", choices=["rtpose", "yolo"],
                    help="rtpose = rtpose_light3d + PAF parsing (the headline workload); yolo = YoloPoseNet + box decode (SURVEY 8a rows 7, 12)")
    ap.add_argument("--h2d", action="store_true", help="hand every batch over from pinned host memory (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--pipeline", type=int, default=3, help="batches in flight per GPU (engines on separate HIP streams)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev)

    import popnet_amd  # noqa: F401
    from popnet_amd import _lib, synth
    from popnet_amd.pipeline import PoseEngine, YoloEngine
    Engine = PoseEngine if args.net == "rtpose" else YoloEngine
    REC = _lib.POSE_FRAME_DTYPE if args.net == "rtpose" else _lib.YOLO_FRAME_DTYPE

    # popnet_amd.pipeline.StreamingEngine: PIPE batches in flight (per slot: engine = activations + parse workspace,
    # static input, record buffers, HIP stream, ONE hipGraph of the whole step).  Batch k runs on slot k % PIPE, so the
    # latency-bound tail of one batch (head convs, pose parsing, the record D2H copy) overlaps with the convolutions of
    # the next.  Every batch still runs the whole path; only its latency, not the work, is hidden.
    from popnet_amd.pipeline import StreamingEngine
    PIPE = max(1, args.pipeline)
    se = StreamingEngine(Engine, depth=PIPE, graph=not args.no_graph, precision=args.precision, device=dev, max_batch=BATCH)
    engines, streams = se.engines, se.streams
    engine = engines[0]
    depth_host = synth.synth_depth(BATCH, 640, 480, seed=1234 + rank)
    for sl in range(PIPE):
        se.input(sl).copy_(torch.from_numpy(depth_host))          # inputs resident in HBM before the timed region
    torch.cuda.synchronize()
    K, W = args.steps, args.warmup
    item = REC.itemsize
    frames_dev = torch.empty((K, BATCH, item), device=dev, dtype=torch.uint8)
    # what crosses xGMI in the closing all-gather: compact pn_pose_wire records (6.2 KB instead of 33 KB per frame) for
    # the PAF path, the (already small) pn_yolo_frame records for the yolo path
    witem = _lib.POSE_WIRE_DTYPE.itemsize if args.net == "rtpose" else item
    wire_dev = torch.empty((K, BATCH, witem), device=dev, dtype=torch.uint8) if world > 1 else None
    gathered = torch.empty((world * K * BATCH, witem), device=dev, dtype=torch.uint8) if world > 1 else None

    pinned = torch.from_numpy(depth_host).pin_memory() if args.h2d else None

    def step(k):
        if pinned is not None:                                         # PCIe-inclusive variant: 19.7 MB per batch on the slot's stream
            sl = se._tickets % PIPE
            with torch.cuda.stream(se.stream(sl)):
                se.input(sl).copy_(pinned, non_blocking=True)
        t = se.submit()
        with torch.cuda.stream(se.stream(t)):
            frames_dev[k].copy_(se.records(t), non_blocking=True)          # keep every step's records (rank-0 statistics)
            if world > 1:                                                  # ... and their wire form for the final gather
                if args.net == "rtpose":
                    engines[t % PIPE].pack(se.records(t), wire_dev[k])
                else:
                    wire_dev[k].copy_(se.records(t), non_blocking=True)

    def join():
        se.join()

    se.capture()
    for i in range(max(W, 2 * PIPE)):
        step(i % K)
    join()
    if world > 1:
        dist.all_gather_into_tensor(gathered, wire_dev.view(K * BATCH, witem))
    torch.cuda.synchronize()

    # ---- timed region: K steps, barrier + device sync on both sides, max over ranks ----
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        step(k)
    join()
    if world > 1:
        dist.all_gather_into_tensor(gathered, wire_dev.view(K * BATCH, witem))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    frames_host = frames_dev.cpu()

    # ---- roofline pass: the same K steps again, eager, every conv launch bracketed by HIP events on
    # the launch stream (event records inside the throughput pass would perturb it) ----
    L = _lib.lib()
    torch.cuda.synchronize()
    L.pn_net_profile_begin(engine.net)                           # only slot 0's net records events: its launches alone
    t1 = time.perf_counter()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
    with torch.cuda.stream(streams[0]):
        for k in range(K):
            nb = engine.preprocess(se.input(0))
            engine.forward(nb)
            ev[k][0].record()
            engine.parse(nb, se.records(0))                  # post-processing kernels of this step, bracketed on their stream
            ev[k][1].record()
            se.host_records(0).copy_(se.records(0), non_blocking=True)
            ev[k][2].record()
    torch.cuda.synchronize()
    elapsed_profiled = time.perf_counter() - t1
    post_ms = sum(ev[k][0].elapsed_time(ev[k][1]) for k in range(K)) / K
    d2h_ms = sum(ev[k][1].elapsed_time(ev[k][2]) for k in range(K)) / K
    conv_ms, other_ms, conv_flops = C.c_double(), C.c_double(), C.c_double()
    conv_n, other_n = C.c_int64(), C.c_int64()
    engine.ctx.check(L.pn_net_profile_end(engine.net, C.byref(conv_ms), C.byref(conv_n), C.byref(conv_flops),
                                          C.byref(other_ms), C.byref(other_n)), "pn_net_profile_end")
    kernels = []                                                # per instantiation, dominant first
    for r in range(16):
        name = C.create_string_buffer(96)
        kms, kfl, kn = C.c_double(), C.c_double(), C.c_int64()
        if L.pn_net_profile_kernel(engine.net, r, name, 96, C.byref(kms), C.byref(kn), C.byref(kfl)) != 0:
            break
        kernels.append({"kernel": name.value.decode(), "launches_per_step": kn.value / K, "us_per_step": round(kms.value * 1e3 / K, 2),
                        "avg_launch_us": round(kms.value * 1e3 / max(kn.value, 1), 3),
                        "tflops": round(kfl.value / (kms.value * 1e-3) / 1e12, 2) if kms.value > 0 else 0.0,
                        "flops_per_launch": round(kfl.value / max(kn.value, 1), 1)})

    if rank == 0:
        recs = frames_host.numpy().view(REC).reshape(K, BATCH)
        raw = frames_host.numpy().reshape(K, -1)
        same = bool(all(np.array_equal(raw[0], raw[k]) for k in range(1, K)))     # same input every step -> same records from every engine
        total_frames = world * K * BATCH
        achieved = conv_flops.value / (conv_ms.value * 1e-3) / 1e12 if conv_ms.value > 0 else 0.0
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else 157.3
        # network outputs read + records written per frame
        post_bytes = BATCH * ((185024 if args.net == "rtpose" else 100 * 14 * 14 * 4) + item)
        dom = kernels[0] if kernels else {"kernel": "none", "us_per_step": 0.0, "avg_launch_us": 0.0, "tflops": 0.0, "flops_per_launch": 0.0, "launches_per_step": 0}
        traffic = None                                          # HBM bytes per launch of the dominant kernel (separate rocprofv3 --pmc passes)
        pmc = os.path.join(ROOT, "profiles", "conv_hbm_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("by_kernel", {}).get(dom["kernel"], {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "depth-frames/sec end-to-end (480x640)", "value": round(total_frames / elapsed, 2),
            "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4), "launch_mode": ("eager" if args.no_graph else "hipGraph replay (one graph per step)") + ", %d batches in flight on separate HIP streams" % PIPE, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic" + (", handed over from pinned host memory every step (PCIe-inclusive)" if args.h2d else ", resident in HBM"),
            "config": {"workload": "BASELINE configs[1]: batch=32 synthetic 480x640 f16 depth frames per GPU per step, "
                                   "resize->224^2, " + ("rtpose_light3d forward + PAF pose parsing" if args.net == "rtpose" else
                                                        "YoloPoseNet forward + box decode / NMS / skeleton read-out (secondary network of the path)") + ", records D2H",
                       "frames_per_step_per_gpu": BATCH, "input": "480x640 f16", "network_input": "224x224",
                       "weights": "seeded random, " + ("heat head calibrated (pipeline.calibrate_heads)" if args.net == "rtpose" else "confidence filters calibrated (pipeline.calibrate_yolo_conf)"),
                       "parallelism": "frames sharded x%d, one all-gather of records" % world},
            "roofline": {"bound": "mfma", "kernel": dom["kernel"] + " (dominant convolution instantiation: %.0f of %.0f conv us/step)" % (dom["us_per_step"], conv_ms.value * 1e3 / K),
                         "achieved": dom["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": round(dom["tflops"] / peak, 4),
                         "traffic": traffic, "avg_launch_us": dom["avg_launch_us"], "flops_per_launch": dom["flops_per_launch"],
                         "launches_per_step": dom["launches_per_step"],
                         "conv_stack": {"achieved": round(achieved, 2), "frac": round(achieved / peak, 4), "launches_per_step": conv_n.value // max(K, 1),
                                        "ms_per_step": round(conv_ms.value / K, 4), "tflops_inside_timed_region": round(conv_flops.value / elapsed / 1e12 / world, 2),
                                        "stem_pool_ms_per_step": round(other_ms.value / K, 4), "by_kernel": kernels},
                         "measured": "HIP events around every conv launch on the launch stream, the same %d steps re-run eagerly on one engine right after the timed region (%.4f ms/step with events)" % (K, elapsed_profiled / K * 1e3)},
            "postproc": {"bound": "hbm", "kernels": "pose parsing (NMS + refine, limb scoring + matching, assembly + read-out)" if args.net == "rtpose" else "box decode + NMS + skeleton read-out",
                         "algorithmic_bytes_per_step": post_bytes, "us_per_step": round(post_ms * 1e3, 2),
                         "achieved": round(post_bytes / (post_ms * 1e-3) / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(post_bytes / (post_ms * 1e-3) / 8e12, 5), "record_d2h_us_per_step": round(d2h_ms * 1e3, 2),
                         "note": "latency-bound: 32 small frames per step; hidden behind the next batch by the StreamingEngine"},
            "frame_stats": {"mean_peaks": round(float(recs['n_peaks' if args.net == "rtpose" else 'n_candidates'].mean()), 2),
                            "mean_persons": round(float(recs['n_persons' if args.net == "rtpose" else 'n_det'].mean()), 3),
                            "overflow_frames": int((recs['status'] != 0).sum()),
                            "records_identical_across_steps_and_engines": same},
        }
        if world == 1 and not args.no_cpu_baseline and args.net == "rtpose":
            out["cpu_baseline"] = cpu_baseline(engine, depth_host)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
