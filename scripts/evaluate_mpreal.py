#!/usr/bin/env python
"""MI355X drop-in for the reference's MP-3DHP evaluation scripts
(tpm/evaluate/evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py and evaluation_yolo_posenet_kdh3d_mpreal.py):
same command line for the arguments that matter, same ``eval_data.json`` in --output-dir, same four metric blocks
(popnet_amd.metrics), frames sharded over the GPUs of one node when launched with torch.distributed.run.

    python scripts/evaluate_mpreal.py --annotations labels.json --image-dir depth_maps --weight best_pose.pth \
        --output-dir out [--net rtpose|yolo] [--precision fp32|bf16] [--batch-size 32] [--drop-last]
"""
import argparse
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # 3 compute streams + copies: one hardware queue each (see bench.py); before torch loads the runtime

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--annotations", "--val-annotations", dest="annotations", required=True)
    ap.add_argument("--image-dir", "--val-image-dir", dest="image_dir", required=True)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--input-size", type=int, default=224)
    ap.add_argument("--w-org", type=int, default=480)
    ap.add_argument("--h-org", type=int, default=640)
    ap.add_argument("--weight", required=True, help="reference checkpoint (state_dict, 'module.'-prefixed or not)")
    ap.add_argument("--output-dir", required=True)
    ap.add_argument("--net", default="rtpose", choices=["rtpose", "yolo"])
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"], help="fp32 = parity mode (1e-3 m), bf16 = throughput mode")
    ap.add_argument("--drop-last", action="store_true", help="skip the tail like the reference's drop_last=True loader")
    ap.add_argument("--no-metrics", action="store_true")
    ap.add_argument("--pipeline", type=int, default=3, help="batches in flight (StreamingEngine); 1 = plain sequential engine")
    ap.add_argument("--gpus", type=int, default=1, help="GPUs of this node to shard the frames over: from a bare shell the script starts that many ranks itself")
    args = ap.parse_args(argv)

    import popnet_amd  # noqa: F401
    from popnet_amd import launch                                   # touches no GPU
    if args.gpus > 1 and not launch.under_torchrun():
        # become the parent of N fresh ranks before any HIP call; they run this same script with the same arguments
        sys.exit(launch.relaunch(os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev)

    import popnet_amd  # noqa: F401
    from popnet_amd import dataset, metrics
    from popnet_amd.pipeline import PoseEngine, StreamingEngine, YoloEngine

    frames = dataset.MP3DHPFrames(args.image_dir, args.annotations)
    sd = torch.load(args.weight, map_location="cpu")
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    Engine = PoseEngine if args.net == "rtpose" else YoloEngine
    kw = dict(precision=args.precision, state_dict=sd, device=dev, max_batch=args.batch_size, input_size=args.input_size,
              w_org=args.w_org, h_org=args.h_org, intrinsics=frames.intrinsics)
    if args.pipeline > 1 and len(frames):
        f0 = frames.load(0)
        se = StreamingEngine(Engine, depth=args.pipeline, frame_hw=f0.shape, frame_dtype=torch.from_numpy(f0[:1]).dtype, **kw)
        se.capture()
        recs = dataset.run_sweep_streaming(se, frames, args.batch_size, rank, world, args.drop_last)
    else:
        recs = dataset.run_sweep(Engine(**kw), frames, args.batch_size, rank, world, args.drop_last)
    out = None
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        data = dataset.eval_data_from_records(recs, frames)
        path = os.path.join(args.output_dir, "eval_data.json")
        json.dump(data, open(path, "w"), indent=4)
        print("wrote %s (%d frames, %d GPUs, %s, drop_last=%s)" % (path, len(recs), world, args.precision, args.drop_last))
        if not args.no_metrics:
            gt = os.path.join(args.output_dir, "labels_used.json")
            json.dump({k: frames.anno_dic[k] for k in ["intrinsics"] * ("intrinsics" in frames.anno_dic) + frames.ids[:len(recs)]}, open(gt, "w"))
            out = metrics.evaluate_mp_human_3d(gt, path)
    if world > 1:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
