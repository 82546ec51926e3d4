"""MP-3DHP ingest and the result schema: the data formats on either side of the hot path.

Upstream of the path -- what the reference's test-mode dataset does before the network sees a frame
(tpm/lib/datasets/datasets_kdh3d_rtpose_mpreal.py:181-246 (CR line endings), data_augmentation_2d3d.py:76-89,507-522):
    ``labels.json``   {"intrinsics": {fx, fy, cx, cy}, "<frame id>": [ {"2d_joints": [15][2], "3d_joints": [15][3], ...}, ... ]}
                      frame ids (dict order, "intrinsics" skipped) are file names under the image directory, ``.npy`` included;
    ``<id>.npy``      one depth frame [H, W] in metres (float16 on disk for MP-3DHP).
Here only the file read stays on the host: resize + clamp + normalise run in pn_preprocess on the raw frame.

Downstream -- the ``eval_data.json`` dictionary the evaluation scripts write and main_evaluate_mp_human_3D.py reads
(evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py:398-409, evaluation_yolo_posenet_kdh3d_mpreal.py:255-262):
parallel lists indexed by frame: human_pred_set_2d[f][p][15][2], _3d[f][p][15][3], _part_conf[f][p][15]
(+ _visibility, and the ground truth copied from the labels).

Sweeps shard frames over ranks as i -> rank i % world, keep global frame order and -- unlike the
reference's ``drop_last=True`` loader -- never drop the tail unless asked to.
"""
import json
import os

import numpy as np

from . import _lib
from .config import INTRINSICS, NUM_PARTS


class MP3DHPFrames:
    """Frame list + annotations of one MP-3DHP split (test-mode KDH3D_Keypoints, datasets_kdh3d_rtpose_mpreal.py:181-236)."""

    def __init__(self, img_dir, ann_file):
        self.img_dir = img_dir
        self.anno_dic = json.load(open(ann_file, "r"))
        self.ids = [k for k in self.anno_dic.keys() if k != "intrinsics"]
        self.intrinsics = dict(self.anno_dic.get("intrinsics", INTRINSICS))

    def __len__(self):
        return len(self.ids)

    def load(self, index):
        """Raw frame as stored: float16 / float32 kept (the device kernel widens exactly like the reference's
        ``astype(np.float)``), anything else (float64, integers) narrowed to float32 as Cvt2ndarray does."""
        a = np.load(os.path.join(self.img_dir, self.ids[index]))
        if a.ndim != 2:
            raise _lib.PopnetError("%s: expected one [H, W] depth frame, got shape %s" % (self.ids[index], a.shape))
        return a if a.dtype in (np.float16, np.float32) else a.astype(np.float32)

    def ground_truth(self):
        """(human_gt_set_2d, human_gt_set_3d) in frame order (main_evaluate_mp_human_3D.py:21-41)."""
        g2 = [[p["2d_joints"] for p in self.anno_dic[k]] for k in self.ids]
        g3 = [[p["3d_joints"] for p in self.anno_dic[k]] for k in self.ids]
        return g2, g3

    def batches(self, indices, batch_size, drop_last=False):
        """Yields (index list, host array [b, H, W]); every frame of a batch must have the same shape and dtype."""
        indices = list(indices)
        for s in range(0, len(indices), batch_size):
            chunk = indices[s:s + batch_size]
            if drop_last and len(chunk) < batch_size:
                return
            frames = [self.load(i) for i in chunk]
            if any(f.shape != frames[0].shape or f.dtype != frames[0].dtype for f in frames):
                raise _lib.PopnetError("frames %s..%s differ in shape or dtype" % (self.ids[chunk[0]], self.ids[chunk[-1]]))
            yield chunk, np.stack(frames)


def pose_records_to_lists(recs, overflow=None):
    """pn_pose_frame records -> the per-frame entries of human_pred_set_{2d,3d,visibility,part_conf} (float64 lists,
    [-1, -1] / Z = -1 for joints a person does not have, exactly what the evaluation script appends).
    overflow: {frame index: result of utils.paf_to_pose.parse_paf_unbounded} for the frames whose record carries an overflow
    status (PoseEngine.predict_lists fills it); without it such a frame raises -- it is never truncated silently."""
    out = {"human_pred_set_2d": [], "human_pred_set_3d": [], "human_pred_set_visibility": [], "human_pred_set_part_conf": []}
    for i, fr in enumerate(recs):
        if int(fr["status"]):
            if overflow is None or i not in overflow:
                raise _lib.PopnetError("pose record overflow (status=%d): more than %d peaks per joint map or %d persons -- re-parse the frame "
                                       "with utils.paf_to_pose.parse_paf_unbounded / PoseEngine.predict_lists" % (int(fr["status"]), _lib.PN_MAX_PEAKS_PER_JOINT, _lib.PN_MAX_PERSONS))
            r = overflow[i]
            out["human_pred_set_2d"].append(r["joints_2d"].tolist())
            out["human_pred_set_3d"].append(r["joints_3d"].tolist())
            out["human_pred_set_visibility"].append((r["person_joint"] >= 0).astype(int).tolist())
            out["human_pred_set_part_conf"].append(r["part_conf"].tolist())
            continue
        n = int(fr["n_persons"])
        out["human_pred_set_2d"].append(np.asarray(fr["joints_2d"][:n], dtype=np.float64).tolist())
        out["human_pred_set_3d"].append(np.asarray(fr["joints_3d"][:n], dtype=np.float64).tolist())
        out["human_pred_set_visibility"].append((np.asarray(fr["person_joint"][:n]) >= 0).astype(int).tolist())
        out["human_pred_set_part_conf"].append(np.asarray(fr["part_conf"][:n], dtype=np.float64).tolist())
    return out


def yolo_records_to_lists(recs):
    """pn_yolo_frame records -> human_pred_set_{2d,3d,part_conf} of evaluation_yolo_posenet_kdh3d_mpreal.py:182-247."""
    out = {"human_pred_set_2d": [], "human_pred_set_3d": [], "human_pred_set_part_conf": []}
    for fr in recs:
        if int(fr["status"]):
            raise _lib.PopnetError("yolo record overflow (status=%d)" % int(fr["status"]))
        n = int(fr["n_det"])
        out["human_pred_set_2d"].append(np.asarray(fr["joints_2d"][:n]).tolist())
        out["human_pred_set_3d"].append(np.asarray(fr["joints_3d"][:n]).tolist())
        out["human_pred_set_part_conf"].append(np.repeat(np.asarray(fr["bbox"][:n, 4:5], dtype=np.float64), NUM_PARTS, axis=1).tolist())
    return out


def run_sweep(engine, frames, batch_size=32, rank=0, world=1, drop_last=False, group=None):
    """Runs `engine` (PoseEngine / YoloEngine) over this rank's shard of `frames` and returns the records of ALL frames
    in global order as a structured numpy array (one all-gather of fixed-size records when world > 1)."""
    import torch
    from .pipeline import gather_records, shard_indices
    n = len(frames)
    if drop_last:
        n -= n % (batch_size * world)
    mine = shard_indices(n, rank, world)
    item = engine.frames.shape[1]
    local = torch.empty((len(mine), item), dtype=torch.uint8, device=engine.device)
    done = 0
    for chunk, host in frames.batches(mine, min(batch_size, engine.max_batch)):
        dev = torch.from_numpy(host).to(engine.device, non_blocking=True)
        local[done:done + len(chunk)].copy_(engine.predict(dev))
        done += len(chunk)
    if world > 1:
        local = gather_records(local, n, rank, world, group)
    dtype = _lib.POSE_FRAME_DTYPE if item == _lib.POSE_FRAME_DTYPE.itemsize else _lib.YOLO_FRAME_DTYPE
    return local.cpu().numpy().view(dtype).reshape(-1)


def run_sweep_streaming(se, frames, batch_size=32, rank=0, world=1, drop_last=False, group=None, gather=True):
    """Same result as run_sweep, through a pipeline.StreamingEngine: batch k+1 is read from disk, pinned and copied to
    the device (on its slot's stream) while batches k, k-1 are still being computed; a slot's records are collected right
    before the slot is reused.  gather=False returns this rank's shard only (device uint8 [n_local, item], frames
    rank, rank + world, ...) without touching torch.distributed."""
    import torch
    from .pipeline import gather_records, shard_indices
    n = len(frames)
    if drop_last:
        n -= n % (batch_size * world)
    mine = shard_indices(n, rank, world)
    if getattr(se, "wire", False):
        raise _lib.PopnetError("run_sweep_streaming collects the full records: build the StreamingEngine with wire=False")
    item = se.recs[0].shape[1]
    bs = min(batch_size, se.max_batch)
    local = torch.empty((len(mine), item), dtype=torch.uint8, device=se.device)
    pending = {}                                   # slot -> (ticket, offset, count)
    pinned = [None] * se.depth

    def collect(slot):
        t, off, cnt = pending.pop(slot)
        se.wait(t)
        with torch.cuda.stream(se.stream(slot)):
            local[off:off + cnt].copy_(se.records(t)[:cnt], non_blocking=True)

    done = 0
    for chunk, host in frames.batches(mine, bs):
        slot = se._tickets % se.depth
        if slot in pending:
            collect(slot)
        buf = se.input(slot)
        if tuple(host.shape[1:]) != tuple(buf.shape[1:]) or torch.from_numpy(host[:1]).dtype != buf.dtype:
            raise _lib.PopnetError("frames are %s %s, the StreamingEngine was built for %s %s" % (host.shape[1:], host.dtype, tuple(buf.shape[1:]), buf.dtype))
        if pinned[slot] is None:
            pinned[slot] = torch.empty((se.max_batch,) + tuple(host.shape[1:]), dtype=buf.dtype).pin_memory()
        se.stream(slot).synchronize()              # the previous H2D copy out of this pinned buffer has finished
        pinned[slot][:len(chunk)].copy_(torch.from_numpy(host))
        with torch.cuda.stream(se.stream(slot)):
            buf[:len(chunk)].copy_(pinned[slot][:len(chunk)], non_blocking=True)
            if len(chunk) < se.max_batch:
                buf[len(chunk):].zero_()          # a ragged tail batch runs full-size; its surplus records are dropped
        t = se.submit()
        pending[slot] = (t, done, len(chunk))
        done += len(chunk)
    for slot in list(pending):
        collect(slot)
    se.join()
    torch.cuda.synchronize(se.device)
    if not gather:
        return local
    if world > 1:
        local = gather_records(local, n, rank, world, group)
    dtype = _lib.POSE_FRAME_DTYPE if item == _lib.POSE_FRAME_DTYPE.itemsize else _lib.YOLO_FRAME_DTYPE
    return local.cpu().numpy().view(dtype).reshape(-1)


def eval_data_from_records(recs, frames):
    """The eval_data.json dictionary for a finished sweep (predictions from the records, ground truth from the labels)."""
    data = pose_records_to_lists(recs) if recs.dtype == _lib.POSE_FRAME_DTYPE else yolo_records_to_lists(recs)
    g2, g3 = frames.ground_truth()
    data["human_gt_set_2d"], data["human_gt_set_3d"] = g2[:len(recs)], g3[:len(recs)]
    return data
