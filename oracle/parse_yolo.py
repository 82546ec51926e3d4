"""Yolo-Pose+ decode + box NMS (ORACLE; test infrastructure -- see oracle/__init__.py).

CPU restatement (NumPy float32, same operation order as the reference's in-place torch
ops) of ``parse_prior_pose``: third_party_methods/lib/utils/prior_pose_align.py:10-168, and
of the per-frame glue of third_party_methods/evaluate/evaluation_yolo_posenet_kdh3d_mpreal.py:
157-217.  Quirks kept on purpose:
  * candidates are anchor-major, then cell (row-major) (:54-77);
  * suppression loop runs over rows 1..n-2 only (:112-115);
  * visibility is an inclusive in-bounds test on [0, w_out-1] x [0, h_out-1] (:160-161).
The reference mutates its input tensor in place; this restatement works on a copy.
"""
import numpy as np

f32 = np.float32


def decode_maps(posemaps, anchors, num_joints, depth_mean, depth_std):
    """prior_pose_align.py:24-52.  posemaps [B, A*(5+3J), h, w] float32 -> [B, A, 5+3J, h*w]."""
    B, _, h, w = posemaps.shape
    A = len(anchors)
    pm = posemaps.astype(np.float32).reshape(B, A, -1, h * w).copy()
    lin_x = np.tile(np.arange(w, dtype=np.float32), h)
    lin_y = np.repeat(np.arange(h, dtype=np.float32), w)
    aw = np.array([a[0] for a in anchors], dtype=np.float32).reshape(1, A, 1)
    ah = np.array([a[1] for a in anchors], dtype=np.float32).reshape(1, A, 1)
    pm[:, :, 0, :] = (pm[:, :, 0, :] + lin_x) / f32(w)
    pm[:, :, 1, :] = (pm[:, :, 1, :] + lin_y) / f32(h)
    pm[:, :, 2, :] = (pm[:, :, 2, :] * aw) / f32(w)
    pm[:, :, 3, :] = (pm[:, :, 3, :] * ah) / f32(h)
    J = num_joints
    aw4 = (aw / f32(2.0)).reshape(1, A, 1, 1)
    ah4 = (ah / f32(2.0)).reshape(1, A, 1, 1)
    pm[:, :, 5:5 + J, :] = (pm[:, :, 5:5 + J, :] * aw4 + lin_x) / f32(w)
    pm[:, :, 5 + J:5 + 2 * J, :] = (pm[:, :, 5 + J:5 + 2 * J, :] * ah4 + lin_y) / f32(h)
    pm[:, :, 5 + 2 * J:5 + 3 * J, :] = pm[:, :, 5 + 2 * J:5 + 3 * J, :] * f32(depth_std) + f32(depth_mean)
    return pm


def box_nms_keep(boxes, nms_threshold):
    """prior_pose_align.py:84-119 for one image.  boxes [n, 5+3J] (candidate order).
    Returns (order, keep_mask_in_sorted_order)."""
    a = boxes[:, :2]
    b = boxes[:, 2:4]
    bb = np.concatenate([a - b / f32(2), a + b / f32(2)], 1).astype(np.float32)
    scores = boxes[:, 4]
    order = np.argsort(-scores, kind='stable')
    x1, y1, x2, y2 = [bb[order][:, i:i + 1] for i in range(4)]
    dx = np.clip(np.minimum(x2, x2.T) - np.maximum(x1, x1.T), 0, None).astype(np.float32)
    dy = np.clip(np.minimum(y2, y2.T) - np.maximum(y1, y1.T), 0, None).astype(np.float32)
    inter = dx * dy
    areas = (x2 - x1) * (y2 - y1)
    unions = (areas + areas.T) - inter
    with np.errstate(divide='ignore', invalid='ignore'):
        ious = inter / unions
    conflicting = np.triu((ious > f32(nms_threshold)).astype(np.int32), 1)
    keep = conflicting.sum(0).astype(np.int32)
    for i in range(1, len(keep) - 1):
        if keep[i] > 0:
            keep -= conflicting[i]
    return order, keep == 0


def parse_prior_pose(posemaps, anchors, num_joints, w_out, h_out, depth_mean, depth_std,
                     conf_threshold=0.35, nms_threshold=0.5, vis_margin=0, pred_vis=False):
    """prior_pose_align.py:10-168.  posemaps: float32 ndarray [B,A*(5+3J),h,w] (the network output after its sigmoid
    casts; [B,A*(5+4J),h,w] with pred_vis).  Returns (bboxes, humans, visibility) as nested lists like the reference:
    bboxes[b][n] float32[5], humans[b][n] float32[J,3], visibility[b][n] bool[J] -- with pred_vis float32[J] =
    the in-bounds test times the predicted visibility channel (:153-157)."""
    posemaps = np.asarray(posemaps, dtype=np.float32)
    if posemaps.ndim == 3:
        posemaps = posemaps[None]
    J = num_joints
    pm = decode_maps(posemaps, anchors, J, depth_mean, depth_std)
    B, A, F, hw = pm.shape
    bboxes_out, humans_out, vis_out = [], [], []
    for bi in range(B):
        det = pm[bi].transpose(0, 2, 1).reshape(A * hw, F)          # anchor-major, then cell
        sel = det[:, 4] > f32(conf_threshold)
        boxes = det[sel].copy()
        if boxes.shape[0] == 0:
            bboxes_out.append([]); humans_out.append([]); vis_out.append([])
            continue
        order, keep = box_nms_keep(boxes, nms_threshold)
        boxes = boxes[order][keep].copy()
        if boxes.shape[0] == 0:
            bboxes_out.append([]); humans_out.append([]); vis_out.append([])
            continue
        boxes[:, 0] *= f32(w_out)
        boxes[:, 2] *= f32(w_out)
        boxes[:, 1] *= f32(h_out)
        boxes[:, 3] *= f32(h_out)
        boxes[:, 0] -= boxes[:, 2] / f32(2)
        boxes[:, 1] -= boxes[:, 3] / f32(2)
        boxes[:, 2] += boxes[:, 0]
        boxes[:, 3] += boxes[:, 1]
        boxes[:, 5:5 + J] *= f32(w_out)
        boxes[:, 5 + J:5 + 2 * J] *= f32(h_out)
        bboxes_out.append([box[:5].copy() for box in boxes])
        hb, vb = [], []
        for box in boxes:
            human = box[5:5 + 3 * J].reshape(3, -1).T.copy()
            hb.append(human)
            inside = np.logical_and(
                np.logical_and(human[:, 0] >= 0 + vis_margin, human[:, 0] <= w_out - 1 - vis_margin),
                np.logical_and(human[:, 1] >= 0 + vis_margin, human[:, 1] <= h_out - 1 - vis_margin))
            vb.append(inside * box[5 + 3 * J:] if pred_vis else inside)
        humans_out.append(hb)
        vis_out.append(vb)
    return bboxes_out, humans_out, vis_out


def frame_glue(bboxes_b, humans_b, num_joints, input_size, w_org, h_org, intrinsics):
    """Per-frame glue of tpm/evaluate/evaluation_yolo_posenet_kdh3d_mpreal.py:182-217 for ONE image of
    parse_prior_pose's output: joints (and box corners, :197-200) rescaled to the original frame,
    back-projected with pos_3d_from_2d_and_depth (tpm/lib/utils/common.py:107-115), part confidence =
    the box confidence repeated for every joint (:185-186).  float32 throughout except part_conf
    (np.float = float64 of a float32 value), exactly as the script computes it.
    Returns dict(humans_2d [n,J,2] f32, humans_3d [n,J,3] f32, part_conf [n,J] f64, bboxes [n,5] f32)."""
    n = len(humans_b)
    J = num_joints
    h2 = np.zeros((n, J, 2), np.float32)
    h3 = np.zeros((n, J, 3), np.float32)
    bb = np.zeros((n, 5), np.float32)
    conf = np.zeros((n, J), np.float64)
    for i in range(n):
        human = np.array(humans_b[i][:, :2])                        # float32 copy (:183,:192)
        depth = humans_b[i][:, 2]
        human[:, 0] = human[:, 0] / input_size * w_org               # :194
        human[:, 1] = human[:, 1] / input_size * h_org               # :195
        box = np.array(bboxes_b[i], dtype=np.float32)
        box[0] = box[0] / input_size * w_org; box[2] = box[2] / input_size * w_org      # :197-198
        box[1] = box[1] / input_size * h_org; box[3] = box[3] / input_size * h_org      # :199-200
        X = (human[:, 0] - intrinsics['cx']) / intrinsics['fx'] * depth                  # common.py:113
        Y = (human[:, 1] - intrinsics['cy']) / intrinsics['fy'] * depth                  # common.py:114
        h2[i], h3[i], bb[i] = human, np.vstack([X, Y, depth]).T, box
        conf[i] = float(bboxes_b[i][4])
    return {"humans_2d": h2, "humans_3d": h3, "part_conf": conf, "bboxes": bb}
