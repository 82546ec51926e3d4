"""MP-3DHP metric evaluation: best-match PCK (2D PCKh-0.5, 3D 10 cm) and MPII-style mAP (2D / 3D).

Host-side step right AFTER the hot path: it consumes the result schema the path
writes (``human_pred_set_2d/_3d/_part_conf``) and the MP-3DHP ``labels.json``.  A from-scratch
NumPy restatement -- per-image array arithmetic instead of the reference's
O(images x preds x gts x 15) Python loops -- of

    util/eval_pck.py   eval_human_dataset_2d_PCKh :80-154, eval_human_dataset_3d :313-374,
                       match_humans_2d/_3d :266-310,377-430, compute_bbox_from_humans :433-449,
                       bbox_ious :452-475, compute_head_size :232-246
    util/eval_mAP.py   assignGTmulti :60-157, getRPC :160-191, VOCap :194-207,
                       eval_ap_mpii_v2 :272-332, eval_ap_3D :335-398
    main_evaluate_mp_human_3D.py :21-99 (parse_gt_labels + the four metric blocks)

with the reference's corner cases kept (SURVEY Appendix B style):
  * a missing predicted joint is the pair (-1, -1); it yields distance -1 in the PCK code but still
    counts as a *prediction* with its confidence in the mAP code (eval_mAP.py:106-108);
  * PCK hits use ``dist < thr`` (strict), mAP matches use ``dist <= thr``;
  * one predicted person without any valid joint empties the whole image's prediction boxes
    (eval_pck.py:441-442 returns early), so no GT of that image is matched;
  * a GT person is matched to the FIRST prediction with the largest box IoU (np.argmax);
  * detections are ranked with ``np.flip(np.argsort(scores))`` (default, non-stable sort) exactly like
    getRPC, so ties are broken identically on the same NumPy;
  * removed NumPy aliases (``np.int``, eval_mAP.py:121) are not used.
Pinned by tests/golden/metrics.json: outputs of the reference's own functions on seeded cases.
"""
import json

import numpy as np

from .config import KEYPOINTS, NUM_PARTS


# ---------------------------------------------------------------------------------------------
# PCK
# ---------------------------------------------------------------------------------------------
def _boxes(humans):
    """eval_pck.py:433-449.  [n,4] (xmin, ymin, xmax, ymax) over valid joints; an EMPTY array as soon
    as one human has no valid joint (the reference returns early)."""
    out = []
    for h in humans:
        a = np.asarray(h, dtype=np.float64).reshape(-1, 2)
        valid = a[~((a[:, 0] == -1) & (a[:, 1] == -1))]
        if len(valid) == 0:
            return np.zeros((0, 4))
        out.append([valid[:, 0].min(), valid[:, 1].min(), valid[:, 0].max(), valid[:, 1].max()])
    return np.array(out, dtype=np.float64).reshape(-1, 4)


def bbox_ious(boxes1, boxes2):
    """eval_pck.py:452-475: [n1,n2] IoU, or a [n1,1] column of -1 when boxes2 is empty."""
    boxes1, boxes2 = np.asarray(boxes1, dtype=np.float64), np.asarray(boxes2, dtype=np.float64)
    if len(boxes2) == 0:
        return -np.ones((len(boxes1), 1))
    dx = np.maximum(np.minimum(boxes1[:, None, 2], boxes2[None, :, 2]) - np.maximum(boxes1[:, None, 0], boxes2[None, :, 0]), 0)
    dy = np.maximum(np.minimum(boxes1[:, None, 3], boxes2[None, :, 3]) - np.maximum(boxes1[:, None, 1], boxes2[None, :, 1]), 0)
    inter = dx * dy
    a1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    a2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / ((a1[:, None] + a2[None, :]) - inter)


def _match(pred2d, gt2d, pred_pts, gt_pts, iou_th, mask_gt2d):
    """match_humans_2d / match_humans_3d: per GT person the joint distances to its best-IoU prediction
    (-1 where unmatched / the predicted joint is missing / (3D only) the GT 2D joint is missing)."""
    n_gt = len(gt2d)
    J = np.asarray(gt_pts[0]).shape[0] if n_gt else 0
    if len(pred2d) == 0:
        return -np.ones((n_gt, J))
    gtb = _boxes(gt2d)
    if len(gtb) != n_gt:
        raise ValueError("a ground-truth person has no valid joint (the reference fails on it too)")
    ious = bbox_ious(gtb, _boxes(pred2d))
    best = np.argmax(ious, axis=1)
    ok = ious[np.arange(n_gt), best] >= iou_th
    P2 = np.asarray(pred2d, dtype=np.float64).reshape(len(pred2d), -1, 2)[best]
    G = np.asarray(gt_pts, dtype=np.float64)
    P = np.asarray(pred_pts, dtype=np.float64)[best]
    d = np.sqrt(np.sum((G - P) ** 2, axis=2))
    d[(P2[:, :, 0] == -1) & (P2[:, :, 1] == -1)] = -1
    if mask_gt2d:
        G2 = np.asarray(gt2d, dtype=np.float64).reshape(n_gt, -1, 2)
        d[(G2[:, :, 0] == -1) & (G2[:, :, 1] == -1)] = -1
    d[~ok] = -1
    return d


def match_humans_2d(humans_pred, humans_gt, iou_th=0.5):
    return list(_match(humans_pred, humans_gt, humans_pred, humans_gt, iou_th, False))


def match_humans_3d(humans_pred_2d, humans_gt_2d, humans_pred_3d, humans_gt_3d, iou_th=0.5):
    return list(_match(humans_pred_2d, humans_gt_2d, humans_pred_3d, humans_gt_3d, iou_th, True))


def compute_head_size(humans, ind1, ind2):
    a = np.asarray(humans, dtype=np.float64).reshape(len(humans), -1, 2)
    return list(2 * np.sqrt((a[:, ind1, 0] - a[:, ind2, 0]) ** 2 + (a[:, ind1, 1] - a[:, ind2, 1]) ** 2))


def _pck_summary(dists, hits, denom):
    avg = []
    with np.errstate(invalid="ignore", divide="ignore"):
        for k in range(dists.shape[1]):
            col = dists[:, k]
            sel = col[col >= 0]
            avg.append(float(np.mean(sel)) if len(sel) else float("nan"))
        kcp = [float(v) for v in hits.sum(axis=0) / denom]
    return avg, kcp


def eval_human_dataset_2d_PCKh(humans_pred_set, humans_gt_set, head_id, neck_id, num_joints=15, h_th=0.5, iou_th=0.5,
                               human_gt_set_visibility=None):
    """eval_pck.py:80-154 -> (joint_avg_dist[J], joint_KCP[J])."""
    assert len(humans_gt_set) == len(humans_pred_set)
    D, H, V = [], [], []
    for i, (gt, pred) in enumerate(zip(humans_gt_set, humans_pred_set)):
        if len(gt) == 0:
            continue
        d = _match(pred, gt, pred, gt, iou_th, False)
        vis = np.ones((len(gt), num_joints)) if human_gt_set_visibility is None else np.asarray(human_gt_set_visibility[i], dtype=np.float64)
        d[vis == 0] = -1
        hsz = np.asarray(compute_head_size(gt, head_id, neck_id))
        D.append(d); V.append(vis)
        H.append((d >= 0) & (d < hsz[:, None] * h_th))
    if not D:
        return [float("nan")] * num_joints, [float("nan")] * num_joints
    D, H, V = np.concatenate(D), np.concatenate(H), np.concatenate(V)
    return _pck_summary(D, H, V.sum(axis=0))


def eval_human_dataset_3d(humans_pred_set_2d, humans_gt_set_2d, humans_pred_set_3d, humans_gt_set_3d, num_joints=15,
                          dist_th=0.1, iou_th=0.5, human_gt_set_visibility=None):
    """eval_pck.py:313-374 -> (joint_avg_dist[J] in metres, joint_KCP[J])."""
    assert len(humans_gt_set_2d) == len(humans_pred_set_2d)
    D, V, samples = [], [], 0
    for i in range(len(humans_gt_set_2d)):
        gt2, pr2 = humans_gt_set_2d[i], humans_pred_set_2d[i]
        samples += len(gt2)
        if len(gt2) == 0:
            continue
        d = _match(pr2, gt2, humans_pred_set_3d[i], humans_gt_set_3d[i], iou_th, True)
        if human_gt_set_visibility is not None:
            vis = np.asarray(human_gt_set_visibility[i], dtype=np.float64)
            d[vis == 0] = -1
            V.append(vis)
        D.append(d)
    if not D:
        return [float("nan")] * num_joints, [float("nan")] * num_joints
    D = np.concatenate(D)
    hits = (D >= 0) & (D < dist_th)
    denom = np.concatenate(V).sum(axis=0) if human_gt_set_visibility is not None else float(samples)
    return _pck_summary(D, hits, denom)


# ---------------------------------------------------------------------------------------------
# mAP
# ---------------------------------------------------------------------------------------------
def assign_gt_multi(humans_pred_set, conf_pred_set, humans_gt_set, gt_visibility_set, ref_dist_set, num_joints, thresh):
    """eval_mAP.py:60-157.  Returns (scores, labels, nGT): per joint the concatenated detection scores and
    0/1 labels in the reference's (image, prediction) order, and the number of annotated GT joints."""
    scores = [[] for _ in range(num_joints)]
    labels = [[] for _ in range(num_joints)]
    nGT = np.zeros(num_joints)
    for img in range(len(humans_gt_set)):
        preds, gts = humans_pred_set[img], humans_gt_set[img]
        nP, nG = len(preds), len(gts)
        if nG:
            vis = np.asarray(gt_visibility_set[img], dtype=np.float64).reshape(nG, num_joints)
            G = np.asarray(gts, dtype=np.float64).reshape(nG, num_joints, -1)
        else:
            vis, G = np.zeros((0, num_joints)), np.zeros((0, num_joints, 2))
        has_gt_final = (vis > 0).astype(np.float64)
        if nP > 0:
            P = np.asarray(preds, dtype=np.float64).reshape(nP, num_joints, -1)
            score = np.asarray(conf_pred_set[img], dtype=np.float64).reshape(nP, num_joints)
            if nG:
                ref = np.asarray(ref_dist_set[img], dtype=np.float64).reshape(nG)
                with np.errstate(divide="ignore", invalid="ignore"):
                    dist = np.sqrt(np.sum((P[:, None] - G[None]) ** 2, axis=3)) / ref[None, :, None]
                dist = np.where(has_gt_final[None] > 0, dist, np.inf)
                match = (dist <= thresh).astype(np.int64)
                with np.errstate(divide="ignore", invalid="ignore"):
                    pck = match.sum(axis=2) / has_gt_final.sum(axis=1)[None, :]
                idx = np.argmax(pck, axis=1)
                keep = np.zeros_like(pck, dtype=bool)
                keep[np.arange(nP), idx] = True
                pck = np.where(keep, pck, 0)
                val = np.max(pck, axis=0)
                pred_to_gt = np.argmax(pck, axis=0)
                pred_to_gt[val == 0] = -1
            else:
                # no GT person: the reference's argmax over an empty axis raises -- an image without ground truth
                # but with predictions is outside what it supports; every detection counts as a false positive here
                match = np.zeros((nP, 0, num_joints), dtype=np.int64)
                pred_to_gt = np.zeros(0, dtype=np.int64)
            for p in range(nP):
                hit = np.where(pred_to_gt == p)[0]
                m = match[p, hit[0]] if len(hit) else np.zeros(num_joints)
                for j in range(num_joints):
                    scores[j].append(score[p, j])
                    labels[j].append(m[j])
        if nP > 0:          # reference quirk (eval_mAP.py:89,153-155): hasGT is only filled inside the prediction loop, so the
            nGT += has_gt_final.sum(axis=0)   # GT joints of an image WITHOUT predictions never enter the recall denominator
    return scores, labels, nGT


def get_rpc(class_margin, true_labels, totalpos):
    """eval_mAP.py:160-191 (cumulative sums instead of the loop)."""
    class_margin, true_labels = np.asarray(class_margin), np.asarray(true_labels)
    n = true_labels.shape[0]
    sortidx = np.flip(np.argsort(class_margin))
    pos = np.cumsum(true_labels[sortidx] == 1) if n else np.zeros(0)
    with np.errstate(divide="ignore", invalid="ignore"):
        return pos / np.arange(1, n + 1), pos / totalpos


def voc_ap(recall, precision):
    """eval_mAP.py:194-207."""
    mrec = np.concatenate([[0.0], recall, [1.0]])
    mpre = np.concatenate([[0.0], precision, [0.0]])
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    idx = np.where((mrec[1:] - mrec[:-1]) > 0)[0] + 1
    return float(np.sum((mrec[idx] - mrec[idx - 1]) * mpre[idx]))


def _eval_ap(humans_pred_set, conf_pred_set, humans_gt_set, gt_visibility_set, ref_dist_set, joint_names, thresh, verbose, title):
    assert len(humans_gt_set) == len(humans_pred_set)
    J = len(joint_names)
    if len(gt_visibility_set) == 0:
        gt_visibility_set = [np.ones((len(g), J)).tolist() for g in humans_gt_set]
    if len(conf_pred_set) == 0:
        conf_pred_set = [np.ones((len(p), J)).tolist() for p in humans_pred_set]
    scores, labels, nGT = assign_gt_multi(humans_pred_set, conf_pred_set, humans_gt_set, gt_visibility_set, ref_dist_set, J, thresh)
    ap = np.zeros(J + 1)
    for j in range(J):
        precision, recall = get_rpc(scores[j], labels[j], nGT[j])
        ap[j] = voc_ap(recall, precision) * 100
    ap[-1] = np.mean(ap[:-1])
    if verbose:
        print(title)
        for j, name in enumerate(joint_names):
            print('    {},  AP: {:03f}'.format(name, ap[j]))
        print('\n     Overall: AP: {:03f}\n'.format(ap[-1]))
    return ap


def eval_ap_mpii_v2(humans_pred_set, conf_pred_set, humans_gt_set, gt_visibility_set, head_id, neck_id, joint_names, thresh=0.5,
                    verbose=True):
    """eval_mAP.py:272-332: 2D AP per joint (+ mean) under the PCKh-`thresh` rule, head size = 2 x |head - neck|."""
    ref = [compute_head_size(g, head_id, neck_id) if len(g) else [] for g in humans_gt_set]
    return _eval_ap(humans_pred_set, conf_pred_set, humans_gt_set, gt_visibility_set, ref, joint_names, thresh, verbose,
                    '2D evaluation in AP evaluation under PCKh-{:01f} rule ...'.format(thresh))


def eval_ap_3D(humans_pred_set, conf_pred_set, humans_gt_set, gt_visibility_set, joint_names, thresh=0.1, verbose=True):
    """eval_mAP.py:335-398: 3D AP per joint (+ mean) under the `thresh`-metre rule."""
    ref = [np.ones(len(g)).tolist() for g in humans_gt_set]
    return _eval_ap(humans_pred_set, conf_pred_set, humans_gt_set, gt_visibility_set, ref, joint_names, thresh, verbose,
                    '3D evaluation in AP under {:01f} meter rule ...'.format(thresh))


# ---------------------------------------------------------------------------------------------
# main_evaluate_mp_human_3D.py
# ---------------------------------------------------------------------------------------------
def parse_gt_labels(anno_dic):
    """main_evaluate_mp_human_3D.py:21-41: labels.json dict -> (gt_2d, gt_3d) in key order, 'intrinsics' skipped."""
    g2, g3 = [], []
    for key, people in anno_dic.items():
        if key == 'intrinsics':
            continue
        g2.append([a['2d_joints'] for a in people])
        g3.append([a['3d_joints'] for a in people])
    return g2, g3


def evaluate_mp_human_3d(gt_file, res_file, verbose=True):
    """The four metric blocks of main_evaluate_mp_human_3D.py:44-99 on a labels.json / results.json pair.
    Returns a dict of the numbers it prints."""
    res = json.load(open(res_file))
    if 'pop' in res_file and 'human_pred_set_2d_aligned' in res:
        p2, p3 = res['human_pred_set_2d_aligned'], res['human_pred_set_3d_aligned']
    else:
        p2, p3 = res['human_pred_set_2d'], res['human_pred_set_3d']
    g2, g3 = parse_gt_labels(json.load(open(gt_file)))
    out = {}
    d2, k2 = eval_human_dataset_2d_PCKh(p2, g2, head_id=0, neck_id=1, num_joints=NUM_PARTS, iou_th=0.5)
    d3, k3 = eval_human_dataset_3d(p2, g2, p3, g3, num_joints=NUM_PARTS, dist_th=0.1, iou_th=0.5)
    out.update(pck2d=k2, err2d=d2, pck3d=k3, err3d=d3)
    if verbose:
        for title, kk, dd, unit in (('2d PCKh-0.5', k2, d2, '2D'), ('3d PCK', k3, d3, '3D')):
            print(title)
            for i, name in enumerate(KEYPOINTS):
                print('     joint: {},  PCK: {:03f}, avg {} error: {:03f}'.format(name, kk[i], unit, dd[i]))
            print('\n     Overall: PCK: {:03f}, avg {} error: {:03f} \n'.format(np.average(kk), unit, np.average(dd)))
    conf = res.get('human_pred_set_part_conf', [])
    out['ap2d'] = eval_ap_mpii_v2(p2, conf, g2, [], 0, 1, KEYPOINTS, 0.5, verbose).tolist()
    out['ap3d'] = eval_ap_3D(p3, conf, g3, [], KEYPOINTS, 0.1, verbose).tolist()
    return out


if __name__ == "__main__":
    import sys
    if len(sys.argv) != 3:
        raise SystemExit("usage: python -m popnet_amd.metrics labels.json results.json")
    evaluate_mp_human_3d(sys.argv[1], sys.argv[2])
