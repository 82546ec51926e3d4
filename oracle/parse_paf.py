"""Bottom-up pose parsing (ORACLE; test infrastructure -- see oracle/__init__.py).

CPU restatement of the reference's Open-Pose+ post-processing:

  find_peaks / NMS              third_party_methods/lib/utils/paf_to_pose.py:33-153
  find_connected_joints         third_party_methods/lib/utils/paf_to_pose.py:156-264
  group_limbs_of_same_person    third_party_methods/lib/utils/paf_to_pose.py:267-351
  paf_to_pose                   third_party_methods/lib/utils/paf_to_pose.py:354-377
  paf_to_human_list             third_party_methods/lib/utils/common.py:5-32
  retrieve_depth_heat_weighted  third_party_methods/lib/utils/common.py:272-293
  per-frame glue (read-out, rescale, back-projection)
        third_party_methods/evaluate/evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py:187-274

Written so that every floating-point operation happens in the same type and the same
order as in the reference's NumPy code (float32 maps, float64 bookkeeping), because
"bit-exact person assignment" depends on threshold comparisons of those values.
cv2.resize is replaced by oracle.cv2_resize (parity unpinned at that boundary).
"""
import numpy as np

from . import cv2_resize

# ---- constants of the path (reference file:line) -------------------------------------------
THRESH_HEATMAP = 0.1          # lib/config/default.py:126
THRESH_PAF = 0.05             # lib/config/default.py:127
NUM_INTERMED_PTS = 10         # lib/config/default.py:128
DOWNSAMPLE = 8                # lib/config/default.py:41 (set by eval script :106)
NUM_KEYPOINTS = 15
WIN_SIZE = 2                  # paf_to_pose.py:108

# limb topology: lib/datasets/datasets_itop_rtpose.py:45-62 == util/util_functions.py:17-34
LIMBS = [[8, 9], [9, 11], [11, 13], [8, 10], [10, 12], [12, 14], [8, 1],
         [1, 2], [2, 4], [4, 6], [1, 3], [3, 5], [5, 7], [1, 0]]
NUM_LIMBS = len(LIMBS)

# MP-3DHP camera + depth normalisation: util/util_functions.py:4,11-13
INTRINSICS = {'fx': 504.1189880371094, 'fy': 504.042724609375,
              'cx': 231.7421875, 'cy': 320.62640380859375}
DEPTH_MEAN, DEPTH_STD, DEPTH_MAX = 3, 2, 6


def find_peaks(thresh, img):
    """paf_to_pose.py:33-46.  4-connected maximum filter with scipy's default 'reflect'
    border (the out-of-image neighbour of an edge pixel is the pixel itself), then
    ``== img`` and ``img > thresh``.  Returns [[x, y]...] in row-major (y, then x) order."""
    h, w = img.shape
    m = img.copy()
    m[1:, :] = np.maximum(m[1:, :], img[:-1, :])
    m[:-1, :] = np.maximum(m[:-1, :], img[1:, :])
    m[:, 1:] = np.maximum(m[:, 1:], img[:, :-1])
    m[:, :-1] = np.maximum(m[:, :-1], img[:, 1:])
    peaks_binary = (m == img) * (img > thresh)
    return np.array(np.nonzero(peaks_binary)[::-1]).T


def nms(heatmaps, upsamp=DOWNSAMPLE, num_keypoints=NUM_KEYPOINTS, thresh=THRESH_HEATMAP):
    """paf_to_pose.py:75-153 with bool_refine_center=True, bool_gaussian_filt=False.
    Returns a list (per joint type) of float64 [n,4] arrays (x, y, score, id)."""
    out = []
    cnt = 0
    for joint in range(num_keypoints):
        map_orig = heatmaps[:, :, joint]
        peak_coords = find_peaks(thresh, map_orig)
        peaks = np.zeros((len(peak_coords), 4))
        for i, peak in enumerate(peak_coords):
            x_min, y_min = np.maximum(0, peak - WIN_SIZE)
            x_max, y_max = np.minimum(np.array(map_orig.T.shape) - 1, peak + WIN_SIZE)
            patch = np.ascontiguousarray(map_orig[y_min:y_max + 1, x_min:x_max + 1])
            map_upsamp = cv2_resize.resize(patch, None, fx=upsamp, fy=upsamp,
                                           interpolation=cv2_resize.INTER_CUBIC)
            loc = np.unravel_index(map_upsamp.argmax(), map_upsamp.shape)   # first max, row-major
            # compute_resized_coords (paf_to_pose.py:49-72): (c + 0.5) * f - 0.5
            center_in_patch = (np.array(peak[::-1] - [y_min, x_min], dtype=float) + 0.5) * upsamp - 0.5
            refined = np.array(loc) - center_in_patch                       # (dy, dx)
            score = map_upsamp[loc]
            base = (np.array(peak_coords[i], dtype=float) + 0.5) * upsamp - 0.5   # (x, y)
            xy = base + refined[::-1]
            peaks[i, :] = (xy[0], xy[1], score, cnt)
            cnt += 1
        out.append(peaks)
    return out


def _pairwise_mean10(s):
    """np.mean of 10 float64 values = numpy pairwise_sum for 8 <= n < 128:
    ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then += the remaining elements, / n."""
    res = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))
    for i in range(8, len(s)):
        res = res + s[i]
    return res / len(s)


def find_connected_joints(paf_upsamp, joint_list_per_joint_type, num_intermed_pts=NUM_INTERMED_PTS,
                          thresh_paf=THRESH_PAF, limbs=LIMBS):
    """paf_to_pose.py:156-264.  Returns a list of NUM_LIMBS float64 [m,5] arrays
    (src_id, dst_id, score, src_idx, dst_idx) ([] when a joint type has no peaks)."""
    connected = []
    H = paf_upsamp.shape[0]
    for limb_type, (jsrc, jdst) in enumerate(limbs):
        joints_src = joint_list_per_joint_type[jsrc]
        joints_dst = joint_list_per_joint_type[jdst]
        if len(joints_src) == 0 or len(joints_dst) == 0:
            connected.append([])
            continue
        cand = []
        cx, cy = 2 * limb_type, 2 * limb_type + 1
        for i, js in enumerate(joints_src):
            for j, jd in enumerate(joints_dst):
                limb_dir = jd[:2] - js[:2]
                limb_dist = np.sqrt(np.sum(limb_dir ** 2)) + 1e-8
                limb_dir = limb_dir / limb_dist
                # np.round = round-half-to-even; np.linspace = i*step+start, last = stop
                xs = np.round(np.linspace(js[0], jd[0], num=num_intermed_pts)).astype(np.intp)
                ys = np.round(np.linspace(js[1], jd[1], num=num_intermed_pts)).astype(np.intp)
                px = paf_upsamp[ys, xs, cx].astype(np.float64)
                py = paf_upsamp[ys, xs, cy].astype(np.float64)
                score_pts = px * limb_dir[0] + py * limb_dir[1]
                score = _pairwise_mean10(score_pts) + min(0.5 * H / limb_dist - 1, 0)
                c1 = np.count_nonzero(score_pts > thresh_paf) > 0.8 * num_intermed_pts
                c2 = score > 0
                if c1 and c2:
                    cand.append([i, j, score, score + js[2] + jd[2]])
        cand = sorted(cand, key=lambda x: x[2], reverse=True)     # stable
        connections = np.empty((0, 5))
        max_conn = min(len(joints_src), len(joints_dst))
        for c in cand:
            i, j, s = c[0:3]
            if i not in connections[:, 3] and j not in connections[:, 4]:
                connections = np.vstack([connections, [joints_src[i][3], joints_dst[j][3], s, i, j]])
                if len(connections) >= max_conn:
                    break
        connected.append(connections)
    return connected


def group_limbs_of_same_person(connected_limbs, joint_list, num_keypoints=NUM_KEYPOINTS, limbs=LIMBS):
    """paf_to_pose.py:267-351, literal (including the ">=3 matches => new person"
    fall-through and the in-place ``+= other + 1`` merge)."""
    persons = []
    for limb_type, (src_t, dst_t) in enumerate(limbs):
        for limb in connected_limbs[limb_type]:
            hit = [p for p, row in enumerate(persons) if row[src_t] == limb[0] or row[dst_t] == limb[1]]
            if len(hit) == 1:
                row = persons[hit[0]]
                if row[dst_t] != limb[1]:
                    row[dst_t] = limb[1]
                    row[-1] += 1
                    row[-2] += joint_list[limb[1].astype(int), 2] + limb[2]
            elif len(hit) == 2:
                r1, r2 = persons[hit[0]], persons[hit[1]]
                membership = ((r1 >= 0) & (r2 >= 0))[:-2]
                if not membership.any():
                    r1[:-2] += (r2[:-2] + 1)
                    r1[-2:] += r2[-2:]
                    r1[-2] += limb[2]
                    persons.pop(hit[1])
                else:
                    r1[dst_t] = limb[1]
                    r1[-1] += 1
                    r1[-2] += joint_list[limb[1].astype(int), 2] + limb[2]
            else:
                row = -1 * np.ones(num_keypoints + 2)
                row[src_t] = limb[0]
                row[dst_t] = limb[1]
                row[-1] = 2
                row[-2] = sum(joint_list[limb[:2].astype(int), 2]) + limb[2]
                persons.append(row)
    keep = [r for r in persons if not (r[-1] < 3 or r[-2] / r[-1] < 0.2)]
    return np.array(keep)


def paf_to_pose(heatmaps, pafs):
    """paf_to_pose.py:354-377.  heatmaps [H',W',J+1] float32, pafs [H',W',2L] float32 (HWC)."""
    per_type = nms(heatmaps)
    joint_list = np.array([tuple(peak) + (jt,) for jt, peaks in enumerate(per_type) for peak in peaks])
    paf_upsamp = cv2_resize.resize(np.ascontiguousarray(pafs), None, fx=DOWNSAMPLE, fy=DOWNSAMPLE,
                                   interpolation=cv2_resize.INTER_CUBIC)
    connected = find_connected_joints(paf_upsamp, per_type)
    assoc = group_limbs_of_same_person(connected, joint_list)
    return joint_list, assoc


def paf_to_pose_cpp(heatmaps, pafs, paflib, num_keypoints=18, thresh=THRESH_HEATMAP):
    """paf_to_pose.py:381-415 (ORACLE restatement): NMS over the 18 COCO part maps -> joint_list [1, N, 5] float32 ->
    INTER_NEAREST x8 of both maps -> process_paf + getters (`paflib`: oracle.pafprocess.restated() or .reference()) ->
    rows [score, 18 x (x / W_up, y / H_up, part score) or -1s], the fields of the reference's Human / BodyPart objects."""
    per_type = nms(heatmaps, num_keypoints=num_keypoints, thresh=thresh)
    joint_list = np.array([tuple(peak) + (jt,) for jt, peaks in enumerate(per_type) for peak in peaks]).astype(np.float32)
    rows = []
    if joint_list.shape[0] > 0:
        hu = cv2_resize.resize(np.ascontiguousarray(heatmaps), None, fx=DOWNSAMPLE, fy=DOWNSAMPLE, interpolation=cv2_resize.INTER_NEAREST)
        pu = cv2_resize.resize(np.ascontiguousarray(pafs), None, fx=DOWNSAMPLE, fy=DOWNSAMPLE, interpolation=cv2_resize.INTER_NEAREST)
        for h in paflib.run(joint_list[None], hu, pu):
            if not h['parts']:
                continue
            row = -np.ones(1 + 18 * 3)
            row[0] = h['score']
            for p, (cid, x, y, sc) in h['parts'].items():
                row[1 + 3 * p:4 + 3 * p] = (float(x) / hu.shape[1], float(y) / hu.shape[0], sc)
            rows.append(row)
    return np.array(rows, dtype=np.float64).reshape(-1, 1 + 18 * 3), per_type


def paf_to_human_list(joint_list, person_to_joint_assoc):
    """common.py:5-32."""
    humans, visibility, conf_vec = [], [], []
    for human in person_to_joint_assoc:
        idx = human[:-2].astype(int)
        joints, conf = [], []
        for ind in idx:
            if ind < 0:
                joints.append([-1, -1])
                conf.append(0)
            else:
                joints.append(joint_list[ind, :2].tolist())
                conf.append(float(joint_list[ind, 2]))
        humans.append(joints)
        visibility.append((idx >= 0).astype(int).tolist())
        conf_vec.append(conf)
    return humans, visibility, conf_vec


def retrieve_depth_heat_weighted(center, depthmap, heatmap, radius=1):
    """common.py:272-293 (the in-place ``heatmap[heatmap < 0] = 0`` included)."""
    heatmap[heatmap < 0] = 0
    gx, gy = depthmap.shape[1], depthmap.shape[0]
    min_x = min(max(int(center[0] - radius), 0), gx - 1)
    max_x = max(min(int(center[0] + radius), gx - 1), 0)
    min_y = min(max(int(center[1] - radius), 0), gy - 1)
    max_y = max(min(int(center[1] + radius), gy - 1), 0)
    xx, yy = np.meshgrid(list(range(min_x, max_x + 1)), list(range(min_y, max_y + 1)))
    w = heatmap[yy, xx] + 0.000000001
    d = depthmap[yy, xx]
    return np.sum(d * w) / np.sum(w)


def frame_to_records(heat_hwc, paf_hwc, z_hwc, w_org=480, h_org=640, input_size=224,
                     intrinsics=INTRINSICS, depth_mean=DEPTH_MEAN, depth_std=DEPTH_STD):
    """Per-frame body of the reference evaluation loop
    (evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py:179-180,187-216,245-262,308-316).

    heat/paf/z are the network outputs in HWC float32 (z still normalised).  Returns a dict of
    Python lists shaped like one entry of the reference's eval_data.json:
      humans_2d [P][15][2], humans_3d [P][15][3], visibility [P][15], conf [P][15],
    plus the raw (joint_list, assoc) for index-exact comparison.
    """
    posedepth = z_hwc.copy()
    posedepth *= depth_std          # float32 in-place, as in the reference (:179-180)
    posedepth += depth_mean
    heat = heat_hwc.copy()
    joint_list, assoc = paf_to_pose(heat, paf_hwc)
    humans_2d, visibility, conf_vec = paf_to_human_list(joint_list, assoc)
    humans_depth = []
    for i, human in enumerate(humans_2d):
        hd = np.ones(NUM_KEYPOINTS) * -1
        for j, joint in enumerate(human):
            if visibility[i][j] > 0.5:
                hd[j] = retrieve_depth_heat_weighted(
                    [int(joint[0] / DOWNSAMPLE), int(joint[1] / DOWNSAMPLE)],
                    posedepth[:, :, j], heat[:, :, j], radius=1)
        humans_depth.append(hd)
    out2d, out3d = [], []
    for i, human in enumerate(humans_2d):
        human = np.array(human)
        vis = np.where(visibility[i])
        human[vis, 0] = human[vis, 0] / input_size * w_org
        human[vis, 1] = human[vis, 1] / input_size * h_org
        X = (human[:, 0] - intrinsics['cx']) * humans_depth[i] / intrinsics['fx']
        Y = (human[:, 1] - intrinsics['cy']) * humans_depth[i] / intrinsics['fy']
        out3d.append(np.vstack([X, Y, humans_depth[i]]).T.tolist())
        out2d.append(human.tolist())
    return {'humans_2d': out2d, 'humans_3d': out3d, 'visibility': visibility, 'conf': conf_vec,
            'joint_list': joint_list, 'assoc': assoc}
