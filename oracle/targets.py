"""Training-target rasterisation and the multi-person depth compositor (ORACLE; test infrastructure -- see oracle/__init__.py).

Restates, cell by cell, what the reference's training dataset computes on the CPU:
  get_ground_truth      third_party_methods/lib/datasets/datasets_kdh3d_rtpose_mpaug.py:318-401 (CR line endings)
    putGaussianMaps     third_party_methods/lib/datasets/heatmap.py:20-36      (sigma 7 px, exponent cut 4.6052, clamp at 1)
    putVecMaps          third_party_methods/lib/datasets/paf.py:18-69          (limb width 1 cell, count-weighted average)
    putJointZ           third_party_methods/lib/datasets/posemap.py:83-106     (window of z_radius cells, nearest person wins)
  z-buffer compositor   datasets_kdh3d_rtpose_mpaug.py:231-266 (CR)            (min depth over the foreground masks, background paste)
Everything is float64 like the reference's NumPy arrays; the final maps are cast to float32 where the reference does
(single_image_processing, :303-306).  Pinned by tests/golden/targets.npz, produced by the reference's own functions.
"""
import numpy as np

DEPTH_MEAN, DEPTH_STD, DEPTH_MAX = 3, 2, 6
LIMBS = [(8, 9), (9, 11), (11, 13), (8, 10), (10, 12), (12, 14), (8, 1), (1, 2), (2, 4), (4, 6), (1, 3), (3, 5), (5, 7), (1, 0)]


def inbounds_mask(kp2d, input_x, input_y):
    """remove_illegal_joint (:308-316): 1 where the joint lies inside the network input."""
    bad = (kp2d[:, :, 0] >= input_x) | (kp2d[:, :, 0] < 0) | (kp2d[:, :, 1] >= input_y) | (kp2d[:, :, 1] < 0)
    return (~bad).astype(np.float64)


def ground_truth(kp2d, kp3d, depth_resize, input_x=224, input_y=224, stride=8, z_radius=2, sigma=7.0):
    """kp2d [P,15,2] input-pixel coordinates, kp3d [P,15,3] metres, depth_resize [h,w] (the clamped input at stride resolution).
    Returns heat [h,w,16], paf [h,w,28], z [h,w,15] (normalised), fg [h,w,15], float64."""
    kp2d = np.asarray(kp2d, dtype=np.float64).reshape(-1, 15, 2)
    kp3d = np.asarray(kp3d, dtype=np.float64).reshape(-1, 15, 3)
    P = kp2d.shape[0]
    gh, gw = int(input_y / stride), int(input_x / stride)
    inb = inbounds_mask(kp2d, input_x, input_y) if P else np.zeros((0, 15))
    ys, xs = np.mgrid[0:gh, 0:gw].astype(np.float64)
    start = stride / 2.0 - 0.5
    heat = np.zeros((gh, gw, 16))
    for i in range(15):
        acc = np.zeros((gh, gw))
        for j in range(P):
            if inb[j, i] <= 0.5:
                continue
            d2 = (xs * stride + start - kp2d[j, i, 0]) ** 2 + (ys * stride + start - kp2d[j, i, 1]) ** 2
            e = d2 / 2.0 / sigma / sigma
            acc = acc + (e <= 4.6052) * np.exp(-e)
            acc[acc > 1.0] = 1.0
        heat[:, :, i] = acc
    heat[:, :, 15] = np.maximum(1 - heat[:, :, :15].max(axis=2), 0.0)

    paf = np.zeros((gh, gw, 28))
    for l, (k1, k2) in enumerate(LIMBS):
        vx, vy, cnt = np.zeros((gh, gw)), np.zeros((gh, gw)), np.zeros((gh, gw))
        for j in range(P):
            if not (inb[j, k1] > 0.5 and inb[j, k2] > 0.5):
                continue
            a, b = kp2d[j, k1] / stride, kp2d[j, k2] / stride
            v = b - a
            n = np.linalg.norm(v)
            if n == 0.0:
                continue
            u = v / n
            x0, x1 = max(int(round(min(a[0], b[0]) - 1)), 0), min(int(round(max(a[0], b[0]) + 1)), gw - 1)
            y0, y1 = max(int(round(min(a[1], b[1]) - 1)), 0), min(int(round(max(a[1], b[1]) + 1)), gh - 1)
            box = (xs >= x0) & (xs <= x1) & (ys >= y0) & (ys <= y1)
            near = np.abs((xs - a[0]) * u[1] - (ys - a[1]) * u[0]) < 1
            m = box & near
            wx, wy = m * u[0], m * u[1]
            hit = (np.abs(wx) > 0) | (np.abs(wy) > 0)
            vx, vy = vx * cnt + wx, vy * cnt + wy
            cnt = cnt + hit
            div = np.where(cnt == 0, 1.0, cnt)
            vx, vy = vx / div, vy / div
        paf[:, :, 2 * l], paf[:, :, 2 * l + 1] = vx, vy

    # the z maps inherit the dtype of depth_resize (np.ones_like, :331): float32 inside __getitem__ (cv2.resize output),
    # float64 when get_ground_truth is handed a float64 map
    zorg = np.repeat(np.asarray(depth_resize)[:, :, None], 15, axis=2)
    if zorg.dtype not in (np.float32, np.float64):
        zorg = zorg.astype(np.float64)
    dt = zorg.dtype.type
    z = np.ones_like(zorg) * 2 * DEPTH_MAX
    fg = np.zeros((gh, gw, 15))
    for j in range(P):
        for k in range(15):
            if inb[j, k] < 0.5:
                continue
            cx, cy = kp2d[j, k] / stride
            x0, x1 = max(int(int(cx - z_radius)), 0), min(int(int(cx + z_radius)), gw - 1)
            y0, y1 = max(int(int(cy - z_radius)), 0), min(int(int(cy + z_radius)), gh - 1)
            win = (xs >= x0) & (xs <= x1) & (ys >= y0) & (ys <= y1)
            pz = np.where(win, dt(kp3d[j, k, 2]), dt(DEPTH_MAX))        # posemap_Z is an array of the map's dtype
            zk = np.minimum(pz, z[:, :, k])
            new = (pz < DEPTH_MAX) & (fg[:, :, k] == 0)
            zk[new] = pz[new]
            z[:, :, k] = zk
            fg[:, :, k] = np.logical_or(fg[:, :, k], new)
    z[fg == 0] = zorg[fg == 0]
    z[z < 0] = 0
    z[z > DEPTH_MAX] = DEPTH_MAX
    z = (z - DEPTH_MEAN) / DEPTH_STD
    return heat, paf, z, fg


def compose_depth(fg_depths, fg_masks, bg, depth_max=DEPTH_MAX):
    """z-buffer composition (:231-266): image starts at 2 * depth_max, every source writes min(depth * mask, image) where its
    mask is set, the union mask selects foreground, the background frame fills the rest."""
    image = np.ones(bg.shape, dtype=np.float64) * 2 * depth_max
    union = np.zeros(bg.shape, dtype=np.float64)
    for d, m in zip(fg_depths, fg_masks):
        d, m = np.asarray(d, dtype=np.float64), np.asarray(m, dtype=np.float64)
        sel = m > 0
        image[sel] = np.minimum((d * m)[sel], image[sel])
        union = np.maximum(union, m)
    return image * union + np.asarray(bg, dtype=np.float64) * (np.ones_like(union) - union), union


def mpaug_item(fg_depths, fg_masks, bg, kp2d_org, kp3d, input_size=224, stride=8, z_radius=2):
    """What KDH3D_Keypoints.__getitem__ (:223-286) returns for a given choice of source frames and background, with the
    evaluation pre-processing Compose([Cvt2ndarray(), Resize(input_size)]) (data_augmentation_2d3d.py:70-89,497-522):
    (image [1,S,S] float32 normalised, heat [16,h,w], paf [28,h,w], z [15,h,w], fg [15,h,w] float32)."""
    from . import cv2_resize
    image, _ = compose_depth(fg_depths, fg_masks, bg)
    h_org, w_org = image.shape
    image = cv2_resize.resize(image.astype(np.float32), (input_size, input_size), interpolation=cv2_resize.INTER_LINEAR)
    kp = np.asarray(kp2d_org, dtype=np.float32).reshape(-1, 15, 2).copy()
    kp[:, :, 0] *= float(input_size) / w_org          # float32 array times a Python float stays float32 (Resize.__call__)
    kp[:, :, 1] *= float(input_size) / h_org
    image[image < 0] = 0
    image[image > DEPTH_MAX] = DEPTH_MAX
    g = int(input_size / stride)
    depth_resize = cv2_resize.resize(image, (g, g), interpolation=cv2_resize.INTER_LINEAR)
    heat, paf, z, fg = ground_truth(kp, kp3d, depth_resize, input_size, input_size, stride, z_radius)
    img = ((image - np.float32(DEPTH_MEAN)) / np.float32(DEPTH_STD))[None].astype(np.float32)
    return img, tuple(a.transpose(2, 0, 1).astype(np.float32) for a in (heat, paf, z, fg))
