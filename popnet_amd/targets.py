"""Training targets on the GPU: the multi-person depth compositor and the ground-truth rasterisers.

Host-side mirror of the reference's training dataset item, batched and on the device:
  KDH3D_Keypoints.__getitem__   third_party_methods/lib/datasets/datasets_kdh3d_rtpose_mpaug.py:223-286 (CR line endings)
  get_ground_truth              ...:318-401 (putGaussianMaps heatmap.py:20-36, putVecMaps paf.py:18-69, putJointZ posemap.py:83-106)
  Compose([Cvt2ndarray, Resize]) third_party_methods/lib/datasets/data_augmentation_2d3d.py:70-89,497-522
The arithmetic lives in csrc/targets.hip (pn_compose_depth, pn_rasterize_targets) and csrc/api.hip (pn_preprocess); this
module only owns tensors and the call order.  There is no CPU fallback: inputs must be device tensors.
"""
import ctypes as C
import random

import numpy as np
import torch

from . import _lib
from .config import DEPTH_MAX, DEPTH_MEAN, DEPTH_STD

NUM_JOINTS, NUM_LIMBS = 15, 14


def _ctx(device):
    return _lib.Context.for_device(torch.device(device).index or 0)


def target_cfg(input_size=224, stride=8, z_radius=2, sigma=7.0):
    cfg = _lib.TargetCfg()
    _lib.lib().pn_target_cfg_default(C.byref(cfg))
    cfg.input_x = cfg.input_y = int(input_size)
    cfg.stride, cfg.z_radius, cfg.sigma = int(stride), int(z_radius), float(sigma)
    return cfg


def compose_depth(fg_depth, fg_mask, n_src, bg, depth_max=DEPTH_MAX):
    """z-buffer composition of up to S source frames per output frame (:231-266).
    fg_depth [B,S,H,W] float16/float32 metres, fg_mask [B,S,H,W] uint8 (0/1), n_src [B] int32 (sources used), bg [B,H,W]
    (same dtype as fg_depth) -> [B,H,W] float32."""
    for t, n in ((fg_depth, "fg_depth"), (fg_mask, "fg_mask"), (n_src, "n_src"), (bg, "bg")):
        _lib.require_cuda_tensor(t, n)
    if fg_depth.dtype not in (torch.float16, torch.float32) or bg.dtype != fg_depth.dtype:
        raise _lib.PopnetError("compose_depth: fg_depth / bg must both be float16 or float32")
    if fg_mask.dtype != torch.uint8 or n_src.dtype != torch.int32:
        raise _lib.PopnetError("compose_depth: fg_mask must be uint8 and n_src int32")
    B, S, H, W = fg_depth.shape
    if fg_mask.shape != fg_depth.shape or bg.shape != (B, H, W) or n_src.shape != (B,):
        raise _lib.PopnetError("compose_depth: shape mismatch")
    fg_depth, fg_mask, bg, n_src = fg_depth.contiguous(), fg_mask.contiguous(), bg.contiguous(), n_src.contiguous()
    out = torch.empty((B, H, W), dtype=torch.float32, device=fg_depth.device)
    ctx = _ctx(fg_depth.device)
    dt = _lib.PN_DEPTH_F16 if fg_depth.dtype == torch.float16 else _lib.PN_DEPTH_F32
    ctx.check(_lib.lib().pn_compose_depth(ctx.handle, C.c_void_p(fg_depth.data_ptr()), C.c_void_p(fg_mask.data_ptr()),
                                          C.c_void_p(n_src.data_ptr()), C.c_void_p(bg.data_ptr()), dt, B, S, H, W, float(depth_max),
                                          C.c_void_p(out.data_ptr()), _lib.current_stream_ptr(fg_depth.device)), "pn_compose_depth")
    return out


def rasterize_targets(kp2d, kp_z, n_persons, depth_resize, input_size=224, stride=8, z_radius=2, sigma=7.0):
    """get_ground_truth for a batch.  kp2d [B,P,15,2] float32 (network-input pixels), kp_z [B,P,15] float64 (metres), n_persons [B]
    int32, depth_resize [B,h,w] float32 -> (heat [B,16,h,w], paf [B,28,h,w], z [B,15,h,w], fg [B,15,h,w]) float32."""
    for t, n in ((kp2d, "kp2d"), (kp_z, "kp_z"), (n_persons, "n_persons"), (depth_resize, "depth_resize")):
        _lib.require_cuda_tensor(t, n)
    if kp2d.dtype != torch.float32 or kp_z.dtype != torch.float64 or n_persons.dtype != torch.int32 or depth_resize.dtype != torch.float32:
        raise _lib.PopnetError("rasterize_targets: kp2d float32, kp_z float64, n_persons int32, depth_resize float32")
    B, P = kp2d.shape[0], kp2d.shape[1]
    g = int(input_size / stride)
    if kp2d.shape != (B, P, NUM_JOINTS, 2) or kp_z.shape != (B, P, NUM_JOINTS) or n_persons.shape != (B,) or depth_resize.shape != (B, g, g):
        raise _lib.PopnetError("rasterize_targets: shape mismatch")
    dev = kp2d.device
    if P == 0:      # nobody annotated anywhere in the batch: one dummy slot that n_persons = 0 never reads
        kp2d, kp_z, P = torch.zeros((B, 1, NUM_JOINTS, 2), dtype=torch.float32, device=dev), torch.zeros((B, 1, NUM_JOINTS), dtype=torch.float64, device=dev), 1
    kp2d, kp_z, n_persons, depth_resize = kp2d.contiguous(), kp_z.contiguous(), n_persons.contiguous(), depth_resize.contiguous()
    heat = torch.empty((B, NUM_JOINTS + 1, g, g), dtype=torch.float32, device=dev)
    paf = torch.empty((B, 2 * NUM_LIMBS, g, g), dtype=torch.float32, device=dev)
    z = torch.empty((B, NUM_JOINTS, g, g), dtype=torch.float32, device=dev)
    fg = torch.empty((B, NUM_JOINTS, g, g), dtype=torch.float32, device=dev)
    cfg = target_cfg(input_size, stride, z_radius, sigma)
    ctx = _ctx(dev)
    ctx.check(_lib.lib().pn_rasterize_targets(ctx.handle, C.c_void_p(kp2d.data_ptr()), C.c_void_p(kp_z.data_ptr()), C.c_void_p(n_persons.data_ptr()),
                                              B, P, C.c_void_p(depth_resize.data_ptr()), C.byref(cfg), C.c_void_p(heat.data_ptr()),
                                              C.c_void_p(paf.data_ptr()), C.c_void_p(z.data_ptr()), C.c_void_p(fg.data_ptr()),
                                              _lib.current_stream_ptr(dev)), "pn_rasterize_targets")
    return heat, paf, z, fg


def _resize(frames, S, depth_max):
    """cv2.resize(INTER_LINEAR) to S x S + the [0, depth_max] clamp, un-normalised (pn_preprocess with mean 0, std 1)."""
    B, H, W = frames.shape
    out = torch.empty((B, 1, S, S), dtype=torch.float32, device=frames.device)
    ctx = _ctx(frames.device)
    ctx.check(_lib.lib().pn_preprocess(ctx.handle, C.c_void_p(frames.data_ptr()), _lib.PN_DEPTH_F32, B, H, W, C.c_void_p(out.data_ptr()), S,
                                       float(depth_max), 0.0, 1.0, _lib.current_stream_ptr(frames.device)), "pn_preprocess")
    return out[:, 0]


def mpaug_batch(fg_depth, fg_mask, n_src, bg, kp2d_org, kp3d, n_persons, input_size=224, stride=8, z_radius=2):
    """A batch of training items as KDH3D_Keypoints.__getitem__ builds them for given source choices (evaluation transform
    Compose([Cvt2ndarray, Resize])).  kp2d_org [B,P,15,2] float32 in ORIGINAL pixel coordinates, kp3d [B,P,15,3] float64 metres.
    Returns image [B,1,S,S] (normalised), heat, paf, z, fg."""
    _lib.require_cuda_tensor(kp2d_org, "kp2d_org")
    _lib.require_cuda_tensor(kp3d, "kp3d")
    image = compose_depth(fg_depth, fg_mask, n_src, bg)
    H, W = image.shape[1:]
    img224 = _resize(image, input_size, DEPTH_MAX)                       # Resize + the clamp of __getitem__ :277-278
    depth_resize = _resize(img224, int(input_size / stride), 3.0e38)     # cv2.resize of the clamped input (:347); no second clamp there
    kp = kp2d_org.to(torch.float32).clone()
    kp[..., 0] *= float(input_size) / W                                  # Resize.__call__: float32 array times a Python float
    kp[..., 1] *= float(input_size) / H
    heat, paf, z, fg = rasterize_targets(kp, kp3d[..., 2].to(torch.float64).contiguous(), n_persons, depth_resize, input_size, stride, z_radius)
    x = ((img224 - float(DEPTH_MEAN)) / float(DEPTH_STD)).unsqueeze(1)
    return x, heat, paf, z, fg


AUG_MODS = [[0, 3], [1, 2], [0, 1], [2, 3], [4]]    # datasets_kdh3d_rtpose_mpaug.py:51 -- which annotation sets may share a frame


class MPAugSampler:
    """The random control flow of KDH3D_Keypoints.__getitem__ (:231-262) on the host: which source frames join item `index`.
    The dataset holds several annotation sets (one list of ids each); a random entry of aug_mods names the sets that may
    contribute, each joins with probability 0.8 (`uniform(0, 1) > 0.8: continue`), set ii contributes its frame
    index % len(ids[ii]); when nobody joined one random set is taken; the background is index % n_backgrounds.  Draws from
    Python's `random` in the reference's order (randint, then one uniform per candidate, then randint), so that a seeded run
    picks the same sources as the reference does."""

    def __init__(self, set_sizes, n_backgrounds, aug_mods=AUG_MODS, p_join=0.8):
        self.set_sizes, self.n_backgrounds, self.aug_mods, self.p_join = list(set_sizes), int(n_backgrounds), [list(m) for m in aug_mods], p_join
        self.max_sources = max(len(m) for m in self.aug_mods)

    def sources(self, index):
        """-> ([(set, frame-in-set), ...], background id)"""
        mod = self.aug_mods[random.randint(0, len(self.aug_mods) - 1)]
        src = []
        for ii in mod:
            if random.uniform(0, 1) > self.p_join:
                continue
            src.append((ii, index % self.set_sizes[ii]))
        if not src:
            ii = random.randint(0, len(self.set_sizes) - 1)
            src.append((ii, index % self.set_sizes[ii]))
        return src, index % self.n_backgrounds

    def batch(self, indices):
        """-> (src [B,S,2] int64 (set, frame) padded with -1, n_src [B] int32, bg [B] int64) for the caller's gather."""
        picks = [self.sources(i) for i in indices]
        src = np.full((len(picks), self.max_sources, 2), -1, dtype=np.int64)
        for r, (s, _) in enumerate(picks):
            src[r, :len(s)] = s
        return src, np.array([len(s) for s, _ in picks], dtype=np.int32), np.array([b for _, b in picks], dtype=np.int64)


class MPAugTrainSet:
    """The files behind KDH3D_Keypoints of the mpaug trainer (datasets_kdh3d_rtpose_mpaug.py:166-221 (CR)): several annotation
    sets (`labels_*.json`: image id -> list of persons with `2d_joints` / `3d_joints`, plus `intrinsics`), depth frames and
    foreground masks as `<id>` .npy files under img_dir / seg_dir, background frames listed in bg_file.  `batch(indices)` draws
    the sources like the reference (MPAugSampler), loads them on the host and returns DEVICE tensors ready for mpaug_batch:
    (fg_depth [B,S,H,W], fg_mask [B,S,H,W] uint8, n_src [B], bg [B,H,W], kp2d_org [B,P,15,2], kp3d [B,P,15,3], n_persons [B]).
    The id lists are shuffled once at construction like the reference's (random.shuffle)."""

    def __init__(self, img_dir, ann_file_list, bg_file, bg_dir, seg_dir, device="cuda:0", shuffle=True):
        import json
        self.img_dir, self.bg_dir, self.seg_dir, self.device = img_dir, bg_dir, seg_dir, torch.device(device)
        self.annos, self.ids = [], []
        for f in ann_file_list:
            a = json.load(open(f, "r"))
            ids = [k for k in a if k != "intrinsics"]
            if shuffle:
                random.shuffle(ids)
            self.annos.append(a)
            self.ids.append(ids)
        self.bg = list(json.load(open(bg_file, "r")).values())
        if shuffle:
            random.shuffle(self.bg)
        self.sampler = MPAugSampler([len(i) for i in self.ids], len(self.bg), aug_mods=[m for m in AUG_MODS if max(m) < len(self.ids)] or [[0]])
        self.max_sources = self.sampler.max_sources

    def __len__(self):
        return max(len(i) for i in self.ids)              # dataset_len (:181)

    def batch(self, indices):
        import os
        src, n_src, bg_id = self.sampler.batch(indices)
        B, S = len(indices), self.max_sources
        frames, masks, bgs, persons = [], [], [], []
        for b in range(B):
            fr, mk, pp = [], [], []
            for s in range(int(n_src[b])):
                ii, f = int(src[b, s, 0]), int(src[b, s, 1])
                image_id = self.ids[ii][f]
                fr.append(np.load(os.path.join(self.img_dir, image_id)))
                mk.append(np.load(os.path.join(self.seg_dir, image_id)))
                pp += self.annos[ii][image_id]
            bgs.append(np.load(os.path.join(self.bg_dir, self.bg[int(bg_id[b])]["file_name"])))
            frames.append(fr)
            masks.append(mk)
            persons.append(pp)
        H, W = bgs[0].shape
        dt = np.float16 if bgs[0].dtype == np.float16 and all(f.dtype == np.float16 for fr in frames for f in fr) else np.float32
        fd = np.zeros((B, S, H, W), dtype=dt)
        fm = np.zeros((B, S, H, W), dtype=np.uint8)
        P = max(1, max(len(p) for p in persons))
        k2 = np.zeros((B, P, NUM_JOINTS, 2), dtype=np.float32)
        k3 = np.zeros((B, P, NUM_JOINTS, 3), dtype=np.float64)
        npers = np.zeros(B, dtype=np.int32)
        for b in range(B):
            for s, (f, m) in enumerate(zip(frames[b], masks[b])):
                fd[b, s], fm[b, s] = f, (np.asarray(m) > 0)
            npers[b] = len(persons[b])
            for p, ann in enumerate(persons[b]):
                k2[b, p] = np.asarray(ann["2d_joints"], dtype=np.float32)
                k3[b, p] = np.asarray(ann["3d_joints"], dtype=np.float64)
        dev = self.device
        t = lambda a: torch.from_numpy(a).to(dev, non_blocking=True)      # noqa: E731
        return t(fd), t(fm), t(n_src), t(np.stack(bgs).astype(dt)), t(k2), t(k3), t(npers)
