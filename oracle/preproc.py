"""Depth-frame pre-processing (ORACLE; test infrastructure -- see oracle/__init__.py).

Restates the test-mode ``KDH3D_Keypoints.__getitem__`` image path:
  np.load(...).astype(float)            lib/datasets/datasets_kdh3d_rtpose_mpreal.py:225 (CR)
  Cvt2ndarray: .astype(np.float32)      lib/datasets/data_augmentation_2d3d.py:89
  Resize: cv2.resize(INTER_LINEAR)      lib/datasets/data_augmentation_2d3d.py:507-510
  clamp to [0, depth_max]               lib/datasets/datasets_kdh3d_rtpose_mpreal.py:238-239 (CR)
  ToTensor + Normalize(mean 3, std 2)   ...:192-194, 242 (CR)
"""
import numpy as np

from . import cv2_resize

DEPTH_MEAN, DEPTH_STD, DEPTH_MAX = 3, 2, 6


def preprocess_frame(frame, input_size=224, depth_max=DEPTH_MAX, depth_mean=DEPTH_MEAN, depth_std=DEPTH_STD):
    """frame: [H, W] float16/float32 metres  ->  float32 [1, input_size, input_size]."""
    img = np.asarray(frame).astype(np.float64).astype(np.float32)
    img = cv2_resize.resize(img, (input_size, input_size), interpolation=cv2_resize.INTER_LINEAR)
    img[img < 0] = 0
    img[img > depth_max] = depth_max
    img = (img - np.float32(depth_mean)) / np.float32(depth_std)     # torchvision Normalize: sub_().div_()
    return img[None].astype(np.float32)


def preprocess_batch(frames, **kw):
    return np.stack([preprocess_frame(f, **kw) for f in frames], 0)
