"""Constants of the path, with the reference file:line each one comes from."""
from types import SimpleNamespace

# MP-3DHP ("kdh3d") dataset constants: util/util_functions.py:4,11-13
INTRINSICS = {'fx': 504.1189880371094, 'fy': 504.042724609375, 'cx': 231.7421875, 'cy': 320.62640380859375}
DEPTH_MEAN, DEPTH_STD, DEPTH_MAX = 3, 2, 6
NUM_PARTS, NUM_LIMBS = 15, 14

# keypoint order: util/util_functions.py:37-55
KEYPOINTS = ['head', 'neck', 'right_shoulder', 'left_shoulder', 'right_elbow', 'left_elbow', 'right_wrist',
             'left_wrist', 'torso', 'right_hip', 'left_hip', 'right_knee', 'left_knee', 'right_ankle', 'left_ankle']
# limbs (src -> dst), PAF channels (2l, 2l+1): util/util_functions.py:17-34
LIMBS = [[8, 9], [9, 11], [11, 13], [8, 10], [10, 12], [12, 14], [8, 1], [1, 2], [2, 4], [4, 6], [1, 3], [3, 5], [5, 7], [1, 0]]

YOLO_ANCHORS = [(6., 3.), (12., 6.)]      # tpm/evaluate/evaluation_yolo_posenet_kdh3d_mpreal.py:47


def default_cfg():
    """The five yacs keys the hot path reads (tpm/lib/config/default.py:40-41,126-128), as set by
    the evaluation script (evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py:103-111)."""
    return SimpleNamespace(
        MODEL=SimpleNamespace(DOWNSAMPLE=8, NUM_KEYPOINTS=15, NUM_LIMBS=14, NUM_STAGES=2, IMAGE_SIZE=[224, 224]),
        TEST=SimpleNamespace(THRESH_HEATMAP=0.1, THRESH_PAF=0.05, NUM_INTERMED_PTS_BETWEEN_KEYPOINTS=10))
