"""popnet_amd -- MI355X-native PoP-Net / MP-3DHP inference path.

``pop-net_amd`` at the repo root is a symlink to this directory (the layout name of the build
contract); import the package as ``popnet_amd``.

Public surface = the reference's own names for this path:
    popnet_amd.network.rtpose_light3d.rtpose_light3d      tpm/lib/network/rtpose_light3d.py:249
    popnet_amd.network.yolo_posenet.YoloPoseNet           tpm/lib/network/yolo_posenet.py:87
    popnet_amd.utils.paf_to_pose.paf_to_pose              tpm/lib/utils/paf_to_pose.py:354
    popnet_amd.utils.common.paf_to_human_list / retrieve_depth_heat_weighted / pos_3d_from_2d_and_depth
    popnet_amd.utils.prior_pose_align.parse_prior_pose    tpm/lib/utils/prior_pose_align.py:10
    popnet_amd.pafprocess                                 tpm/lib/pafprocess (SWIG module)
plus the batched engine popnet_amd.pipeline.PoseEngine (depth frames in, pose records out).
Everything computes through libpopnet_hip.so; there is no CPU fallback.
"""
from . import config  # noqa: F401

__all__ = ["config"]
