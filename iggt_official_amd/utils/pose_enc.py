"""Pose-encoding decode on the GPU -- reference iggt/utils/pose_enc.py:65-130 (caller: demo.py:340)."""
import torch

from .. import _C


def pose_encoding_to_extri_intri(pose_encoding, image_size_hw=None, pose_encoding_type="absT_quaR_FoV",
                                 build_intrinsics=True):
    """pose_encoding [B, S, 9] (T, quaternion xyzw, fov_h, fov_w) -> (extrinsics [B, S, 3, 4] = [R | T] camera-from-world,
    OpenCV convention; intrinsics [B, S, 3, 3] in pixels with the principal point at the image centre, or None).
    One HIP kernel (csrc/smallops.hip) instead of ~25 small tensor ops; same fp32 formulas as the reference
    (quat_to_mat, rotation.py:14-44; fy = (H/2) / tan(fov_h/2))."""
    if pose_encoding_type != "absT_quaR_FoV":
        raise NotImplementedError
    if not torch.is_tensor(pose_encoding) or not pose_encoding.is_cuda:
        raise _C.HipExtensionError("pose_encoding_to_extri_intri runs on the GPU (no CPU fallback)")
    if build_intrinsics:
        H, W = image_size_hw
    else:
        H, W = 1, 1
    return _C.pose_to_extri_intri(pose_encoding, H, W, build_intrinsics)
