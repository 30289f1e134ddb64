"""Depth unprojection on the GPU -- reference iggt/utils/geometry.py:151-330 (callers: demo.py:281,352).

The reference loops over the frames in numpy on the host (geometry.py:173-179); here one HIP kernel maps
depth [S, H, W] + [R | t] + K to world points [S, H, W, 3] (csrc/smallops.hip: fp64 arithmetic inside, like numpy's
promotion in the reference, camera coordinates rounded to fp32 at the same place).  Inputs may be numpy arrays, CPU or GPU
tensors; like the reference the functions return numpy arrays -- fp32 (the reference's fp64 arrays hold the same values to
fp32 precision); pass `as_tensor=True` to keep the result on the GPU."""
import numpy as np
import torch

from .. import _C


def _dev_tensor(a, device):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    return t.to(device=device, dtype=torch.float32).contiguous()


def _device_of(*xs):
    for x in xs:
        if torch.is_tensor(x) and x.is_cuda:
            return x.device
    if not torch.cuda.is_available():
        raise _C.HipExtensionError("depth unprojection runs on the GPU (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def unproject_depth_map_to_point_map(depth_map, extrinsics_cam, intrinsics_cam, as_tensor=False):
    """depth_map (S, H, W, 1) or (S, H, W); extrinsics (S, 3, 4) camera-from-world; intrinsics (S, 3, 3) -> (S, H, W, 3)."""
    dev = _device_of(depth_map, extrinsics_cam, intrinsics_cam)
    d = _dev_tensor(depth_map, dev)
    if d.dim() == 4:
        d = d[..., 0].contiguous()
    pts = _C.unproject_depth(d, _dev_tensor(extrinsics_cam, dev)[:, :3, :4].contiguous(), _dev_tensor(intrinsics_cam, dev))
    return pts if as_tensor else pts.cpu().numpy()


def depth_to_cam_coords_points(depth_map, intrinsic):
    """(H, W) depth + (3, 3) intrinsics -> camera coordinates (H, W, 3) fp32."""
    depth = np.asarray(depth_map.cpu() if torch.is_tensor(depth_map) else depth_map)
    K = np.asarray(intrinsic.cpu() if torch.is_tensor(intrinsic) else intrinsic)
    assert K.shape == (3, 3) and K[0, 1] == 0 and K[1, 0] == 0, "Intrinsic matrix must be 3x3 with zero skew"
    H, W = depth.shape
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    return np.stack(((u - K[0, 2]) * depth / K[0, 0], (v - K[1, 2]) * depth / K[1, 1], depth), axis=-1).astype(np.float32)


def depth_to_world_coords_points(depth_map, extrinsic, intrinsic, z_far: float = 100.0, eps=1e-8):
    """One frame: (world points (H, W, 3), camera points (H, W, 3), validity mask (H, W))."""
    if depth_map is None:
        return None, None, None
    depth = np.asarray(depth_map.cpu() if torch.is_tensor(depth_map) else depth_map)
    mask = depth > eps
    if z_far > 0:
        mask = mask & (depth < z_far)
    cam = depth_to_cam_coords_points(depth, intrinsic)
    ext = np.asarray(extrinsic.cpu() if torch.is_tensor(extrinsic) else extrinsic)
    K = np.asarray(intrinsic.cpu() if torch.is_tensor(intrinsic) else intrinsic)
    world = unproject_depth_map_to_point_map(depth[None], ext[None], K[None])[0]
    return world, cam, mask


def closed_form_inverse_se3(se3, R=None, T=None):
    """Inverse of a batch of rigid transforms (N, 4, 4) or (N, 3, 4): [R | t]^-1 = [R^T | -R^T t]; numpy in, numpy out;
    tensor in, tensor out (always 4 x 4)."""
    if se3.shape[-2:] not in ((4, 4), (3, 4)):
        raise ValueError(f"se3 must be of shape (N,4,4), got {se3.shape}.")
    R = se3[:, :3, :3] if R is None else R
    T = se3[:, :3, 3:] if T is None else T
    if isinstance(se3, np.ndarray):
        Rt = np.transpose(R, (0, 2, 1))
        out = np.tile(np.eye(4), (len(R), 1, 1))
        out[:, :3, :3] = Rt
        out[:, :3, 3:] = -np.matmul(Rt, T)
        return out
    Rt = R.transpose(1, 2)
    out = torch.eye(4, dtype=R.dtype, device=R.device)[None].repeat(len(R), 1, 1)
    out[:, :3, :3] = Rt
    out[:, :3, 3:] = -torch.bmm(Rt, T)
    return out
