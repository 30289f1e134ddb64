"""CPU restatement of the callers' side of the forward path (TEST INFRASTRUCTURE, see oracle/__init__.py):
pose decoding, depth unprojection and image loading, each following the reference line by line in plain numpy / torch / PIL.
Pinned against outputs of the REFERENCE functions by tests/test_utils_golden.py (fixture tests/golden/utils_pose_geometry.pt,
written by oracle/make_golden_utils.py from /root/reference/iggt/utils/{pose_enc,geometry}.py).  load_fn.py itself cannot be
imported anywhere (torchvision is not installed): its restatement uses PIL exactly as the reference does and restates
torchvision's ToTensor (uint8 HWC -> float CHW / 255), SURVEY.md section 8c."""
import numpy as np
import torch
from PIL import Image


def quat_to_mat(q):
    """iggt/utils/rotation.py:14-44 (scalar-last quaternion)."""
    i, j, k, r = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def pose_encoding_to_extri_intri(pose, image_size_hw=None, build_intrinsics=True):
    """iggt/utils/pose_enc.py:104-130."""
    T, quat, fov_h, fov_w = pose[..., :3], pose[..., 3:7], pose[..., 7], pose[..., 8]
    extr = torch.cat([quat_to_mat(quat), T[..., None]], dim=-1)
    intr = None
    if build_intrinsics:
        H, W = image_size_hw
        intr = torch.zeros(pose.shape[:2] + (3, 3), dtype=pose.dtype)
        intr[..., 0, 0] = (W / 2.0) / torch.tan(fov_w / 2.0)
        intr[..., 1, 1] = (H / 2.0) / torch.tan(fov_h / 2.0)
        intr[..., 0, 2] = W / 2
        intr[..., 1, 2] = H / 2
        intr[..., 2, 2] = 1.0
    return extr, intr


def unproject_depth_map_to_point_map(depth, extri, intri):
    """iggt/utils/geometry.py:151-181 + 183-236 + 238-268 + 271-330 (numpy, frame by frame; float64 result)."""
    depth, extri, intri = (np.asarray(a) for a in (depth, extri, intri))
    out = []
    for f in range(depth.shape[0]):
        d = depth[f].squeeze(-1) if depth[f].ndim == 3 else depth[f]
        H, W = d.shape
        K = intri[f]
        u, v = np.meshgrid(np.arange(W), np.arange(H))
        cam = np.stack(((u - K[0, 2]) * d / K[0, 0], (v - K[1, 2]) * d / K[1, 1], d), axis=-1).astype(np.float32)
        R, t = extri[f][:3, :3], extri[f][:3, 3:]
        inv = np.tile(np.eye(4), (1, 1, 1))[0]
        inv[:3, :3] = R.T
        inv[:3, 3:] = -np.matmul(R.T, t)
        out.append(np.dot(cam, inv[:3, :3].T) + inv[:3, 3])
    return np.stack(out, 0)


def load_and_preprocess_images(paths, mode="crop", resize_target_size=None):
    """iggt/utils/load_fn.py:12-128 with torchvision's ToTensor restated."""
    target = 518
    images, shapes = [], set()
    for p in paths:
        img = Image.open(p)
        if img.mode == "RGBA":
            img = Image.alpha_composite(Image.new("RGBA", img.size, (255, 255, 255, 255)), img)
        img = img.convert("RGB")
        width, height = img.size
        if mode == "pad":
            if width >= height:
                new_w, new_h = target, round(height * (target / width) / 14) * 14
            else:
                new_h, new_w = target, round(width * (target / height) / 14) * 14
        elif mode == "resize":
            new_w, new_h = resize_target_size
        else:
            new_w, new_h = target, round(height * (target / width) / 14) * 14
        img = img.resize((new_w, new_h), Image.Resampling.BICUBIC)
        t = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float().div(255)     # ToTensor
        if mode == "crop" and new_h > target:
            y0 = (new_h - target) // 2
            t = t[:, y0:y0 + target, :]
        elif mode == "pad":
            hp, wp = target - t.shape[1], target - t.shape[2]
            if hp > 0 or wp > 0:
                t = torch.nn.functional.pad(t, (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2), value=1.0)
        shapes.add((t.shape[1], t.shape[2]))
        images.append(t)
    if len(shapes) > 1:
        mh, mw = max(s[0] for s in shapes), max(s[1] for s in shapes)
        images = [torch.nn.functional.pad(t, ((mw - t.shape[2]) // 2, (mw - t.shape[2]) - (mw - t.shape[2]) // 2,
                                               (mh - t.shape[1]) // 2, (mh - t.shape[1]) - (mh - t.shape[1]) // 2), value=1.0)
                  for t in images]
    return torch.stack(images)
