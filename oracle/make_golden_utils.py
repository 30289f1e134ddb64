#!/usr/bin/env python3
"""Golden vectors for the pose / geometry utilities, produced by the REFERENCE functions (build container only).

TEST INFRASTRUCTURE.  Usage: python oracle/make_golden_utils.py  ->  tests/golden/utils_pose_geometry.pt
Imports /root/reference/iggt/utils/pose_enc.py and geometry.py through oracle/ref_shim.py; geometry.py pulls in
iggt.utils.misc / iggt.utils.device (cv2, torch_geometric, ...: not installed) only for helpers the functions on this path
never call, so those two modules are replaced by inert stubs first."""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "utils_pose_geometry.pt")


def main():
    ref_shim.install()
    import iggt.utils  # noqa: F401  (namespace package of the reference)
    for name, attrs in (("iggt.utils.misc", ("invalid_to_zeros", "invalid_to_nans")), ("iggt.utils.device", ("to_numpy",))):
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, lambda *x, **k: (_ for _ in ()).throw(RuntimeError("stub")))
        sys.modules[name] = m
    from iggt.utils.geometry import closed_form_inverse_se3, depth_to_world_coords_points, unproject_depth_map_to_point_map
    from iggt.utils.pose_enc import pose_encoding_to_extri_intri

    for f in (pose_encoding_to_extri_intri, unproject_depth_map_to_point_map):
        ref_shim.assert_reference(f)
    g = torch.Generator().manual_seed(7)
    S, H, W = 5, 28, 42
    pose = torch.randn(1, S, 9, generator=g)
    pose[..., 3:7] = torch.nn.functional.normalize(pose[..., 3:7], dim=-1) * (1 + 0.2 * torch.rand(1, S, 1, generator=g))
    pose[..., 7:] = 0.5 + torch.rand(1, S, 2, generator=g)          # fields of view 0.5 .. 1.5 rad
    extri, intri = pose_encoding_to_extri_intri(pose, (H, W))
    extri_only, none = pose_encoding_to_extri_intri(pose, None, build_intrinsics=False)
    assert none is None and torch.equal(extri_only, extri)
    depth = torch.rand(S, H, W, 1, generator=g) * 5 + 0.2
    depth[0, :3, :3] = 0.0                                            # invalid depths stay finite
    world = unproject_depth_map_to_point_map(depth, extri[0], intri[0])
    w0, c0, m0 = depth_to_world_coords_points(depth[1, ..., 0].numpy(), extri[0, 1].numpy(), intri[0, 1].numpy())
    inv = closed_form_inverse_se3(extri[0].numpy())
    torch.save(dict(pose=pose, H=H, W=W, extri=extri, intri=intri, depth=depth, world=torch.from_numpy(world),
                    frame1_world=torch.from_numpy(w0), frame1_cam=torch.from_numpy(c0), frame1_mask=torch.from_numpy(m0),
                    inv_se3=torch.from_numpy(inv), numpy=np.__version__, torch=torch.__version__), OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; world dtype", world.dtype)


if __name__ == "__main__":
    main()
