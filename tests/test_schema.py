"""State-dict contract: the HIP model exposes exactly the reference's parameter/buffer names and shapes
(tests/golden/state_dict_schema.json was dumped from the reference IGGT by oracle/make_golden.py)."""
import torch


def test_state_dict_matches_reference_schema(schema):
    from iggt.models.vggt import IGGT

    with torch.device("meta"):
        model = IGGT()
    mine = {k: (list(v.shape), str(v.dtype)) for k, v in model.state_dict().items()}
    ref = {k: (v["shape"], v["dtype"]) for k, v in schema.items()}
    assert sorted(mine) == sorted(ref)            # all 2 053 tensors, track_head.* (394 of them) included
    for k in ref:
        assert mine[k] == ref[k], (k, mine[k], ref[k])
    assert sum(1 for k in mine if k.startswith("track_head.")) == 394 and len(mine) == 2053


def test_relative_position_buffers_match_reference():
    import os

    from conftest import GOLDEN
    from iggt_official_amd.heads.window_sa import SwinCA, SwinSA

    ints = torch.load(os.path.join(GOLDEN, "int_buffers.pt"), weights_only=False)
    sa = SwinSA(img_size=512, out_chans=128, embed_dim=128, num_heads=4, window_size=8)
    ca = SwinCA(img_size=128, out_chans=256, embed_dim=256, num_heads=4, window_size=8)
    assert torch.equal(sa.relative_position_index_SA, ints["part_head.window_self_atten.relative_position_index_SA"])
    assert torch.equal(ca.relative_position_index_OCA,
                       ints["part_head.window_cross_attention.relative_position_index_OCA"])
    assert torch.equal(ca.relative_position_index_SA,
                       ints["part_head.window_cross_attention.relative_position_index_SA"])


def test_alias_package_falls_through_to_reference_for_off_path_modules():
    """demo.py imports iggt.models.vggt and iggt.utils.{load_fn,pose_enc,geometry} (provided here) AND other iggt.utils.*
    modules (not provided: off the hot path).  With this repository in front of a reference checkout on PYTHONPATH the first
    group resolves here, the rest to the reference -- and the functions of pose_enc.py this repository does not re-implement
    are filled in from the reference's module of the same name."""
    import os
    import subprocess
    import sys

    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "iggt", "utils")):
        import pytest

        pytest.skip("no reference checkout on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import iggt.models.vggt as v, iggt.utils.rotation as rot, iggt.heads.dpt_head as d, iggt.utils.pose_enc as pe, "
            "utils.model as um;"
            "print(v.__file__); print(rot.__file__); print(d.__file__); print(pe.__file__); print(um.__file__);"
            "print(pe.pose_encoding_to_extri_intri.__module__); print(pe.extri_intri_to_pose_encoding.__module__)")
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + ref)
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()[-7:]
    assert lines[0].startswith(root) and lines[2].startswith(root) and lines[3].startswith(root) and lines[4].startswith(root)
    assert lines[1].startswith(ref), lines
    assert lines[5] == "iggt_official_amd.utils.pose_enc", lines       # the hot-path function: this repository's
    assert lines[6] == "iggt.utils._reference_pose_enc", lines                 # an off-path one: the reference's, filled in
