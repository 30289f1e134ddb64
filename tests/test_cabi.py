"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads through ctypes and exports
every symbol include/iggt_hip.h declares (no compute: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from iggt_official_amd import build_ext

    return ctypes.CDLL(build_ext.build(verbose=False))


def _declared():
    src = open(os.path.join(ROOT, "include", "iggt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long)\s+(iggt_\w+)\s*\(", src)))


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/iggt_hip.h but not exported"


def test_binding_matches_header():
    from iggt_official_amd import _C

    assert sorted(_C.exported_symbols()) == _declared()
    src = open(os.path.join(ROOT, "include", "iggt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for name, argtypes in _C._SIGNATURES.items():
        m = re.search(r"\b(?:int|long)\s+" + name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(argtypes), (name, len(params), len(argtypes))


def test_abi_version_and_argument_contract(lib):
    from iggt_official_amd import _C

    assert lib.iggt_hip_abi_version() == _C.ABI_VERSION
    # argument-contract violations are rejected on the host before any launch (negative code)
    lib.iggt_gemm_bf16.restype = ctypes.c_int
    rc = lib.iggt_gemm_bf16(None, ctypes.c_long(8), None, ctypes.c_long(8), 4, 4, 63, None, None, None, None,
                            ctypes.c_long(4), 1, 0, 0, 0, 0, 0, None)
    assert rc < 0


def test_product_raises_without_gpu():
    """No silent CPU fallback: CPU tensors must raise."""
    import torch

    from iggt_official_amd import _C

    with pytest.raises(_C.HipExtensionError):
        _C.gemm_bf16(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(4, 64, dtype=torch.bfloat16),
                     torch.zeros(4, 4))


def test_track_head_raises_without_gpu():
    """The query_points path has no CPU fallback either: tracker and sampling kernels refuse CPU tensors."""
    import torch

    from iggt.heads.track_modules.base_track_predictor import BaseTrackerPredictor
    from iggt.heads.track_modules.blocks import CorrBlock
    from iggt_official_amd import _C

    tr = BaseTrackerPredictor(stride=2, corr_levels=7, hidden_size=384, latent_dim=128).eval()
    with pytest.raises(_C.HipExtensionError):
        tr(query_points=torch.zeros(1, 3, 2), fmaps=torch.zeros(1, 2, 128, 64, 64), iters=1)
    with pytest.raises(_C.HipExtensionError):
        CorrBlock(torch.zeros(2, 64, 64, 128), num_levels=7, radius=4)
    with pytest.raises(_C.HipExtensionError):
        tr.updateformer(torch.zeros(1, 3, 2, 388))


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the product packages may reference it."""
    bad = []
    for pkg in ("iggt_official_amd", "iggt"):
        for d, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(d, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_attention_dispatcher_tile_choices(lib):
    """Which instantiation the attention dispatcher picks per shape (host logic of csrc/attention.hip, no GPU needed; 256 CUs
    assumed without a device): 256-row tiles with 128-key macro tiles for deep grids -- the global attention, and since round 3 the
    frame / DINOv2 attention of 32 views although its 1 374-token sequences pad 12 % -- 128-row tiles for the few-view frame
    attention, key ranges + combine for the per-rank global attention of an 8-GPU run."""
    from iggt_official_amd import _C

    lab = lambda B, Nq, Nk, sb: _C.attn_kernel_label(B, 16, Nq, Nk, "f16", static_bound=sb, with_part_ws=True)
    assert "QB=2,KVM=2" in lab(1, 43968, 43968, True) and "static-bound" in lab(1, 43968, 43968, True)
    assert "QB=2,KVM=2" in lab(32, 1374, 1374, True) and "QB=2,KVM=2" in lab(32, 1370, 1370, False)
    assert "QB=1,KVM=1" in lab(4, 1374, 1374, True) and "QB=1,KVM=1" in lab(8, 1374, 1374, False)
    per_rank = lab(1, 5496, 43968, True)
    assert "key ranges" in per_rank and "attn_combine_kernel" in per_rank
    assert "online-max" in lab(1, 5496, 43968, False)


def test_estimated_shift_workspace_layout_matches_the_python_views(lib):
    """The typed views the tests, probes and reports take into an est_ws buffer (_C.static_attn_est_views) must end exactly
    where iggt_flash_attn_static_est_ws_bytes says the buffer ends (csrc/attention_common.h est_offsets), for ragged and
    aligned shapes; the slot table of the 256-row kernel covers whole tiles."""
    import torch

    from iggt_official_amd import _C

    for B, H, Nq, Nk in [(1, 16, 43968, 43968), (1, 16, 5496, 43968), (2, 4, 300, 700), (3, 16, 1374, 1374), (1, 1, 1, 1)]:
        n = _C.static_attn_est_ws_bytes(B, H, Nq, Nk)
        v = _C.static_attn_est_views(torch.zeros(n, dtype=torch.uint8), B, H, Nq, Nk)
        assert v["bytes"] == n, (B, H, Nq, Nk, v["bytes"], n)
        assert v["slotrow"].shape == (B * H, (Nq + 255) // 256 * 256)
        assert v["rowshift"].shape == (B * H, Nq) and v["rowflag"].shape[1] % 16 == 0
