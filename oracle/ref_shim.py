"""Import shim for the *reference* implementation (build container only).

TEST INFRASTRUCTURE.  Used only by oracle/make_golden.py (fixture generation) and by
tests that are skipped when /root/reference is absent (it never exists on the GPU box).

The reference's model files import four third-party packages that are not installed
and carry no hot-path arithmetic:
  detectron2.layers.ShapeSpec          (iggt/heads/adaptor.py:6; only in output_shape())
  sam2.modeling.position_encoding      (adaptor.py:7; result discarded at vggt.py:208)
  apex.normalization.FusedRMSNorm      (heads/block.py:38; never instantiated, qk_norm=False)
  basicsr.archs.arch_util              (window_sa.py:4; init-time helpers)
They are replaced by inert stand-ins before `iggt` is imported.
"""
import collections
import importlib.util
import itertools
import os
import sys
import types

REF_ROOT = os.environ.get("IGGT_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "iggt", "models"))


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install():
    """Make `import iggt...` resolve to the reference tree. Idempotent."""
    import torch

    if "iggt" in sys.modules and getattr(sys.modules["iggt"], "__file__", "").startswith(REF_ROOT):
        return
    # a product-side `iggt` alias package may already be imported: drop it
    for k in [k for k in sys.modules if k == "iggt" or k.startswith("iggt.")]:
        del sys.modules[k]
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    d2 = _mod("detectron2")
    d2l = _mod("detectron2.layers")
    d2l.ShapeSpec = collections.namedtuple("ShapeSpec", "channels height width stride", defaults=(None,) * 4)
    d2.layers = d2l

    apex = _mod("apex")
    apn = _mod("apex.normalization")

    class FusedRMSNorm(torch.nn.Module):  # never instantiated on the path
        def __init__(self, dim, elementwise_affine=True, eps=1e-6):
            super().__init__()
            self.eps = eps
            self.weight = torch.nn.Parameter(torch.ones(dim))

        def forward(self, x):
            return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps) * self.weight

    apn.FusedRMSNorm = FusedRMSNorm
    apex.normalization = apn

    bs = _mod("basicsr")
    bsa = _mod("basicsr.archs")
    bsu = _mod("basicsr.archs.arch_util")

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else tuple(itertools.repeat(x, 2))

    bsu.to_2tuple = to_2tuple
    bsu.trunc_normal_ = torch.nn.init.trunc_normal_
    bs.archs = bsa
    bsa.arch_util = bsu

    s2 = _mod("sam2")
    s2m = _mod("sam2.modeling")
    spec = importlib.util.spec_from_file_location(
        "sam2.modeling.position_encoding", os.path.join(REF_ROOT, "sam2", "modeling", "position_encoding.py")
    )
    pe = importlib.util.module_from_spec(spec)
    sys.modules["sam2.modeling.position_encoding"] = pe
    spec.loader.exec_module(pe)
    s2.modeling = s2m
    s2m.position_encoding = pe


def build_reference_iggt(fast_init=True):
    """Construct the reference IGGT().eval().  fast_init skips trunc_normal_ (weights are
    overwritten by oracle.weights.fill_state_dict afterwards)."""
    import torch

    install()
    saved = torch.nn.init.trunc_normal_
    if fast_init:
        torch.nn.init.trunc_normal_ = lambda t, *a, **k: t
        import iggt.layers.vision_transformer as vt

        vt.trunc_normal_ = lambda t, *a, **k: t
    try:
        from iggt.models.vggt import IGGT

        model = IGGT().eval()
    finally:
        torch.nn.init.trunc_normal_ = saved
    return model
