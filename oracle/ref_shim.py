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


def _is_reference(mod) -> bool:
    """True when `mod` (the `iggt` package object) resolves inside REF_ROOT and nowhere else."""
    paths = [os.path.realpath(p) for p in list(getattr(mod, "__path__", []) or [])]
    root = os.path.realpath(os.path.join(REF_ROOT, "iggt"))
    return bool(paths) and all(p == root for p in paths)


def install():
    """Make `import iggt...` resolve to the REFERENCE tree and to nothing else.  Idempotent.

    The repository ships its own regular package `iggt/` (the drop-in alias of the product), which wins over the
    reference's namespace package whatever the order of sys.path.  So the top-level `iggt` module is built here by
    hand with `submodule_search_locations = [REF_ROOT/iggt]`: every `iggt.*` import below it can only be served from
    the reference checkout.  `assert_reference()` re-checks that after the model classes are imported."""
    import importlib.machinery

    import torch

    if "iggt" in sys.modules and _is_reference(sys.modules["iggt"]):
        return
    # a product-side `iggt` alias package may already be imported: drop it (and every sub-module)
    for k in [k for k in sys.modules if k == "iggt" or k.startswith("iggt.")]:
        del sys.modules[k]
    spec = importlib.machinery.ModuleSpec("iggt", None, is_package=True)
    spec.submodule_search_locations = [os.path.join(REF_ROOT, "iggt")]
    sys.modules["iggt"] = importlib.util.module_from_spec(spec)
    # inspect.getmodule() (torch.library calls it while the reference imports) keeps a file-name cache per module name; an
    # `iggt*` entry left by the product alias makes it call getfile() on the reference's file-less namespace packages
    import inspect

    cache = getattr(inspect, "_filesbymodname", None)
    if isinstance(cache, dict):
        for k in [k for k in cache if k == "iggt" or k.startswith("iggt.")]:
            del cache[k]
    if REF_ROOT not in sys.path:
        sys.path.append(REF_ROOT)   # for the reference's own top-level helpers; `iggt` itself no longer uses sys.path

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


def uninstall():
    """Forget the reference modules so that `import iggt` resolves to the product alias again."""
    for k in [k for k in sys.modules if k == "iggt" or k.startswith("iggt.")]:
        del sys.modules[k]
    if REF_ROOT in sys.path:
        sys.path.remove(REF_ROOT)


def assert_reference(obj):
    """Raise unless the class / module `obj` was loaded from the reference checkout (never from this repository)."""
    import inspect

    f = os.path.realpath(inspect.getsourcefile(obj) or "")
    root = os.path.realpath(REF_ROOT) + os.sep
    if not f.startswith(root):
        raise RuntimeError(f"oracle/ref_shim: {obj!r} was loaded from {f}, not from the reference tree {root}")


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

        assert_reference(IGGT)
        model = IGGT().eval()
        for sub in (model.aggregator, model.camera_head, model.point_head, model.part_adaptor, model.part_head,
                    model.aggregator.patch_embed, model.aggregator.frame_blocks[0].attn):
            assert_reference(type(sub))
    finally:
        torch.nn.init.trunc_normal_ = saved
    return model
