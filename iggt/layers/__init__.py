from iggt_official_amd.layers import *  # noqa: F401,F403
from iggt_official_amd.layers import Mlp, PatchEmbed, NestedTensorBlock, MemEffAttention  # noqa: F401

# Sub-modules this repository does not provide (iggt.utils, iggt.datasets, iggt.metrics, iggt.heads.track_head, ...:
# everything off the forward hot path that demo.py imports) resolve to the reference checkout further down sys.path;
# modules that exist here win because this directory comes first in __path__.
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
