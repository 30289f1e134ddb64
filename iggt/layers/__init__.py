from iggt_official_amd.layers import *  # noqa: F401,F403
from iggt_official_amd.layers import Mlp, PatchEmbed, NestedTensorBlock, MemEffAttention  # noqa: F401
