from iggt_official_amd.layers.blocks import Attention, MemEffAttention  # noqa: F401
