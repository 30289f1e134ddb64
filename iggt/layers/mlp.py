from iggt_official_amd.layers.blocks import Mlp  # noqa: F401
