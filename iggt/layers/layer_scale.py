from iggt_official_amd.layers.blocks import LayerScale  # noqa: F401
