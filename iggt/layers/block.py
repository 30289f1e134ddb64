from iggt_official_amd.layers.blocks import Block, NestedTensorBlock  # noqa: F401
