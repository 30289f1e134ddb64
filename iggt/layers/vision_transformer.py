from iggt_official_amd.layers.vision_transformer import *  # noqa: F401,F403
