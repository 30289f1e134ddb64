from iggt_official_amd.layers.patch_embed import *  # noqa: F401,F403
