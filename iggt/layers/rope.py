from iggt_official_amd.layers.rope import *  # noqa: F401,F403
