from iggt_official_amd.heads.track_modules.blocks import *  # noqa: F401,F403
