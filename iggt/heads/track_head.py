from iggt_official_amd.heads.track_head import *  # noqa: F401,F403
