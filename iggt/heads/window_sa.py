from iggt_official_amd.heads.window_sa import *  # noqa: F401,F403
