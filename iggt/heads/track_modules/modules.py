from iggt_official_amd.heads.track_modules.modules import *  # noqa: F401,F403
