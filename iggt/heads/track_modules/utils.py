from iggt_official_amd.heads.track_modules.utils import *  # noqa: F401,F403
