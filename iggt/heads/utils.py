from iggt_official_amd.heads.utils import *  # noqa: F401,F403
