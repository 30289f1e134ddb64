from iggt_official_amd.heads.adaptor import *  # noqa: F401,F403
