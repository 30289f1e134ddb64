from iggt_official_amd.heads.dpt_head import *  # noqa: F401,F403
