from iggt_official_amd.heads.part_head import *  # noqa: F401,F403
