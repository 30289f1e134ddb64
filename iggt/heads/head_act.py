from iggt_official_amd.heads.head_act import *  # noqa: F401,F403
