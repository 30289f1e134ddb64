from iggt_official_amd.heads.block import *  # noqa: F401,F403
