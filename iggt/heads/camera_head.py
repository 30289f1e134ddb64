from iggt_official_amd.heads.camera_head import *  # noqa: F401,F403
