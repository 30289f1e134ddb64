from iggt_official_amd.models.vggt import *  # noqa: F401,F403
