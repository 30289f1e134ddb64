from iggt_official_amd.heads.track_modules.base_track_predictor import *  # noqa: F401,F403
