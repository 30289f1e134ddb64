from iggt_official_amd.models.aggregator import *  # noqa: F401,F403
