from iggt_official_amd.utils.pose_enc import *  # noqa: F401,F403
from iggt_official_amd.utils._fallthrough import fill_missing as _fill

_fill("iggt.utils", "pose_enc", globals())   # names this repository does not re-implement come from the reference checkout, if any
