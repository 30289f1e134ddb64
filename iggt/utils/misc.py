from iggt_official_amd.utils.misc import *  # noqa: F401,F403
from iggt_official_amd.utils._fallthrough import fill_missing as _fill

_fill("iggt.utils", "misc", globals())   # names this repository does not re-implement come from the reference checkout, if any
