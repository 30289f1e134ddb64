from iggt_official_amd.utils.model import *  # noqa: F401,F403
from iggt_official_amd.utils.model import align_and_update_state_dicts, load_checkpoint  # noqa: F401
