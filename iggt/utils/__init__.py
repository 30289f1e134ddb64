"""`iggt.utils.*` as demo.py imports it: load_fn / pose_enc / geometry resolve to the GPU implementations in
iggt_official_amd.utils; every other sub-module (misc, visual_util, rotation, ...) to the reference checkout behind this
repository on sys.path."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
