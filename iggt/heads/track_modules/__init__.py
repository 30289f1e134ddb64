

# Sub-modules this repository does not provide (iggt.utils, iggt.datasets, iggt.metrics, iggt.heads.track_head, ...:
# everything off the forward hot path that demo.py imports) resolve to the reference checkout further down sys.path;
# modules that exist here win because this directory comes first in __path__.
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
