"""Top-level `utils` package of the reference checkout as demo.py imports it (`from utils.model import
align_and_update_state_dicts`, demo.py:51): `utils.model` resolves here, any other sub-module to the reference behind this
repository on sys.path."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
