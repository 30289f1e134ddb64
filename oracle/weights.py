"""Seeded synthetic weights / images used to pin the oracle (TEST INFRASTRUCTURE, see oracle/__init__.py).

The generator itself lives in iggt_official_amd/synthetic.py (a dependency-free integer hash) because bench.py and
__graft_entry__.smoke() feed the product model with the very same synthetic checkpoint the reference fixtures were
produced with, and the product side may not import from oracle/.  This module re-exports it under the name the
fixture generator and the tests have always used.
"""
from iggt_official_amd.synthetic import (  # noqa: F401
    _is_conv_transpose, _name_seed, fill_state_dict, hash_uniform, make_images, make_tensor)
