"""`import config` / `from config import get_config` of the reference scripts -> yolact_minimal_amd.config (same names)."""
import sys

import yolact_minimal_amd.config as _impl

sys.modules[__name__] = _impl
