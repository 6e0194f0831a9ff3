"""`utils` of the reference checkout: the modules on the hot path (and its neighbours) are served by yolact_minimal_amd.utils,
everything else (`labelme2coco`, `pascal2coco`, ...) falls through to the checkout's own `utils/` directory."""
import importlib
import os
import sys

import yolact_minimal_amd.utils as _impl

_here = os.path.dirname(os.path.abspath(__file__))
_checkout = [os.path.join(p or os.getcwd(), 'utils') for p in sys.path
             if os.path.isdir(os.path.join(p or os.getcwd(), 'utils')) and os.path.abspath(os.path.join(p or os.getcwd(), 'utils')) != _here]
__path__ = [_here] + _checkout
for _name in ('box_utils', 'augmentations', 'coco', 'common_utils', 'timer'):
    _m = importlib.import_module(f'yolact_minimal_amd.utils.{_name}')
    sys.modules[f'{__name__}.{_name}'] = _m
    globals()[_name] = _m
