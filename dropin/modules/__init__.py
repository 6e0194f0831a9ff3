"""`modules` of the reference checkout, with `yolact`, `resnet` and `swin_transformer` served by yolact_minimal_amd.modules;
anything else falls through to the checkout's own `modules/` directory."""
import os
import sys

import yolact_minimal_amd.modules as _impl
from yolact_minimal_amd.modules import resnet, swin_transformer, yolact  # noqa: F401

_here = os.path.dirname(os.path.abspath(__file__))
__path__ = [_here] + list(_impl.__path__) + [os.path.join(p or os.getcwd(), 'modules') for p in sys.path
                                            if os.path.isdir(os.path.join(p or os.getcwd(), 'modules')) and os.path.abspath(os.path.join(p or os.getcwd(), 'modules')) != _here]
for _name in ('yolact', 'resnet', 'swin_transformer'):
    sys.modules[f'{__name__}.{_name}'] = sys.modules[f'yolact_minimal_amd.modules.{_name}']
