"""`utils.output_utils` for the reference scripts: `nms` / `after_nms` are the HIP path; the drawing helpers stay the checkout's
own cv2 code (`/root/reference/utils/output_utils.py:276-369`, host side, out of the hot path) and are loaded from there on
first use."""
import importlib.util
import os
import sys

from yolact_minimal_amd.utils.output_utils import nms, after_nms  # noqa: F401

_host = None


def _checkout_module():
    global _host
    if _host is None:
        here = os.path.dirname(os.path.abspath(__file__))
        for p in sys.path:
            cand = os.path.join(p or os.getcwd(), 'utils', 'output_utils.py')
            if os.path.isfile(cand) and os.path.dirname(os.path.abspath(cand)) != here:
                spec = importlib.util.spec_from_file_location('_reference_output_utils', cand)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)      # needs the checkout's own dependencies (cv2, cython_nms)
                _host = mod
                break
        else:
            raise ImportError('draw_img / draw_lincomb are the reference checkout\'s host-side cv2 helpers: run from inside a '
                              'Yolact_minimal checkout (its utils/output_utils.py was not found on sys.path)')
    return _host


def draw_img(*args, **kwargs):
    return _checkout_module().draw_img(*args, **kwargs)


def draw_lincomb(*args, **kwargs):
    return _checkout_module().draw_lincomb(*args, **kwargs)
