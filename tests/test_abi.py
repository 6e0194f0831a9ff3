"""The C-ABI library loads and exports every symbol include/yolact_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from tests.conftest import REPO


def _declared():
    text = open(os.path.join(REPO, 'include', 'yolact_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ym_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_match_binding():
    from yolact_minimal_amd import hip
    assert sorted(hip.ABI_SYMBOLS) == _declared()


def test_library_exports_every_symbol():
    from yolact_minimal_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.ym_abi_version.restype = ctypes.c_int
    assert lib.ym_abi_version() == 1


def test_struct_layout_matches_c():
    """ctypes mirrors of ym_conv_desc / ym_conv_seg / ym_nms_cfg have the C sizes (x86-64 SysV)."""
    from yolact_minimal_amd import hip
    assert ctypes.sizeof(hip.ConvSeg) == 32
    assert ctypes.sizeof(hip.ConvDesc) == 40 + 13 * 4 + 4 + 3 * 32 + 6 * 4 + 16 + 8 + 44 + 8 + 4 + 4 + 4 + 6 * 8 + 8   # (... mma, bnb_relu, padding, 6 bnb_* pointers, grid_wgs + padding)
    assert hip.lib().ym_sizeof_conv_desc() == ctypes.sizeof(hip.ConvDesc)
    assert ctypes.sizeof(hip.WgradDesc) == 24 + 14 * 4 + 4 + 8 + 4 + 16 + 8   # + accumulate, row_end[2], padding, dw_seg[2], lds_buffers (+pad)
    assert ctypes.sizeof(hip.NmsCfg) == 32
    assert ctypes.sizeof(hip.WgradReduceItem) == 80


def test_product_fails_loudly_on_cpu():
    import torch
    from yolact_minimal_amd.config import build_cfg
    from yolact_minimal_amd.modules.yolact import Yolact
    from yolact_minimal_amd.utils.output_utils import nms
    net = Yolact(build_cfg('res50_coco', 'val', 64)).eval()
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError):
        nms(torch.zeros(1, 10, 81), torch.zeros(1, 10, 4), torch.zeros(1, 10, 32), torch.zeros(1, 8, 8, 32),
            [0.0] * 40, net.cfg)
