"""CPU: the configuration surface (`yolact_minimal_amd.config`: every cfg class x mode x two argument variants, plus the module
constants) against snapshots of the REAL reference's config.py (oracle/make_golden_config.py)."""
import json
import os

import pytest

from oracle.make_golden_config import CFGS, VARIANTS, make_args, snapshot, jsonable
from yolact_minimal_amd import config as C

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'config.json')))


@pytest.mark.parametrize('name', CFGS)
def test_cfg_attributes_match_reference(name):
    for mode in ('train', 'val', 'detect'):
        for vi, variant in enumerate(VARIANTS):
            want = GOLD[f'{name}|{mode}|{vi}']
            got = snapshot(getattr(C, name)(make_args(name, mode, variant)))
            missing = sorted(set(want) - set(got))
            assert not missing, (name, mode, vi, 'attributes the reference has and the mirror lacks', missing)
            for k, v in want.items():
                assert got[k] == v, (name, mode, vi, k, got[k], v)


def test_module_constants_match_reference():
    for k, v in GOLD['__module__'].items():
        assert jsonable(getattr(C, k)) == v, k
