import os
import sys

import pytest

# RequestPipeline (bench.py's headline mode) overlaps requests on separate HIP streams; ROCm multiplexes streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable when the runtime starts -> set it before torch touches HIP
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session', autouse=True)
def _oracle_built():
    """The C half of the oracle is a build product (oracle/Makefile)."""
    import subprocess
    so = os.path.join(REPO, 'oracle', 'libgreedy_nms.so')
    if not os.path.exists(so):
        subprocess.check_call(['make', '-C', os.path.join(REPO, 'oracle')])
