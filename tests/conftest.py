import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# debug mode of the library for every test: each convolution launch first replays its operand addresses on the host and refuses
# (LLV_E_EXTENT) to run if one falls outside the buffers its descriptor declares (include/llv.h, llv_extents)
os.environ.setdefault('LLV_CHECK_EXTENTS', '1')
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
