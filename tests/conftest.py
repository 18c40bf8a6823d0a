import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (HIP device); run with `-m gpu` on the GPU box')


@pytest.fixture(scope='session')
def params():
    from deepcv_amd.utils import load_parameters
    return load_parameters()


@pytest.fixture(scope='session')
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    return torch.device('cuda:0')
