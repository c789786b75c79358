import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def configs():
    from imgcomp_cvpr_amd import config_parser as cp
    ae, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
    pc, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    return ae, pc


@pytest.fixture(scope='session')
def syn_weights(configs):
    from imgcomp_cvpr_amd import weights as W
    return W.synthetic_weights(*configs)


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    return torch.device('cuda:0')
