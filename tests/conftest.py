import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tf-nas_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
# the fused per-image route of the late cells (csrc/fx_kernels.hip) is exercised by the whole GPU suite, whatever the library's default
os.environ.setdefault('TFNAS_FX', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run through gpurun)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (build container only)')


def pytest_collection_modifyitems(config, items):
    import torch
    has_gpu = torch.cuda.is_available()
    has_ref = os.path.isdir('/root/reference/models')
    for item in items:
        if 'gpu' in item.keywords and not has_gpu:
            item.add_marker(pytest.mark.skip(reason='no GPU in this container'))
        if 'reference' in item.keywords and not has_ref:
            item.add_marker(pytest.mark.skip(reason='/root/reference not present'))
