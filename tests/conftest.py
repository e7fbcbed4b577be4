import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the driver on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (build container only)')


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir('/root/reference/wavernn')
    skip_ref = pytest.mark.skip(reason='/root/reference not present on this box')
    have_gpu = None
    for item in items:
        if 'reference' in item.keywords and not have_ref:
            item.add_marker(skip_ref)
        if 'gpu' in item.keywords:
            if have_gpu is None:
                try:
                    import torch
                    have_gpu = bool(torch.cuda.is_available())
                except Exception:
                    have_gpu = False
            if not have_gpu:
                item.add_marker(pytest.mark.skip(reason='no HIP device visible (gpu tests run on the MI355X box)'))
