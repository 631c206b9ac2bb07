import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible, so a plain
    # `pytest tests/` in the CPU container stays green.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session', autouse=True)
def _built_libraries():
    """Test infrastructure: make sure libgist_hip.so (hipcc cross-compiles without a GPU) and
    the oracle's C helper exist and are not older than their sources.  The PRODUCT loader
    (gist_amd/_lib.py) never builds anything: a missing library is a loud error there."""
    try:
        from gist_amd import build as hip_build
        hip_build.build()
    except Exception as e:          # surface as a failing export test, not a collection error
        print('WARNING: could not build libgist_hip.so: %r' % (e,))
    try:
        from oracle import build as oracle_build
        oracle_build.build()
    except Exception as e:
        print('WARNING: could not build the oracle helper: %r' % (e,))
    yield


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
