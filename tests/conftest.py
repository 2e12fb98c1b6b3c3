import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


FORKSERVER = None


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')
    # Multi-process GPU tests start their ranks from a fork server that is created HERE, before anything in
    # this process has touched the GPU: a process that has initialised HIP must not exec another program,
    # and the fork server (a clean interpreter that never touches the GPU) does the forking for it.
    global FORKSERVER
    import multiprocessing
    try:
        FORKSERVER = multiprocessing.get_context('forkserver')
        from multiprocessing import forkserver
        forkserver.ensure_running()
    except Exception:                         # platform without forkserver: those tests skip
        FORKSERVER = None


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)
