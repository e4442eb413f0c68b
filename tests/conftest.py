import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(fname):
    """tests/golden/<fname> -> {case: {key: ndarray}} (fixtures made by tests/golden/make_golden.py)."""
    z = np.load(os.path.join(GOLDEN, fname))
    cases = {}
    for full in z.files:
        tag, key = full.split('/', 1)
        cases.setdefault(tag, {})[key] = z[full]
    return cases


def rel_err(a, b):
    """max|a-b| / max|b| -- the parity measure used throughout."""
    a = np.asarray(a)
    b = np.asarray(b)
    den = float(np.max(np.abs(b))) if b.size else 0.0
    num = float(np.max(np.abs(a - b))) if b.size else 0.0
    return num / den if den > 0 else num


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(fname):
        if fname not in cache:
            cache[fname] = load_golden(fname)
        return cache[fname]
    return get


def free_port():
    """A TCP port nobody listens on right now (a fixed rendezvous port can still be in TIME_WAIT from the previous run)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]
