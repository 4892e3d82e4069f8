"""Shared pytest fixtures.  GPU tests are marked @pytest.mark.gpu; everything else runs on CPU."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def _group(npz):
    out = {}
    for key in npz.files:
        case, field = key.split('/', 1)
        out.setdefault(case, {})[field] = npz[key]
    return out


@pytest.fixture(scope='session')
def es2005a():
    return dict(np.load(os.path.join(GOLDEN, 'es2005a.npz')))


@pytest.fixture(scope='session')
def synth_cases():
    return _group(np.load(os.path.join(GOLDEN, 'synth_cases.npz')))


@pytest.fixture(scope='session')
def fb_cases():
    return _group(np.load(os.path.join(GOLDEN, 'fb_cases.npz')))


@pytest.fixture(scope='session')
def fb_dense_cases():
    return _group(np.load(os.path.join(GOLDEN, 'fb_dense_cases.npz')))
