import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pkg(sub=None):
    """Import `3dgan_amd[.sub]` (the package name is not a valid identifier)."""
    return importlib.import_module('3dgan_amd' + ('.' + sub if sub else ''))


@pytest.fixture(scope='session')
def tdg():
    return pkg
