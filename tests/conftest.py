"""pytest configuration: markers and import paths.

The product package lives in a directory whose (mandated) name is not a Python identifier, so it is put on
``sys.path`` here; it then provides the reference-compatible ``src`` package (``src.model.nets.RefineNet``,
``src.main`` ...) and the ``hipvsr`` engine package.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
