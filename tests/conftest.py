import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def wad_path():
    from util import ensure_wad
    return ensure_wad()


@pytest.fixture(scope='session')
def oracle_levels(wad_path):
    """Oracle-built levels, cached per session: index -> BuiltLevel (oracle/wad_oracle.py)."""
    from oracle import wad_oracle
    from util import META_PATH
    cache = {}

    def get(index):
        if index not in cache:
            cache[index] = wad_oracle.build_level(wad_path, META_PATH, index)
        return cache[index]
    return get
