import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # a fresh checkout has no built artefacts (they are git-ignored): build them once, exactly as
    # __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU; about two minutes)
    lib = os.path.join(ROOT, 'rust-doom_amd', 'librdoom_hip.so')
    oracle_so = os.path.join(ROOT, 'oracle', '_build', 'liboracle_raster.so')
    if not (os.path.exists(lib) and os.path.exists(oracle_so)):
        import __graft_entry__
        __graft_entry__.build()


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
