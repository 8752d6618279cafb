"""The reference's own metadata file (assets/meta/doom.toml: sky table, animations, linedef specials, things) read by
the product's C++ TOML-subset reader (wad_meta.cpp, for wad/src/meta.rs) and by the oracle (tomllib): both must
build identical levels from it.  Runs where /root/reference is mounted (this container); skipped elsewhere -- nothing
on the GPU box reads the reference.  The file is used in place, never copied."""
import os

import numpy as np
import pytest

import rust_doom_amd as rd
from oracle import wad_oracle

DOOM_TOML = '/root/reference/assets/meta/doom.toml'
ARRAYS = ['static_vertices', 'static_indices', 'sky_vertices', 'sky_indices', 'decor_vertices', 'decor_indices',
          'draws', 'flat_atlas', 'wall_atlas', 'decor_atlas', 'sky_texture']

pytestmark = pytest.mark.skipif(not os.path.exists(DOOM_TOML), reason='reference checkout not mounted')


@pytest.mark.parametrize('index', [0, 3, 7])
def test_levels_built_with_the_reference_metadata_agree(wad_path, index):
    built = rd.Wad(wad_path, DOOM_TOML).build_level(index)
    got = built.arrays()
    want = wad_oracle.build_level(wad_path, DOOM_TOML, index)
    for name in ARRAYS:
        assert np.asarray(got[name]).tobytes() == np.asarray(getattr(want, name)).tobytes(), name
    c = built.counters()
    for k, v in want.counters.items():
        assert c[k] == v, k
    assert c['num_objects'] == want.num_objects
    for t in (0.0, 2.25):
        assert np.array_equal(built.lights_at(t), want.lights.fill_buffer_at(t))
