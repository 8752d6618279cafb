"""BASELINE config 1: the CPU-only geometry build -> triangle dump (tools/dump_geometry.py).  The dump is
re-read from disk and compared with BuiltLevel.arrays(); the timings the reference logs for the same phases
(wad/src/tex.rs:67-88, game/src/level.rs:333, 384-396) are present and plausible."""
import json
import os
import subprocess
import sys

import numpy as np

import rust_doom_amd as rd
from util import META_PATH, ROOT


def test_dump_round_trips(wad_path, tmp_path):
    out = str(tmp_path / 'dump')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'dump_geometry.py'), wad_path, META_PATH, '0', out, '--repeat', '3'],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    assert p.stdout.startswith('Level built in ') and '\tnum_wall_quads = ' in p.stdout   # the reference's log wording
    built = rd.Wad(wad_path, META_PATH).build_level(0, gpu_tessellation=False)
    a = built.arrays()
    verts = np.fromfile(os.path.join(out, 'verts.bin'), rd.STATIC_VERTEX)
    assert verts.tobytes() == a['static_vertices'].tobytes() and verts.dtype.itemsize == 48
    assert np.fromfile(os.path.join(out, 'sky_verts.bin'), np.float32).tobytes() == a['sky_vertices'].astype(np.float32).tobytes()
    assert np.fromfile(os.path.join(out, 'decor_verts.bin'), rd.SPRITE_VERTEX).tobytes() == a['decor_vertices'].tobytes()
    kinds = {0: 'flat', 1: 'wall', 2: 'decor', 3: 'sky'}
    source = {0: a['static_indices'], 1: a['static_indices'], 2: a['decor_indices'], 3: a['sky_indices']}
    expected_files = {'verts.bin', 'sky_verts.bin', 'decor_verts.bin', 'counters.json'}
    n_static_tris = 0
    for kind, obj, first, count in a['draws']:
        name = 'indices_%d_%s.bin' % (obj, kinds[int(kind)])
        expected_files.add(name)
        got = np.fromfile(os.path.join(out, name), np.uint32)
        assert np.array_equal(got, source[int(kind)][first:first + count]), name
        limit = {0: len(verts), 1: len(verts), 2: len(a['decor_vertices']), 3: len(a['sky_vertices'])}[int(kind)]
        assert len(got) % 3 == 0 and (got < limit).all(), name
        n_static_tris += len(got) // 3 if int(kind) in (0, 1) else 0
    assert set(os.listdir(out)) == expected_files
    info = json.load(open(os.path.join(out, 'counters.json')))
    assert info['counters'] == built.counters() and info['counters']['num_static_tris'] == n_static_tris
    assert 0 < info['t_load_ms'] < 5000 and 0 < info['t_walk_ms'] < 5000
    assert abs(info['t_load_ms'] - sum(info['phases_ms'][k] for k in ('open_ms', 'textures_ms', 'level_lumps_ms', 'atlases_ms'))) < 0.01
    assert abs(info['t_walk_ms'] - sum(info['phases_ms'][k] for k in ('analysis_ms', 'walk_ms'))) < 0.01


def test_timings_through_the_c_abi(wad_path):
    wad = rd.Wad(wad_path, META_PATH)
    t = wad.timings()
    assert t['open_ms'] > 0 and t['textures_ms'] > 0 and t['walk_ms'] == 0
    b = wad.build_level(3).timings()
    assert b['walk_ms'] > 0 and b['atlases_ms'] > 0 and b['level_lumps_ms'] > 0 and b['analysis_ms'] >= 0 and b['open_ms'] == 0
