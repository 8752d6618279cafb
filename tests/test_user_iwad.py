"""One-command acceptance run for an IWAD the USER supplies ("missing" 2): every file this repository has read so
far came from its own generator (tools/mkwad.py) or from mutants of such files -- DOOM1.WAD / DOOM2.WAD cannot exist in the build
image.  The day one is supplied:

    RDOOM_IWAD=/path/DOOM1.WAD [RDOOM_META=/path/rust-doom/assets/meta/doom.toml] python -m pytest tests/test_user_iwad.py -q -s            # CPU half
    RDOOM_IWAD=... python -m pytest tests/test_user_iwad.py -q -s -m gpu                                                                   # on an MI355X

(skipped entirely unless RDOOM_IWAD is set; RDOOM_META defaults to the reference checkout's doom.toml, /root/reference/assets/meta/doom.toml;
RDOOM_IWAD_LEVELS=0,3,8 restricts the levels.)  Per level of the file:
  * product == oracle, byte for byte: every array `game::level::Builder` hands to the renderer (wad/src/archive.rs:36-106,
    tex.rs:53-107, visitor.rs:541-1259, game/src/level.rs:275-794), counters, start, light tables;
  * the map-side invariants of tests/test_cpu_half_invariants.py (sub-sector polygons tile their sectors, wall quads tile their
    linedef sides, flat triangles carry their sector's attributes) -- decoded from the map lumps by tests/mapcheck.py, which
    shares nothing with oracle or product.  (Their rarity bounds were tuned on generated maps: a failure on a hand-made map
    names the invariant and the place; it is a finding to look at, not necessarily a bug);
  * the counters the reference logs after a build (game/src/level.rs:384-422), printed in its wording -- to be put next to a
    Rust run's (docs/RUST_CROSSCHECK.md);
  * -m gpu: HIP == oracle, bit for bit, framebuffers and winning primitive ids, for the spawn pose (the reference's own binary32
    camera arithmetic) + 16 poses of the seeded sweep at 320x200, and the spawn pose + 4 of the sweep at 1920x1080;
  * -m gpu: `bench.py --iwad` lines for the first level and, if the file has one, MAP29 at 3840x2160 time-varying (BASELINE
    config 5): value > 0, no pose overflowed its tile lists (`config.paths`).
Green here when pointed at the one kind of file available -- the generator's "shapes" IWAD written to disk
(tests/golden/make_golden_shapes.py) with assets/meta/synth.toml -- which is how the committed suite exercises this module
(test_this_module_on_the_shapes_iwad)."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import rust_doom_amd as rd
from util import ROOT

IWAD = os.environ.get('RDOOM_IWAD')
META = os.environ.get('RDOOM_META', '/root/reference/assets/meta/doom.toml')
ARRAYS = ['static_vertices', 'static_indices', 'sky_vertices', 'sky_indices', 'decor_vertices', 'decor_indices',
          'draws', 'flat_atlas', 'wall_atlas', 'decor_atlas', 'sky_texture', 'colormap']
COUNTERS = ('num_wall_quads', 'num_floor_polys', 'num_ceil_polys', 'num_sky_wall_quads', 'num_sky_floor_polys',
            'num_sky_ceil_polys', 'num_decors', 'num_static_tris', 'num_sky_tris', 'num_sprite_tris')


def _levels():
    if not IWAD:
        return []
    n = rd.Wad(IWAD, META).num_levels()
    pick = os.environ.get('RDOOM_IWAD_LEVELS')
    return [int(x) for x in pick.split(',')] if pick else list(range(n))


LEVELS = _levels()
needs_iwad = pytest.mark.skipif(not IWAD, reason='set RDOOM_IWAD (and RDOOM_META) to run the acceptance checks on a real IWAD')


# ---- the checks, as functions of (iwad, metadata, level): the parametrised tests below and the self-test both call them ----------
def check_product_equals_oracle(iwad, meta, index):
    from oracle import wad_oracle
    built = rd.Wad(iwad, meta).build_level(index)
    want = wad_oracle.build_level(iwad, meta, index)
    got = built.arrays()
    for name in ARRAYS:
        a, b = got[name], np.asarray(getattr(want, name))
        assert a.shape == b.shape, (index, name, a.shape, b.shape)
        assert a.tobytes() == b.tobytes(), (index, name)
    assert got['palette'].tobytes() == np.asarray(want.palette).tobytes()
    c = built.counters()
    for k, v in want.counters.items():
        assert c[k] == v, (index, k)
    assert c['num_objects'] == want.num_objects and c['num_lights'] == len(want.lights.lights)
    pos, yaw = built.start()
    assert pos.tobytes() == np.array(want.start_pos, np.float32).tobytes() and yaw == np.float32(want.start_yaw)
    for t in (0.0, 0.31, 1.7, 12.5):
        assert np.array_equal(built.lights_at(t), want.lights.fill_buffer_at(t)), (index, t)
    # the device tessellation kernels produce the same arrays (GPU only: the flag needs a device)
    return built, want


def check_map_invariants(iwad, meta, index):
    inv = importlib.import_module('test_cpu_half_invariants')
    wads = {'user': (iwad, meta)}
    inv.test_subsector_polygons_tile_their_sectors(wads, 'user', index)
    inv.test_wall_quads_tile_their_linedef_sides(wads, 'user', index)
    inv.test_flat_triangles_carry_their_sectors_attributes(wads, 'user', index)


def print_counters(iwad, meta, index):
    wad = rd.Wad(iwad, meta)
    c = wad.build_level(index).counters()
    print('\n%s:\nLevel built:' % wad.level_name(index))   # game/src/level.rs:384-396's wording and order
    for key in COUNTERS:
        print('\t%s = %d' % (key, c[key]))
    return c


def check_hip_equals_oracle(iwad, meta, index, sizes=((320, 200, 16), (1920, 1080, 4))):
    from oracle import camera, raster, wad_oracle
    from util import render_checked
    sharding = importlib.import_module('rust-doom_amd.sharding')
    built = rd.Wad(iwad, meta).build_level(index, gpu_tessellation=True)
    lv = wad_oracle.build_level(iwad, meta, index)
    got = built.arrays()
    for name in ARRAYS:   # SSECTOR -> polygon and SEG -> quad kernels == host path == oracle
        assert got[name].tobytes() == np.asarray(getattr(lv, name)).tobytes(), (index, name, 'device tessellation')
    ro = raster.RasterOracle(lv)
    level = rd.DeviceLevel(built)
    lights = built.lights_at(0.0)
    for w, h, n_sweep in sizes:
        poses = np.zeros(1 + n_sweep, rd.POSE)
        pos, yaw = built.start()
        poses[0]['modelview'], poses[0]['projection'] = camera.pose_from_player(pos, yaw, 1e-8, w, h)   # the reference's spawn view
        poses[1:] = sharding.pose_sweep(rd, built, n_sweep, w, h)
        batch = rd.Batch(level, w, h, len(poses))
        fb_plain, fb, prim = render_checked(batch, poses, lights)
        for i, p in enumerate(poses):
            ofb, oprim = ro.render(p['modelview'], p['projection'], 0.0, lights, w, h, want_prim=True)
            bad = int((ofb != fb[i]).sum()), int((ofb != fb_plain[i]).sum()), int((oprim != prim[i]).sum())
            assert bad == (0, 0, 0), (index, (w, h), i, bad)
        assert (fb != 0).any(), (index, (w, h), 'every frame is empty')
        batch.close()


def bench_line(iwad, meta, extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--iwad', iwad, '--metadata', meta, '--other', 'off', '--cpu-sample', '0',
                          '--steps', '5', '--warmup', '2'] + extra, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    print('\n' + json.dumps({k: line[k] for k in ('metric', 'value', 'unit', 'ms_per_step')} | {'paths': line['config']['paths'], 'kernels_ms': line['config']['kernels_ms']}))
    assert line['value'] > 0
    assert line['config']['paths']['bins_overflowed_poses'] == 0, 'poses overflowed their tile lists: rasterised from the sorted list (slow) -- ' + json.dumps(line['config']['paths'])
    return line


# ---- the user's file ------------------------------------------------------------------------------------------------------------
@needs_iwad
@pytest.mark.parametrize('index', LEVELS)
def test_product_equals_oracle(index):
    check_product_equals_oracle(IWAD, META, index)


@needs_iwad
@pytest.mark.parametrize('index', LEVELS)
def test_map_invariants(index):
    check_map_invariants(IWAD, META, index)


@needs_iwad
def test_counters_in_the_reference_s_wording():
    for index in LEVELS:
        c = print_counters(IWAD, META, index)
        assert c['num_static_tris'] > 0, index


@needs_iwad
@pytest.mark.gpu
@pytest.mark.parametrize('index', LEVELS)
def test_hip_equals_oracle(index):
    check_hip_equals_oracle(IWAD, META, index)


@needs_iwad
@pytest.mark.gpu
def test_bench_lines():
    wad = rd.Wad(IWAD, META)
    names = [wad.level_name(i) for i in range(wad.num_levels())]
    bench_line(IWAD, META, ['--level', '0', '--poses', '256'])                                       # E1M1 / MAP01 at 1080p
    if 'MAP29' in names:                                                                             # BASELINE config 5
        bench_line(IWAD, META, ['--level', str(names.index('MAP29')), '--poses', '64', '--width', '3840', '--height', '2160', '--time-varying'])


# ---- the committed suite: this module on the one kind of file that exists here ---------------------------------------------------
@pytest.fixture(scope='module')
def shapes_on_disk(tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from make_golden_shapes import build_shapes_wad
    path = str(tmp_path_factory.mktemp('user_iwad') / 'shapes.wad')
    build_shapes_wad(path)
    return path


def test_this_module_on_the_shapes_iwad(shapes_on_disk, capsys):
    """TEXTURE2, duplicated lump names, MAPxx markers, paired-rotation sprites, many-patch textures -- the lump shapes real IWADs
    have (tests/test_iwad_shapes.py) -- through exactly the functions a user's file goes through"""
    from util import META_PATH
    n = rd.Wad(shapes_on_disk, META_PATH).num_levels()
    assert n >= 3
    for index in range(n):
        check_product_equals_oracle(shapes_on_disk, META_PATH, index)
        check_map_invariants(shapes_on_disk, META_PATH, index)
        print_counters(shapes_on_disk, META_PATH, index)
    assert 'num_wall_quads = ' in capsys.readouterr().out


@pytest.mark.gpu
def test_this_module_on_the_shapes_iwad_gpu(shapes_on_disk):
    from util import META_PATH
    for index in range(rd.Wad(shapes_on_disk, META_PATH).num_levels()):
        check_hip_equals_oracle(shapes_on_disk, META_PATH, index, sizes=((320, 200, 8), (1920, 1080, 1)))
    bench_line(shapes_on_disk, META_PATH, ['--level', '0', '--poses', '64'])
