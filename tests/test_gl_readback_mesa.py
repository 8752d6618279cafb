"""The reference-executed pin of the GPU half, SECOND GL implementation: Mesa's llvmpipe (desktop OpenGL 4.5 core profile, opened
headless through tests/mesa_headless.c) runs the reference's six GLSL programs UNPATCHED -- the engine's own `#version 140` line
(engine/src/platform.rs:5, engine/src/shaders.rs:45), `samplerBuffer u_lights` as game/src/game_shaders.rs:152-159 binds it --
on the same 204 frames as tests/test_gl_readback.py (SwiftShader, OpenGL ES 3.0, three documented text substitutions).

Why a second one: round 5's review -- "bit-exact vs GL readback is 99.16-99.54 %, not 100 %, and it is ONE GL implementation".
Mesa interpolates perspective in full binary32 and snaps vertices to 1/256 pixel (SwiftShader: ~13 bits, 1/16 pixel), so the
oracle's frames differ from ITS readbacks in a fifteenth to a fortieth as many pixels (0.022 % of the stored, 0.031 % of the extended ones; bounds below), every
one again explained by a discontinuity GL leaves open (tests/gl_census.py, unchanged: `other` == 0 in all frames), and the
fragment stage on its own is exact again: the oracle's binary32 static.frag / sprite.frag arithmetic on the varyings Mesa
itself interpolated reproduces Mesa's colour at every flat / wall / decor pixel.

Committed (generator: tests/golden/make_gl_readback_mesa.py): the census of every frame, and Mesa's readbacks (RGB + winning
primitive) of the 27 golden poses at 320x200 (level 0 pose 0 = BASELINE config 2) and of pose 0 of the benchmark sweep at
1920x1080 -- what the oracle (here) and the HIP renderer (GPU box) are held against without Mesa or the reference present."""
import json
import os

import numpy as np
import pytest

import gl_census
import gl_readback
import rust_doom_amd as rd
from oracle import raster
from util import GOLDEN, META_PATH

OUT = os.path.join(GOLDEN, 'gl_readback')
CENSUS = json.load(open(os.path.join(OUT, 'census_mesa.json')))
FRAMES = np.load(os.path.join(OUT, 'frames_mesa.npz'))
SS = json.load(open(os.path.join(OUT, 'census.json')))   # SwiftShader's census of the same frames
STORED = sorted(k[:-4] for k in FRAMES.files if k.endswith('_rgb'))

MAX_MISMATCH_TOTAL = 0.0005         # of all pixels (measured: 0.00022 stored, 0.00031 extended; SwiftShader's bound: 0.02)
MAX_MISMATCH_FRAME = 0.01           # of one frame's pixels, coplanar-wall depth ties aside (SwiftShader's: 0.05)
MAX_WINNER_MISMATCH_TOTAL = 0.001


def test_same_frames_as_the_swiftshader_census():
    assert set(CENSUS['frames']) == set(SS['frames']) and set(CENSUS['extended']) == set(SS['extended'])
    assert len(CENSUS['frames']) == 46 and len(CENSUS['extended']) >= 158
    assert 'Mesa' in CENSUS['gl_version'] and 'llvmpipe' in CENSUS['gl_renderer'] and CENSUS['shader_head'] == '#version 140\n'
    assert CENSUS['subpixel_bits'] == 8 and SS['subpixel_bits'] == 4
    assert CENSUS['total']['pixels'] == SS['total']['pixels'] and CENSUS['extended_total']['pixels'] == SS['extended_total']['pixels']


@pytest.mark.parametrize('part,total', [('frames', 'total'), ('extended', 'extended_total')])
def test_census_is_clean_and_bounded(part, total):
    tot = CENSUS[total]
    assert tot['other'] == 0 and all(f['other'] == 0 for f in CENSUS[part].values())
    assert tot['mismatch'] <= MAX_MISMATCH_TOTAL * tot['pixels'], tot
    assert tot['winner_mismatch'] <= MAX_WINNER_MISMATCH_TOTAL * tot['pixels'], tot
    assert tot['mismatch'] * 10 <= SS[total]['mismatch']       # a different rasteriser and interpolator: far fewer open discontinuities hit
    for k, f in CENSUS[part].items():
        assert f['mismatch'] - f['depth tie'] <= MAX_MISMATCH_FRAME * f['pixels'], (k, f)
        assert sum(f[c] for c in gl_census.CLASSES) == f['mismatch'], k


def test_fragment_stage_is_exact():
    """static.frag:18-28 / sprite.frag:15-27 in binary32 on Mesa's own varyings == Mesa's colour at every drawn pixel -- but for 17 of
    145 M: GLSL does not require a correctly rounded quotient, Mesa evaluates DIST_SCALE / (v_dist + DIST_SCALE) as a product with a
    reciprocal (an ulp off the IEEE quotient in a quarter of all cases), and where (1 - light) * 32 then sits within 2^-17 of a
    COLORMAP-row boundary the palette fetch picks the row across it.  Each such pixel is reproduced through that neighbouring row
    with the same texel (gl_census.fragment_exact: row_division_boundary); nothing else disagrees."""
    for name, part in (('fragment_exact_total', 'frames'), ('extended_fragment_exact_total', 'extended')):
        tot = CENSUS[name]
        lit = tot['by_kind']['flat'] + tot['by_kind']['wall'] + tot['by_kind']['decor']
        assert lit == tot['row_division_boundary'] and lit <= 2e-7 * tot['pixels'], (name, tot)
        assert tot['by_kind']['sky'] == tot['sky_sampler_boundary'], (name, tot)
        assert tot['disagree'] == lit + tot['by_kind']['sky'] and tot['disagree'] <= 2e-5 * tot['pixels'], (name, tot)
        assert tot['pixels'] >= 0.85 * sum(f['pixels'] for f in CENSUS[part].values())
        for k, f in CENSUS[part].items():
            fe = f['fragment_exact']
            assert fe['disagree'] == fe['sky_sampler_boundary'] + fe['row_division_boundary'], (k, fe)
    assert CENSUS['fragment_exact_total']['pixels'] >= 9_000_000 and CENSUS['extended_fragment_exact_total']['pixels'] >= 110_000_000


def frame_inputs(lv, key):
    c = CENSUS['frames'][key]
    pose = FRAMES[key + '_pose']
    mv, pr, t = pose[:16], pose[16:32], float(pose[32])
    assert c['objects_seed'] is None
    return c, mv, pr, t, lv.lights.fill_buffer_at(t)


def mismatch_counts(lv, key, fb, prim):
    pal = np.asarray(lv.palette, np.uint8).reshape(256, 3)
    ours = pal[fb]
    ours[prim == 0xFFFFFFFF] = gl_readback.CLEAR_RGB
    return (int((ours != FRAMES[key + '_rgb']).any(-1).sum()),
            int(((prim & 0xFFFFFF) != (FRAMES[key + '_prim'] & 0xFFFFFF)).sum()))


def test_stored_readbacks():
    assert len(STORED) == 28 and 'L0_P0' in STORED and 'L0_bench0_1080p' in STORED
    assert CENSUS['frames']['L0_P0']['width'] == 320 and CENSUS['frames']['L0_bench0_1080p']['width'] == 1920   # BASELINE configs 2 and 3


@pytest.mark.parametrize('key', STORED)
def test_oracle_against_committed_mesa_readback(oracle_levels, key):
    lv = oracle_levels(CENSUS['frames'][key]['level'])
    c, mv, pr, t, lights = frame_inputs(lv, key)
    fb, prim = raster.RasterOracle(lv).render(mv, pr, t, lights, c['width'], c['height'], want_prim=True)
    assert mismatch_counts(lv, key, fb, prim) == (c['mismatch'], c['winner_mismatch'])
    assert c['mismatch'] <= 0.004 * c['pixels']   # these 28 frames: 99.6 % or more of every one identical to Mesa's


needs_mesa = pytest.mark.skipif(not gl_readback.available('mesa'), reason='needs Mesa (swrast_dri.so + dri_interface.h) and the reference checkout (/root/reference)')


@needs_mesa
def test_the_reference_shaders_run_unpatched(oracle_levels):
    """what Mesa compiles is the engine's version line + the file's bytes, for all six programs"""
    glref = gl_readback.GLReference(oracle_levels(0), backend='mesa')
    for name in ('static', 'sky', 'sprite'):
        for stage in ('vert', 'frag'):
            ref = open(os.path.join(gl_readback.REFERENCE_SHADERS, '%s.%s' % (name, stage))).read()
            assert glref.sources[(name, 'colour')][stage] == '#version 140\n' + ref
    assert 'samplerBuffer' in glref.sources[('static', 'colour')]['vert']


@needs_mesa
@pytest.mark.parametrize('key', ['L0_P0', 'L1_P1', 'L4_P2', 'L8_P0', 'L0_bench0_1080p'])
def test_mesa_regenerates_the_committed_readbacks(oracle_levels, key):
    lv = oracle_levels(CENSUS['frames'][key]['level'])
    c, mv, pr, t, lights = frame_inputs(lv, key)
    w, h = c['width'], c['height']
    glref = gl_readback.GLReference(lv, backend='mesa')
    rgb = glref.render(mv, pr, t, lights, w, h)
    gid = glref.render(mv, pr, t, lights, w, h, mode='ids')
    var = glref.render(mv, pr, t, lights, w, h, mode='varyings')
    assert np.array_equal(rgb, FRAMES[key + '_rgb']) and np.array_equal(gid, FRAMES[key + '_prim'])
    fb, prim = raster.RasterOracle(lv).render(mv, pr, t, lights, w, h, want_prim=True)
    got = gl_census.census(lv, mv, pr, t, lights, w, h, fb, prim, rgb, gid, var)
    assert got['other'] == 0
    for k in ('mismatch', 'winner_mismatch') + gl_census.CLASSES:
        assert got[k] == c[k], (k, got[k], c[k])
    fe = gl_census.fragment_exact(raster.RasterOracle(lv), lv, t, lights, rgb, gid, var)
    assert {k: fe[k] for k in ('pixels', 'disagree', 'by_kind', 'sky_sampler_boundary', 'row_division_boundary')} == c['fragment_exact']


@needs_mesa
@pytest.mark.parametrize('key', ['L2_P0_objects', 'L0_sky1', 'L0_decor3', 'L0_anim3'])
def test_mesa_regenerates_counts_of_other_stored_frames(oracle_levels, key):
    """moving objects, sky, decorations, animated / scrolling textures: counts only (no Mesa readback is stored for these)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_gl_readback_mesa', os.path.join(GOLDEN, 'make_gl_readback_mesa.py'))
    mgen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mgen)
    c = CENSUS['frames'][key]
    _k, index, w, h, pose, seed = [f for f in mgen.stored_frames() if f[0] == key][0]
    lv = oracle_levels(index)
    got, _ = mgen.one_frame(lv, gl_readback.GLReference(lv, backend='mesa'), raster.RasterOracle(lv), np.asarray(pose, np.float32), w, h, seed)
    assert got['other'] == 0
    for k in ('mismatch', 'winner_mismatch') + gl_census.CLASSES:
        assert got[k] == c[k], (k, got[k], c[k])
    assert got['fragment_exact'] == c['fragment_exact']


@pytest.mark.gpu
def test_hip_against_committed_mesa_readback(wad_path, oracle_levels):
    """the HIP renderer differs from Mesa's frames of the reference's shaders in exactly the pixels the oracle does"""
    wad = rd.Wad(wad_path, META_PATH)
    by_level = {}
    for k in STORED:
        by_level.setdefault((CENSUS['frames'][k]['level'], CENSUS['frames'][k]['width']), []).append(k)
    for (index, width), keys in sorted(by_level.items()):
        lv = oracle_levels(index)
        built = wad.build_level(index)
        level = rd.DeviceLevel(built)
        height = CENSUS['frames'][keys[0]]['height']
        batch = rd.Batch(level, width, height, 1)
        batch.enable_primitive_ids()
        for key in keys:
            c, mv, pr, t, lights = frame_inputs(lv, key)
            pose = np.zeros(1, rd.POSE)
            pose[0]['modelview'], pose[0]['projection'], pose[0]['time'] = mv, pr, t
            batch.render(pose, built.lights_at(t))
            fb, prim = batch.read_framebuffer()[0], batch.read_primitive_ids()[0]
            assert mismatch_counts(lv, key, fb, prim) == (c['mismatch'], c['winner_mismatch']), key


# ---- a wider net: 360 frames of IWADs no committed fixture was produced from (counts only) ---------------------------------------
EXTRA = json.load(open(os.path.join(OUT, 'census_mesa_extra.json')))


def test_wider_net_over_fresh_seeds():
    """60 IWADs of fresh generator seeds x 3 levels x 2 poses, random sizes from 320x200 to 1366x768, random times, half of them
    with every door / lift displaced: 190.6 M pixels against Mesa.  99.949 % identical (depth ties of coplanar displaced walls are
    three quarters of the rest), every other pixel attributed to a discontinuity: `other` == 0 in all 360 frames; the fragment stage
    on Mesa's varyings disagrees at 11 wall pixels, all on a COLORMAP-row boundary.
    THIS NET IS WHAT CHANGED THE SPECIFICATION (round 6): with the binary32 triangle set-up of rounds 1-5 it held 13 pixels the
    census could not attribute -- same winner on both sides, within a quarter of a pixel of an edge of a triangle that is thin on
    the screen, Mesa's varyings within 5e-4 texels of the float64 value, the oracle's 0.02 to 3 texels off.  The error was the
    set-up's alone (det and the plane numerators lose their leading digits there): S3..S5 now run in binary64 on the same binary32
    inputs, which removed all 13, a third of ALL differences from Mesa on the stored frames (3 023 -> 2 412) and a
    sixth on the extended ones (50 811 -> 43 102).  `others` (per-pixel descriptions of unattributed pixels) is therefore empty."""
    tot = EXTRA['total']
    assert len(EXTRA['frames']) == 360 and tot['pixels'] >= 190_000_000
    assert tot['mismatch'] <= 0.001 * tot['pixels'] and tot['mismatch'] - tot['depth tie'] <= 0.0002 * tot['pixels'], tot
    assert tot['other'] == 0 and all(f['other'] == 0 and f['others'] == [] for f in EXTRA['frames'].values()), tot
    fe = EXTRA['fragment_exact_total']
    assert fe['disagree'] == fe['row_division_boundary'] + fe['sky_sampler_boundary'] and fe['disagree'] <= 2e-7 * fe['pixels'], fe
    for k, f in EXTRA['frames'].items():
        assert sum(f[c] for c in gl_census.CLASSES) == f['mismatch'], k


@needs_mesa
@pytest.mark.parametrize('key', ['seed31007_L0_sweep745_t18.1_1280x720', 'seed31012_L1_sweep'])
def test_mesa_regenerates_counts_of_the_wider_net(key):
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_gl_readback_mesa', os.path.join(GOLDEN, 'make_gl_readback_mesa.py'))
    mgen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mgen)
    from oracle import wad_oracle
    k, index, w, h, pose, seed = [f for f in mgen.extra_frames() if f[0].startswith(key)][0]
    c = EXTRA['frames'][k]
    path, li = mgen.gen.wad_of(index)
    lv = wad_oracle.build_level(path, META_PATH, li)
    got, _ = mgen.one_frame(lv, gl_readback.GLReference(lv, backend='mesa'), raster.RasterOracle(lv), np.asarray(pose, np.float32), w, h, seed, others=True)
    for name in ('mismatch', 'winner_mismatch') + gl_census.CLASSES:
        assert got[name] == c[name], (name, got[name], c[name])
    assert got['fragment_exact'] == c['fragment_exact'] and len(got['others']) == c['other']


# ---- poses the sweep rarely produces (counts only) -------------------------------------------------------------------------------
EXTREME = json.load(open(os.path.join(OUT, 'census_mesa_extreme.json')))


def test_extreme_poses_against_mesa():
    """180 frames of tests/stress_extreme_poses.py's kind -- eyes within 2 cm of a vertex of the level, on the floor plane itself, far
    outside looking back, pitches up to straight up / down -- on all nine levels, 75 M pixels: 99.935 % identical to Mesa's (87 % of the
    rest are depth ties: from far outside the whole level collapses into a few depth steps), the fragment stage exact everywhere.
    FOUR pixels of ONE frame (the eye 2 cm from a vertex, a wall that crosses the eye plane filling the frame) stay unattributed, and
    are kept as such: the same primitive wins, texel coordinates agree to 1e-3 texels, but the oracle's 1/w -- the binary32 plane
    evaluated at absolute pixel coordinates (F1), steep next to the eye plane -- is 4.3e-4 (relative) off the float64 value, just enough
    to cross a COLORMAP-row boundary that the census's margin (2^-11 of v_dist) does not reach.  The per-pixel evaluation, not the
    set-up (which is binary64 since this round); a pose the reference's player cannot take (its radius keeps the eye 16 map units
    from a wall)."""
    tot = EXTREME['total']
    assert len(EXTREME['frames']) == 180 and tot['pixels'] >= 75_000_000
    assert tot['mismatch'] - tot['depth tie'] <= 0.0002 * tot['pixels'] and tot['mismatch'] <= 0.001 * tot['pixels'], tot
    fe = EXTREME['fragment_exact_total']
    assert fe['disagree'] == 0 and fe['pixels'] >= 30_000_000, fe
    assert tot['other'] <= 4 and sum(1 for f in EXTREME['frames'].values() if f['other']) <= 1, tot
    for k, f in EXTREME['frames'].items():
        assert sum(f[c] for c in gl_census.CLASSES) == f['mismatch'] and len(f['others']) == f['other'], k
        for o in f['others']:
            assert o['same_winner'] and o['crosses_eye_plane'] and o['oracle_uv_off_texels'] <= 2e-3 and o['gl_uv_off_texels'] <= 2e-4, (k, o)
            assert 1e-4 <= o['oracle_dist_off_rel'] <= 1e-3, (k, o)


def test_the_large_level_against_mesa():
    """40 frames of the 10 x E1M1 level (38 262 triangles, lists of hundreds of entries at the horizon) at 640x400, four of every ten
    time-varying: 10.2 M pixels, 99.954 % identical to Mesa's, every other pixel attributed, the fragment stage exact but for one
    COLORMAP-row-boundary pixel (counts only; tests/golden/make_gl_readback_mesa.py --large)"""
    c = json.load(open(os.path.join(OUT, 'census_mesa_large.json')))
    tot, fe = c['total'], c['fragment_exact_total']
    assert len(c['frames']) == 40 and tot['pixels'] == 40 * 640 * 400
    assert tot['other'] == 0 and tot['mismatch'] <= 0.001 * tot['pixels'], tot
    assert fe['disagree'] == fe['row_division_boundary'] + fe['sky_sampler_boundary'] <= 2, fe
    for k, f in c['frames'].items():
        assert sum(f[x] for x in gl_census.CLASSES) == f['mismatch'] and f['others'] == [], k
