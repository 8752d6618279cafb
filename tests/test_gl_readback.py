"""The reference-executed pin of the GPU half (rows a17-a19, f1, f2): frames of the reference's OWN shaders
(/root/reference/assets/shaders/*.{vert,frag}), run headless by SwiftShader (tests/gl_readback.py), against the oracle
and the HIP renderer.

Committed under tests/golden/gl_readback/ (generator: tests/golden/make_gl_readback.py): the GL readbacks (RGB), the
primitive SwiftShader's rasteriser chose per pixel, and the mismatch census of the oracle's frames against them --
46 frames: the 27 golden poses at 320x200 (level 0 pose 0 = BASELINE config 2: E1M1, spawn pose, 320x200), three frames
with moving objects, twelve targeted views (sky, decorations, scrolling / animated textures), poses 0, 341, 682 and 1000 of
the benchmark sweep at 1920x1080.  An EXTENDED census (counts only, no stored readbacks, 136 M pixels in 149 frames: 32
more poses of that sweep at 1920x1080, eight poses of every other level at 640x400, three time-varying frames per level
with every door / lift displaced, four 3840x2160 frames, four frames of the 10 x E1M1 level, eighteen frames of IWADs
written from other seeds) widens the net for systematic differences.

What is asserted:
  * every mismatching pixel is explained by a discontinuity GL leaves to the implementation (tests/gl_census.py):
    `other` == 0 in every frame, and the totals stay under stated bounds;
  * the oracle reproduces the committed census pixel for pixel (no SwiftShader needed: runs anywhere);
  * where SwiftShader and the reference checkout are present (this container), the readbacks and the census are
    regenerated from the reference's shader files and must equal the committed ones;
  * (gpu) the HIP renderer's frames have exactly the oracle's mismatch sets against the GL readbacks;
  * the FRAGMENT STAGE with zero tolerance (gl_census.fragment_exact): the oracle's binary32 fragment code on the varyings
    SwiftShader itself interpolated, for the primitive that won there, reproduces SwiftShader's colour at EVERY drawn
    pixel of every frame -- 0 disagreements on flats, walls and decorations (9.7 M stored + 131 M extended pixels); the only
    disagreements are sky pixels within 1/64 texel of a texel boundary of sky.frag's REPEAT / NEAREST fetch, whose address
    arithmetic GL leaves to the sampler (38 stored pixels), each reproduced through the neighbouring texel.

Bounds (measured: 0.83 % of the 10 982 400 stored pixels and 0.45 % of the 138 035 200 extended ones differ; 95 % of those
are texel-boundary picks caused by SwiftShader's ~13-bit perspective interpolation, 2.4 % lie on primitive edges; winners
differ on 0.03 - 0.05 %):"""
import importlib.util
import json
import os

import numpy as np
import pytest

import gl_census
import gl_readback
import rust_doom_amd as rd
from oracle import raster
from util import GOLDEN, META_PATH

MAX_MISMATCH_TOTAL = 0.02      # of all pixels of all frames
MAX_MISMATCH_FRAME = 0.05      # of one frame's pixels
MAX_WINNER_MISMATCH_TOTAL = 0.002

OUT = os.path.join(GOLDEN, 'gl_readback')
CENSUS = json.load(open(os.path.join(OUT, 'census.json')))
FRAMES = np.load(os.path.join(OUT, 'frames.npz'))
KEYS = sorted(CENSUS['frames'])
SMALL = [k for k in KEYS if CENSUS['frames'][k]['width'] == 320]

_spec = importlib.util.spec_from_file_location('make_gl_readback', os.path.join(GOLDEN, 'make_gl_readback.py'))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


def frame_inputs(lv, key):
    c = CENSUS['frames'][key]
    pose = FRAMES[key + '_pose']
    mv, pr, t = pose[:16], pose[16:32], float(pose[32])
    om = None if c['objects_seed'] is None else gen.moving_object_views(lv, mv, c['objects_seed'])
    return c, mv, pr, t, lv.lights.fill_buffer_at(t), om


def mismatch_counts(lv, key, fb, prim):
    pal = np.asarray(lv.palette, np.uint8).reshape(256, 3)
    ours = pal[fb]
    ours[prim == 0xFFFFFFFF] = gl_readback.CLEAR_RGB
    return (int((ours != FRAMES[key + '_rgb']).any(-1).sum()),
            int(((prim & 0xFFFFFF) != (FRAMES[key + '_prim'] & 0xFFFFFF)).sum()))


def test_extended_census_is_clean_and_bounded():
    tot = CENSUS['extended_total']
    assert tot['other'] == 0 and all(f['other'] == 0 for f in CENSUS['extended'].values())
    assert tot['pixels'] >= 138_000_000 and len(CENSUS['extended']) >= 158
    assert tot['mismatch'] <= MAX_MISMATCH_TOTAL * tot['pixels'], tot
    assert tot['winner_mismatch'] <= MAX_WINNER_MISMATCH_TOTAL * tot['pixels'], tot
    for k, f in CENSUS['extended'].items():
        # (frames with displaced doors / lifts: a displaced sector's walls stay in the vertical planes of the static walls
        # next to them, and where two COPLANAR walls overlap the winner is anybody's rounding -- whole wall sections of ties)
        assert f['mismatch'] - f['depth tie'] <= MAX_MISMATCH_FRAME * f['pixels'], (k, f)
        assert sum(f[c] for c in gl_census.CLASSES) == f['mismatch'], k
    assert sum(1 for f in CENSUS['extended'].values() if f['width'] == 1920) >= 32
    assert {f['level'] for f in CENSUS['extended'].values()} == (set(range(9)) | {'big'} | {'seed%d:%d' % (s, i) for s in gen.OTHER_SEEDS for i in range(3)} |
                                                                 {'shapes:%d' % i for i in range(3)})
    assert sum(1 for f in CENSUS['extended'].values() if f['width'] == 3840) >= 4
    assert sum(1 for f in CENSUS['extended'].values() if f['objects_seed'] is not None and f['time'] > 0) >= 27


def test_fragment_stage_is_exact():
    """static.frag:18-28 / sprite.frag:15-27 in binary32 on GL's own varyings == GL's colour, at every drawn pixel"""
    for name, frames in (('fragment_exact_total', CENSUS['frames']), ('extended_fragment_exact_total', CENSUS['extended'])):
        tot = CENSUS[name]
        assert tot['by_kind']['flat'] == 0 and tot['by_kind']['wall'] == 0 and tot['by_kind']['decor'] == 0, (name, tot)
        assert tot['disagree'] == tot['by_kind']['sky'] == tot['sky_sampler_boundary'], (name, tot)
        assert tot['disagree'] <= 2e-5 * tot['pixels'], (name, tot)
        assert tot['pixels'] >= 0.85 * sum(f['pixels'] for f in frames.values())   # nearly every pixel of every frame is drawn
        for k, f in frames.items():
            fe = f['fragment_exact']
            assert fe['disagree'] == fe['by_kind']['sky'] == fe['sky_sampler_boundary'], (k, fe)
    assert CENSUS['fragment_exact_total']['pixels'] >= 9_000_000 and CENSUS['extended_fragment_exact_total']['pixels'] >= 110_000_000


def test_census_is_clean_and_bounded():
    tot = CENSUS['total']
    assert tot['other'] == 0
    assert all(f['other'] == 0 for f in CENSUS['frames'].values())
    assert tot['mismatch'] <= MAX_MISMATCH_TOTAL * tot['pixels'], tot
    assert tot['winner_mismatch'] <= MAX_WINNER_MISMATCH_TOTAL * tot['pixels'], tot
    for k, f in CENSUS['frames'].items():
        assert f['mismatch'] <= MAX_MISMATCH_FRAME * f['pixels'], (k, f)
        assert sum(f[c] for c in gl_census.CLASSES) == f['mismatch'], k
    assert 'L0_P0' in CENSUS['frames'] and CENSUS['frames']['L0_P0']['width'] == 320   # BASELINE config 2's frame
    assert any(f['width'] == 1920 and f['height'] == 1080 for f in CENSUS['frames'].values())


@pytest.mark.parametrize('key', KEYS)
def test_oracle_against_committed_gl_readback(oracle_levels, key):
    lv = oracle_levels(CENSUS['frames'][key]['level'])
    c, mv, pr, t, lights, om = frame_inputs(lv, key)
    fb, prim = raster.RasterOracle(lv).render(mv, pr, t, lights, c['width'], c['height'], want_prim=True, object_modelviews=om)
    assert mismatch_counts(lv, key, fb, prim) == (c['mismatch'], c['winner_mismatch'])


@pytest.mark.skipif(not gl_readback.available(), reason='needs SwiftShader and the reference checkout (/root/reference)')
@pytest.mark.parametrize('key', ['L0_P0', 'L0_P2', 'L1_P1', 'L2_P0_objects', 'L5_P1', 'L7_P2', 'L8_P0', 'L0_sky1', 'L0_decor3', 'L0_anim3'])
def test_swiftshader_runs_the_reference_shaders(oracle_levels, key):
    """regenerates readback + census from the reference's shader files; both must equal the committed fixtures"""
    lv = oracle_levels(CENSUS['frames'][key]['level'])
    c, mv, pr, t, lights, om = frame_inputs(lv, key)
    w, h = c['width'], c['height']
    glref = gl_readback.GLReference(lv)
    rgb = glref.render(mv, pr, t, lights, w, h, object_modelviews=om)
    gid = glref.render(mv, pr, t, lights, w, h, mode='ids', object_modelviews=om)
    var = glref.render(mv, pr, t, lights, w, h, mode='varyings', object_modelviews=om)
    assert np.array_equal(rgb, FRAMES[key + '_rgb']) and np.array_equal(gid, FRAMES[key + '_prim'])
    fb, prim = raster.RasterOracle(lv).render(mv, pr, t, lights, w, h, want_prim=True, object_modelviews=om)
    got = gl_census.census(lv, mv, pr, t, lights, w, h, fb, prim, rgb, gid, var, object_modelviews=om)
    assert got['other'] == 0
    for k in ('mismatch', 'winner_mismatch') + gl_census.CLASSES:
        assert got[k] == c[k], (k, got[k], c[k])
    fe = gl_census.fragment_exact(raster.RasterOracle(lv), lv, t, lights, rgb, gid, var)
    assert {k: fe[k] for k in ('pixels', 'disagree', 'by_kind', 'sky_sampler_boundary')} == c['fragment_exact']
    assert fe['by_kind']['flat'] == fe['by_kind']['wall'] == fe['by_kind']['decor'] == 0 and fe['disagree'] == fe['sky_sampler_boundary']


@pytest.mark.skipif(not gl_readback.available(), reason='needs SwiftShader and the reference checkout (/root/reference)')
@pytest.mark.parametrize('key', ['L0_bench624_1080p', 'L5_sweep128_640', 'L3_sweep424_t12.1_objects_640', 'L7_sweep808_t34.5_objects_640',
                                 'seed4242_L0_sweep523_t5.3_640', 'shapes_L1_sweep301_t4.1_640', 'shapes_L2_sweep7_t0.0_640'])
def test_swiftshader_extended_census(key):
    """regenerates the counts of extended frames (nothing stored but the counts) from the reference's shader files"""
    c = CENSUS['extended'][key]
    _k, level_key, w, h, pose, seed = [f for f in gen.extended_frames() if f[0] == key][0]
    lv = gen.extended_level({}, level_key)
    got = gen.extended_census(lv, gl_readback.GLReference(lv), raster.RasterOracle(lv), pose, w, h, seed)
    assert got['other'] == 0
    for k in ('mismatch', 'winner_mismatch') + gl_census.CLASSES:
        assert got[k] == c[k], (k, got[k], c[k])
    assert got['fragment_exact'] == c['fragment_exact']


@pytest.mark.skipif(not gl_readback.available(), reason='needs the reference checkout (/root/reference)')
def test_shader_patch_is_only_what_glsl_es_forces():
    """undoing the three documented substitutions gives back the reference's text byte for byte"""
    for name in ('static', 'sky', 'sprite'):
        for stage in ('vert', 'frag'):
            ref = open(os.path.join(gl_readback.REFERENCE_SHADERS, '%s.%s' % (name, stage))).read()
            got = gl_readback.patch_shader(ref, stage)
            head = '#version 300 es\nprecision highp float;\nprecision highp int;\nprecision highp sampler2D;\n'
            assert got.startswith(head)
            body = got[len(head):]
            body = body.replace('uniform sampler2D u_lights;', 'uniform samplerBuffer u_lights;')
            body = body.replace('texelFetch(u_lights, ivec2(a_light, 0), 0)', 'texelFetch(u_lights, a_light)')
            assert body == ref.replace('precision mediump float;', ''), (name, stage)


@pytest.mark.gpu
def test_hip_against_committed_gl_readback(wad_path, oracle_levels):
    """the HIP renderer differs from the reference's GL frames in exactly the pixels the census explains"""
    wad = rd.Wad(wad_path, META_PATH)
    by_level = {}
    for k in KEYS:
        by_level.setdefault((CENSUS['frames'][k]['level'], CENSUS['frames'][k]['width']), []).append(k)
    for (index, width), keys in sorted(by_level.items()):
        lv = oracle_levels(index)
        built = wad.build_level(index)
        level = rd.DeviceLevel(built)
        height = CENSUS['frames'][keys[0]]['height']
        batch = rd.Batch(level, width, height, 1)
        batch.enable_primitive_ids()
        for key in keys:
            c, mv, pr, t, lights, om = frame_inputs(lv, key)
            pose = np.zeros(1, rd.POSE)
            pose[0]['modelview'], pose[0]['projection'], pose[0]['time'] = mv, pr, t
            batch.render(pose, built.lights_at(t), object_modelviews=None if om is None else om[None])
            fb, prim = batch.read_framebuffer()[0], batch.read_primitive_ids()[0]
            assert mismatch_counts(lv, key, fb, prim) == (c['mismatch'], c['winner_mismatch']), key
            # not just the same COUNTS: the same pixels -- the HIP frame and its winners are the oracle's, so its mismatch
            # mask against the GL readback is the one the census classified pixel by pixel
            ofb, oprim = raster.RasterOracle(lv).render(mv, pr, t, lights, c['width'], c['height'], want_prim=True, object_modelviews=om)
            pal = np.asarray(lv.palette, np.uint8).reshape(256, 3)
            gl_rgb, gl_prim = FRAMES[key + '_rgb'], FRAMES[key + '_prim']

            def masks(f, p):
                rgb = pal[f]
                rgb[p == 0xFFFFFFFF] = gl_readback.CLEAR_RGB
                return (rgb != gl_rgb).any(-1), (p & 0xFFFFFF) != (gl_prim & 0xFFFFFF)

            for got, want in zip(masks(fb, prim), masks(ofb, oprim)):
                assert np.array_equal(got, want), key
