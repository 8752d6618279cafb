"""The lump shapes real IWADs have and the nine default synthetic levels lack, written by
tools/mkwad.py build_wad(..., shapes=True) into a second IWAD with its own seed and committed digests
(tests/golden/digests_shapes.json, generator tests/golden/make_golden_shapes.py):
  * a TEXTURE2 lump next to TEXTURE1 (wad/src/tex.rs:71-88, 356) -- one texture only there, one RE-DEFINING a TEXTURE1 name
    (IndexMap::insert replaces the image: the later definition is the one the level shows);
  * duplicated lump names: a decoy patch / flat FIRST, the real lump later -- a name finds the LAST lump of that name
    (wad/src/archive.rs:85 index_map.insert; flats are read in directory order into a map, tex.rs:432-470);
  * MAP01.. level markers with the full ten-lump block (levels are found by their THINGS lump, archive.rs:92-97; the sky by the
    `MAP..` pattern of the metadata);
  * sprite lumps with paired rotations (eight-character names, every lump between S_START and S_END goes into the texture map
    under its own name, tex.rs:475-497) and a sprite family WITHOUT an ..A0 lump (the decor looks ..A0 up, then ..A1,
    visitor.rs:1071-1083);
  * textures of 3-6 overlapping patches: transparent posts over opaque ones, a first patch with holes (blitted with
    ignore_transparency: the holes stay, tex.rs:570 / image.rs:171-252), patches hanging over every edge.
Checks: product == numpy oracle byte for byte; an INDEPENDENT composition of the textures from the generator's own patch
pixels (textbook rules, nothing shared with oracle or product) found in the product's atlases, decoys absent; committed
digests; the sanitized host library on mutants of the new lumps; on the GPU, HIP == C oracle and the committed frame digests."""
import hashlib
import json
import os
import struct
import sys

import numpy as np
import pytest

import rust_doom_amd as rd
from oracle import raster, wad_oracle
from test_host_builder_parity import ARRAYS
from util import GOLDEN, META_PATH, ROOT

sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, GOLDEN)
import mkwad  # noqa: E402
from make_golden import H, W, level_poses, sha  # noqa: E402
from make_golden_shapes import SEED, SPECS, build_shapes_wad  # noqa: E402

G = json.load(open(os.path.join(GOLDEN, 'digests_shapes.json')))


@pytest.fixture(scope='module')
def shapes_wad(tmp_path_factory):
    path = str(tmp_path_factory.mktemp('shapes') / 'shapes.wad')
    assert build_shapes_wad(path) == G['wad_sha256'], 'tools/mkwad.py no longer writes the committed shapes IWAD'
    return path


def _directory(path):
    raw = open(path, 'rb').read()
    _magic, n, off = struct.unpack_from('<4sii', raw, 0)
    return raw, [struct.unpack_from('<ii8s', raw, off + 16 * i) for i in range(n)]


def test_the_file_has_the_shapes(shapes_wad):
    raw, entries = _directory(shapes_wad)
    names = [e[2].rstrip(b'\0').decode() for e in entries]
    assert 'TEXTURE1' in names and 'TEXTURE2' in names
    assert names.count('WALL02_1') == 2 and names.count('STEP1') == 2 and names.count('FLAT14') == 2 and names.count('FLOOR4_8') == 2
    assert [n for n in names if n.startswith('MAP')] == ['MAP01', 'MAP02', 'MAP03']
    for m in ('MAP01', 'MAP02', 'MAP03'):
        i = names.index(m)
        assert names[i + 1:i + 11] == ['THINGS', 'LINEDEFS', 'SIDEDEFS', 'VERTEXES', 'SEGS', 'SSECTORS', 'NODES', 'SECTORS', 'REJECT', 'BLOCKMAP']
    assert {'COLUA2A8', 'COLUA3A7', 'COLUA4A6', 'POSSA1', 'POSSA2A8'} <= set(names) and 'POSSA0' not in names
    # TEXTURE2 re-defines BROWN1 and holds T2ONLY
    t2 = [e for e in entries if e[2].rstrip(b'\0') == b'TEXTURE2'][0]
    lump = raw[t2[0]:t2[0] + t2[1]]
    n = struct.unpack_from('<I', lump, 0)[0]
    t2_names = [lump[o:o + 8].rstrip(b'\0').decode() for o in struct.unpack_from('<%dI' % n, lump, 4)]
    assert t2_names == ['T2ONLY', 'BROWN1']


def test_product_equals_oracle_and_the_committed_digests(shapes_wad):
    product = rd.Wad(shapes_wad, META_PATH)
    assert product.num_levels() == 3 and [product.level_name(i) for i in range(3)] == ['MAP01', 'MAP02', 'MAP03']
    for index in range(3):
        got = product.build_level(index).arrays()
        want = wad_oracle.build_level(shapes_wad, META_PATH, index)
        for name in ARRAYS:
            a, b = got[name], np.asarray(getattr(want, name))
            assert a.shape == b.shape and a.tobytes() == b.tobytes(), (index, name)
            if name in G['levels'][index]['arrays']:
                assert sha(a) == G['levels'][index]['arrays'][name], (index, name)
        c = product.build_level(index).counters()
        for k, v in G['levels'][index]['counters'].items():
            assert c[k] == v == want.counters[k], (index, k)
        assert float(want.sky_band) == 1.0   # the `MAP..` sky entry of the metadata, not the E1M. one


# ---- an independent reading of "how a texture is composed" and "which lump a name finds" -------------------------------
def _compose(w, h, prefs, pnames, last_patch):
    """wad/src/tex.rs:560-592 + image.rs:171-252 in textbook form: background 0xFF00; the FIRST patch is copied as it is,
    transparent pixels (0xFFFF in a decoded picture) included; later patches copy their opaque pixels only; everything is
    clipped to the texture; a negative origin_y counts as 0 (tex.rs:583)."""
    img = np.full((h, w), 0xFF00, np.uint16)
    for k, (ox, oy, pi) in enumerate(prefs):
        pix = last_patch.get(pnames[pi])
        if pix is None:
            continue
        src = np.where(pix < 0, 0xFFFF, pix.astype(np.int32) & 0xFF).astype(np.uint16)
        oy = max(oy, 0)
        for y in range(src.shape[0]):
            for x in range(src.shape[1]):
                X, Y = ox + x, oy + y
                if 0 <= X < w and 0 <= Y < h and (k == 0 or src[y, x] != 0xFFFF):
                    img[Y, X] = src[y, x]
    return img


def _used_names(path, level):
    raw, entries = _directory(path)
    marks = [i - 1 for i, e in enumerate(entries) if e[2].rstrip(b'\0') == b'THINGS']
    side = entries[marks[level] + 3]
    sd = raw[side[0]:side[0] + side[1]]
    walls = set()
    for i in range(len(sd) // 30):
        for o in (4, 12, 20):
            walls.add(sd[i * 30 + o:i * 30 + o + 8].rstrip(b'\0').decode())
    return walls - {'-'}


def _regions(atlas, verts, wall=True):
    """every distinct (atlas position, tile size) the vertices name, as the pixels found there"""
    out = []
    for au, av, sx, sy in {(float(v['a_atlas_uv'][0]), float(v['a_atlas_uv'][1]), float(v['a_tile_size'][0]), float(v['a_tile_size'][1])) for v in verts}:
        x, y, w, h = int(au), int(av), int(sx), int(sy)
        if w > 0 and h > 0 and y + h <= atlas.shape[0] and x + w <= atlas.shape[1]:
            out.append(atlas[y:y + h, x:x + w])
    return out


def test_names_find_the_last_lump_texture2_wins_and_patches_overlap_as_the_reference_blits(shapes_wad):
    patches, pnames, textures, flats, sprites, textures2 = mkwad.make_graphics(mkwad.Rng(SEED), True)   # build_wad's first use of its generator
    defs = {}
    for name, w, h, prefs in list(textures) + list(textures2):   # TEXTURE1, then TEXTURE2: the later definition stays
        defs[name] = (w, h, prefs)
    product = rd.Wad(shapes_wad, META_PATH)
    seen_textures, seen_flats, poss = set(), set(), 0
    for index in range(3):
        arrays = product.build_level(index).arrays()
        walls = arrays['static_vertices']
        atlas = np.asarray(arrays['wall_atlas'])
        regions = _regions(atlas, walls)
        for name in _used_names(shapes_wad, index) & {'BROWN1', 'ROCK1', 'OVERLAP3', 'GRATEMIX', 'T2ONLY', 'STEP1', 'STONE2'}:
            w, h, prefs = defs[name]
            want = _compose(w, h, prefs, pnames, patches)           # `patches` holds the REAL pixels (the decoys are full of 251)
            if any(r.shape == want.shape and np.array_equal(r, want) for r in regions):
                seen_textures.add(name)
            decoy = _compose(w, h, prefs, pnames, {k: (np.full_like(v, 251) if k in ('WALL02_1', 'STEP1') else v) for k, v in patches.items()})
            if not np.array_equal(decoy, want):
                assert not any(r.shape == decoy.shape and np.array_equal(r, decoy) for r in regions), (index, name, 'the FIRST lump of a duplicated name was used')
        fatlas = np.asarray(arrays['flat_atlas'])
        for r in _regions(fatlas, walls):
            if r.shape == (64, 64):
                assert not (r == 176).all(), (index, 'a decoy flat was used')
                for fname in ('FLAT14', 'FLOOR4_8'):
                    if np.array_equal(r, flats[fname].reshape(64, 64)):
                        seen_flats.add(fname)
        dv = arrays['decor_vertices']
        poss += int(sum(1 for v in dv if (float(v['a_tile_size'][0]), float(v['a_tile_size'][1])) == (38.0, 56.0)))   # POSSA1: 38 x 56
    assert {'BROWN1', 'OVERLAP3', 'GRATEMIX', 'T2ONLY'} <= seen_textures, seen_textures   # TEXTURE2's BROWN1, the many-patch textures
    assert 'ROCK1' in seen_textures or 'STONE2' in seen_textures                                # built from the REAL WALL02_1
    assert seen_flats == {'FLAT14', 'FLOOR4_8'}
    assert poss >= 4                                                                           # things 3004 draw ..A1: there is no ..A0


# ---- the sanitized host library on mutants of the new lumps -------------------------------------------------------------
def _shape_mutations():
    def t2(f):
        def g(w):
            f(w.named('TEXTURE2')[1])
        return g
    out = [('texture2_truncated', t2(lambda d: d.__delitem__(slice(len(d) - 3, len(d))))),
           ('texture2_empty', t2(lambda d: d.__delitem__(slice(0, len(d))))),
           ('texture2_count_too_large', t2(lambda d: struct.pack_into('<I', d, 0, 4000))),
           ('texture2_offset_beyond_the_lump', t2(lambda d: struct.pack_into('<I', d, 4, 0x00FFFFF0))),
           ('texture2_patch_index_out_of_range', t2(lambda d: struct.pack_into('<H', d, struct.unpack_from('<I', d, 4)[0] + 22 + 4, 999))),
           ('texture2_zero_sized_texture', t2(lambda d: struct.pack_into('<HH', d, struct.unpack_from('<I', d, 8)[0] + 12, 0, 0)))]

    def sprite_cut(w):
        del w.named('POSSA2A8')[1][-9:]
    out.append(('paired_rotation_sprite_truncated', sprite_cut))

    def a1_cut(w):
        del w.named('POSSA1')[1][-9:]
    out.append(('the_only_front_sprite_truncated', a1_cut))

    def decoy_damage(w):
        first = [e for e in w.lumps if e[0].rstrip(b'\0') == b'WALL02_1'][0]
        del first[1][8:]
    out.append(('decoy_patch_corrupt', decoy_damage))
    for seed in range(6):
        def noise(w, seed=seed):
            rng = np.random.RandomState(9000 + seed)
            pool = [w.named('TEXTURE2'), w.named('POSSA3A7'), w.named('COLUA2A8')]
            for _ in range(int(rng.randint(1, 6))):
                data = pool[int(rng.randint(len(pool)))][1]
                data[int(rng.randint(len(data) - 1))] = int(rng.randint(256))
        out.append(('shapes_noise_%02d' % seed, noise))
    return out


SHAPE_MUTATIONS = _shape_mutations()


@pytest.mark.parametrize('name', [m[0] for m in SHAPE_MUTATIONS])
def test_malformed_shapes_under_the_sanitizers(shapes_wad, tmp_path, name):
    import test_host_sanitizers as ths
    driver = ths._build_driver_locked() if hasattr(ths, '_build_driver_locked') else None
    w = ths.WadFile(shapes_wad)
    dict(SHAPE_MUTATIONS)[name](w)
    path = str(tmp_path / 'mutant.wad')
    w.write(path)
    got = ths._driver_outcome(driver, path)
    want = ths._oracle_outcome(path)
    assert got[0] == want[0], (name, got, want)
    if got[0] == 'level':
        assert got[1] == want[1], (name, got, want)
    else:
        assert got[1] in (-5, -6, -7), (name, got, want)
    if name in ('decoy_patch_corrupt', 'paired_rotation_sprite_truncated', 'texture2_patch_index_out_of_range'):
        assert got[0] == 'level', (name, got)   # nothing the level shows is touched (a decoy, an unused rotation, a skipped patch)


# ---- GPU ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_frames_of_the_shapes_iwad(shapes_wad):
    from util import render_checked
    product = rd.Wad(shapes_wad, META_PATH)
    for index in range(3):
        built = product.build_level(index, gpu_tessellation=True)
        lv = wad_oracle.build_level(shapes_wad, META_PATH, index)
        ref = level_poses(lv, index)
        poses = np.zeros(len(ref), rd.POSE)
        lights = np.zeros((len(ref), 256), np.uint8)
        for i, p in enumerate(ref):
            poses[i]['modelview'], poses[i]['projection'], poses[i]['time'] = p[:16], p[16:32], p[32]
            lights[i] = built.lights_at(float(p[32]))
        batch = rd.Batch(rd.DeviceLevel(built), W, H, len(ref))
        fb_plain, fb, prim = render_checked(batch, poses, lights)
        ro = raster.RasterOracle(lv)
        for i, g in enumerate(G['levels'][index]['frames']):
            ofb, oprim = ro.render(ref[i][:16], ref[i][16:32], float(ref[i][32]), lights[i], W, H, want_prim=True)
            assert np.array_equal(fb[i], ofb) and np.array_equal(fb_plain[i], ofb) and np.array_equal(prim[i], oprim), (index, i)
            assert sha(fb[i]) == g['fb'] and sha(prim[i]) == g['prim'], (index, i)
