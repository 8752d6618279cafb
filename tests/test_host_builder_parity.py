"""Host logic parity (CPU, no GPU needed): the product's C++ `wad` loader + `game::level` builder
(behind the C ABI) vs the numpy oracle -- every array byte-for-byte, every level of the synthetic IWAD."""
import numpy as np
import pytest

import rust_doom_amd as rd
from util import META_PATH

ARRAYS = ['static_vertices', 'static_indices', 'sky_vertices', 'sky_indices', 'decor_vertices', 'decor_indices',
          'draws', 'flat_atlas', 'wall_atlas', 'decor_atlas', 'sky_texture', 'colormap']


@pytest.fixture(scope='module')
def product_wad(wad_path):
    return rd.Wad(wad_path, META_PATH)


def test_level_directory(product_wad):
    assert product_wad.num_levels() == 9
    assert [product_wad.level_name(i) for i in range(9)] == ['E1M%d' % (i + 1) for i in range(9)]


@pytest.mark.parametrize('index', range(9))
def test_built_level_matches_oracle(product_wad, oracle_levels, index):
    built = product_wad.build_level(index)
    got = built.arrays()
    want = oracle_levels(index)
    for name in ARRAYS:
        a, b = got[name], np.asarray(getattr(want, name))
        assert a.shape == b.shape, (name, a.shape, b.shape)
        assert a.tobytes() == b.tobytes(), name
    assert got['palette'].tobytes() == np.asarray(want.palette).tobytes()
    assert np.float32(got['sky_band']) == np.float32(want.sky_band)
    c = built.counters()
    for k, v in want.counters.items():
        assert c[k] == v, k
    assert c['num_objects'] == want.num_objects and c['num_lights'] == len(want.lights.lights)
    pos, yaw = built.start()
    assert pos.tobytes() == np.array(want.start_pos, np.float32).tobytes() and yaw == np.float32(want.start_yaw)
    for t in (0.0, 0.31, 1.7, 12.5):
        assert np.array_equal(built.lights_at(t), want.lights.fill_buffer_at(t)), t


def test_e1m1_is_e1m1_sized(product_wad):
    c = product_wad.build_level(0).counters()
    assert 2500 <= c['num_static_tris'] <= 6000 and c['num_sky_tris'] > 0 and c['num_objects'] > 1


def _repack(src_path, dst_path, seed):
    """An IWAD with the same directory ENTRIES (names, order, sizes) but a physical layout tools/mkwad.py never writes:
    lump data in shuffled order with junk-filled gaps in between, the directory in the middle of the file, zero-length
    foreign marker lumps sprinkled between the non-level lumps, and two lumps sharing one copy of identical bytes."""
    import struct
    rng = np.random.RandomState(seed)
    raw = open(src_path, 'rb').read()
    magic, n, dir_off = struct.unpack_from('<4sII', raw, 0)
    entries = [struct.unpack_from('<II8s', raw, dir_off + 16 * i) for i in range(n)]
    level_lumps = {b'THINGS', b'LINEDEFS', b'SIDEDEFS', b'VERTEXES', b'SEGS', b'SSECTORS', b'NODES', b'SECTORS', b'REJECT', b'BLOCKMAP'}
    out_entries, blobs = [], []
    for pos, size, name in entries:
        nm = name.rstrip(b'\0')
        is_level_part = nm in level_lumps or (len(nm) == 4 and nm[0:1] == b'E' and nm[2:3] == b'M')
        if not is_level_part and size and rng.rand() < 0.05:
            out_entries.append((None, 0, b'XX_JUNK\0'))   # a marker lump no reader asks for
        out_entries.append((len(blobs), size, name))
        blobs.append(raw[pos:pos + size])
    order = list(rng.permutation(len(blobs)))
    body = bytearray(b'\0' * 12)
    offsets, seen = {}, {}
    half = len(order) // 2
    dir_at = None
    for k, bi in enumerate(order):
        if k == half:   # the directory sits in the middle of the file
            dir_at = len(body)
            body += b'\0' * (16 * len(out_entries))
        blob = blobs[bi]
        if blob and blob in seen and rng.rand() < 0.5:   # identical bytes stored once, referenced twice
            offsets[bi] = seen[blob]
            continue
        body += bytes(rng.randint(0, 256, rng.randint(0, 37)).astype(np.uint8))   # junk gap
        offsets[bi] = len(body)
        seen.setdefault(blob, len(body))
        body += blob
    struct.pack_into('<4sII', body, 0, magic, len(out_entries), dir_at)
    for i, (bi, size, name) in enumerate(out_entries):
        struct.pack_into('<II8s', body, dir_at + 16 * i, 0 if bi is None else offsets[bi], size, name)
    open(dst_path, 'wb').write(bytes(body))


@pytest.mark.parametrize('seed', [1, 2])
def test_foreign_physical_layout_reads_the_same(wad_path, oracle_levels, tmp_path, seed):
    """the reader (archive.rs:62-106: header, directory anywhere, lumps by (offset, size)) must not depend on the layout
    its sibling writer produces"""
    from oracle import wad_oracle
    path = str(tmp_path / ('repacked%d.wad' % seed))
    _repack(wad_path, path, seed)
    assert open(path, 'rb').read() != open(wad_path, 'rb').read()
    wad = rd.Wad(path, META_PATH)
    assert wad.num_levels() == 9
    for index in (0, 4, 8):
        got = wad.build_level(index).arrays()
        want = oracle_levels(index)   # built from the ORIGINAL file
        for name in ARRAYS:
            assert got[name].tobytes() == np.asarray(getattr(want, name)).tobytes(), (index, name)
    again = wad_oracle.build_level(path, META_PATH, 4)   # the oracle's reader on the repacked file
    for name in ARRAYS:
        assert np.asarray(getattr(again, name)).tobytes() == np.asarray(getattr(oracle_levels(4), name)).tobytes(), name
