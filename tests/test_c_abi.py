"""C-ABI boundary (CPU, no compute): librdoom_hip.so loads, exports every function include/rdoom.h declares,
and reports errors the way the header promises (status code + thread-local message, never a crash)."""
import ctypes
import os
import re

import numpy as np
import pytest

import rust_doom_amd as rd
from util import META_PATH, ROOT

HEADER = os.path.join(ROOT, 'include', 'rdoom.h')


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rdoom_[a-z0-9_]+)\s*\(', text)))


def test_header_and_python_mirror_agree():
    assert declared_functions() == sorted(rd.API_SYMBOLS)


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(rd.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_layouts_match_header():
    """sizes the reference fixes: StaticVertex 48 B (game/src/vertex.rs:5-16), SpriteVertex 44 B (:30-40)"""
    assert rd.STATIC_VERTEX.itemsize == 48 and rd.SPRITE_VERTEX.itemsize == 44
    assert rd.POSE.itemsize == 136 and ctypes.sizeof(rd.Timings) == 40


def test_null_arguments_are_rejected():
    L = rd.lib()
    assert L.rdoom_device_count(None) == -1
    assert b'null' in L.rdoom_last_error()
    assert L.rdoom_level_create(None, None) == -1
    assert L.rdoom_batch_create(None, 64, 64, 1, None) == -1
    assert L.rdoom_wad_open(None, None, None) == -1
    assert L.rdoom_built_desc(None, None) == -1


def test_missing_wad_is_an_io_error(tmp_path):
    """Archive::open on a missing file -> ErrorKind::Io (wad/src/archive.rs:36-46, errors.rs:9-19)"""
    with pytest.raises(rd.RdoomError) as e:
        rd.Wad(str(tmp_path / 'nope.wad'), META_PATH)
    assert e.value.status == -5


def test_corrupt_wad_is_reported(tmp_path):
    """bad magic -> ErrorKind::CorruptWad (archive.rs:67-76)"""
    p = tmp_path / 'bad.wad'
    p.write_bytes(b'NOPE' + b'\0' * 64)
    with pytest.raises(rd.RdoomError) as e:
        rd.Wad(str(p), META_PATH)
    assert e.value.status == -6


def test_corrupt_metadata_is_reported(tmp_path, wad_path):
    p = tmp_path / 'bad.toml'
    p.write_text('[[sky]\nthis is not toml')
    with pytest.raises(rd.RdoomError) as e:
        rd.Wad(wad_path, str(p))
    assert e.value.status == -7


def test_level_index_out_of_range(wad_path):
    wad = rd.Wad(wad_path, META_PATH)
    with pytest.raises(rd.RdoomError):
        wad.build_level(99)


def test_pose_look_matches_numpy_camera():
    """rdoom_pose_look == the cgmath restatement used by the tests (player.rs:84-89, projections.rs:93-101)"""
    from util import reference_projection, view_matrix
    p = rd.pose_look((1.0, 0.5, -2.0), 0.7, -0.2, 1920, 1080, 0.25)
    assert np.allclose(p['projection'], reference_projection(1920, 1080), rtol=0, atol=1e-6)
    assert np.allclose(p['modelview'], view_matrix((1.0, 0.5, -2.0), 0.7, -0.2), rtol=0, atol=1e-6)
    assert p['time'] == np.float32(0.25)


def test_level_create_validates_before_touching_the_device():
    """bad descriptors are rejected with RDOOM_BAD_ARG (-1) on a box without a GPU too"""
    from test_kat_analytic import kat_level
    lvl, _ = kat_level()
    bad = dict(lvl)
    bad['wall_atlas'] = np.zeros((100, 64), np.uint16)  # not a power of two (tex.rs:183-200 guarantees one)
    with pytest.raises(rd.RdoomError) as e:
        rd.DeviceLevel(bad)
    assert e.value.status == -1 and 'power of two' in str(e.value)
    bad = dict(lvl)
    bad['draws'] = np.array([[1, 0, 0, 5]], np.uint32)  # index count not a multiple of 3
    with pytest.raises(rd.RdoomError) as e:
        rd.DeviceLevel(bad)
    assert e.value.status == -1
    bad = dict(lvl)
    bad['draws'] = np.array([[7, 0, 0, 3]], np.uint32)  # unknown kind
    with pytest.raises(rd.RdoomError) as e:
        rd.DeviceLevel(bad)
    assert e.value.status == -1
    bad = dict(lvl)
    bad['static_indices'] = np.array([0, 1, 999] + [0] * 12, np.uint32)  # vertex index out of range
    with pytest.raises(rd.RdoomError) as e:
        rd.DeviceLevel(bad)
    assert e.value.status == -1


def test_integration_binding_is_the_generated_one():
    """INTEGRATION.md's Rust block == tools/gen_rust_binding.py's output for the current header (no drift), and it
    declares every entry point"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('gen_rust_binding', os.path.join(ROOT, 'tools', 'gen_rust_binding.py'))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text = gen.generate()
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    block = doc[doc.index(gen.BEGIN) + len(gen.BEGIN):doc.index(gen.END)]
    assert block.strip() == ('```rust\n' + text + '```').strip()
    assert sorted(re.findall(r'pub fn (rdoom_\w+)\(', text)) == declared_functions()


def test_level_create_rejects_sky_draws_without_a_sky_texture_and_empty_atlases():
    """a descriptor the reference could never produce must come back as RDOOM_BAD_ARG, not fault on the device"""
    from test_kat_analytic import kat_level
    lvl, _ = kat_level()
    bad = dict(lvl)
    bad['sky_vertices'] = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    bad['sky_indices'] = np.array([0, 1, 2], np.uint32)
    bad['draws'] = np.concatenate([np.asarray(lvl['draws'], np.uint32).reshape(-1, 4), [[3, 0, 0, 3]]]).astype(np.uint32)
    bad['sky_texture'] = np.zeros((0, 0), np.uint16)
    with pytest.raises(rd.RdoomError) as e:
        rd.DeviceLevel(bad)
    assert e.value.status == -1 and 'sky' in str(e.value)
    desc, _keep = rd.make_desc(lvl)
    desc.wall_w = 0  # pointer set, size zero
    with pytest.raises(rd.RdoomError) as e:
        rd.DeviceLevel(desc)
    assert e.value.status == -1


def test_documents_state_the_real_number_of_entry_points():
    """DESIGN.md / INTEGRATION.md quote how many functions include/rdoom.h declares: the numbers may not drift"""
    import re
    n = len(rd.API_SYMBOLS)
    header = open(os.path.join(ROOT, 'include', 'rdoom.h')).read()
    assert len(re.findall(r'^(?:rdoom_status|void|const char \*)\s*rdoom_\w+\(', header, flags=re.M)) == n
    for doc, pattern in (('INTEGRATION.md', r'all (\d+)\s+entry points'), ('DESIGN.md', r'(\d+) C entry points')):
        text = open(os.path.join(ROOT, doc)).read()
        found = re.findall(pattern, text)
        assert found and all(int(x) == n for x in found), (doc, found, n)
