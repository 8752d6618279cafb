"""rust-doom_amd -- MI355X-native pose-batch renderer for Doom WAD levels.

Thin Python mirror of the C ABI in include/rdoom.h (ctypes over librdoom_hip.so).  Names follow the
reference: `Wad` ~ wad::Archive + TextureDirectory (wad/src/archive.rs, tex.rs), `BuiltLevel` ~ what
game::level::Builder + GameShaders::load_level hand to glium (game/src/level.rs:424-496),
`DeviceLevel` / `Batch` ~ the GL buffers/textures and Renderer::update's draw loop
(engine/src/renderer.rs:98-157) -- executed by hand-written HIP kernels.

There is NO CPU fallback: if librdoom_hip.so is missing or a HIP call fails, this raises.
(The directory name contains '-', so import it with importlib.import_module('rust-doom_amd') or via
the `rust_doom_amd` shim at the repository root.)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'librdoom_hip.so')

KIND_FLAT, KIND_WALL, KIND_DECOR, KIND_SKY = 0, 1, 2, 3
ALL_KINDS = 0xF
NO_PRIMITIVE = 0xFFFFFFFF

STATIC_VERTEX = np.dtype([('a_pos', '<f4', 3), ('a_atlas_uv', '<f4', 2), ('a_tile_uv', '<f4', 2),
                          ('a_tile_size', '<f4', 2), ('a_scroll_rate', '<f4'), ('a_row_height', '<f4'),
                          ('a_num_frames', 'u1'), ('a_light', 'u1'), ('_pad', 'u1', 2)])
SPRITE_VERTEX = np.dtype([('a_pos', '<f4', 3), ('a_atlas_uv', '<f4', 2), ('a_tile_uv', '<f4', 2),
                          ('a_tile_size', '<f4', 2), ('a_local_x', '<f4'), ('a_num_frames', 'u1'),
                          ('a_light', 'u1'), ('_pad', 'u1', 2)])
POSE = np.dtype([('modelview', '<f4', 16), ('projection', '<f4', 16), ('time', '<f4'), ('_pad', '<f4')])
assert STATIC_VERTEX.itemsize == 48 and SPRITE_VERTEX.itemsize == 44 and POSE.itemsize == 136


class RdoomError(RuntimeError):
    def __init__(self, status, message):
        super().__init__('rdoom status %d: %s' % (status, message))
        self.status = status


class LevelDesc(ctypes.Structure):
    _fields_ = [
        ('static_verts', ctypes.c_void_p), ('n_static_verts', ctypes.c_uint32),
        ('static_indices', ctypes.c_void_p), ('n_static_indices', ctypes.c_uint32),
        ('sky_verts', ctypes.c_void_p), ('n_sky_verts', ctypes.c_uint32),
        ('sky_indices', ctypes.c_void_p), ('n_sky_indices', ctypes.c_uint32),
        ('decor_verts', ctypes.c_void_p), ('n_decor_verts', ctypes.c_uint32),
        ('decor_indices', ctypes.c_void_p), ('n_decor_indices', ctypes.c_uint32),
        ('draws', ctypes.c_void_p), ('n_draws', ctypes.c_uint32),
        ('flat_atlas', ctypes.c_void_p), ('flat_w', ctypes.c_uint32), ('flat_h', ctypes.c_uint32),
        ('wall_atlas', ctypes.c_void_p), ('wall_w', ctypes.c_uint32), ('wall_h', ctypes.c_uint32),
        ('decor_atlas', ctypes.c_void_p), ('decor_w', ctypes.c_uint32), ('decor_h', ctypes.c_uint32),
        ('sky_texture', ctypes.c_void_p), ('sky_w', ctypes.c_uint32), ('sky_h', ctypes.c_uint32),
        ('sky_tiled_band_size', ctypes.c_float),
        ('playpal', ctypes.c_void_p), ('colormap', ctypes.c_void_p)]


class Timings(ctypes.Structure):
    _fields_ = [('setup_ms', ctypes.c_float), ('raster_ms', ctypes.c_float), ('fragment_ms', ctypes.c_float),
                ('total_ms', ctypes.c_float), ('pixels', ctypes.c_uint64), ('visible_triangles', ctypes.c_uint64),
                ('fixup_pixels', ctypes.c_uint64)]


class PathStats(ctypes.Structure):
    _fields_ = [('poses', ctypes.c_uint32), ('bins_overflowed_poses', ctypes.c_uint32), ('tiles', ctypes.c_uint64), ('split_tiles', ctypes.c_uint64),
                ('tile_entries', ctypes.c_uint64), ('quadrants', ctypes.c_uint64), ('described_quadrants', ctypes.c_uint64)]


class HostTimings(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ('open_ms', 'textures_ms', 'level_lumps_ms', 'atlases_ms', 'analysis_ms', 'walk_ms')]


class Counters(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in
                ('num_wall_quads', 'num_floor_polys', 'num_ceil_polys', 'num_sky_wall_quads', 'num_sky_floor_polys',
                 'num_sky_ceil_polys', 'num_decors', 'num_static_tris', 'num_sky_tris', 'num_sprite_tris',
                 'num_objects', 'num_lights')]


# every symbol include/rdoom.h declares (tests check the library exports all of them)
API_SYMBOLS = [
    'rdoom_last_error', 'rdoom_device_count', 'rdoom_set_device', 'rdoom_level_create', 'rdoom_level_destroy',
    'rdoom_batch_create', 'rdoom_batch_destroy', 'rdoom_batch_render', 'rdoom_batch_render_timed', 'rdoom_batch_render_profiled', 'rdoom_batch_collect_timings',
    'rdoom_batch_framebuffer_device', 'rdoom_batch_finish', 'rdoom_batch_read_framebuffer', 'rdoom_batch_read_primitive_ids',
    'rdoom_wad_open', 'rdoom_wad_close', 'rdoom_wad_num_levels', 'rdoom_wad_level_name',
    'rdoom_wad_name_from_bytes', 'rdoom_wad_build_level', 'rdoom_built_destroy', 'rdoom_built_desc',
    'rdoom_built_counters', 'rdoom_built_lights_at', 'rdoom_built_start', 'rdoom_built_floor_centroids',
    'rdoom_pose_look', 'rdoom_selftest_fastmath', 'rdoom_debug_set', 'rdoom_wad_walk', 'rdoom_wad_build_level_chained', 'rdoom_batch_render_objects', 'rdoom_level_num_objects', 'rdoom_batch_enable_primitive_ids',
    'rdoom_wad_timings', 'rdoom_built_timings', 'rdoom_pose_from_player', 'rdoom_batch_framebuffer_pitch', 'rdoom_batch_path_stats',
    'rdoom_levelset_create', 'rdoom_level_num_levels', 'rdoom_batch_render_levels']

_lib = None


def lib():
    """Loads librdoom_hip.so (built by rust-doom_amd/build.py).  Raises if it is missing: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('%s is missing: run `python rust-doom_amd/build.py` (hipcc, gfx950). '
                              'There is no CPU fallback.' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.rdoom_last_error.restype = ctypes.c_char_p
        for name in API_SYMBOLS:
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            if name not in ('rdoom_last_error', 'rdoom_level_destroy', 'rdoom_batch_destroy', 'rdoom_wad_close',
                            'rdoom_built_destroy'):
                fn.restype = ctypes.c_int32
            elif name != 'rdoom_last_error':
                fn.restype = None
        _lib = L
    return _lib


def _check(status):
    if status != 0:
        raise RdoomError(status, lib().rdoom_last_error().decode('utf-8', 'replace'))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None and a.size else None


def device_count():
    n = ctypes.c_int32(0)
    _check(lib().rdoom_device_count(ctypes.byref(n)))
    return n.value


def set_device(i):
    _check(lib().rdoom_set_device(int(i)))


def debug_set(name, value=1):
    """rdoom_debug_set: test hooks selecting equivalent kernel paths (include/rdoom.h); 'reset' restores the defaults."""
    _check(lib().rdoom_debug_set(name.encode('ascii'), int(value)))


def selftest_fastmath():
    """rdoom_selftest_fastmath: exhaustive on-device check of the fragment kernel's exact division forms."""
    out = (ctypes.c_uint64 * 8)()
    _check(lib().rdoom_selftest_fastmath(out))
    keys = ('rcp_mismatches', 'div09_mismatches', 'inputs_swept', 'mod_violations', 'mod_samples', 'mod_certified',
            'mod_floor_differs', 'packed_mismatches')
    return dict(zip(keys, [int(x) for x in out]))


def wad_name(value):
    """WadName::from_bytes (wad/src/name.rs:41-75) -> 8 bytes; raises RdoomError on invalid names."""
    if isinstance(value, str):
        value = value.encode('utf-8')
    out = (ctypes.c_uint8 * 8)()
    _check(lib().rdoom_wad_name_from_bytes(bytes(value), len(value), out))
    return bytes(out)


def pose_look(eye, yaw, pitch, width, height, time=0.0):
    """rdoom_pose_look: the reference camera (player.rs:84-89,325-345) as one POSE record."""
    pose = np.zeros(1, POSE)
    e = (ctypes.c_float * 3)(*[float(x) for x in eye])
    _check(lib().rdoom_pose_look(e, ctypes.c_float(yaw), ctypes.c_float(pitch), int(width), int(height),
                                 ctypes.c_float(time), pose.ctypes.data_as(ctypes.c_void_p)))
    return pose[0]


def pose_from_player(pos, yaw, pitch, width, height, time=0.0):
    """rdoom_pose_from_player: the same camera in the reference's own binary32 arithmetic (Decomposed / Quaternion of cgmath);
    `pos` is the player's position -- the camera height (0.12) is added by the helper."""
    pose = np.zeros(1, POSE)
    e = (ctypes.c_float * 3)(*[float(x) for x in pos])
    _check(lib().rdoom_pose_from_player(e, ctypes.c_float(yaw), ctypes.c_float(pitch), int(width), int(height),
                                        ctypes.c_float(time), pose.ctypes.data_as(ctypes.c_void_p)))
    return pose[0]


def make_desc(arrays):
    """Builds a LevelDesc from a dict/object of numpy arrays (same field names as oracle BuiltLevel).
    Returns (desc, keepalive)."""
    g = (lambda k, d=None: arrays.get(k, d)) if isinstance(arrays, dict) else (lambda k, d=None: getattr(arrays, k, d))
    c = np.ascontiguousarray
    keep = dict(
        sv=c(g('static_vertices')), si=c(g('static_indices'), np.uint32),
        kv=c(g('sky_vertices'), np.float32).reshape(-1, 3), ki=c(g('sky_indices'), np.uint32),
        dv=c(g('decor_vertices', np.zeros(0, SPRITE_VERTEX))), di=c(g('decor_indices', np.zeros(0, np.uint32)), np.uint32),
        dr=c(g('draws'), np.uint32).reshape(-1, 4), fa=c(g('flat_atlas'), np.uint8), wa=c(g('wall_atlas'), np.uint16),
        da=c(g('decor_atlas', np.zeros((0, 0), np.uint16)), np.uint16),
        st=c(g('sky_texture'), np.uint16), pp=c(g('palette'), np.uint8), cm=c(g('colormap'), np.uint8))
    k = keep
    assert k['sv'].dtype.itemsize == 48, 'static vertices must be 48-byte StaticVertex records'
    assert k['dv'].dtype.itemsize == 44 or k['dv'].size == 0
    h2 = lambda a: (a.shape[1], a.shape[0]) if a.ndim == 2 and a.size else (0, 0)
    d = LevelDesc()
    d.static_verts, d.n_static_verts = _ptr(k['sv']), len(k['sv'])
    d.static_indices, d.n_static_indices = _ptr(k['si']), len(k['si'])
    d.sky_verts, d.n_sky_verts = _ptr(k['kv']), len(k['kv'])
    d.sky_indices, d.n_sky_indices = _ptr(k['ki']), len(k['ki'])
    d.decor_verts, d.n_decor_verts = _ptr(k['dv']), len(k['dv'])
    d.decor_indices, d.n_decor_indices = _ptr(k['di']), len(k['di'])
    d.draws, d.n_draws = _ptr(k['dr']), len(k['dr'])
    d.flat_atlas = _ptr(k['fa'])
    d.flat_w, d.flat_h = h2(k['fa'])
    d.wall_atlas = _ptr(k['wa'])
    d.wall_w, d.wall_h = h2(k['wa'])
    d.decor_atlas = _ptr(k['da'])
    d.decor_w, d.decor_h = h2(k['da'])
    d.sky_texture = _ptr(k['st'])
    d.sky_w, d.sky_h = h2(k['st'])
    d.sky_tiled_band_size = float(g('sky_band', 0.0))
    d.playpal, d.colormap = _ptr(k['pp']), _ptr(k['cm'])
    return d, keep


# ---- trait wad::LevelVisitor over the C ABI (include/rdoom.h: rdoom_visitor_vtbl) -----------------------------------
class LightInfo(ctypes.Structure):
    _fields_ = [('level', ctypes.c_float), ('has_effect', ctypes.c_int32), ('effect_kind', ctypes.c_int32),
                ('alt_level', ctypes.c_float), ('speed', ctypes.c_float), ('duration', ctypes.c_float), ('sync', ctypes.c_float)]


class StaticQuad(ctypes.Structure):
    _fields_ = [('object_id', ctypes.c_uint32), ('v1', ctypes.c_float * 2), ('v2', ctypes.c_float * 2),
                ('tex_start', ctypes.c_float * 2), ('tex_end', ctypes.c_float * 2), ('height_range', ctypes.c_float * 2),
                ('light_info', ctypes.POINTER(LightInfo)), ('scroll', ctypes.c_float), ('has_tex_name', ctypes.c_int32),
                ('tex_name', ctypes.c_uint8 * 8), ('blocker', ctypes.c_int32)]


class StaticPoly(ctypes.Structure):
    _fields_ = [('object_id', ctypes.c_uint32), ('vertices', ctypes.POINTER(ctypes.c_float)), ('n_vertices', ctypes.c_uint32),
                ('height', ctypes.c_float), ('light_info', ctypes.POINTER(LightInfo)), ('tex_name', ctypes.c_uint8 * 8)]


class SkyQuad(ctypes.Structure):
    _fields_ = [('object_id', ctypes.c_uint32), ('v1', ctypes.c_float * 2), ('v2', ctypes.c_float * 2),
                ('height_range', ctypes.c_float * 2)]


class SkyPoly(ctypes.Structure):
    _fields_ = [('object_id', ctypes.c_uint32), ('vertices', ctypes.POINTER(ctypes.c_float)), ('n_vertices', ctypes.c_uint32),
                ('height', ctypes.c_float)]


class Decor(ctypes.Structure):
    _fields_ = [('object_id', ctypes.c_uint32), ('low', ctypes.c_float * 3), ('high', ctypes.c_float * 3),
                ('half_width', ctypes.c_float), ('light_info', ctypes.POINTER(LightInfo)), ('tex_name', ctypes.c_uint8 * 8)]


class Line2f(ctypes.Structure):
    _fields_ = [('origin', ctypes.c_float * 2), ('displace', ctypes.c_float * 2), ('length', ctypes.c_float)]


_VP = ctypes.c_void_p
_VISITOR_SIGNATURES = [
    ('visit_wall_quad', (ctypes.POINTER(StaticQuad),)), ('visit_floor_poly', (ctypes.POINTER(StaticPoly),)),
    ('visit_ceil_poly', (ctypes.POINTER(StaticPoly),)), ('visit_floor_sky_poly', (ctypes.POINTER(SkyPoly),)),
    ('visit_ceil_sky_poly', (ctypes.POINTER(SkyPoly),)), ('visit_sky_quad', (ctypes.POINTER(SkyQuad),)),
    ('visit_marker', (ctypes.POINTER(ctypes.c_float), ctypes.c_float, ctypes.c_int32, ctypes.c_uint32)),
    ('visit_decor', (ctypes.POINTER(Decor),)), ('visit_bsp_root', (ctypes.POINTER(Line2f),)),
    ('visit_bsp_node', (ctypes.POINTER(Line2f), ctypes.c_int32)), ('visit_bsp_leaf', (ctypes.c_int32,)),
    ('visit_bsp_leaf_end', ()), ('visit_bsp_node_end', ())]
_VISITOR_TYPES = {name: ctypes.CFUNCTYPE(None, _VP, *args) for name, args in _VISITOR_SIGNATURES}


class VisitorVtbl(ctypes.Structure):
    _fields_ = [(name, _VISITOR_TYPES[name]) for name, _ in _VISITOR_SIGNATURES]


def make_visitor(obj):
    """rdoom_visitor_vtbl from any object: a method named like a callback of trait LevelVisitor (visitor.rs:65-116)
    receives the payload (ctypes structure pointers are dereferenced); missing methods stay NULL = the trait's default.
    Returns (vtbl, keepalive)."""
    vt, keep = VisitorVtbl(), []
    for name, args in _VISITOR_SIGNATURES:
        fn = getattr(obj, name, None)
        if fn is None:
            continue

        def thunk(_user, *a, _fn=fn):
            _fn(*[x.contents if hasattr(x, 'contents') and not isinstance(x, ctypes.POINTER(ctypes.c_float)) else x for x in a])
        cb = _VISITOR_TYPES[name](thunk)
        keep.append(cb)
        setattr(vt, name, cb)
    return vt, keep


def _close_quietly(obj):
    """__del__ of the handle classes: at interpreter shutdown the module's globals may already be gone"""
    try:
        obj.close()
    except Exception:
        pass


class Wad:
    """wad::Archive + TextureDirectory behind rdoom_wad_open (wad/src/archive.rs:36-60, tex.rs:53-107)."""

    def __init__(self, wad_path, metadata_path):
        self._h = ctypes.c_void_p()
        _check(lib().rdoom_wad_open(os.fsencode(wad_path), os.fsencode(metadata_path), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            lib().rdoom_wad_close(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        _close_quietly(self)

    def timings(self):
        """open_ms, textures_ms of rdoom_wad_open (single thread, steady clock inside the library)"""
        t = HostTimings()
        _check(lib().rdoom_wad_timings(self._h, ctypes.byref(t)))
        return {n: getattr(t, n) for n, _ in HostTimings._fields_}

    def num_levels(self):
        n = ctypes.c_uint32()
        _check(lib().rdoom_wad_num_levels(self._h, ctypes.byref(n)))
        return n.value

    def level_name(self, index):
        buf = ctypes.create_string_buffer(9)
        _check(lib().rdoom_wad_level_name(self._h, int(index), buf))
        return buf.value.decode('ascii')

    def build_level(self, index, gpu_tessellation=False, visitor=None):
        """visitor: an object with LevelVisitor methods, chained after the Builder (game/src/level.rs:378-382)"""
        return BuiltLevel(self, index, gpu_tessellation, visitor)

    def walk(self, index, visitor):
        """WadSystem::walk (game/src/wad_system.rs:47-56) with the caller's visitor only"""
        vt, keep = make_visitor(visitor)
        _check(lib().rdoom_wad_walk(self._h, int(index), ctypes.byref(vt), None))
        del keep


class BuiltLevel:
    """Result of game::level::Builder::build + GameShaders::load_level."""

    def __init__(self, wad, index, gpu_tessellation=False, visitor=None):
        self._h = ctypes.c_void_p()
        self._wad = wad
        if visitor is None:
            _check(lib().rdoom_wad_build_level(wad._h, int(index), int(bool(gpu_tessellation)), ctypes.byref(self._h)))
        else:
            vt, keep = make_visitor(visitor)
            _check(lib().rdoom_wad_build_level_chained(wad._h, int(index), int(bool(gpu_tessellation)), ctypes.byref(vt), None,
                                                       ctypes.byref(self._h)))
            del keep
        self.desc = LevelDesc()
        _check(lib().rdoom_built_desc(self._h, ctypes.byref(self.desc)))

    def close(self):
        if self._h:
            lib().rdoom_built_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        _close_quietly(self)

    def _view(self, ptr, count, dtype):
        if not ptr or count == 0:
            return np.zeros(0, dtype)
        dt = np.dtype(dtype)
        buf = (ctypes.c_uint8 * (count * dt.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dt, count).copy()

    def arrays(self):
        """Copies of every array of the descriptor, keyed like the oracle's BuiltLevel fields."""
        d = self.desc
        return dict(
            static_vertices=self._view(d.static_verts, d.n_static_verts, STATIC_VERTEX),
            static_indices=self._view(d.static_indices, d.n_static_indices, np.uint32),
            sky_vertices=self._view(d.sky_verts, d.n_sky_verts * 3, np.float32).reshape(-1, 3),
            sky_indices=self._view(d.sky_indices, d.n_sky_indices, np.uint32),
            decor_vertices=self._view(d.decor_verts, d.n_decor_verts, SPRITE_VERTEX),
            decor_indices=self._view(d.decor_indices, d.n_decor_indices, np.uint32),
            draws=self._view(d.draws, d.n_draws * 4, np.uint32).reshape(-1, 4),
            flat_atlas=self._view(d.flat_atlas, d.flat_w * d.flat_h, np.uint8).reshape(d.flat_h, d.flat_w),
            wall_atlas=self._view(d.wall_atlas, d.wall_w * d.wall_h, np.uint16).reshape(d.wall_h, d.wall_w),
            decor_atlas=self._view(d.decor_atlas, d.decor_w * d.decor_h, np.uint16).reshape(d.decor_h, d.decor_w),
            sky_texture=self._view(d.sky_texture, d.sky_w * d.sky_h, np.uint16).reshape(d.sky_h, d.sky_w),
            sky_band=np.float32(d.sky_tiled_band_size),
            palette=self._view(d.playpal, 768, np.uint8), colormap=self._view(d.colormap, 32 * 256, np.uint8))

    def counters(self):
        c = Counters()
        _check(lib().rdoom_built_counters(self._h, ctypes.byref(c)))
        return {n: getattr(c, n) for n, _ in Counters._fields_}

    def timings(self):
        """level_lumps_ms, atlases_ms, analysis_ms, walk_ms of this build"""
        t = HostTimings()
        _check(lib().rdoom_built_timings(self._h, ctypes.byref(t)))
        return {n: getattr(t, n) for n, _ in HostTimings._fields_}

    def lights_at(self, time):
        out = np.zeros(256, np.uint8)
        _check(lib().rdoom_built_lights_at(self._h, ctypes.c_float(time), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def start(self):
        pos = (ctypes.c_float * 3)()
        yaw = ctypes.c_float()
        _check(lib().rdoom_built_start(self._h, pos, ctypes.byref(yaw)))
        return np.array(pos[:], np.float32), np.float32(yaw.value)

    def floor_centroids(self):
        p = ctypes.c_void_p()
        n = ctypes.c_uint32()
        _check(lib().rdoom_built_floor_centroids(self._h, ctypes.byref(p), ctypes.byref(n)))
        return self._view(p.value, n.value * 3, np.float32).reshape(-1, 3)


class DeviceLevel:
    """Level arrays resident in HBM (replaces the GL vertex/index buffers and textures)."""

    def __init__(self, source):
        if isinstance(source, BuiltLevel):
            desc, self._keep = source.desc, source
        elif isinstance(source, LevelDesc):
            desc, self._keep = source, None
        else:
            desc, self._keep = make_desc(source)
        self._h = ctypes.c_void_p()
        _check(lib().rdoom_level_create(ctypes.byref(desc), ctypes.byref(self._h)))

    def num_objects(self):
        n = ctypes.c_uint32()
        _check(lib().rdoom_level_num_objects(self._h, ctypes.byref(n)))
        return n.value

    def close(self):
        if self._h:
            lib().rdoom_level_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        _close_quietly(self)


class DeviceLevelSet(DeviceLevel):
    """Several levels resident in HBM as ONE handle (rdoom_levelset_create): a Batch created on it renders poses of different
    levels in one launch set (Batch.render(..., level_of_pose=...)).  `sources`: BuiltLevel / LevelDesc / array dicts."""

    def __init__(self, sources):
        descs, self._keep = [], []
        for source in sources:
            if isinstance(source, BuiltLevel):
                desc, keep = source.desc, source
            elif isinstance(source, LevelDesc):
                desc, keep = source, None
            else:
                desc, keep = make_desc(source)
            descs.append(desc)
            self._keep.append((desc, keep))
        arr = (ctypes.POINTER(LevelDesc) * len(descs))(*[ctypes.pointer(d) for d in descs])
        self._h = ctypes.c_void_p()
        _check(lib().rdoom_levelset_create(arr, len(descs), ctypes.byref(self._h)))

    def num_levels(self):
        n = ctypes.c_uint32()
        _check(lib().rdoom_level_num_levels(self._h, ctypes.byref(n)))
        return n.value


class Batch:
    """A pose batch: device scratch + the three kernels (setup, tiled raster, fragment)."""

    def __init__(self, level, width, height, max_poses):
        self.level, self.width, self.height, self.max_poses = level, int(width), int(height), int(max_poses)
        self._h = ctypes.c_void_p()
        _check(lib().rdoom_batch_create(level._h, self.width, self.height, self.max_poses, ctypes.byref(self._h)))
        self.last_n = 0

    def close(self):
        if self._h:
            lib().rdoom_batch_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        _close_quietly(self)

    @staticmethod
    def _prep(poses, lights):
        poses = np.ascontiguousarray(poses, POSE).reshape(-1)
        lights = np.ascontiguousarray(lights, np.uint8)
        if lights.size == 256:
            stride = 0
        else:
            assert lights.size == 256 * len(poses), 'lights must be (256,) or (n_poses, 256)'
            stride = 256
        return poses, lights, stride

    def _render_levels(self, poses, lights, stride, kinds, stream, level_of_pose, object_modelviews, profiled):
        lop = np.ascontiguousarray(level_of_pose, np.uint32).reshape(-1)
        assert len(lop) == len(poses), 'level_of_pose: one level index per pose'
        om, n_obj = None, 0
        if object_modelviews is not None:
            om = np.ascontiguousarray(object_modelviews, np.float32).reshape(len(poses), -1, 16)
            n_obj = om.shape[1]
        _check(lib().rdoom_batch_render_levels(
            self._h, poses.ctypes.data_as(ctypes.c_void_p), lop.ctypes.data_as(ctypes.c_void_p), lights.ctypes.data_as(ctypes.c_void_p),
            stride, len(poses), int(kinds), ctypes.c_void_p(stream or 0), om.ctypes.data_as(ctypes.c_void_p) if om is not None else None,
            n_obj, 1 if profiled else 0))

    def render_profiled(self, poses, lights, kinds=ALL_KINDS, stream=None, level_of_pose=None):
        """rdoom_batch_render_profiled: asynchronous, the per-kernel events stay pending (at most 64 renders)"""
        poses, lights, stride = self._prep(poses, lights)
        self.last_n = len(poses)
        if level_of_pose is not None:
            return self._render_levels(poses, lights, stride, kinds, stream, level_of_pose, None, True)
        _check(lib().rdoom_batch_render_profiled(self._h, poses.ctypes.data_as(ctypes.c_void_p), lights.ctypes.data_as(ctypes.c_void_p),
                                                 stride, len(poses), int(kinds), ctypes.c_void_p(stream or 0)))

    def collect_timings(self):
        """rdoom_batch_collect_timings: sums over the pending profiled renders + their number"""
        t, n = Timings(), ctypes.c_uint32()
        _check(lib().rdoom_batch_collect_timings(self._h, ctypes.byref(t), ctypes.byref(n)))
        out = {k: getattr(t, k) for k, _ in Timings._fields_}
        out['renders'] = n.value
        return out

    def render(self, poses, lights, kinds=ALL_KINDS, stream=None, timed=False, object_modelviews=None, level_of_pose=None):
        """rdoom_batch_render(_timed): asynchronous unless timed; returns Timings fields when timed.
        object_modelviews: optional (n_poses, n_objects, 16) u_modelview per object -> rdoom_batch_render_objects.
        level_of_pose: (n_poses,) level index of every pose, for a batch on a DeviceLevelSet -> rdoom_batch_render_levels."""
        poses, lights, stride = self._prep(poses, lights)
        self.last_n = len(poses)
        if level_of_pose is not None:
            assert not timed, 'timed renders of a level set: use render_profiled + collect_timings'
            return self._render_levels(poses, lights, stride, kinds, stream, level_of_pose, object_modelviews, False)
        if object_modelviews is not None:
            om = np.ascontiguousarray(object_modelviews, np.float32).reshape(len(poses), -1, 16)
            _check(lib().rdoom_batch_render_objects(
                self._h, poses.ctypes.data_as(ctypes.c_void_p), lights.ctypes.data_as(ctypes.c_void_p), stride,
                len(poses), int(kinds), ctypes.c_void_p(stream or 0), om.ctypes.data_as(ctypes.c_void_p), om.shape[1]))
            return None
        args = (self._h, poses.ctypes.data_as(ctypes.c_void_p), lights.ctypes.data_as(ctypes.c_void_p), stride,
                len(poses), int(kinds), ctypes.c_void_p(stream or 0))
        if not timed:
            _check(lib().rdoom_batch_render(*args))
            return None
        t = Timings()
        _check(lib().rdoom_batch_render_timed(*args, ctypes.byref(t)))
        return {n: getattr(t, n) for n, _ in Timings._fields_}

    def finish(self):
        """rdoom_batch_finish: wait for the last render and raise if the device found a problem"""
        _check(lib().rdoom_batch_finish(self._h))

    def framebuffer_device_ptr(self):
        p = ctypes.c_void_p()
        _check(lib().rdoom_batch_framebuffer_device(self._h, ctypes.byref(p)))
        return p.value

    def path_stats(self):
        """rdoom_batch_path_stats: which paths the last render took (overflowed poses, split tile lists, described quadrants)"""
        s = PathStats()
        _check(lib().rdoom_batch_path_stats(self._h, ctypes.byref(s)))
        return {n: getattr(s, n) for n, _ in PathStats._fields_}

    def framebuffer_pitch(self):
        """bytes between rows of the device framebuffer (the width, or the next multiple of 8 when width % 4 != 0)"""
        n = ctypes.c_uint32()
        _check(lib().rdoom_batch_framebuffer_pitch(self._h, ctypes.byref(n)))
        return n.value

    def read_framebuffer(self, first=0, count=None):
        count = self.last_n - first if count is None else count
        out = np.zeros((count, self.height, self.width), np.uint8)
        _check(lib().rdoom_batch_read_framebuffer(self._h, int(first), int(count), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def enable_primitive_ids(self):
        _check(lib().rdoom_batch_enable_primitive_ids(self._h))

    def read_primitive_ids(self, first=0, count=None):
        count = self.last_n - first if count is None else count
        out = np.zeros((count, self.height, self.width), np.uint32)
        _check(lib().rdoom_batch_read_primitive_ids(self._h, int(first), int(count),
                                                    out.ctypes.data_as(ctypes.c_void_p)))
        return out
