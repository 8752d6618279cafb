"""ORACLE (test infrastructure only): ctypes front-end of oracle/raster_oracle.c."""
import ctypes
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle_raster.so')
KIND_FLAT, KIND_WALL, KIND_DECOR, KIND_SKY = 0, 1, 2, 3
ALL_KINDS = 0xF
NO_PRIM = 0xFFFFFFFF


def build(force=False):
    src = os.path.join(_HERE, 'raster_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, '-B', '_build/liboracle_raster.so'])
    return _SO


class _Level(ctypes.Structure):
    _fields_ = [('static_verts', ctypes.c_void_p), ('static_indices', ctypes.c_void_p),
                ('sky_verts', ctypes.c_void_p), ('sky_indices', ctypes.c_void_p),
                ('draws', ctypes.c_void_p), ('n_draws', ctypes.c_uint32),
                ('flat_atlas', ctypes.c_void_p), ('flat_w', ctypes.c_uint32), ('flat_h', ctypes.c_uint32),
                ('wall_atlas', ctypes.c_void_p), ('wall_w', ctypes.c_uint32), ('wall_h', ctypes.c_uint32),
                ('sky_tex', ctypes.c_void_p), ('sky_w', ctypes.c_uint32), ('sky_h', ctypes.c_uint32),
                ('sky_band', ctypes.c_float), ('colormap', ctypes.c_void_p),
                ('decor_verts', ctypes.c_void_p), ('decor_indices', ctypes.c_void_p),
                ('decor_atlas', ctypes.c_void_p), ('decor_w', ctypes.c_uint32), ('decor_h', ctypes.c_uint32)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        try:
            _lib = ctypes.CDLL(build())
        except OSError:
            _lib = ctypes.CDLL(build(force=True))
        _lib.oracle_render.restype = ctypes.c_int
        _lib.oracle_render_objects.restype = ctypes.c_int
        _lib.oracle_render_batch.restype = ctypes.c_int
        _lib.oracle_render_varyings.restype = ctypes.c_int
        _lib.oracle_shade_varyings.restype = ctypes.c_int
    return _lib


class RasterOracle:
    """Holds the level arrays (any object exposing the BuiltLevel fields of wad_oracle.build_level, or
    a dict with the same keys) and renders poses on the CPU."""

    def __init__(self, lvl):
        g = (lambda k, d=None: lvl.get(k, d)) if isinstance(lvl, dict) else (lambda k, d=None: getattr(lvl, k, d))
        c = np.ascontiguousarray
        self._keep = dict(
            sv=c(g('static_vertices')), si=c(g('static_indices'), np.uint32),
            kv=c(g('sky_vertices'), np.float32), ki=c(g('sky_indices'), np.uint32),
            dr=c(g('draws'), np.uint32), fa=c(g('flat_atlas'), np.uint8), wa=c(g('wall_atlas'), np.uint16),
            st=c(g('sky_texture'), np.uint16), cm=c(g('colormap'), np.uint8),
            dv=c(g('decor_vertices', np.zeros(0, np.uint8))), di=c(g('decor_indices', np.zeros(0, np.uint32)), np.uint32),
            da=c(g('decor_atlas', np.zeros((0, 0), np.uint16)), np.uint16))
        k = self._keep
        assert k['sv'].dtype.itemsize == 48 and k['cm'].size == 32 * 256
        assert k['dv'].size == 0 or k['dv'].dtype.itemsize == 44, 'decor vertices must be 44-byte SpriteVertex records'
        for a in (k['fa'], k['wa'], k['da']):
            for d in a.shape:
                assert d == 0 or (d & (d - 1)) == 0, 'atlas sizes must be powers of two'
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self.level = _Level(p(k['sv']), p(k['si']), p(k['kv']), p(k['ki']), p(k['dr']), len(k['dr']),
                            p(k['fa']), k['fa'].shape[1] if k['fa'].ndim == 2 else 0, k['fa'].shape[0],
                            p(k['wa']), k['wa'].shape[1] if k['wa'].ndim == 2 else 0, k['wa'].shape[0],
                            p(k['st']), k['st'].shape[1], k['st'].shape[0], float(g('sky_band')), p(k['cm']),
                            p(k['dv']), p(k['di']), p(k['da']),
                            k['da'].shape[1] if k['da'].ndim == 2 and k['da'].size else 0,
                            k['da'].shape[0] if k['da'].ndim == 2 and k['da'].size else 0)

    def render(self, modelview, projection, time, lights, width, height, kinds=ALL_KINDS, want_prim=False,
               object_modelviews=None):
        """object_modelviews: optional (n_objects, 16) u_modelview per object (renderer.rs:120-132)."""
        lib = _load()
        mv = np.ascontiguousarray(modelview, np.float32).reshape(16)
        pr = np.ascontiguousarray(projection, np.float32).reshape(16)
        li = np.ascontiguousarray(lights, np.uint8).reshape(256)
        fb = np.zeros((height, width), np.uint8)
        prim = np.zeros((height, width), np.uint32) if want_prim else None
        om = None if object_modelviews is None else np.ascontiguousarray(object_modelviews, np.float32).reshape(-1, 16)
        rc = lib.oracle_render_objects(ctypes.byref(self.level), mv.ctypes.data_as(ctypes.c_void_p),
                                       pr.ctypes.data_as(ctypes.c_void_p), ctypes.c_float(float(time)),
                                       li.ctypes.data_as(ctypes.c_void_p), int(width), int(height), ctypes.c_uint32(kinds),
                                       fb.ctypes.data_as(ctypes.c_void_p),
                                       prim.ctypes.data_as(ctypes.c_void_p) if want_prim else None,
                                       om.ctypes.data_as(ctypes.c_void_p) if om is not None else None,
                                       ctypes.c_uint32(0 if om is None else len(om)))
        if rc:
            raise MemoryError('oracle_render failed')
        return (fb, prim) if want_prim else fb

    def render_varyings(self, modelview, projection, time, lights, width, height, kinds=ALL_KINDS):
        """(fb, prim, var): the frame, the winning primitive ids and, per pixel, (v_tile_uv.x, v_tile_uv.y, v_dist) as the
        oracle's binary32 arithmetic evaluates them for the winning fragment (census support, tests/gl_census.py)."""
        lib = _load()
        mv = np.ascontiguousarray(modelview, np.float32).reshape(16)
        pr = np.ascontiguousarray(projection, np.float32).reshape(16)
        li = np.ascontiguousarray(lights, np.uint8).reshape(256)
        fb = np.zeros((height, width), np.uint8)
        prim = np.zeros((height, width), np.uint32)
        var = np.zeros((height, width, 3), np.float32)
        rc = lib.oracle_render_varyings(ctypes.byref(self.level), mv.ctypes.data_as(ctypes.c_void_p),
                                        pr.ctypes.data_as(ctypes.c_void_p), ctypes.c_float(float(time)),
                                        li.ctypes.data_as(ctypes.c_void_p), int(width), int(height), ctypes.c_uint32(kinds),
                                        fb.ctypes.data_as(ctypes.c_void_p), prim.ctypes.data_as(ctypes.c_void_p),
                                        var.ctypes.data_as(ctypes.c_void_p))
        if rc:
            raise MemoryError('oracle_render_varyings failed')
        return fb, prim, var

    def shade_varyings(self, time, lights, prim, var):
        """The oracle's binary32 fragment stage (F2..F6) on varyings from outside: prim (h, w) u32 winners (NO_PRIM: none),
        var (h, w, 3) float32 = (v_tile_uv, v_dist), or the folded sky uv.  Returns (h, w) u16: palette index, 0x100 =
        discarded by the alpha test, 0xFFFF = no primitive."""
        lib = _load()
        prim = np.ascontiguousarray(prim, np.uint32)
        var = np.ascontiguousarray(var, np.float32)
        h, w = prim.shape
        li = np.ascontiguousarray(lights, np.uint8).reshape(256)
        out = np.zeros((h, w), np.uint16)
        rc = lib.oracle_shade_varyings(ctypes.byref(self.level), ctypes.c_float(float(time)), li.ctypes.data_as(ctypes.c_void_p),
                                       int(w), int(h), prim.ctypes.data_as(ctypes.c_void_p), var.ctypes.data_as(ctypes.c_void_p),
                                       out.ctypes.data_as(ctypes.c_void_p))
        if rc:
            raise MemoryError('oracle_shade_varyings failed')
        return out

    def render_batch(self, poses, lights, width, height, kinds=ALL_KINDS, threads=1):
        """poses: (n,33) float32 [modelview16, projection16, time]; lights: (n,256) u8."""
        lib = _load()
        poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 33)
        lights = np.ascontiguousarray(lights, np.uint8).reshape(-1, 256)
        n = len(poses)
        out = np.zeros((n, height, width), np.uint8)

        def run(lo, hi):
            if hi > lo:
                rc = lib.oracle_render_batch(ctypes.byref(self.level), poses[lo:hi].ctypes.data_as(ctypes.c_void_p),
                                             lights[lo:hi].ctypes.data_as(ctypes.c_void_p), hi - lo, int(width),
                                             int(height), ctypes.c_uint32(kinds),
                                             out[lo:hi].ctypes.data_as(ctypes.c_void_p))
                if rc:
                    raise MemoryError('oracle_render_batch failed')

        threads = max(1, min(threads, n))
        if threads == 1:
            run(0, n)
        else:
            cuts = [n * t // threads for t in range(threads + 1)]
            with ThreadPoolExecutor(threads) as ex:
                list(ex.map(lambda t: run(cuts[t], cuts[t + 1]), range(threads)))
        return out
