"""Shared test helpers: synthetic IWAD location, pose generation (numpy f32), PNG dump."""
import hashlib
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import importlib  # noqa: E402

_syn = importlib.import_module('rust-doom_amd.synthetic')
GOLDEN, WAD_PATH, BIG_WAD_PATH, META_PATH = _syn.GOLDEN, _syn.WAD_PATH, _syn.BIG_WAD_PATH, _syn.META_PATH
ensure_wad, ensure_big_wad, wad_digest = _syn.ensure_wad, _syn.ensure_big_wad, _syn.wad_digest
F = np.float32


def perspective(fovy_deg, aspect, near, far):
    """cgmath::perspective (engine/src/projections.rs:93-101), column-major 16 floats."""
    f = F(1.0) / F(np.tan(np.float64(fovy_deg) * np.pi / 360.0))
    m = np.zeros((4, 4), np.float32)  # m[c][r]
    m[0][0] = f / F(aspect)
    m[1][1] = f
    m[2][2] = (F(far) + F(near)) / (F(near) - F(far))
    m[2][3] = F(-1.0)
    m[3][2] = (F(2.0) * F(far) * F(near)) / (F(near) - F(far))
    return m.reshape(16)


def reference_projection(width, height):
    """game/src/player.rs:84-89,336-344: fovy 65 deg, aspect*1.2, near 0.01, far 100."""
    return perspective(65.0, (F(width) / F(height)) * F(1.2), 0.01, 100.0)


def view_matrix(eye, yaw, pitch):
    """inverse(T(eye) * Ry(yaw) * Rx(pitch)), column-major 16 floats (engine/src/renderer.rs:78-87)."""
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], np.float64)
    rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]], np.float64)
    r = ry @ rx
    m = np.eye(4)
    m[:3, :3] = r.T
    m[:3, 3] = -(r.T @ np.asarray(eye, np.float64))
    return m.T.astype(np.float32).reshape(16)  # row-major transpose -> column-major


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b''.join(b'\0' + rgb[y].tobytes() for y in range(h))

    def chunk(t, d):
        c = struct.pack('>I', len(d)) + t + d
        return c + struct.pack('>I', zlib.crc32(t + d) & 0xFFFFFFFF)

    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def fb_to_png(path, fb, playpal):
    """fb: (h,w) palette indices with row 0 = bottom; playpal: 768 bytes."""
    pal = np.asarray(playpal, np.uint8).reshape(256, 3)
    write_png(path, pal[fb[::-1]])


def dirtying_poses(poses):
    """ANOTHER pose set of the same size: every frame slot gets its neighbour's pose, turned by one radian about the view's
    y axis (new modelview = Ry(1) o modelview; column-major).  Rendered into a batch before the checked poses, it leaves
    another frame's visibility words, quadrant table and tile lists wherever the checked render does not write them."""
    out = np.roll(np.array(poses, copy=True), 1)
    c, s = np.float32(np.cos(1.0)), np.float32(np.sin(1.0))
    ry = np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], np.float32)
    for p in out.reshape(-1):
        m = np.asarray(p['modelview'], np.float32).reshape(4, 4).T  # column-major storage -> matrix
        p['modelview'] = (ry @ m).T.reshape(16)
    return out


def render_checked(batch, poses, lights, want_prim=True, **kw):
    """The render every HIP-vs-oracle assertion goes through.  Since the rasteriser leaves out the
    visibility words of quadrants its table describes, a reader of a word nobody wrote must meet ANOTHER frame's record, not
    the zeros of a fresh allocation -- so the batch is dirtied by a render of other poses first; and both instantiations are
    checked: the one without primitive ids (the path bench.py times), then, dirtied again, the one with them.
    Returns (framebuffers of the plain path, framebuffers of the id path, primitive ids) -- callers compare all three with the
    oracle.  kw: kinds=, object_modelviews= as Batch.render takes them."""
    other = dirtying_poses(poses)
    olights = lights
    if getattr(lights, 'ndim', 1) == 2 and len(lights) == len(np.atleast_1d(poses)):
        olights = np.roll(lights, 1, axis=0)
    dirty_kw = {k: v for k, v in kw.items() if k != 'object_modelviews'}   # (the dirtying frames need no displaced doors)
    batch.render(other, olights, **dirty_kw)
    batch.render(poses, lights, **kw)
    fb_plain = batch.read_framebuffer()
    if not want_prim:
        return fb_plain, None, None
    batch.enable_primitive_ids()
    batch.render(other, olights, **dirty_kw)
    batch.render(poses, lights, **kw)
    return fb_plain, batch.read_framebuffer(), batch.read_primitive_ids()


def apply_stress_hooks():
    """hand-run stress scripts: RDOOM_STRESS_HOOKS="qpath=1 frag_bw=2" sets rdoom_debug_set hooks (equivalent paths: same images)
    before anything renders -- the library itself never reads the environment"""
    import rust_doom_amd as rd
    hooks = os.environ.get('RDOOM_STRESS_HOOKS', '').split()
    for item in hooks:
        name, _, value = item.partition('=')
        rd.debug_set(name, int(value or 1))
    return hooks
