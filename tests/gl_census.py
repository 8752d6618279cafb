"""Mismatch census: oracle / HIP frame vs the GL readback of the reference's own shaders (tests/gl_readback.py).

OpenGL does not define results to the bit: SwiftShader snaps vertices to 1/16 pixel (GL_SUBPIXEL_BITS = 4), builds its
interpolation planes from the snapped positions and evaluates them in its own operation order.  Every legitimate
difference therefore sits on a discontinuity of the pipeline -- a primitive edge, a texel boundary, a COLORMAP-row
boundary, the alpha test flipping on a texel boundary, or two surfaces closer in depth than the depth buffer resolves.
This module labels each mismatching pixel with the discontinuity that explains it, or `other` if none does.

The judge of "explains" is an independent float64 evaluation of the shader text at the pixel centre and at sample
positions displaced by up to JITTER pixels (the reach of a 1/16-pixel vertex snap plus interpolation rounding):
  * winners differ and the centre is within JITTER of an edge of either winner, or either winner is thinner than
    JITTER there (nearly collinear fan triangles: facing and coverage are decided by rounding) -> `edge`
  * winners differ, an alpha-tested winner's opacity flips within JITTER                 -> `alpha texel boundary`
  * winners differ, both cover robustly, depths within DEPTH_TIE at the centre or at
    samples displaced by up to JITTER (grazing surfaces: many depth steps per pixel)       -> `depth tie`
  * same winner; the reference's fragment arithmetic applied to the varyings GL ITSELF interpolated at the pixel
    (auxiliary pass, within VARYING_TOL of the float64 value), or to a displaced sample, reproduces the GL colour
    through another texel                                                                -> `texel boundary`
  * same winner, ... through another COLORMAP row                                        -> `colormap-row boundary`
  * anything else                                                                        -> `other`
A systematic difference (half-texel offset, wrong fill rule, wrong depth rounding, wrong row formula) produces `other`
pixels in bulk or blows the total bound; neither is tolerated by tests/test_gl_readback.py.
"""
import math

import numpy as np

KIND_FLAT, KIND_WALL, KIND_DECOR, KIND_SKY = 0, 1, 2, 3
JITTER = 0.0625          # pixels = 2^-GL_SUBPIXEL_BITS: what GL guarantees about vertex positions
# 3 units of the 24-bit depth buffer + 5 binary32 roundoffs of a depth near 1 (one unit each) in GL's own plane evaluation
# (3840x2160: two surfaces 7.7 units apart over the jitter square went the other way; 1920x1080 and below: never more than 3)
DEPTH_TIE = 8.0 / 16777215.0
VARYING_TOL_ABS = 1.0 / 16.0      # texels: how far GL's own interpolated v_tile_uv may sit from the float64 value ...
VARYING_TOL_REL = 1.0 / 128.0     # ... plus this fraction of its magnitude (and of v_dist): SwiftShader's measured envelope
ROUNDING_MARGIN = 2.0 ** -11      # binary32 v_dist vs float64 at the pixel centre, relative (far slivers: measured 2.1e-4)
# binary32 v_tile_uv vs float64 at the pixel centre: ROUNDING_K unit roundoffs of the planes' condition sums
# (|a x| + |b y| + |c|) / rw for u/w, plus |tu| times the same for 1/w.  Measured on 100 000 pixels of five 1920x1080
# frames through oracle_render_varyings: median 0.3, 99.9 % below 8.7, maximum 36 (a wall seen edge-on: 128 texels over
# 4 pixels, condition sum 1.4e5, 0.025 texels off).  Well-conditioned pixels get a far smaller margin than before.
ROUNDING_K = 64.0
ROW_DIVISION_MARGIN = 2.0 ** -17   # COLORMAP rows: see fragment_exact
CLASSES = ('edge', 'alpha texel boundary', 'depth tie', 'texel boundary', 'colormap-row boundary', 'other')


def _mod(x, y):
    return x - y * math.floor(x / y)


class F64Frame:
    """float64 evaluation of the reference's shaders for single (primitive, sample position) pairs of one frame."""

    def __init__(self, lvl, modelview, projection, time, lights, width, height, object_modelviews=None):
        g = (lambda k, d=None: lvl.get(k, d)) if isinstance(lvl, dict) else (lambda k, d=None: getattr(lvl, k, d))
        self.sv, self.si = g('static_vertices'), np.asarray(g('static_indices'))
        self.kv, self.ki = np.asarray(g('sky_vertices'), np.float64).reshape(-1, 3), np.asarray(g('sky_indices'))
        self.dv, self.di = g('decor_vertices'), np.asarray(g('decor_indices'))
        self.draws = np.asarray(g('draws')).reshape(-1, 4)
        self.atlas = {KIND_FLAT: np.asarray(g('flat_atlas')), KIND_WALL: np.asarray(g('wall_atlas')),
                      KIND_DECOR: np.asarray(g('decor_atlas'))}
        self.sky_tex, self.sky_band = np.asarray(g('sky_texture')), float(g('sky_band'))
        self.cmap = np.asarray(g('colormap')).reshape(32, 256)
        self.P = np.asarray(projection, np.float64).reshape(4, 4).T
        self.M0 = np.asarray(modelview, np.float64).reshape(4, 4).T
        self.OM = None if object_modelviews is None else np.asarray(object_modelviews, np.float64).reshape(-1, 4, 4)
        self.time, self.lights, self.w, self.h = float(time), np.asarray(lights), width, height
        self.first = np.cumsum([0] + [int(c) // 3 for c in self.draws[:, 3]])
        self._cache = {}

    def _tri(self, pid):
        if pid in self._cache:
            return self._cache[pid]
        d = int(np.searchsorted(self.first, pid, side='right') - 1)
        kind, obj, first, _count = (int(x) for x in self.draws[d])
        t = pid - int(self.first[d])
        M = self.M0 if self.OM is None else self.OM[obj].T
        PM = self.P @ M
        s = {'kind': kind}
        if kind == KIND_SKY:
            pos = self.kv[self.ki[first + 3 * t:first + 3 * t + 3]]
            clip = np.concatenate([pos, np.ones((3, 1))], 1) @ PM.T
            u = v = np.zeros(3)
            s['vr'] = (math.atan2(PM[0, 2], PM[2, 2]), PM[1, 2] / PM[3, 2])
        elif kind == KIND_DECOR:
            vs = self.dv[self.di[first + 3 * t:first + 3 * t + 3]]
            right = M[0, :3]
            pos = vs['a_pos'].astype(np.float64) + np.outer(vs['a_local_x'].astype(np.float64), right)
            clip = (np.concatenate([pos, np.ones((3, 1))], 1) @ M.T) @ self.P.T
            u, v = vs['a_tile_uv'][:, 0].astype(np.float64), vs['a_tile_uv'][:, 1].astype(np.float64)
            pv = vs[2]
        else:
            vs = self.sv[self.si[first + 3 * t:first + 3 * t + 3]]
            clip = np.concatenate([vs['a_pos'].astype(np.float64), np.ones((3, 1))], 1) @ PM.T
            u = vs['a_tile_uv'][:, 0].astype(np.float64) + self.time * vs['a_scroll_rate'].astype(np.float64)
            v = vs['a_tile_uv'][:, 1].astype(np.float64)
            pv = vs[2]
        if kind != KIND_SKY:
            aw = float(self.atlas[kind].shape[1])
            sx, sy = float(pv['a_tile_size'][0]), float(pv['a_tile_size'][1])
            au, av = float(pv['a_atlas_uv'][0]), float(pv['a_atlas_uv'][1])
            nf = int(pv['a_num_frames'])
            if nf != 1:   # static.vert:27-39 / sprite.vert:25-37
                fi = math.floor(_mod(self.time / (8.0 / 35.0), float(nf)))
                atlas_u = au + fi * sx
                rows = math.ceil((atlas_u + sx) / aw) - 1.0
                atlas_u += _mod(aw - au, sx) * rows
                step = sy if kind == KIND_DECOR else float(pv['a_row_height'])
                au, av = atlas_u, av + rows * step
            s.update(size=(sx, sy), atlas_uv=(au, av), light=float(self.lights[int(pv['a_light'])]) / 255.0)
        w = clip[:, 3]
        xw, yw = (clip[:, 0] + w) * (self.w / 2), (clip[:, 1] + w) * (self.h / 2)
        e = np.zeros((3, 3))
        for i in range(3):
            j, k = (i + 1) % 3, (i + 2) % 3
            e[i] = (yw[j] * w[k] - yw[k] * w[j], xw[k] * w[j] - xw[j] * w[k], xw[j] * yw[k] - xw[k] * yw[j])
        det = w[0] * e[0, 2] + yw[0] * e[0, 1] + xw[0] * e[0, 0]
        # thinner than the vertex snap: whether it faces the viewer, and which pixel centres it covers, is decided by
        # rounding (nearly collinear fan triangles of the sub-sector polygons)
        s['sliver'] = False
        if (w > 0).all():
            px, py = xw / w, yw / w
            longest = max(math.hypot(px[i] - px[j], py[i] - py[j]) for i, j in ((0, 1), (1, 2), (0, 2)))
            s['sliver'] = longest > 0 and abs(det / (w[0] * w[1] * w[2])) / longest <= JITTER
        s['culled'] = bool((w <= 0).all() or not det > 0)
        s['wmin'] = float(w.min())
        s['e'] = e
        s['norm'] = np.hypot(e[:, 0], e[:, 1])
        if not s['culled']:
            s['zp'], s['wp'] = (clip[:, 2] @ e) / det, (np.ones(3) @ e) / det
            s['up'], s['vp'] = (u @ e) / det, (v @ e) / det
            # the near plane z = -w is an edge of the clipped primitive as well: zw = 0
        self._cache[pid] = s
        return s

    def fragment(self, pid, tu, tv, dist, binary32=False):
        """static.frag:19-26 / sprite.frag:19-26 in float64 on given varyings (v_tile_uv, v_dist).  binary32: the texel
        coordinate `mod(v_tile_uv, u_tile_size) + u_atlas_uv` rounded the way the shader's highp floats round it -- just
        below a tile's far edge the sum rounds UP onto the first texel of the neighbouring atlas entry."""
        s = self._tri(pid)
        if binary32:
            f = np.float32
            with np.errstate(all='ignore'):
                m = [f(t) - f(z) * np.floor(f(t) / f(z)) for t, z in ((tu, s['size'][0]), (tv, s['size'][1]))]
                uvx, uvy = float(f(m[0]) + f(s['atlas_uv'][0])), float(f(m[1]) + f(s['atlas_uv'][1]))
        else:
            uvx, uvy = _mod(tu, s['size'][0]) + s['atlas_uv'][0], _mod(tv, s['size'][1]) + s['atlas_uv'][1]
        a = self.atlas[s['kind']]
        ix, iy = int(math.floor(uvx)) & (a.shape[1] - 1), int(math.floor(uvy)) & (a.shape[0] - 1)
        texel = int(a[iy, ix])
        if s['kind'] == KIND_DECOR:
            light = min(s['light'], s['light'] * 2.0 - min(1.0, 1.0 - 1.0 / (dist + 1.0)))
        else:
            light = s['light'] * 2.0 - min(1.0, 1.0 - 0.9 / (dist + 0.9))
        t = (1.0 - light) * 32.0
        row = int(min(max(math.floor(t), 0), 31))
        return {'texel': (ix, iy), 'row': row, 'colour': int(self.cmap[row, texel & 255]), 'tuv': (tu, tv), 'dist': dist,
                'opaque': s['kind'] == KIND_FLAT or not texel & 0x8000,
                # distances to the nearest discontinuity: texel units / COLORMAP-row units
                'texel_margin': min(uvx - math.floor(uvx), math.ceil(uvx) - uvx, uvy - math.floor(uvy), math.ceil(uvy) - uvy),
                'row_margin': min(t - math.floor(t), math.ceil(t) - t) if 0.0 < t < 32.0 else 1.0}

    def fragment_sky(self, uvx, uvy):
        """sky.frag:24-25 on the folded uv: REPEAT / NEAREST fetch, palette row 0."""
        sh, sw = self.sky_tex.shape
        fx, fy = (uvx - math.floor(uvx)) * sw, (uvy - math.floor(uvy)) * sh
        ix, iy = min(int(math.floor(fx)), sw - 1), min(int(math.floor(fy)), sh - 1)
        return {'texel': (ix, iy), 'row': 0, 'colour': int(self.cmap[0, int(self.sky_tex[iy, ix]) & 255]),
                'tuv': (uvx * sw, uvy * sh), 'dist': 1.0, 'opaque': True, 'row_margin': 1.0,
                'texel_margin': min(fx - math.floor(fx), math.ceil(fx) - fx, fy - math.floor(fy), math.ceil(fy) - fy)}

    def sample(self, pid, x, y):
        """Evaluates primitive `pid` at window position (x, y).  Returns None if the primitive is culled, else a dict:
        margin (pixels to the nearest edge incl. the near / far planes, negative outside), z (window depth), and for
        covered samples the fields of fragment() / fragment_sky()."""
        s = self._tri(pid)
        if s['culled']:
            return None
        ev = lambda p: p[0] * x + p[1] * y + p[2]  # noqa: E731
        vals = np.array([ev(s['e'][i]) for i in range(3)])
        with np.errstate(divide='ignore', invalid='ignore'):
            margin = float(np.min(np.where(s['norm'] > 0, vals / np.where(s['norm'] > 0, s['norm'], 1), np.inf)))
        z = 0.5 * ev(s['zp']) + 0.5
        rw = ev(s['wp'])
        gz = 0.5 * math.hypot(s['zp'][0], s['zp'][1])
        if gz > 0:   # distance to the near / far clip planes in pixels
            margin = min(margin, z / gz, (1.0 - z) / gz)
        out = {'margin': margin, 'z': z, 'kind': s['kind'], 'opaque': True}
        if not rw > 0:
            out['margin'] = min(margin, -1.0)
            return out
        if s['kind'] == KIND_SKY:   # sky.frag:13-23
            ndc_x, ndc_y = x / (self.w / 2) - 1.0, y / (self.h / 2) - 1.0
            uvx = ndc_x - 4.0 * s['vr'][0] / 3.14159265358
            uvy = -ndc_y + 1.0 + s['vr'][1]
            b = self.sky_band
            if (uvy < 0.0 or uvy >= 2.0) and not b > 0:
                return out
            if uvy < 0.0:
                uvy = abs(_mod(-uvy + b, b * 2.0) - b)
            elif uvy >= 2.0:
                uvy = abs(_mod(uvy - 2.0 + b, b * 2.0) - b)
            elif uvy >= 1.0:
                uvy = 1.0 - uvy
            out.update(self.fragment_sky(uvx, uvy))
            return out
        wq = 1.0 / rw
        out.update(self.fragment(pid, ev(s['up']) * wq, ev(s['vp']) * wq, wq))
        ab = lambda p: abs(p[0] * x) + abs(p[1] * y) + abs(p[2])  # noqa: E731
        out['uv_cond'] = max(ab(s['up']) + abs(out['tuv'][0]) * ab(s['wp']), ab(s['vp']) + abs(out['tuv'][1]) * ab(s['wp'])) * wq
        return out


def _grid(n):
    """(2n+1)^2 sample displacements covering [-JITTER, JITTER]^2, centre first"""
    pts = [(dx * JITTER / n, dy * JITTER / n) for dy in range(-n, n + 1) for dx in range(-n, n + 1)]
    return sorted(pts, key=lambda o: max(abs(o[0]), abs(o[1])))


_OFFSETS = _grid(2)


def _offsets_for(f64, pid, x, y):
    """Minified textures put many texels under the jitter square: sample it densely enough to meet each of them."""
    tuv = [s['tuv'] for s in (f64.sample(pid, x + dx, y + dy) for dx in (-JITTER, JITTER) for dy in (-JITTER, JITTER))
           if s is not None and 'tuv' in s]
    if len(tuv) < 2:
        return _OFFSETS
    span = max(max(t[i] for t in tuv) - min(t[i] for t in tuv) for i in (0, 1))
    return _grid(int(min(32, max(2, math.ceil(2.0 * span)))))


def _edges_within_jitter(f64, pid, x, y):
    """How many of the primitive's edge lines pass within JITTER of (x, y): two or more = the triangle is thinner than
    the vertex snap HERE (nearly collinear fan triangles; also ones float64 culls as back-facing by a hair), so which
    pixel centres it covers, if any, is decided by rounding."""
    s = f64._tri(pid)
    n = 0
    for i in range(3):
        if s['norm'][i] > 0 and abs((s['e'][i][0] * x + s['e'][i][1] * y + s['e'][i][2]) / s['norm'][i]) <= JITTER:
            n += 1
    return n


def _tile_edge_opacities(f64, pid, around):
    """Opacity of the texels a fragment of `pid` can fetch where v_tile_uv crosses a multiple of the tile size inside the
    jitter square: just below the multiple, the shader's binary32 sum `mod(v_tile_uv, size) + atlas_uv` rounds UP to
    atlas_uv + size, the first texel of the neighbouring atlas entry (the zone is a fraction of a binary32 step wide -- no
    sample grid meets it, so the neighbouring texel is looked up directly)."""
    pts = [s for s in around if s is not None and 'tuv' in s and s['kind'] != KIND_SKY]
    out = set()
    if not pts:
        return out
    tri = f64._tri(pid)
    if tri['kind'] == KIND_FLAT:
        return out   # flats are never alpha-tested
    size, atlas_uv, a = tri['size'], tri['atlas_uv'], f64.atlas[tri['kind']]
    for axis in (0, 1):
        lo, hi = min(s['tuv'][axis] for s in pts), max(s['tuv'][axis] for s in pts)
        if math.ceil(lo / size[axis]) * size[axis] <= hi:   # a tile boundary inside the square
            for s in pts:
                uv = [_mod(s['tuv'][0], size[0]) + atlas_uv[0], _mod(s['tuv'][1], size[1]) + atlas_uv[1]]
                uv[axis] = atlas_uv[axis] + size[axis]
                texel = int(a[int(math.floor(uv[1])) & (a.shape[0] - 1), int(math.floor(uv[0])) & (a.shape[1] - 1)])
                out.add(not texel & 0x8000)
    return out


def _which_boundary(centre, other):
    if other['texel'] != centre['texel']:
        return 'texel boundary'
    if other['row'] != centre['row']:
        return 'colormap-row boundary'
    # float64 agrees with GL at the centre: the binary32 oracle rounded across a boundary float64 resolves the other way
    if centre['texel_margin'] <= ROUNDING_K * 2.0 ** -23 * centre.get('uv_cond', 0.0):
        return 'texel boundary'
    if centre['row_margin'] <= ROUNDING_MARGIN * 32.0:
        return 'colormap-row boundary'
    return None


def classify_pixel(f64, ix, iy, oracle_prim, gl_prim, gl_rgb, gl_var, rgb_of_index, stats=None):
    """The label of one mismatching pixel (module docstring).  gl_var: the varyings SwiftShader interpolated at this
    pixel (auxiliary pass), or None.  rgb_of_index: (256, 3) PLAYPAL."""
    x, y = ix + 0.5, iy + 0.5
    none = (0xFFFFFFFF, 0xFFFFFF)
    want = tuple(int(c) for c in gl_rgb)
    if (oracle_prim & 0xFFFFFF) != (gl_prim & 0xFFFFFF):
        if any(f64._tri(int(p))['sliver'] or _edges_within_jitter(f64, int(p), x, y) >= 2
               for p in (oracle_prim, gl_prim) if p not in none):
            return 'edge'
        samples = [f64.sample(int(p), x, y) for p in (oracle_prim, gl_prim) if p not in none]
        if any(s is None for s in samples):
            return 'other'
        if any(abs(s['margin']) <= JITTER for s in samples):
            return 'edge'
        for p in (oracle_prim, gl_prim):
            if p in none:
                continue
            around = [f64.sample(int(p), x + dx, y + dy) for dx, dy in _OFFSETS]
            opacities = {bool(s['opaque']) for s in around if s is not None}
            # ... and with the texel coordinate rounded as the shader's binary32 sum rounds it: within 2^-16 of a tile's far
            # edge `mod(v_tile_uv, size) + atlas_uv` lands on the neighbouring atlas entry, whose texel may be transparent
            opacities |= _tile_edge_opacities(f64, int(p), around)
            if len(opacities) > 1:
                return 'alpha texel boundary'
        if len(samples) == 2:
            # depths closer than the depth buffer resolves, at the centre or anywhere a 1/16-pixel vertex snap can move
            # the sample to: on surfaces seen at a grazing angle the depth changes by many buffer steps per pixel, so the
            # snap alone shifts it by several (1920x1080 frames: up to 19 steps measured between near-coplanar walls)
            spans = []
            for p in (oracle_prim, gl_prim):
                zs = [s['z'] for s in (f64.sample(int(p), x + dx, y + dy) for dx, dy in _OFFSETS) if s is not None]
                spans.append((min(zs), max(zs)))
            if spans[0][0] - DEPTH_TIE <= spans[1][1] and spans[1][0] - DEPTH_TIE <= spans[0][1]:
                return 'depth tie'
        return 'other'
    if oracle_prim in none:
        return 'other'
    pid = int(oracle_prim)
    centre = f64.sample(pid, x, y)
    if centre is None or 'texel' not in centre:
        return 'other'
    # 1. the reference's fragment arithmetic on the varyings GL itself interpolated at this pixel
    if gl_var is not None:
        gv = [float(v) for v in gl_var]
        if centre['kind'] == KIND_SKY:
            g = f64.fragment_sky(gv[0], gv[1]) if gv[2] == -1.0 else None
        else:
            g = f64.fragment(pid, gv[0], gv[1], gv[2]) if gv[2] > 0.0 else None
        if g is not None and centre['kind'] != KIND_SKY and tuple(rgb_of_index[g['colour']]) != want:
            g32 = f64.fragment(pid, gv[0], gv[1], gv[2], binary32=True)
            if tuple(rgb_of_index[g32['colour']]) == want:
                g = g32
        if g is not None and tuple(rgb_of_index[g['colour']]) == want:
            duv = max(abs(g['tuv'][0] - centre['tuv'][0]), abs(g['tuv'][1] - centre['tuv'][1]))
            scale = max(abs(centre['tuv'][0]), abs(centre['tuv'][1]))
            ddist = abs(g['dist'] - centre['dist']) / centre['dist']
            if duv <= VARYING_TOL_ABS + VARYING_TOL_REL * scale and ddist <= VARYING_TOL_REL:
                label = _which_boundary(centre, g)
                if label is not None:
                    if stats is not None:
                        stats['max_varying_uv_deviation_texels'] = max(stats.get('max_varying_uv_deviation_texels', 0.0), duv)
                        stats['max_varying_uv_deviation_rel'] = max(stats.get('max_varying_uv_deviation_rel', 0.0), duv / max(scale, 1.0))
                        stats['max_varying_dist_deviation_rel'] = max(stats.get('max_varying_dist_deviation_rel', 0.0), ddist)
                    return label
    # 2. sample positions displaced by up to JITTER: a 1/16-pixel vertex snap
    label = None
    for dx, dy in _offsets_for(f64, pid, x, y):
        s = f64.sample(pid, x + dx, y + dy)
        if s is None or 'texel' not in s:
            continue
        if tuple(rgb_of_index[s['colour']]) == want:
            this = _which_boundary(centre, s)
            if this is None:
                continue
            label = this
            if this == 'texel boundary':
                break
    return label or 'other'


def census(lvl, modelview, projection, time, lights, width, height, ours_index, ours_prim, gl_rgb, gl_prim,
           gl_varyings=None, object_modelviews=None, detail=False):
    """Counts and labels the pixels where the oracle / HIP frame (palette indices + winning primitive ids) and the GL
    readback (RGB; primitive ids and interpolated varyings of the auxiliary passes) differ.
    Returns {'pixels', 'mismatch', 'winner_mismatch', classes..., deviation statistics}."""
    g = (lambda k: lvl[k]) if isinstance(lvl, dict) else (lambda k: getattr(lvl, k))
    playpal = np.asarray(g('palette'), np.uint8).reshape(256, 3)
    from gl_readback import CLEAR_RGB
    ours_rgb = playpal[ours_index]
    ours_rgb[ours_prim == 0xFFFFFFFF] = CLEAR_RGB
    mis = (ours_rgb != gl_rgb).any(-1)
    out = {'pixels': int(width * height), 'mismatch': int(mis.sum()),
           'winner_mismatch': int(((ours_prim & 0xFFFFFF) != (gl_prim & 0xFFFFFF)).sum())}
    out.update({c: 0 for c in CLASSES})
    f64 = F64Frame(lvl, modelview, projection, time, lights, width, height, object_modelviews)
    where = []
    for iy, ix in zip(*np.nonzero(mis)):
        label = classify_pixel(f64, int(ix), int(iy), int(ours_prim[iy, ix]), int(gl_prim[iy, ix]), gl_rgb[iy, ix],
                               None if gl_varyings is None else gl_varyings[iy, ix], playpal, out)
        out[label] += 1
        if detail:
            where.append((int(ix), int(iy), label))
    if detail:
        out['where'] = where
    return out


def fragment_exact(oracle, lvl, time, lights, gl_rgb, gl_prim, gl_var):
    """ZERO-TOLERANCE pin of the fragment stage.  For every pixel SwiftShader drew: the oracle's binary32 fragment code
    (oracle/raster_oracle.c: oracle_shade_varyings = static.frag:18-28, sprite.frag:15-27, sky.frag:24-25 as restated
    there) applied to the varyings SwiftShader ITSELF interpolated at that pixel, for the primitive that won THERE, must
    give the colour SwiftShader wrote -- no jitter, no envelope, no classifier.  Returns the counts:
      pixels    drawn pixels compared
      disagree  pixels whose colour differs (RGB through PLAYPAL 0, as GL writes it), or that the alpha test would have
                discarded although GL drew them
      by_kind   the disagreements per primitive kind
    `oracle`: oracle.raster.RasterOracle of the level; gl_prim / gl_var: the auxiliary passes of tests/gl_readback.py."""
    g = (lambda k, d=None: lvl.get(k, d)) if isinstance(lvl, dict) else (lambda k, d=None: getattr(lvl, k, d))
    playpal = np.asarray(g('palette'), np.uint8).reshape(-1, 3)[:256]
    out = oracle.shade_varyings(time, lights, gl_prim, gl_var)
    drawn = out != 0xFFFF
    kept = drawn & (out < 0x100)
    rgb = np.zeros(gl_rgb.shape, np.uint8)
    rgb[kept] = playpal[out[kept]]
    bad = drawn & (~kept | (rgb != gl_rgb).any(axis=-1))
    draws = np.asarray(g('draws')).reshape(-1, 4)
    first = np.cumsum([0] + [int(c) // 3 for c in draws[:, 3]])
    kinds = draws[np.clip(np.searchsorted(first, gl_prim[bad], side='right') - 1, 0, len(draws) - 1), 0]
    by_kind = {name: int((kinds == k).sum()) for k, name in ((KIND_FLAT, 'flat'), (KIND_WALL, 'wall'), (KIND_DECOR, 'decor'), (KIND_SKY, 'sky'))}
    # Sky disagreements: sky.frag hands `uv` straight to a REPEAT / NEAREST sampler, whose texel ADDRESS arithmetic GL leaves
    # to the implementation (SwiftShader quantises the normalised coordinate to 16 fractional bits: 1/256 of a texel of the
    # 256-texel-wide sky).  Such a pixel counts as explained only if the coordinate lies within 1/64 texel of a texel
    # boundary AND the neighbouring texel across that boundary gives exactly the colour GL wrote.
    sky_explained = 0
    if by_kind['sky']:
        sky = np.asarray(g('sky_texture'))
        sh, sw = sky.shape
        cmap0 = np.asarray(g('colormap')).reshape(32, 256)[0]
        all_kinds = draws[np.clip(np.searchsorted(first, gl_prim, side='right') - 1, 0, len(draws) - 1), 0]
        for y, x in zip(*np.nonzero(bad & (all_kinds == KIND_SKY) & (gl_prim != 0xFFFFFFFF))):
            u, v = float(gl_var[y, x, 0]), float(gl_var[y, x, 1])
            fx, fy = (u - math.floor(u)) * sw, (v - math.floor(v)) * sh
            near = []
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    if (dx and abs(fx - round(fx)) > 1.0 / 64) or (dy and abs(fy - round(fy)) > 1.0 / 64):
                        continue
                    ix, iy = int(math.floor(fx + dx * 1.0 / 32)) % sw, int(math.floor(fy + dy * 1.0 / 32)) % sh
                    near.append(tuple(int(c) for c in playpal[cmap0[int(sky[iy, ix]) & 255]]))
            sky_explained += tuple(int(c) for c in gl_rgb[y, x]) in near
    # Flat / wall / decor disagreements (none with SwiftShader; a handful per hundred million pixels with Mesa's llvmpipe): GLSL does
    # not ask for a correctly rounded quotient -- Mesa evaluates `DIST_SCALE / (v_dist + DIST_SCALE)` as DIST_SCALE * (1 / x), an
    # ulp away from the IEEE quotient in a quarter of all cases -- and the palette's NEAREST fetch turns `1.0 - light` into a row
    # by the sampler's own floor.  Such a pixel counts as explained only if (1 - light) * 32, evaluated in binary32 as the
    # shader text reads, lies within ROW_DIVISION_MARGIN (2^-17 of a row: three ulps of a value below 32) of a row boundary
    # AND the row across that boundary, with the SAME texel, gives exactly the colour GL wrote.
    row_explained = 0
    if by_kind['flat'] + by_kind['wall'] + by_kind['decor']:
        f64 = F64Frame(lvl, np.eye(4, dtype=np.float32).reshape(16), np.eye(4, dtype=np.float32).reshape(16), time, lights, 16, 16)
        cmap = np.asarray(g('colormap')).reshape(32, 256)
        all_kinds = draws[np.clip(np.searchsorted(first, gl_prim, side='right') - 1, 0, len(draws) - 1), 0]
        for y, x in zip(*np.nonzero(bad & (all_kinds != KIND_SKY) & (gl_prim != 0xFFFFFFFF))):
            fr = f64.fragment(int(gl_prim[y, x]), float(gl_var[y, x, 0]), float(gl_var[y, x, 1]), float(gl_var[y, x, 2]), binary32=True)
            if not fr['opaque'] or fr['row_margin'] > ROW_DIVISION_MARGIN:
                continue
            a = f64.atlas[f64._tri(int(gl_prim[y, x]))['kind']]
            texel = int(a[fr['texel'][1], fr['texel'][0]]) & 255
            # (fr['row'] is the float64 evaluation's row: the oracle's binary32 one -- which disagreed -- is it or a neighbour)
            near = [tuple(int(c) for c in playpal[cmap[r, texel]]) for r in (fr['row'] - 1, fr['row'], fr['row'] + 1) if 0 <= r < 32]
            row_explained += tuple(int(c) for c in gl_rgb[y, x]) in near
    return {'pixels': int(drawn.sum()), 'disagree': int(bad.sum()), 'by_kind': by_kind, 'sky_sampler_boundary': int(sky_explained),
            'row_division_boundary': int(row_explained), 'mask': bad}
