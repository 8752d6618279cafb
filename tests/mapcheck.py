"""An INDEPENDENT view of a Doom level for the CPU-half invariants (tests/test_cpu_half_invariants.py).

Nothing here is shared with the two implementations under test (oracle/wad_oracle.py, csrc/host/*.cpp) and nothing here
restates wad/src/visitor.rs: the level lumps are decoded with a dozen lines of `struct` from the published Doom formats
(VERTEXES 4 B, LINEDEFS 14 B, SIDEDEFS 30 B, SECTORS 26 B -- the same layouts wad/src/types.rs:33-150 declares), and
every quantity the tests compare against is derived from the MAP (linedefs / sidedefs / sectors), never from the BSP
lumps (SEGS / SSECTORS / NODES) the code under test walks:

  * sector area        shoelace over the directed boundary edges of the sector (front side = right of v1 -> v2);
  * sector of a point  ray casting: the nearest linedef crossed by a ray from the point decides, through the side
                       that faces the point (all arithmetic float64);
  * convex clipping    Sutherland-Hodgman, for pairwise overlap areas.

World coordinates are the reference's (wad/src/util.rs:12-22): x = -wad_y / 100, z = -wad_x / 100, y = height / 100.
"""
import struct

import numpy as np

POLY_BIAS = 0.64 * 3e-4  # wad/src/visitor.rs: POLY_BIAS = 0.64 * 3e-4 world units (= 0.0192 map units)


def read_directory(path):
    data = open(path, 'rb').read()
    magic, n, off = struct.unpack_from('<4sii', data, 0)
    assert magic in (b'IWAD', b'PWAD')
    lumps = []
    for i in range(n):
        pos, size, name = struct.unpack_from('<ii8s', data, off + 16 * i)
        lumps.append((name.split(b'\0')[0].upper(), pos, size))
    return data, lumps


def level_markers(lumps):
    """lump indices of the level markers: the lump before a THINGS lump"""
    return [i - 1 for i, (name, _, _) in enumerate(lumps) if name == b'THINGS' and i > 0]


def wall_texture_names(path):
    """names of the composed wall textures the IWAD defines (TEXTURE1 / TEXTURE2: count, offsets, then per texture an
    8-byte name followed by its size and patch list)"""
    data, lumps = read_directory(path)
    names = set()
    for name, pos, size in lumps:
        if name in (b'TEXTURE1', b'TEXTURE2') and size >= 4:
            n, = struct.unpack_from('<i', data, pos)
            for off in struct.unpack_from('<%di' % n, data, pos + 4):
                names.add(data[pos + off:pos + off + 8].split(b'\0')[0].upper())
    return names


def flat_lumps(path):
    """name -> the 4096 raw bytes of every flat between F_START and F_END (row-major 64 x 64)"""
    data, lumps = read_directory(path)
    out, inside = {}, False
    for name, pos, size in lumps:
        if name in (b'F_START', b'FF_START'):
            inside = True
        elif name in (b'F_END', b'FF_END'):
            inside = False
        elif inside and size == 4096:
            out[name] = data[pos:pos + 4096]
    return out


class Map:
    """the MAP lumps of one level, decoded independently"""

    def __init__(self, path, level_index):
        data, lumps = read_directory(path)
        m = level_markers(lumps)[level_index]
        self.name = lumps[m][0].decode()
        by_name = {lumps[m + k][0]: lumps[m + k] for k in range(1, 11) if m + k < len(lumps)}

        def rec(name, fmt):
            _, pos, size = by_name[name]
            st = struct.Struct(fmt)
            assert size % st.size == 0
            return [st.unpack_from(data, pos + i * st.size) for i in range(size // st.size)]

        self.vertices = np.array(rec(b'VERTEXES', '<hh'), np.float64).reshape(-1, 2)
        self.linedefs = rec(b'LINEDEFS', '<HHHHHHH')    # v1, v2, flags, special, tag, right (front), left (back)
        self.sidedefs = rec(b'SIDEDEFS', '<hh8s8s8sH')  # x_off, y_off, upper, lower, middle, sector
        self.sectors = rec(b'SECTORS', '<hh8s8sHHH')    # floor, ceil, floor tex, ceil tex, light, type, tag
        self.things = rec(b'THINGS', '<hhHHH')
        self.n_ssectors = by_name[b'SSECTORS'][2] // 4
        self.n_segs = by_name[b'SEGS'][2] // 12

    @staticmethod
    def tex(raw):
        return raw.split(b'\0')[0].upper()

    def side_sector(self, side):
        return None if side == 0xFFFF or side >= len(self.sidedefs) else self.sidedefs[side][5]

    def edges(self):
        """(x1, y1, x2, y2, front sector or -1, back sector or -1) per linedef, map units"""
        out = []
        for v1, v2, _, _, _, right, left in self.linedefs:
            f, b = self.side_sector(right), self.side_sector(left)
            out.append((self.vertices[v1][0], self.vertices[v1][1], self.vertices[v2][0], self.vertices[v2][1],
                        -1 if f is None else f, -1 if b is None else b))
        return np.array(out, np.float64)

    def sector_areas(self):
        """|area| of every sector in map units^2: half the sum of the cross products of its directed boundary edges.  The
        front sector lies to the RIGHT of v1 -> v2 (clockwise boundary), the back sector to the left; an edge with the
        same sector on both sides cancels."""
        area = np.zeros(len(self.sectors))
        for x1, y1, x2, y2, f, b in self.edges():
            c = x1 * y2 - x2 * y1
            if f >= 0:
                area[int(f)] -= c
            if b >= 0:
                area[int(b)] += c
        return 0.5 * area   # positive for a properly oriented closed sector

    def _sector_by_ray(self, e, pts):
        """one horizontal ray towards +x per point: the nearest linedef crossed decides, through the side facing the point"""
        x1, y1, x2, y2 = e[:, 0], e[:, 1], e[:, 2], e[:, 3]
        px, py = pts[:, 0:1], pts[:, 1:2]
        with np.errstate(divide='ignore', invalid='ignore'):
            # crossing of the ray y = py with the segment (half-open rule so that a vertex is counted once)
            cross = ((y1 <= py) & (py < y2)) | ((y2 <= py) & (py < y1))
            t = (py - y1) / (y2 - y1)
            xc = x1 + t * (x2 - x1)
        ahead = cross & (xc > px)
        dist = np.where(ahead, xc - px, np.inf)
        k = np.argmin(dist, axis=1)
        hit = np.isfinite(dist[np.arange(len(pts)), k])
        # which side of linedef k is the point on?  right of v1 -> v2 = front
        dx, dy = (x2 - x1)[k], (y2 - y1)[k]
        side = dx * (pts[:, 1] - y1[k]) - dy * (pts[:, 0] - x1[k])  # > 0: left
        sec = np.where(side < 0, e[k, 4], e[k, 5]).astype(np.int64)
        return np.where(hit, sec, -1)

    def sector_at(self, pts):
        """sector index of each map point (n, 2), -1 outside the map (or undecided), and the distance to the nearest
        linedef.  Ray casting in the four axis directions (the map rotated by quarter turns); the answer at least two
        rays agree on wins -- hand-made maps are not watertight (a vertex a unit off a T-junction lets one ray slip
        through a gap)."""
        e = self.edges()
        pts = np.asarray(pts, np.float64).reshape(-1, 2)
        votes = []
        for q in range(4):
            er, pr = e.copy(), pts.copy()
            for _ in range(q):  # quarter turn: (x, y) -> (-y, x); orientation (hence front / back) is preserved
                er = np.stack([-er[:, 1], er[:, 0], -er[:, 3], er[:, 2], er[:, 4], er[:, 5]], axis=1)
                pr = np.stack([-pr[:, 1], pr[:, 0]], axis=1)
            votes.append(self._sector_by_ray(er, pr))
        votes = np.stack(votes, axis=1)
        sec = np.full(len(pts), -1, np.int64)
        for i, v in enumerate(votes):
            vals, counts = np.unique(v[v >= 0], return_counts=True)
            if len(vals) and counts.max() >= 2 and (counts == counts.max()).sum() == 1:
                sec[i] = vals[np.argmax(counts)]
        # distance to the nearest linedef (to exclude samples in the bias ring around boundaries)
        x1, y1, x2, y2 = e[:, 0], e[:, 1], e[:, 2], e[:, 3]
        px, py = pts[:, 0:1], pts[:, 1:2]
        ex, ey = x2 - x1, y2 - y1
        ll = ex * ex + ey * ey
        with np.errstate(divide='ignore', invalid='ignore'):
            u = np.clip(((px - x1) * ex + (py - y1) * ey) / np.where(ll > 0, ll, 1.0), 0.0, 1.0)
        d = np.hypot(px - (x1 + u * ex), py - (y1 + u * ey)).min(axis=1)
        return sec, d


def world_to_map(xz):
    """(x, z) world -> (wad_x, wad_y) map units: x = -wad_y / 100, z = -wad_x / 100"""
    xz = np.asarray(xz, np.float64).reshape(-1, 2)
    return np.stack([-xz[:, 1] * 100.0, -xz[:, 0] * 100.0], axis=1)


def poly_area(p):
    """signed shoelace area of an (n, 2) polygon"""
    p = np.asarray(p, np.float64)
    x, y = p[:, 0], p[:, 1]
    return 0.5 * float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))


def poly_perimeter(p):
    p = np.asarray(p, np.float64)
    return float(np.sum(np.hypot(*(np.roll(p, -1, axis=0) - p).T)))


def convexity(p):
    """(min, max) of the turn cross products of consecutive edges, normalised by the edge lengths (sine of the turn)"""
    p = np.asarray(p, np.float64)
    a = np.roll(p, -1, axis=0) - p
    b = np.roll(a, -1, axis=0)
    cr = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    nm = np.hypot(a[:, 0], a[:, 1]) * np.hypot(b[:, 0], b[:, 1])
    s = cr / np.where(nm > 0, nm, 1.0)
    return float(s.min()), float(s.max())


def concavity_depth(p, sign):
    """how far (map units) the worst reflex vertex of polygon p lies inside the chord of its neighbours; 0 for a convex
    polygon of orientation `sign` (+1 counter-clockwise, -1 clockwise)"""
    p = np.asarray(p, np.float64)
    prev, nxt = np.roll(p, 1, axis=0), np.roll(p, -1, axis=0)
    ch = nxt - prev
    ln = np.hypot(ch[:, 0], ch[:, 1])
    # signed distance of the vertex from the chord prev -> next: for a convex counter-clockwise polygon it lies to the right
    d = (ch[:, 0] * (p[:, 1] - prev[:, 1]) - ch[:, 1] * (p[:, 0] - prev[:, 0])) / np.where(ln > 0, ln, 1.0)
    return float(max(0.0, (sign * d).max()))


def thickness(p):
    """smallest width of the point set p over the directions of its point pairs (map units): a sliver's short side"""
    p = np.asarray(p, np.float64)
    best = np.inf
    for i in range(len(p)):
        for j in range(i + 1, len(p)):
            d = p[j] - p[i]
            ln = np.hypot(d[0], d[1])
            if ln > 0:
                w = ((p[:, 0] - p[i, 0]) * d[1] - (p[:, 1] - p[i, 1]) * d[0]) / ln
                best = min(best, float(w.max() - w.min()))
    return best


def clip_convex(subject, clip):
    """Sutherland-Hodgman: subject polygon clipped to the convex polygon `clip` (both counter-clockwise)"""
    out = [tuple(q) for q in subject]
    c = [tuple(q) for q in clip]
    for i in range(len(c)):
        if not out:
            break
        ax, ay = c[i]
        bx, by = c[(i + 1) % len(c)]
        inp, out = out, []

        def side(q):
            return (bx - ax) * (q[1] - ay) - (by - ay) * (q[0] - ax)

        for j in range(len(inp)):
            p, q = inp[j], inp[(j + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp >= 0) != (sq >= 0):
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return np.array(out, np.float64).reshape(-1, 2)


def overlap_area(p, q):
    """area of the intersection of two convex polygons of either winding"""
    p, q = np.asarray(p, np.float64), np.asarray(q, np.float64)
    if poly_area(p) < 0:
        p = p[::-1]
    if poly_area(q) < 0:
        q = q[::-1]
    r = clip_convex(p, q)
    return abs(poly_area(r)) if len(r) >= 3 else 0.0
