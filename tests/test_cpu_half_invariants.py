"""Independent checks of the CPU half's GEOMETRY (LevelAnalysis, LevelWalker::{subsector,
points_to_polygon, seg, wall_quad, flat_poly}, game::level::Builder) -- wad/src/visitor.rs:621-1008, 1184-1259,
game/src/level.rs:513-794.

The oracle (numpy) and the product (C++) agree byte for byte, but they share one author's reading of visitor.rs; no
rustc exists here to run the reference.  These tests do NOT share that reading: what they compare against is derived
from the MAP lumps (LINEDEFS / SIDEDEFS / SECTORS / VERTEXES) by tests/mapcheck.py with textbook geometry, never from the
BSP lumps the walker consumes.  What each would catch is listed in DESIGN section 2.

  (a) sub-sector polygons tile their sector: convex, consistently wound, pairwise disjoint up to the POLY_BIAS ring, and
      per sector their areas sum to the sector's shoelace area;
  (b) wall quads tile their linedef side: the s-ranges of the quads of one side and one texture part cover
      [x_offset, x_offset + length] without gap or overlap; lower / upper / one-sided middle quads span exactly the
      heights the adjoining sectors leave open;
  (c) the Builder's flat TRIANGLES (vertices + indices as handed to the renderer), sampled on a grid of map points: the
      floor / ceiling triangle over a point carries the height, light level and flat of the sector that an independent
      ray cast finds there.
All on the product's C++ path (rdoom_wad_walk / rdoom_wad_build_level through the C ABI), on the nine levels of the
synthetic IWAD, nine more from other generator seeds and the 10 x E1M1 level.

A one-off sweep over many more generator seeds: RDOOM_EXTRA_SEEDS=first:count (three levels per seed).  Random small maps
with slanted walls hit the reference's own tolerances far more often than the committed levels: a split vertex is stored
rounded to integer map units, a short seg between two such vertices points a few degrees off its linedef, and the
reference tests every candidate corner of a leaf against the seg's infinite LINE with SEG_TOLERANCE (visitor.rs:676-688,
1159) -- a legitimate corner a few hundred units away is then rejected (a hole in the floor the reference would show as
well), or a notch shallower than the tolerance is paved over.  Every exception is still checked one by one (within the
tolerance of its sector's seg lines; slivers only), the RARITY bounds are wider for swept levels (_swept); the committed
levels keep the sharp ones (no hole, no dented polygon, every polygon placed).  DESIGN section 2 has the sweep's result."""
import os

import numpy as np
import pytest

import mapcheck as mc
import rust_doom_amd as rd
from test_other_seeds import SEEDS, _wad
from test_visitor_abi import Recorder
from util import META_PATH, ensure_big_wad

BIAS_MAP = mc.POLY_BIAS * 100.0  # POLY_BIAS in map units
SEG_TOLERANCE_MAP = 0.1 * 100.0   # visitor.rs:1159 SEG_TOLERANCE, world units -> map units


def _extra_seeds():
    """RDOOM_EXTRA_SEEDS=first:count widens the run to more generator seeds (a one-off sweep, not the committed suite)"""
    spec = os.environ.get('RDOOM_EXTRA_SEEDS', '')
    if not spec:
        return ()
    first, count = (int(x) for x in spec.split(':'))
    return tuple(range(first, first + count))


def _swept(which):
    """a level of the one-off sweep: the rarity bounds are wider there (random small maps with slanted walls hit the
    reference's own tolerances more often than the committed levels do; every single exception is still checked)"""
    if which == 'user':   # tests/test_user_iwad.py: a file the committed bounds were not tuned on
        return True
    return which.startswith('seed') and int(which[4:]) in _extra_seeds()


def _cases():
    out = [('synth', i) for i in range(9)]
    out += [('seed%d' % s, i) for s in SEEDS + _extra_seeds() for i in range(3)]
    out += [('big', 0)]
    return out


@pytest.fixture(scope='module')
def wads(wad_path, tmp_path_factory):
    paths = {'synth': wad_path, 'big': ensure_big_wad()}
    for s in SEEDS + _extra_seeds():
        paths['seed%d' % s] = _wad(tmp_path_factory, s)
    return paths


_cache = {}


def _path(wads, which):
    return wads[which][0] if isinstance(wads[which], tuple) else wads[which]


def _load(wads, which, index):
    key = (which, index, _path(wads, which))
    if key not in _cache:
        path, meta = wads[which] if isinstance(wads[which], tuple) else (wads[which], META_PATH)   # (tests/test_user_iwad.py brings its own metadata)
        rec = Recorder()
        wad = rd.Wad(path, meta)
        wad.walk(index, rec)
        built = wad.build_level(index)  # CPU-only path (use_gpu_tessellation = 0)
        arrays = built.arrays()
        arrays['lights0'] = built.lights_at(0.0)
        _cache[key] = (mc.Map(path, index), rec.events, arrays)
    return _cache[key]


def _leaves(events):
    """events grouped per BSP leaf: list of lists"""
    out, cur = [], None
    for e in events:
        if e[0] == 'leaf':
            cur = []
        elif e[0] == 'leaf_end':
            out.append(cur)
            cur = None
        elif cur is not None:
            cur.append(e)
    return out


def _poly_map(e):
    """polygon of a floor / ceil / *_sky event in map units, (n, 2)"""
    return mc.world_to_map(np.array(e[2], np.float64).reshape(-1, 2))


def _possibly_dynamic(m):
    """sectors a door / lift / floor special could move -- decided from the map alone, generously: a sector whose tag a
    linedef with a special refers to, and the back sector of any linedef with a special (manual doors act on it)"""
    tags = {tag for _, _, _, special, tag, _, _ in m.linedefs if special != 0 and tag != 0}
    dyn = {i for i, s in enumerate(m.sectors) if s[6] != 0 and s[6] in tags}
    for _, _, _, special, _, right, left in m.linedefs:
        if special != 0:
            sec = m.side_sector(left)
            if sec is not None:
                dyn.add(sec)
    return dyn


@pytest.mark.parametrize('which,index', _cases())
def test_subsector_polygons_tile_their_sectors(wads, which, index):
    m, events, _ = _load(wads, which, index)
    leaves = _leaves(events)
    assert len(leaves) == m.n_ssectors
    floors, ceils = [], []
    for lv in leaves:
        f = [e for e in lv if e[0] in ('floor', 'floor_sky')]
        c = [e for e in lv if e[0] in ('ceil', 'ceil_sky')]
        assert len(f) <= 1 and len(c) <= 1 and len(f) == len(c)
        if f:
            floors.append(f[0])
            ceils.append(c[0])
            # the ceiling is the floor's polygon (Builder reverses it for the winding, level.rs:721)
            assert f[0][2] == c[0][2]
    # the reference skips a sub-sector whose polygon degenerates (visitor.rs:622-643); it must stay the exception
    assert len(floors) >= 0.97 * m.n_ssectors
    polys = [_poly_map(e) for e in floors]
    areas = np.array([mc.poly_area(p) for p in polys])
    # consistently wound: every polygon has the same orientation, and is convex up to rounding
    assert (areas > 0).all() or (areas < 0).all()
    sign = 1.0 if areas[0] > 0 else -1.0
    dented = 0
    for p in polys:
        # convex -- up to the map's own integer vertices: a seg endpoint that sits ON a neighbour's edge (a T-junction) is
        # rounded to integer map units, up to half a unit off that edge, and every seg endpoint is a polygon point.
        # Beyond that only what the reference's own tolerance explains: a point up to SEG_TOLERANCE on the wrong side of a
        # seg's line is accepted (visitor.rs:683), so a notch shallower than that dents the polygon, and a sub-sector
        # thinner than that may collect a point in its interior, which the angular sort keeps when it comes first or last
        # (visitor.rs:1226-1245 drops reflex points between the ends only).  Both must stay the exception.
        assert len(p) >= 3
        depth = mc.concavity_depth(p, sign)
        if depth > 1.0:
            dented += 1
            assert depth <= SEG_TOLERANCE_MAP + 2 * BIAS_MAP or mc.thickness(p) <= SEG_TOLERANCE_MAP + 2 * BIAS_MAP, (which, index, p)
    assert dented <= (max(2, 0.04 * len(polys)) if _swept(which) else 0), (which, index, dented)
    # which sector is each polygon in?  asked of the map (ray casts) at the polygon's centroid and at its vertices pulled
    # 5 % towards the centroid.  A polygon may PROTRUDE from its sector: the reference accepts an implicit point up to
    # SEG_TOLERANCE = 0.1 WORLD units = 10 map units beyond a seg's line (visitor.rs:683, 1159), so a notch shallower than
    # that is filled in (E1M1 of the synthetic IWAD has two).  Such polygons are attributed by the majority of their
    # sample points, must stay within that tolerance of their sector, and must remain rare.
    sec = np.full(len(polys), -1, np.int64)
    protrudes = np.zeros(len(polys), bool)
    for i, p in enumerate(polys):
        c = p.mean(axis=0)
        samples = np.concatenate([[c], c + 0.95 * (p - c)])
        ss, _ = m.sector_at(samples)
        vals, counts = np.unique(ss[ss >= 0], return_counts=True)
        if not len(vals):
            # every sample on a linedef (the rays do not decide): only a sliver along a wall can do that; it is left out of
            # the sums below and its area added to what a sector may be short of
            assert mc.thickness(p) <= SEG_TOLERANCE_MAP + 2 * BIAS_MAP, (which, index, i, p)
            continue
        sec[i] = vals[np.argmax(counts)]
        protrudes[i] = bool((ss != sec[i]).any())
        if protrudes[i]:
            e_ = m.edges()
            mine = (e_[:, 4] == sec[i]) | (e_[:, 5] == sec[i])
            x1, y1, x2, y2 = e_[mine, 0], e_[mine, 1], e_[mine, 2], e_[mine, 3]
            for q in samples[ss != sec[i]]:
                # (the half-plane test is against the seg's infinite LINE: a sliver sub-sector may run along it)
                ex, ey = x2 - x1, y2 - y1
                dline = np.abs(ex * (q[1] - y1) - ey * (q[0] - x1)) / np.maximum(np.hypot(ex, ey), 1e-12)
                assert dline.min() <= SEG_TOLERANCE_MAP + 2 * BIAS_MAP, (which, index, i, q, dline.min())
    assert protrudes.sum() <= (max(4, 0.15 * len(polys)) if _swept(which) else max(3, 0.08 * len(polys))), (which, index, int(protrudes.sum()))
    placed = sec >= 0
    unplaced_area = float(np.abs(areas[~placed]).sum())
    assert (~placed).sum() <= (max(1, 0.02 * len(polys)) if _swept(which) else 0), (which, index, int((~placed).sum()))
    # height and flat of each polygon are its sector's
    misplaced = 0
    for i, (e, s) in enumerate(zip(floors, sec)):
        if s < 0:
            continue
        if _swept(which) and abs(e[3] * 100.0 - m.sectors[s][0]) >= 1e-3:
            misplaced += 1  # a sliver whose sample points mostly fell into the neighbour: the vote, not the walker
            unplaced_area += abs(areas[i])
            sec[i] = -1
            continue
        assert abs(e[3] * 100.0 - m.sectors[s][0]) < 1e-3, (e[3], m.sectors[s][0])
        if e[0] == 'floor':
            assert e[5].rstrip(b'\0') == mc.Map.tex(m.sectors[s][2])
        else:
            assert mc.Map.tex(m.sectors[s][2]) == b'F_SKY1'
    assert misplaced <= max(1, 0.02 * len(polys)), (which, index, misplaced)
    dyn = _possibly_dynamic(m)
    for e, s in zip(ceils, sec):
        if s < 0:
            continue
        if e[0] == 'ceil':
            if s not in dyn:
                assert abs(e[3] * 100.0 - m.sectors[s][1]) < 1e-3
            assert e[5].rstrip(b'\0') == mc.Map.tex(m.sectors[s][3])
        else:
            assert mc.Map.tex(m.sectors[s][3]) == b'F_SKY1'
    # per sector: the polygons' areas add up to the sector's shoelace area.  NO HOLES is the sharp direction -- the sum may
    # fall short only by the rounding of the map's own vertices; an excess is bounded by the bias ring (each polygon is
    # grown by POLY_BIAS per vertex: at most perimeter x bias) plus, where polygons protrude (above), 3 % of the sector
    want = m.sector_areas()
    got = np.zeros(len(m.sectors))
    ring = np.zeros(len(m.sectors))
    for p, a, s in zip(polys, areas, sec):
        if s < 0:
            continue
        got[s] += abs(a)
        ring[s] += mc.poly_perimeter(p) * BIAS_MAP * 1.5 + 1e-6
    missing = m.n_ssectors - len(floors)
    # The map's own vertices bound the shortfall: where the node builder split a linedef it stored the new vertex rounded
    # to integer map units, up to 0.71 units off the line, and the polygons follow the stored vertex -- along that linedef
    # a sliver of at most length x 0.71 / 2 is nobody's (a 202-unit edge split once: 45 units^2 of an 11 648-unit^2 sector)
    boundary = np.zeros(len(m.sectors))
    for x1, y1, x2, y2, f, b in m.edges():
        for side in {int(f), int(b)} - {-1}:
            boundary[side] += np.hypot(x2 - x1, y2 - y1)
    for s in range(len(m.sectors)):
        if want[s] <= 0 and got[s] == 0:
            continue  # a sector no linedef refers to
        # (sweep: a short seg between two rounded vertices points a few degrees off its linedef, and the half-plane test
        # against its LINE then rejects a legitimate corner of the leaf a few hundred units away, visitor.rs:676-688 -- a
        # hole of a few per cent of the sector that the reference would show as well)
        assert got[s] >= want[s] * (0.94 if _swept(which) else 1.0) - 0.36 * boundary[s] - 1.0 - unplaced_area - (64.0 if missing else 0.0), (which, index, s, got[s], want[s], boundary[s])
        assert got[s] <= want[s] * (1.06 if _swept(which) else 1.03) + ring[s], (which, index, s, got[s], want[s], ring[s])
    assert got.sum() <= want[want > 0].sum() * 1.01
    # polygons do not overlap beyond the bias ring, but for the protrusions: in total at most 1 % of the level's area
    lo = np.array([p.min(axis=0) for p in polys]) - BIAS_MAP
    hi = np.array([p.max(axis=0) for p in polys]) + BIAS_MAP
    order = np.argsort(lo[:, 0])
    excess = 0.0
    for ii, i in enumerate(order):
        for j in order[ii + 1:]:
            if lo[j, 0] > hi[i, 0]:
                break
            if lo[j, 1] > hi[i, 1] or hi[j, 1] < lo[i, 1]:
                continue
            ov = mc.overlap_area(polys[i], polys[j])
            allowed = (mc.poly_perimeter(polys[i]) + mc.poly_perimeter(polys[j])) * BIAS_MAP * 1.5 + 1e-6
            if ov > allowed:
                assert protrudes[i] or protrudes[j] or ov <= allowed + 0.002 * min(abs(areas[i]), abs(areas[j])) + 64.0, (which, index, int(i), int(j), ov, allowed)
                excess += ov - allowed
    assert excess <= 0.01 * np.abs(areas).sum(), (which, index, excess)


@pytest.mark.parametrize('which,index', _cases())
def test_wall_quads_tile_their_linedef_sides(wads, which, index):
    m, events, _ = _load(wads, which, index)
    walls = [e for e in events if e[0] == 'wall']
    assert walls
    e_ = m.edges()
    p1, p2 = e_[:, 0:2], e_[:, 2:4]
    d = p2 - p1
    length = np.hypot(d[:, 0], d[:, 1])
    dyn = _possibly_dynamic(m)
    # attribute every wall quad to a linedef SIDE geometrically: both endpoints on the linedef's segment (within the
    # bias), direction v1 -> v2 along the linedef for the front side, against it for the back side
    groups = {}
    for q in walls:
        a, b = mc.world_to_map([q[2]])[0], mc.world_to_map([q[3]])[0]
        ok = length > 0
        nx, ny = -d[:, 1] / np.where(ok, length, 1), d[:, 0] / np.where(ok, length, 1)
        da = np.abs((a[0] - p1[:, 0]) * nx + (a[1] - p1[:, 1]) * ny)
        db = np.abs((b[0] - p1[:, 0]) * nx + (b[1] - p1[:, 1]) * ny)
        ta = ((a[0] - p1[:, 0]) * d[:, 0] + (a[1] - p1[:, 1]) * d[:, 1]) / np.where(ok, length, 1)
        tb = ((b[0] - p1[:, 0]) * d[:, 0] + (b[1] - p1[:, 1]) * d[:, 1]) / np.where(ok, length, 1)
        # (a seg may end at a vertex the BSP builder added where a partition line cuts the linedef: rounded to integer map
        # units, up to a unit off the linedef's line)
        tol = 1.0 + 4 * BIAS_MAP
        tm = 0.5 * (ta + tb)  # the quad's midpoint must project INTO the linedef (collinear neighbours share an end point)
        on = ok & (da < tol) & (db < tol) & (np.minimum(ta, tb) > -tol) & (np.maximum(ta, tb) < length + tol) & (tm > 0) & (tm < length)
        cand = np.nonzero(on)[0]
        assert len(cand) >= 1, q
        k = int(cand[np.argmin(np.maximum(da, db)[cand])])  # the linedef the quad lies closest to
        front = tb[k] > ta[k]
        side = m.linedefs[k][5] if front else m.linedefs[k][6]
        assert side != 0xFFFF and side < len(m.sidedefs), q
        lo_h, hi_h = q[6]
        groups.setdefault((k, front), []).append((min(ta[k], tb[k]), max(ta[k], tb[k]), q, side))
    checked_heights = 0
    for (k, front), items in groups.items():
        side = items[0][3]
        x_off, y_off, up, low, mid, sec = m.sidedefs[side]
        other = m.linedefs[k][6] if front else m.linedefs[k][5]
        osec = m.side_sector(other)
        # parts are told apart by their height range
        parts = {}
        for t0, t1, q, _ in items:
            key = (round(q[6][0], 4), round(q[6][1], 4))
            parts.setdefault(key, []).append((t0, t1, q))
        for key, segs in parts.items():
            segs.sort(key=lambda s: s[0])
            # along the linedef: the quads of one part cover [0, length] of the side without gap or overlap
            # (a seg may be missing where the BSP builder dropped it: allowed only at the ends, never inside)
            for (a0, a1, _), (b0, b1, _) in zip(segs, segs[1:]):
                assert abs(b0 - a1) <= 4 * BIAS_MAP + 0.01, (which, index, k, front, key, a1, b0)
            # texture s runs with the distance from the side's first vertex: s = x_offset + distance, growing v1 -> v2
            # of the SEG (the seg's direction is the side's own direction)
            for t0, t1, q in segs:
                s1, s2 = q[4][0], q[5][0]
                start = t0 if front else length[k] - t1
                # (SEGS stores the offset as an integer: the BSP builder rounded the distance -- once per split, and the
                # split vertex itself sits up to 0.71 units off: the sweep's slanted walls reach one unit)
                assert abs(s1 - (x_off + start)) <= (1.01 if _swept(which) else 0.51) + 2 * BIAS_MAP, (which, index, k, front, s1, x_off, start)
                assert abs((s2 - s1) - (t1 - t0)) <= 0.05 + 4 * BIAS_MAP
        # heights: only where no sector involved can move (the walker uses the movement ranges there)
        if sec in dyn or (osec is not None and osec in dyn):
            continue
        ff, fc = m.sectors[sec][0], m.sectors[sec][1]
        spans = sorted(parts.keys())
        tolh = 2 * mc.POLY_BIAS + 1e-5
        if osec is None:
            # one-sided: one middle quad from floor to ceiling
            assert len(spans) == 1, (which, index, k, spans)
            assert abs(spans[0][0] - ff / 100.0) <= tolh + 1e-4 and abs(spans[0][1] - fc / 100.0) <= tolh + 1e-4
            checked_heights += 1
        else:
            bf, bc = m.sectors[osec][0], m.sectors[osec][1]
            sky_back = mc.Map.tex(m.sectors[osec][3]) == b'F_SKY1'
            want = []
            if bf > ff:
                want.append((ff / 100.0, bf / 100.0))
            if bc < fc and not sky_back:
                want.append((bc / 100.0, fc / 100.0))
            lowers_uppers = [s for s in spans if any(abs(s[0] - w[0]) <= tolh + 1e-4 and abs(s[1] - w[1]) <= tolh + 1e-4 for w in want)]
            assert len(lowers_uppers) == len(want), (which, index, k, front, spans, want)
            # what is left is the middle texture of a two-sided line: drawn once, not tiled -- inside the opening, or hanging
            # from its ceiling / standing on its floor (shifted by the side's y offset) at the texture's own height
            lo_open, hi_open = max(ff, bf), min(fc, bc)
            for s in spans:
                if s in lowers_uppers:
                    continue
                inside = s[0] >= lo_open / 100.0 - tolh - 1e-4 and s[1] <= hi_open / 100.0 + tolh + 1e-4
                stands = abs(s[0] - (lo_open + y_off) / 100.0) <= tolh + 1e-4
                hangs = abs(s[1] - (hi_open + y_off) / 100.0) <= tolh + 1e-4
                assert inside or stands or hangs, (which, index, k, front, s, ff, bf, fc, bc, y_off)
            checked_heights += 1
    assert checked_heights > 0.2 * len(groups) or len(groups) < 50, (checked_heights, len(groups))
    # every one-sided linedef with a middle texture the IWAD defines, in a sector with room between floor and ceiling, is
    # drawn (the reference skips a quad whose texture is unknown, visitor.rs:855-872, and one of no height, :849-851)
    known = mc.wall_texture_names(_path(wads, which))
    for k, (_, _, _, _, _, right, left) in enumerate(m.linedefs):
        if left != 0xFFFF or right == 0xFFFF or length[k] == 0:
            continue
        _, _, up, low, mid, sec = m.sidedefs[right]
        expect = mc.Map.tex(mid) in known and m.sectors[sec][0] < m.sectors[sec][1]  # (a closed door's track: only if it can open)
        if expect:
            assert (k, True) in groups, (which, index, k, mc.Map.tex(mid), m.sectors[sec])
        if mc.Map.tex(mid) != b'-' and mc.Map.tex(mid) not in known:
            assert (k, True) not in groups, (which, index, k)


@pytest.mark.parametrize('which,index', _cases())
def test_flat_triangles_carry_their_sectors_attributes(wads, which, index):
    """(c): a top-down look at the Builder's output.  For a grid of map points the sector is found by ray casting; the
    static flat TRIANGLE (vertex + index arrays as handed to the renderer) over the point must be at the sector's floor
    height (and one at its ceiling height), with the light level of the sector (a_light indexes the light table whose
    base level is (light >> 3) / 31, wad/src/light.rs:113-115)."""
    m, _, arr = _load(wads, which, index)
    v, idx, draws = arr['static_vertices'], arr['static_indices'], arr['draws']
    tri = []
    for kind, obj, first, count in draws:
        if kind == 0:  # RDOOM_KIND_FLAT
            tri.append(idx[first:first + count].reshape(-1, 3))
    tri = np.concatenate(tri)
    P = v['a_pos'][tri].astype(np.float64)                      # (n, 3, 3)
    xy = mc.world_to_map(P[:, :, [0, 2]].reshape(-1, 2)).reshape(-1, 3, 2)
    hgt = P[:, :, 1] * 100.0
    assert np.allclose(hgt, hgt[:, :1], atol=1e-3)              # horizontal
    def cross2(a, b):
        return a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]

    area2 = cross2(xy[:, 1] - xy[:, 0], xy[:, 2] - xy[:, 0])
    keep = np.abs(area2) > 1e-9                                 # the fan's degenerate leading triangle (level.rs:636-645)
    xy, hgt, tri_k = xy[keep], hgt[keep, 0], tri[keep]
    lo, hi = m.vertices.min(axis=0), m.vertices.max(axis=0)
    rng = np.random.RandomState(1234 + index)
    n = 4000 if which != 'big' else 8000
    pts = lo + rng.rand(n, 2) * (hi - lo)
    sec, dist = m.sector_at(pts)
    inside = (sec >= 0) & (dist > 1.0)     # a map unit away from every linedef: far outside the POLY_BIAS ring
    pts, sec = pts[inside], sec[inside]
    assert len(pts) > n // 8
    dyn = _possibly_dynamic(m)
    lights0 = arr['lights0']
    flats = mc.flat_lumps(_path(wads, which))
    atlas = arr['flat_atlas']
    holes = doubles = checked = 0
    for c0 in range(0, len(pts), 256):
        pc = pts[c0:c0 + 256][:, None, :]                       # (c, 1, 2) against (n, 2)
        d0 = cross2(xy[None, :, 1] - xy[None, :, 0], pc - xy[None, :, 0])
        d1 = cross2(xy[None, :, 2] - xy[None, :, 1], pc - xy[None, :, 1])
        d2 = cross2(xy[None, :, 0] - xy[None, :, 2], pc - xy[None, :, 2])
        overs = ((d0 >= 0) & (d1 >= 0) & (d2 >= 0)) | ((d0 <= 0) & (d1 <= 0) & (d2 <= 0))
        for j in range(overs.shape[0]):
            over = np.nonzero(overs[j])[0]                      # triangles over the point (barycentric signs)
            s = sec[c0 + j]
            hs = hgt[over]
            floor_h, ceil_h, ftex, ctex, light = (m.sectors[s][0], m.sectors[s][1], mc.Map.tex(m.sectors[s][2]),
                                                  mc.Map.tex(m.sectors[s][3]), m.sectors[s][4])
            want = []
            if ftex != b'F_SKY1':
                want.append((floor_h, ftex))
            if ctex != b'F_SKY1' and s not in dyn:
                want.append((ceil_h, ctex))
            for h, tex in want:
                hit = np.nonzero(np.abs(hs - h) < 1e-3)[0]
                if len(hit) == 0:
                    holes += 1  # nothing drawn over a point of the map: a hole in the floor / ceiling
                    continue
                if len(hit) > 1:
                    doubles += 1  # two sub-sectors of one sector overlap here (protrusions, see the polygon test): same height
                    continue
                pv = v[tri_k[over[hit[0]]][2]]                  # the provoking vertex carries the flat attributes
                checked += 1
                # light: in a sector without a light effect (type 0) the vertex's table entry holds, at any time,
                # u8((light >> 3) / 31 clamped * 255) -- wad/src/light.rs:113-115, game/src/lights.rs:26-30, in binary32
                if m.sectors[s][5] == 0:
                    lvl = np.float32(light >> 3) / np.float32(31.0)
                    expect = int(np.float32(min(max(lvl, np.float32(0)), np.float32(1))) * np.float32(255.0))
                    assert int(lights0[int(pv['a_light'])]) == expect, (which, index, int(s), int(pv['a_light']), expect)
                # flat: the 64 x 64 tile the vertex points at in the flat atlas holds the bytes of the sector's flat lump
                # (an animated flat points at the first frame of its sequence: some flat lump of the IWAD)
                au, av = int(pv['a_atlas_uv'][0]), int(pv['a_atlas_uv'][1])
                assert tuple(pv['a_tile_size']) == (64.0, 64.0)
                tile = atlas[av:av + 64, au:au + 64].tobytes()
                if int(pv['a_num_frames']) == 1:
                    assert tile == flats[tex], (which, index, int(s), tex)
                else:
                    assert tile in flats.values(), (which, index, int(s), tex)
            # nothing else over the point but (for moving sectors) the ceiling at another height
            extra = [h for h in hs if not any(abs(h - w) < 1e-3 for w, _ in want)]
            if s not in dyn:
                assert not extra, (which, index, pts[c0 + j], int(s), extra, want)
    # no holes -- but for what the polygon test allows the reference: a sliver sub-sector whose polygon comes out dented is
    # fanned into triangles that miss part of it (one sample in a few thousand); overlaps stay within the 3 % of that test
    assert holes <= (max(1, 0.015 * checked) if _swept(which) else 0), (which, index, holes, len(pts))
    assert doubles <= (0.04 if _swept(which) else 0.02) * (checked + doubles), (which, index, doubles, checked)
    assert checked > len(pts)
