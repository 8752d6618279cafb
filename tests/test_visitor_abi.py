"""The reference's plug-in point -- trait wad::LevelVisitor (wad/src/visitor.rs:65-127) -- across the C ABI:
rdoom_visitor_vtbl / rdoom_wad_walk / rdoom_wad_build_level_chained (include/rdoom.h).  A non-C++ host (here: Python
callbacks through ctypes) is a second implementor next to the Builder, the way game::world::WorldBuilder is
(game/src/world.rs:306), chained as in game/src/level.rs:378-382.  Every event and payload is compared with the ones the
oracle's LevelWalker emits for its own recording visitor."""
import numpy as np
import pytest

import rust_doom_amd as rd
from oracle import wad_oracle
from util import META_PATH


class Recorder:
    """collects every event; payloads are copied out (they are borrowed for the duration of the call)"""

    def __init__(self):
        self.events = []

    @staticmethod
    def _light(p):
        if not p:
            return None
        li = p.contents
        return (li.level, (li.effect_kind, li.alt_level, li.speed, li.duration, li.sync) if li.has_effect else None)

    def visit_wall_quad(self, q):
        self.events.append(('wall', q.object_id, tuple(q.v1), tuple(q.v2), tuple(q.tex_start), tuple(q.tex_end),
                            tuple(q.height_range), self._light(q.light_info), q.scroll,
                            bytes(q.tex_name) if q.has_tex_name else None, bool(q.blocker)))

    def _poly(self, tag, p):
        pts = tuple(p.vertices[i] for i in range(2 * p.n_vertices))
        self.events.append((tag, p.object_id, pts, p.height) + ((self._light(p.light_info), bytes(p.tex_name)) if hasattr(p, 'tex_name') else ()))

    def visit_floor_poly(self, p): self._poly('floor', p)
    def visit_ceil_poly(self, p): self._poly('ceil', p)
    def visit_floor_sky_poly(self, p): self._poly('floor_sky', p)
    def visit_ceil_sky_poly(self, p): self._poly('ceil_sky', p)

    def visit_sky_quad(self, q):
        self.events.append(('sky_quad', q.object_id, tuple(q.v1), tuple(q.v2), tuple(q.height_range)))

    def visit_marker(self, pos, yaw, marker, player):
        self.events.append(('marker', (pos[0], pos[1], pos[2]), yaw, marker, player))

    def visit_decor(self, d):
        self.events.append(('decor', d.object_id, tuple(d.low), tuple(d.high), d.half_width, self._light(d.light_info), bytes(d.tex_name)))

    def visit_bsp_root(self, line): self.events.append(('root', tuple(line.origin), tuple(line.displace), line.length))
    def visit_bsp_node(self, line, branch): self.events.append(('node', tuple(line.origin), tuple(line.displace), line.length, branch))
    def visit_bsp_leaf(self, branch): self.events.append(('leaf', branch))
    def visit_bsp_leaf_end(self): self.events.append(('leaf_end',))
    def visit_bsp_node_end(self): self.events.append(('node_end',))


class OracleRecorder(wad_oracle.LevelVisitor):
    """the same events from the oracle's walker (oracle/wad_oracle.py: LevelWalker), normalised to the tuples above"""

    def __init__(self):
        self.events = []

    @staticmethod
    def _light(li):
        if li is None:
            return None
        if li.effect is None:
            return (float(li.level), None)
        alt, speed, dur, sync, kind = li.effect
        return (float(li.level), (int(kind), float(alt), float(speed), float(dur), float(sync)))

    @staticmethod
    def _f(seq):
        return tuple(float(x) for x in seq)

    def visit_wall_quad(self, q):
        self.events.append(('wall', int(q['object_id']), self._f(q['vertices'][0]), self._f(q['vertices'][1]), self._f(q['tex_start']),
                            self._f(q['tex_end']), self._f(q['height_range']), self._light(q['light_info']), float(q['scroll']),
                            None if q['tex_name'] is None else bytes(q['tex_name']), bool(q['blocker'])))

    def _poly(self, tag, p, textured):
        pts = tuple(float(c) for v in p['vertices'] for c in v)
        self.events.append((tag, int(p['object_id']), pts, float(p['height'])) +
                           ((self._light(p['light_info']), bytes(p['tex_name'])) if textured else ()))

    def visit_floor_poly(self, p): self._poly('floor', p, True)
    def visit_ceil_poly(self, p): self._poly('ceil', p, True)
    def visit_floor_sky_poly(self, p): self._poly('floor_sky', p, False)
    def visit_ceil_sky_poly(self, p): self._poly('ceil_sky', p, False)

    def visit_sky_quad(self, q):
        self.events.append(('sky_quad', int(q['object_id']), self._f(q['vertices'][0]), self._f(q['vertices'][1]), self._f(q['height_range'])))

    def visit_marker(self, pos, yaw, marker):
        kind, player = marker
        kind = {'StartPos': 0, 'TeleportStart': 1, 'TeleportEnd': 2}[kind]
        self.events.append(('marker', self._f(pos), float(yaw), int(kind), int(player)))

    def visit_decor(self, d):
        self.events.append(('decor', int(d['object_id']), self._f(d['low']), self._f(d['high']), float(d['half_width']),
                            self._light(d['light_info']), bytes(d['tex_name'])))

    def visit_bsp_root(self, line): self.events.append(('root',) + self._line(line))
    def visit_bsp_node(self, line, branch): self.events.append(('node',) + self._line(line) + ({'Positive': 0, 'Negative': 1}[branch],))
    def visit_bsp_leaf(self, branch): self.events.append(('leaf', {'Positive': 0, 'Negative': 1}[branch]))
    def visit_bsp_leaf_end(self): self.events.append(('leaf_end',))
    def visit_bsp_node_end(self): self.events.append(('node_end',))

    def _line(self, line):
        return ((float(line.ox), float(line.oy)), (float(line.dx), float(line.dy)), float(line.length))


def oracle_events(lv):
    rec = OracleRecorder()
    wad_oracle.LevelWalker(lv.level, lv.analysis, lv.tex, lv.wad.meta, rec).walk()
    return rec.events


@pytest.mark.parametrize('index', [0, 3, 7])
def test_walk_delivers_the_oracles_events(wad_path, oracle_levels, index):
    rec = Recorder()
    rd.Wad(wad_path, META_PATH).walk(index, rec)
    want = oracle_events(oracle_levels(index))
    assert len(rec.events) == len(want)
    for got, exp in zip(rec.events, want):
        assert got == exp, (got, exp)
    kinds = {e[0] for e in rec.events}
    assert {'wall', 'floor', 'ceil', 'sky_quad', 'marker', 'decor', 'root', 'node', 'leaf', 'leaf_end', 'node_end'} <= kinds


def test_chained_visitor_sees_what_the_builder_counts(wad_path):
    """builder.chain(second) (game/src/level.rs:378-382): same events, and the level built is the unchained one"""
    wad = rd.Wad(wad_path, META_PATH)
    rec = Recorder()
    built = wad.build_level(0, visitor=rec)
    c = built.counters()
    n = lambda tag: sum(1 for e in rec.events if e[0] == tag)  # noqa: E731
    assert (n('wall'), n('floor'), n('ceil'), n('sky_quad'), n('floor_sky'), n('ceil_sky'), n('decor')) == (
        c['num_wall_quads'], c['num_floor_polys'], c['num_ceil_polys'], c['num_sky_wall_quads'], c['num_sky_floor_polys'],
        c['num_sky_ceil_polys'], c['num_decors'])
    plain = wad.build_level(0).arrays()
    for k, a in built.arrays().items():
        assert np.array_equal(np.asarray(a), np.asarray(plain[k])), k


def test_partial_visitor_and_bad_arguments(wad_path):
    class OnlyLeaves:
        n = 0

        def visit_bsp_leaf(self, branch):
            OnlyLeaves.n += 1

    wad = rd.Wad(wad_path, META_PATH)
    wad.walk(1, OnlyLeaves())
    assert OnlyLeaves.n > 0
    with pytest.raises(rd.RdoomError):
        wad.walk(99, OnlyLeaves())
