#!/usr/bin/env python3
"""Synthetic IWAD writer (fixture generator).

No DOOM1.WAD / DOOM2.WAD exists in the build container or on the GPU box and
none can be downloaded, so every test, the smoke run and the bench use an IWAD
written by this script.  It emits a structurally complete IWAD:

  PLAYPAL (14 palettes), COLORMAP (34 maps), PNAMES, TEXTURE1, patch lumps
  (P_START..P_END), flats (F_START..F_END, incl. the animated NUKAGE1-3 trio
  and F_SKY1), sprites (S_START..S_END) and levels E1M1..E1M9, each with the 10
  classic lumps.  Levels are authored on a cell grid (rooms, corridors, doors,
  lifts, stairs, windows, pools, pillars, 45-degree corner cuts, jittered
  vertices) and run through the BSP node builder below, which follows the
  on-disk conventions the reference reader expects (wad/src/types.rs:33-150,
  wad/src/level.rs:13-20, wad/src/visitor.rs:590-619: right child = right side
  of the partition direction, segs carry their sector on the right).

E1M1 is sized like the shareware E1M1 (a few hundred linedefs / subsectors);
E1M2 is a tiny hand-authored level used for analytic known-answer tests.

Everything is seeded: the same arguments always give byte-identical output
(tests/golden/synth_wad.sha256 pins it).
"""
import argparse
import hashlib
import math
import struct
import sys

import numpy as np

CELL = 64


# --------------------------------------------------------------------------------------
# palette / colormap
# --------------------------------------------------------------------------------------
def make_playpal():
    ramps = [  # (r,g,b) anchors for 16 ramps of 16 shades
        (255, 255, 255), (255, 80, 60), (255, 170, 90), (200, 150, 100),
        (120, 255, 110), (90, 140, 60), (110, 130, 255), (255, 240, 90),
        (190, 120, 70), (150, 150, 170), (90, 200, 200), (220, 110, 220),
        (130, 100, 80), (255, 120, 0), (70, 90, 60), (180, 190, 160),
    ]
    pal0 = np.zeros((256, 3), np.float64)
    for r, anchor in enumerate(ramps):
        for s in range(16):
            f = (16 - s) / 16.0
            pal0[r * 16 + s] = [c * f for c in anchor]
    pal0[0] = (0, 0, 0)
    pals = []
    for i in range(14):
        if i == 0:
            p = pal0
        elif i < 9:  # red pain tints
            p = pal0 + (np.array([255, 0, 0]) - pal0) * (i / 9.0)
        elif i < 13:  # pickup tints
            p = pal0 + (np.array([215, 186, 69]) - pal0) * ((i - 8) / 8.0)
        else:  # radiation suit
            p = pal0 + (np.array([0, 255, 0]) - pal0) * 0.125
        pals.append(np.clip(np.rint(p), 0, 255).astype(np.uint8))
    return pals


def nearest_index(pal, rgb):
    d = ((pal[None, :, :].astype(np.int32) - rgb[:, None, :].astype(np.int32)) ** 2).sum(-1)
    return d.argmin(1).astype(np.uint8)


def make_colormap(pal0):
    maps = []
    base = pal0.astype(np.float64)
    for i in range(32):
        scaled = np.rint(base * (32 - i) / 32.0)
        maps.append(nearest_index(pal0, scaled))
    grey = (base * [0.299, 0.587, 0.114]).sum(1)
    inv = np.repeat((255 - grey)[:, None], 3, 1)
    maps.append(nearest_index(pal0, np.rint(inv)))  # 32: invulnerability
    maps.append(np.zeros(256, np.uint8))  # 33: black
    return maps


# --------------------------------------------------------------------------------------
# pictures
# --------------------------------------------------------------------------------------
def encode_picture(pix, xoff=0, yoff=0):
    """pix: int16 array [h][w], -1 = transparent.  Doom column/post picture format."""
    h, w = pix.shape
    assert h <= 254
    cols = []
    for x in range(w):
        col = bytearray()
        y = 0
        while y < h:
            if pix[y, x] < 0:
                y += 1
                continue
            y0 = y
            while y < h and pix[y, x] >= 0 and y - y0 < 128:
                y += 1
            run = pix[y0:y, x].astype(np.uint8).tobytes()
            col += bytes([y0, len(run), run[0]]) + run + bytes([run[-1]])
        col.append(255)
        cols.append(bytes(col))
    head = struct.pack('<HHhh', w, h, xoff, yoff)
    off = len(head) + 4 * w
    table = b''
    for c in cols:
        table += struct.pack('<I', off)
        off += len(c)
    return head + table + b''.join(cols)


class Rng:
    """xorshift32 so output does not depend on numpy/python RNG versions."""

    def __init__(self, seed):
        self.s = (seed & 0xFFFFFFFF) or 1

    def u32(self):
        s = self.s
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        self.s = s
        return s

    def randint(self, lo, hi):  # inclusive
        return lo + self.u32() % (hi - lo + 1)

    def chance(self, p):
        return (self.u32() & 0xFFFF) < p * 65536

    def choice(self, seq):
        return seq[self.u32() % len(seq)]


def tex_pattern(kind, w, h, ramp, rng):
    """Procedural palette-index pattern; ramp = 16-colour ramp number."""
    yy, xx = np.mgrid[0:h, 0:w]
    noise = np.array([[rng.u32() & 3 for _ in range(w)] for _ in range(h)])
    if kind == 'brick':
        row = yy // 16
        bx = (xx + (row % 2) * 16) % 32
        shade = 4 + noise + np.where((yy % 16 == 0) | (bx == 0), 6, 0)
    elif kind == 'panel':
        shade = 3 + noise + np.where((xx % 32 < 2) | (yy % 64 < 2), 7, 0) + (yy * 3 // h)
    elif kind == 'stripe':
        shade = 2 + ((xx // 8 + yy // 8) % 2) * 5 + noise
    elif kind == 'rock':
        shade = 3 + noise * 2 + ((xx * 7 + yy * 13) % 5)
    elif kind == 'grad':
        shade = (xx * 12 // max(w, 1)) + (yy * 3 // max(h, 1)) + (noise >> 1)
    elif kind == 'door':
        shade = 4 + noise + np.where((xx < 4) | (xx >= w - 4) | (yy < 4) | (yy % 24 == 0), 6, 0)
    elif kind == 'checker':
        shade = 2 + ((xx // 16 + yy // 16) % 2) * 8 + (noise >> 1)
    else:
        shade = 5 + noise
    shade = np.clip(shade, 0, 15)
    return (ramp * 16 + shade).astype(np.int16)


def make_graphics(rng, shapes=False):
    patches = {}  # name -> int16 [h][w]

    def patch(name, kind, w, h, ramp):
        patches[name] = tex_pattern(kind, w, h, ramp, rng)

    patch('WALL00_1', 'brick', 64, 128, 3)
    patch('WALL00_2', 'brick', 64, 128, 8)
    patch('WALL01_1', 'panel', 128, 128, 9)
    patch('WALL02_1', 'rock', 64, 128, 12)
    patch('WALL03_1', 'stripe', 64, 72, 6)
    patch('W13_1', 'panel', 24, 128, 15)       # non power-of-two width
    patch('DOOR2_1', 'door', 64, 72, 10)
    patch('STEP1', 'grad', 32, 16, 9)
    patch('STEP2', 'grad', 32, 8, 2)
    patch('SW1S0', 'checker', 32, 32, 7)
    patch('SLAD1', 'rock', 64, 128, 4)
    patch('SLAD2', 'rock', 64, 128, 5)
    patch('SLAD3', 'rock', 64, 128, 14)
    patch('FIRE1', 'grad', 128, 128, 1)
    patch('FIRE2', 'grad', 128, 128, 13)
    patch('SKY1', 'grad', 256, 128, 6)
    # sky: make it less uniform (mountain silhouette)
    sky = patches['SKY1']
    for x in range(256):
        top = 70 + int(20 * math.sin(x * 2 * math.pi / 256 * 3) + 10 * math.sin(x * 2 * math.pi / 256 * 7))
        sky[top:, x] = 12 * 16 + np.clip(4 + (np.arange(128 - top) // 6), 0, 15)
    # masked grate: holes are transparent
    g = tex_pattern('panel', 64, 128, 9, rng)
    yy, xx = np.mgrid[0:128, 0:64]
    g[((xx % 16) >= 4) & ((yy % 16) >= 4)] = -1
    patches['GRATE1'] = g
    # a decal patch with transparent border for composite textures
    d = tex_pattern('checker', 48, 48, 1, rng)
    yy, xx = np.mgrid[0:48, 0:48]
    d[((xx - 24) ** 2 + (yy - 24) ** 2) > 23 ** 2] = -1
    patches['DECAL1'] = d

    pnames = list(patches.keys()) + ['MISSING1']  # last one has no lump (missing_patches path)

    def T(name, w, h, prefs):
        return (name, w, h, [(ox, oy, pnames.index(p)) for ox, oy, p in prefs])

    textures = [
        T('STARTAN3', 128, 128, [(0, 0, 'WALL00_1'), (64, 0, 'WALL00_2')]),
        T('BROWN1', 64, 128, [(0, 0, 'WALL00_2')]),
        T('COMPTALL', 256, 128, [(0, 0, 'WALL01_1'), (128, 0, 'WALL01_1'), (40, 40, 'DECAL1'),
                                 (168, -8, 'DECAL1')]),          # negative origin_y -> 0
        T('ROCK1', 64, 128, [(0, 0, 'WALL02_1')]),
        T('STONE2', 128, 128, [(0, 0, 'WALL02_1'), (64, 0, 'WALL00_1'), (100, 30, 'DECAL1')]),  # clipped right
        T('TEKWALL1', 128, 128, [(0, 0, 'WALL01_1')]),
        T('LITE3', 64, 72, [(0, 0, 'WALL03_1')]),
        T('SUPPORT2', 24, 128, [(0, 0, 'W13_1')]),
        T('DOOR1', 64, 72, [(0, 0, 'DOOR2_1')]),
        T('DOORTRAK', 8, 128, [(-4, 0, 'W13_1')]),                # negative origin_x clip
        T('STEP1', 32, 16, [(0, 0, 'STEP1')]),
        T('STEP2', 32, 8, [(0, 0, 'STEP2')]),
        T('SW1COMP', 64, 128, [(0, 0, 'WALL00_2'), (16, 72, 'SW1S0')]),
        T('SLADRIP1', 64, 128, [(0, 0, 'SLAD1')]),
        T('SLADRIP2', 64, 128, [(0, 0, 'SLAD2')]),
        T('SLADRIP3', 64, 128, [(0, 0, 'SLAD3')]),
        T('FIREBLU1', 128, 128, [(0, 0, 'FIRE1')]),
        T('FIREBLU2', 128, 128, [(0, 0, 'FIRE2')]),
        T('MIDGRATE', 64, 128, [(0, 0, 'GRATE1')]),
        T('BADPATCH', 64, 64, [(0, 0, 'MISSING1')]),              # PatchRef .. is missing
        T('SKY1', 256, 128, [(0, 0, 'SKY1')]),
        T('BIGDOOR2', 128, 128, [(0, 0, 'WALL01_1'), (32, 28, 'DOOR2_1'), (0, 0, 'DECAL1')]),
    ]

    flats = {}
    for name, kind, ramp in [
        ('FLOOR0_1', 'checker', 3), ('FLOOR4_8', 'panel', 9), ('FLAT5_4', 'rock', 12),
        ('CEIL3_5', 'panel', 15), ('CEIL5_1', 'stripe', 8), ('FLAT14', 'brick', 6),
        ('FLAT20', 'grad', 9), ('STEP_F', 'stripe', 2), ('TLITE6_4', 'checker', 7),
        ('NUKAGE1', 'rock', 4), ('NUKAGE2', 'rock', 5), ('NUKAGE3', 'rock', 14),
        ('FLAT1', 'grad', 11), ('DEM1_5', 'brick', 1), ('F_SKY1', 'plain', 6),
    ]:
        flats[name] = tex_pattern(kind, 64, 64, ramp, rng).astype(np.uint8)

    sprites = {}

    def sprite(name, w, h, ramp, shape):
        p = tex_pattern('grad', w, h, ramp, rng)
        yy, xx = np.mgrid[0:h, 0:w]
        if shape == 'ellipse':
            mask = ((xx - w / 2 + .5) / (w / 2)) ** 2 + ((yy - h / 2 + .5) / (h / 2)) ** 2 > 1
        elif shape == 'column':
            mask = (np.abs(xx - w / 2 + .5) > w / 4) & (yy > 6) & (yy < h - 6)
        else:
            mask = (np.abs(xx - w / 2 + .5) * h > (yy + 1) * w / 2)
        p[mask] = -1
        sprites[name] = (p, w // 2, h - 4)

    sprite('BAR1A0', 23, 32, 8, 'ellipse')
    sprite('COLUA0', 18, 48, 7, 'column')
    sprite('ELECA0', 38, 120, 9, 'column')
    sprite('TRE1A0', 56, 76, 5, 'tri')
    sprite('GOR1A0', 20, 68, 1, 'column')
    sprite('CANDA1', 8, 15, 7, 'ellipse')      # only the ..1 rotation exists (sprite1 lookup path)
    textures2 = []
    if shapes:
        # Lump shapes real IWADs have and the nine default levels do not:
        #  * sprite lumps with paired rotations (eight-character names: wad/src/tex.rs:475-497 inserts every lump between
        #    S_START and S_END under its own name), a family without an ..A0 lump (the decor looks ..A0 up, then ..A1,
        #    wad/src/visitor.rs:1071-1083);
        for name, w, h in (('COLUA2A8', 18, 48), ('COLUA3A7', 16, 48), ('COLUA4A6', 14, 48), ('COLUA5', 18, 48),
                           ('POSSA1', 38, 56), ('POSSA2A8', 34, 56), ('POSSA3A7', 30, 55), ('POSSA4A6', 26, 56), ('POSSA5', 36, 54),
                           ('POSSB1', 38, 56)):
            sprite(name, w, h, 4 + len(sprites) % 9, 'ellipse' if name.startswith('POSS') else 'column')
        #  * textures of three and more overlapping patches, transparent posts over opaque ones, the FIRST patch with holes
        #    (blitted with ignore_transparency: its holes stay holes, wad/src/tex.rs:570, wad/src/image.rs:171-252), patches
        #    hanging over every edge of the texture;
        textures += [
            T('OVERLAP3', 128, 128, [(0, 0, 'WALL01_1'), (20, 10, 'GRATE1'), (40, 30, 'DECAL1'), (-10, 100, 'GRATE1'),
                                     (100, -20, 'DECAL1'), (90, 90, 'STEP1')]),
            T('GRATEMIX', 64, 128, [(0, 0, 'GRATE1'), (8, 8, 'STEP1'), (30, 60, 'DECAL1'), (-40, -40, 'WALL03_1')]),
        ]
        #  * a TEXTURE2 lump next to TEXTURE1 (wad/src/tex.rs:71-88, 356), one of its textures only there, one re-defining
        #    a TEXTURE1 name (IndexMap::insert: the later definition replaces the image, tex.rs:590).
        textures2 = [
            T('T2ONLY', 128, 72, [(0, 0, 'WALL02_1'), (64, 0, 'WALL03_1'), (32, 20, 'SW1S0')]),
            T('BROWN1', 64, 128, [(0, 0, 'WALL02_1'), (0, 64, 'STEP1'), (32, 64, 'STEP2')]),
        ]
    return patches, pnames, textures, flats, sprites, textures2


def rich_graphics(seed, n_walls=320, n_flats=192):
    """The texture-rich stand-in (build_wad(rich=True)): `n_walls` wall textures of the sizes real IWADs use, each with a patch
    of its own (every fourth one composed of two), and `n_flats` flats -- so that a level whose every linedef and sector picks
    its own ends up with a wall atlas of 2048 x 2048 and more (wad/src/tex.rs:168-271) and a flat atlas of 1024 x 1024
    (tex.rs:273-333): a texel store several times one XCD's 4 MiB of L2, where the nine default levels keep theirs at 1.1 MB.
    Patterns as tex_pattern's, with the noise from an integer hash of (texture, x, y) (numpy; a Python-level xorshift per texel
    would take minutes).  Returns (patches, textures as (name, w, h, [(ox, oy, patch name)]), flats)."""
    sizes = [(64, 128), (128, 128), (128, 128), (256, 128), (64, 128), (128, 128), (64, 72), (256, 128), (128, 64), (64, 64), (24, 128), (128, 96)]
    kinds = ['brick', 'panel', 'stripe', 'rock', 'grad', 'door', 'checker']

    def pattern(kind, w, h, ramp, salt):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.uint32)
        v = (xx * np.uint32(0x9E3779B1)) ^ (yy * np.uint32(0x85EBCA77)) ^ np.uint32((salt * 0xC2B2AE3D + seed) & 0xFFFFFFFF)
        v ^= v >> np.uint32(15)
        v = v * np.uint32(0x2C1B3C6D)
        v ^= v >> np.uint32(12)
        noise = (v & np.uint32(3)).astype(np.int64)
        xx, yy = xx.astype(np.int64), yy.astype(np.int64)
        k = salt % 7 + 1
        if kind == 'brick':
            row = yy // 16
            bx = (xx + (row % 2) * 16) % 32
            shade = 4 + noise + np.where((yy % 16 == 0) | (bx == 0), 6, 0)
        elif kind == 'panel':
            shade = 3 + noise + np.where((xx % 32 < 2) | (yy % 64 < 2), 7, 0) + (yy * 3 // h)
        elif kind == 'stripe':
            shade = 2 + ((xx // (4 * k) + yy // 8) % 2) * 5 + noise
        elif kind == 'rock':
            shade = 3 + noise * 2 + ((xx * (5 + k) + yy * 13) % 5)
        elif kind == 'grad':
            shade = (xx * 12 // max(w, 1)) + (yy * 3 // max(h, 1)) + (noise >> 1)
        elif kind == 'door':
            shade = 4 + noise + np.where((xx < 4) | (xx >= w - 4) | (yy < 4) | (yy % 24 == 0), 6, 0)
        else:
            shade = 2 + ((xx // (8 + 2 * k) + yy // 16) % 2) * 8 + (noise >> 1)
        return (ramp * 16 + np.clip(shade, 0, 15)).astype(np.int16)

    patches, textures, flats = {}, [], {}
    for i in range(n_walls):
        w, h = sizes[i % len(sizes)]
        name, pname = 'RW%03d' % i, 'RP%03d' % i
        patches[pname] = pattern(kinds[i % len(kinds)], w, h, 1 + i % 15, i)
        prefs = [(0, 0, pname)]
        if i % 4 == 3 and i >= 4:   # a second patch over the first, from an earlier texture (clipped where it hangs over)
            prefs.append((w // 4, h // 4, 'RP%03d' % (i - 4)))
        textures.append((name, w, h, prefs))
    for i in range(n_flats):
        flats['RF%03d' % i] = pattern(kinds[(i * 3) % len(kinds)], 64, 64, 1 + (i * 7) % 15, 1000 + i).astype(np.uint8)
    return patches, textures, flats


# when build_wad(rich=True) is at work: every linedef side picks ITS OWN wall texture from this list (None: the sector's)
RICH_WALLS = None


# --------------------------------------------------------------------------------------
# level authoring on a cell grid
# --------------------------------------------------------------------------------------
class Sector:
    def __init__(self, floor, ceil, ftex, ctex, light, special=0, tag=0, wall='STARTAN3'):
        self.floor, self.ceil, self.ftex, self.ctex = floor, ceil, ftex, ctex
        self.light, self.special, self.tag, self.wall = light, special, tag, wall
        self.upper = wall
        self.lower = wall
        self.kind = 'room'


class GridLevel:
    """cells[y][x] = (sector_a, sector_b, diag) ; diag 0 = whole cell is sector_a;
    diag 1 = split along (x0,y0)-(x1,y1) ('/' in y-up coords... a below/right, b above/left);
    diag 2 = split along (x0,y1)-(x1,y0).  -1 = void."""

    def __init__(self, w, h):
        self.w, self.h = w, h
        self.a = -np.ones((h, w), np.int32)
        self.b = -np.ones((h, w), np.int32)
        self.diag = np.zeros((h, w), np.int32)
        self.sectors = []
        self.things = []
        self.line_special = {}   # (sector_front, sector_back) -> (special, tag)
        self.jitter = {}         # (gx,gy) -> (dx,dy)
        self.mid_masked = set()  # (sa,sb) unordered pairs that get a masked middle texture
        self.scroll_sectors = set()

    def add_sector(self, *a, **k):
        self.sectors.append(Sector(*a, **k))
        return len(self.sectors) - 1

    def fill(self, x0, y0, x1, y1, s):
        self.a[y0:y1, x0:x1] = s
        self.b[y0:y1, x0:x1] = s
        self.diag[y0:y1, x0:x1] = 0

    def vertex(self, gx, gy):
        dx, dy = self.jitter.get((gx, gy), (0, 0))
        return (gx * CELL + dx, gy * CELL + dy)


FLOORS = ['FLOOR0_1', 'FLOOR4_8', 'FLAT5_4', 'FLAT14', 'FLAT20', 'FLAT1', 'DEM1_5']
CEILS = ['CEIL3_5', 'CEIL5_1', 'FLAT20', 'TLITE6_4', 'FLAT1']
WALLS = ['STARTAN3', 'BROWN1', 'COMPTALL', 'ROCK1', 'STONE2', 'TEKWALL1', 'SLADRIP1', 'FIREBLU1']
LIGHT_SPECIALS = [1, 2, 3, 8, 12, 13, 17]
DECOR_TYPES = [2035, 2028, 48, 43, 34, 9999]  # 9999 has no metadata (skip path)


def gen_level(seed, grid=52, n_rooms=14):
    rng = Rng(seed)
    L = GridLevel(grid, grid)
    rooms = []
    tries = 0
    while len(rooms) < n_rooms and tries < 4000:
        tries += 1
        w, h = rng.randint(4, 10), rng.randint(4, 9)
        x0, y0 = rng.randint(2, grid - w - 2), rng.randint(2, grid - h - 2)
        if any(x0 < r[2] + 2 and r[0] < x0 + w + 2 and y0 < r[3] + 2 and r[1] < y0 + h + 2 for r in rooms):
            continue
        rooms.append((x0, y0, x0 + w, y0 + h))
    next_tag = [1]
    room_sec = []
    for i, (x0, y0, x1, y1) in enumerate(rooms):
        floor = rng.randint(-2, 4) * 16
        height = rng.choice([96, 112, 128, 160, 192])
        sky = rng.chance(0.2) and i > 0
        s = L.add_sector(floor, floor + (256 if sky else height), rng.choice(FLOORS),
                         'F_SKY1' if sky else rng.choice(CEILS), rng.choice([112, 144, 160, 176, 192, 208, 224, 255]),
                         wall=rng.choice(WALLS))
        if rng.chance(0.25):
            L.sectors[s].special = rng.choice(LIGHT_SPECIALS)
        L.fill(x0, y0, x1, y1, s)
        room_sec.append(s)

    # corridors along a chain + a few extra links
    links = [(i, i + 1) for i in range(len(rooms) - 1)]
    for _ in range(len(rooms) // 3):
        a, b = rng.randint(0, len(rooms) - 1), rng.randint(0, len(rooms) - 1)
        if a != b:
            links.append((a, b))
    for a, b in links:
        ax, ay = (rooms[a][0] + rooms[a][2]) // 2, (rooms[a][1] + rooms[a][3]) // 2
        bx, by = (rooms[b][0] + rooms[b][2]) // 2, (rooms[b][1] + rooms[b][3]) // 2
        fa, fb = L.sectors[room_sec[a]].floor, L.sectors[room_sec[b]].floor
        floor = min(fa, fb)
        cs = L.add_sector(floor, floor + rng.choice([80, 96, 128]), rng.choice(FLOORS), rng.choice(CEILS),
                          rng.choice([96, 128, 144, 160]), wall=rng.choice(WALLS))
        L.sectors[cs].kind = 'corridor'
        width = rng.randint(1, 2)
        path = []
        x, y = ax, ay
        horizontal_first = rng.chance(0.5)
        for phase in range(2):
            if (phase == 0) == horizontal_first:
                while x != bx:
                    path.append((x, y))
                    x += 1 if bx > x else -1
            else:
                while y != by:
                    path.append((x, y))
                    y += 1 if by > y else -1
        path.append((bx, by))
        for (x, y) in path:
            for ox in range(width):
                for oy in range(width):
                    cx, cy = x + ox, y + oy
                    if 0 < cx < grid - 1 and 0 < cy < grid - 1 and L.a[cy, cx] < 0:
                        L.a[cy, cx] = L.b[cy, cx] = cs

    # doorway / door / step sectors where corridors meet rooms
    def neighbours(x, y):
        return [(x + 1, y), (x - 1, y), (x, y + 1), (x, y - 1)]

    door_cells = []
    for y in range(1, grid - 1):
        for x in range(1, grid - 1):
            s = L.a[y, x]
            if s >= 0 and L.sectors[s].kind == 'corridor':
                rs = [L.a[ny, nx] for nx, ny in neighbours(x, y)
                      if L.a[ny, nx] >= 0 and L.sectors[L.a[ny, nx]].kind == 'room']
                if rs:
                    door_cells.append((x, y, s, rs[0]))
    for (x, y, cs, rs) in door_cells:
        if L.a[y, x] != cs:
            continue
        r = rng.u32() % 10
        room, cor = L.sectors[rs], L.sectors[cs]
        floor = max(room.floor, cor.floor)
        if r < 2:      # closed manual door (tag 0, special 1 on its lines, door = left side)
            ds = L.add_sector(floor, floor, cor.ftex, 'FLAT20', cor.light, wall='DOORTRAK')
            L.sectors[ds].kind = 'door'
            L.sectors[ds].upper = 'BIGDOOR2' if rng.chance(0.5) else 'DOOR1'
        elif r < 3:    # tagged lift
            tag = next_tag[0]
            next_tag[0] += 1
            ds = L.add_sector(floor, cor.ceil, 'FLAT14', cor.ctex, cor.light, tag=tag, wall='SUPPORT2')
            L.sectors[ds].kind = 'lift'
        elif r < 6:    # doorway: lowered ceiling, raised step
            ds = L.add_sector(floor + 8, min(room.ceil, cor.ceil) - 16, 'STEP_F', 'FLAT20', 128, wall='LITE3')
            L.sectors[ds].kind = 'doorway'
            L.sectors[ds].lower = 'STEP1'
            L.sectors[ds].upper = 'LITE3'
        else:
            continue
        L.a[y, x] = L.b[y, x] = ds

    # room decoration
    for i, (x0, y0, x1, y1) in enumerate(rooms):
        rs = room_sec[i]
        room = L.sectors[rs]
        w, h = x1 - x0, y1 - y0
        kind = rng.u32() % 6
        if kind == 0 and w >= 5 and h >= 5:      # pool with animated flat + glow
            ps = L.add_sector(room.floor - 16, room.ceil, 'NUKAGE1', room.ctex, room.light, special=8,
                              wall='SLADRIP1')
            L.sectors[ps].lower = 'SLADRIP1'
            L.fill(x0 + 1, y0 + 1, x1 - 1, y1 - 1, ps)
            isl = L.add_sector(room.floor + 8, room.ceil, 'FLAT5_4', room.ctex, min(255, room.light + 32),
                               wall='ROCK1')
            L.fill(x0 + 2, y0 + 2, x0 + 3, y0 + 3, isl)
        elif kind == 1 and w >= 5 and h >= 5:    # raised platform with stairs
            base = room.floor
            for k in range(3):
                st = L.add_sector(base + 8 * (k + 1), room.ceil, 'STEP_F', room.ctex, room.light, wall='BROWN1')
                L.sectors[st].lower = 'STEP2'
                if x0 + 1 + k < x1 - 1:
                    L.fill(x0 + 1 + k, y0 + 1, x0 + 2 + k, y1 - 1, st)
            pl = L.add_sector(base + 32, room.ceil - 16 if room.ctex != 'F_SKY1' else room.ceil, 'FLAT14',
                              room.ctex, min(255, room.light + 16), wall='STONE2')
            if x0 + 4 < x1 - 1:
                L.fill(x0 + 4, y0 + 1, x1 - 1, y1 - 1, pl)
        elif kind == 2:                           # pillars
            for px in range(x0 + 1, x1 - 1, 2):
                for py in range(y0 + 1, y1 - 1, 2):
                    if rng.chance(0.45):
                        if rng.chance(0.5):
                            L.fill(px, py, px + 1, py + 1, -1)
                        else:
                            bl = L.add_sector(room.floor + rng.choice([24, 48, 64]), room.ceil, 'FLAT20',
                                              room.ctex, room.light, wall='TEKWALL1')
                            L.fill(px, py, px + 1, py + 1, bl)
        elif kind == 3 and w >= 4 and h >= 4:    # light patch w/ effect + grate divider
            ls = L.add_sector(room.floor, room.ceil - 8 if room.ctex != 'F_SKY1' else room.ceil, room.ftex,
                              'TLITE6_4', 255, special=rng.choice(LIGHT_SPECIALS), wall=room.wall)
            L.fill(x0 + 1, y0 + 1, x0 + 3, y0 + 3, ls)
            L.mid_masked.add((min(rs, ls), max(rs, ls)))
        elif kind == 4:                           # scrolling wall room
            L.scroll_sectors.add(rs)
            room.wall = 'FIREBLU1'
        # corner cuts (45 degree walls)
        if rng.chance(0.6):
            for (cx, cy, dg, keep_a) in [(x0, y0, 2, False), (x1 - 1, y0, 1, False), (x0, y1 - 1, 1, True),
                                         (x1 - 1, y1 - 1, 2, True)]:
                if rng.chance(0.5) and L.a[cy, cx] == rs and L.diag[cy, cx] == 0:
                    nb = [L.a[ny, nx] for nx, ny in neighbours(cx, cy)]
                    if sum(1 for v in nb if v == rs) == 2 and sum(1 for v in nb if v < 0) == 2:
                        L.diag[cy, cx] = dg
                        # a = lower part, b = upper part (see cell_halves)
                        if keep_a:
                            L.a[cy, cx], L.b[cy, cx] = rs, -1
                        else:
                            L.a[cy, cx], L.b[cy, cx] = -1, rs
                        # orientation fix: which half touches the room is decided in cell_halves
                        L.diag[cy, cx] = -dg  # negative = "auto": solved below
        # things
        for _ in range(rng.randint(1, 4)):
            tx, ty = rng.randint(x0, x1 - 1), rng.randint(y0, y1 - 1)
            if L.a[ty, tx] >= 0 and L.diag[ty, tx] == 0:
                L.things.append((tx * CELL + 32, ty * CELL + 32, rng.randint(0, 7) * 45, rng.choice(DECOR_TYPES), 7))
        if i == 1:
            L.things.append(((x0 + 1) * CELL + 32, (y0 + 1) * CELL + 32, 0, 63, 7))  # hanging decor

    # windows between rooms separated by a single void cell
    for y in range(2, grid - 2):
        for x in range(2, grid - 2):
            if L.a[y, x] >= 0:
                continue
            for (dx, dy) in [(1, 0), (0, 1)]:
                s1, s2 = L.a[y - dy, x - dx], L.a[y + dy, x + dx]
                if s1 >= 0 and s2 >= 0 and s1 != s2 and L.diag[y - dy, x - dx] == 0 and L.diag[y + dy, x + dx] == 0 \
                        and L.sectors[s1].kind == 'room' and L.sectors[s2].kind == 'room' and rng.chance(0.35):
                    # void on the two other sides?
                    o1, o2 = L.a[y - dx, x - dy], L.a[y + dx, x + dy]
                    if o1 < 0 and o2 < 0 and L.a[y, x] < 0:
                        a, b = L.sectors[s1], L.sectors[s2]
                        fl = max(a.floor, b.floor) + 32
                        ce = min(a.ceil, b.ceil) - 24
                        if ce - fl >= 24:
                            ws = L.add_sector(fl, ce, 'FLAT20', 'FLAT20', 160, wall='SUPPORT2')
                            L.sectors[ws].kind = 'window'
                            L.a[y, x] = L.b[y, x] = ws

    # resolve auto diagonals: the solid half must be the one away from the room
    for y in range(grid):
        for x in range(grid):
            if L.diag[y, x] < 0:
                dg = -L.diag[y, x]
                rs = max(L.a[y, x], L.b[y, x])
                # diag 1: split along (x0,y0)-(x1,y1): 'a' = below the diagonal (touches bottom & right edges),
                #                                      'b' = above (touches top & left edges)
                # diag 2: split along (x0,y1)-(x1,y0): 'a' = below (touches bottom & left), 'b' = above (top & right)
                below_touch = [(x, y - 1), (x + 1, y)] if dg == 1 else [(x, y - 1), (x - 1, y)]
                room_below = all(L.a[ny, nx] == rs or L.b[ny, nx] == rs for nx, ny in below_touch)
                L.diag[y, x] = dg
                if room_below:
                    L.a[y, x], L.b[y, x] = rs, -1
                else:
                    L.a[y, x], L.b[y, x] = -1, rs

    # vertex jitter on some vertices that only touch void/room boundaries
    for gy in range(1, grid):
        for gx in range(1, grid):
            if rng.chance(0.10):
                around = {int(L.a[yy, xx]) for yy in (gy - 1, gy) for xx in (gx - 1, gx)} | \
                         {int(L.b[yy, xx]) for yy in (gy - 1, gy) for xx in (gx - 1, gx)}
                kinds = {L.sectors[s].kind for s in around if s >= 0}
                if kinds <= {'room'} and len(around) >= 2:
                    L.jitter[(gx, gy)] = (rng.randint(-10, 10), rng.randint(-10, 10))

    # player start: centre of room 0 (kept free of diagonals by construction: centre cell)
    x0, y0, x1, y1 = rooms[0]
    cx, cy = (x0 + x1) // 2, (y0 + y1) // 2
    if L.a[cy, cx] < 0:
        cx, cy = x0 + 1, y0 + 1
    L.things.insert(0, (cx * CELL + 32, cy * CELL + 32, 90, 1, 7))
    # make sure at least one tagged sector exists (reference precondition)
    if next_tag[0] == 1:
        L.sectors[room_sec[-1]].tag = 1
        next_tag[0] = 2
        L.line_special[('anytag',)] = (88, 1)
    L.max_tag = next_tag[0] - 1
    return L


def kat_level():
    """E1M2: two axis-aligned rooms joined by an opening; exact, jitter-free geometry for analytic KATs.
    Room A: cells x 2..6, y 2..6 (256x256 units) floor 0 ceil 128 light 255;
    room B: cells x 6..9, y 3..5 floor 16 ceil 112 light 160, tagged (lift) so an object exists."""
    L = GridLevel(12, 9)
    a = L.add_sector(0, 128, 'FLOOR0_1', 'CEIL3_5', 255, wall='STARTAN3')
    b = L.add_sector(16, 112, 'FLAT14', 'TLITE6_4', 160, tag=1, wall='BROWN1')
    L.sectors[b].lower = 'STEP1'
    L.sectors[b].upper = 'LITE3'
    L.fill(2, 2, 6, 6, a)
    L.fill(6, 3, 9, 5, b)
    L.things.append((4 * CELL, 4 * CELL, 0, 1, 7))
    L.things.append((7 * CELL + 32, 4 * CELL, 180, 2035, 7))
    L.line_special[('anytag',)] = (88, 1)
    L.max_tag = 1
    return L


# --------------------------------------------------------------------------------------
# grid -> linedefs/sidedefs
# --------------------------------------------------------------------------------------
def cell_halves(L, x, y):
    """Return list of (sector, polygon[(gx,gy)...] CCW in y-up coords) for a cell."""
    a, b, dg = int(L.a[y, x]), int(L.b[y, x]), int(L.diag[y, x])
    p00, p10, p11, p01 = (x, y), (x + 1, y), (x + 1, y + 1), (x, y + 1)
    if dg == 0:
        return [(a, [p00, p10, p11, p01])]
    if dg == 1:   # diagonal p00-p11 : a = below (p00,p10,p11), b = above (p00,p11,p01)
        return [(a, [p00, p10, p11]), (b, [p00, p11, p01])]
    return [(a, [p00, p10, p01]), (b, [p10, p11, p01])]  # diagonal p10-p01


def build_lines(L, rng):
    """Every polygon edge whose two sides differ in sector becomes (part of) a linedef."""
    edges = {}  # (p,q) directed grid edge -> sector on its LEFT (polygon is CCW so interior is on the left)
    for y in range(L.h):
        for x in range(L.w):
            for sec, poly in cell_halves(L, x, y):
                if sec < 0:
                    continue
                n = len(poly)
                for i in range(n):
                    edges[(poly[i], poly[(i + 1) % n])] = sec
    lines = []  # (p, q, right_sector, left_sector or -1)  : right side of p->q holds right_sector
    done = set()
    for (p, q), sec in edges.items():
        if (p, q) in done:
            continue
        other = edges.get((q, p), -1)
        if other == sec:
            continue
        done.add((p, q))
        done.add((q, p))
        # interior of 'sec' is on the LEFT of p->q, so 'sec' is on the RIGHT of q->p.
        if other < 0:
            lines.append((q, p, sec, -1))
        else:
            # two-sided: deterministic orientation; doors want the door sector on the LEFT
            ks, ko = L.sectors[sec].kind, L.sectors[other].kind
            if ks == 'door' and ko != 'door':
                lines.append((p, q, other, sec))
            elif ko == 'door' and ks != 'door':
                lines.append((q, p, sec, other))
            elif sec < other:
                lines.append((q, p, sec, other))
            else:
                lines.append((p, q, other, sec))
    # merge collinear runs (same sectors, same direction, contiguous, no jitter at the joint, length <= 4 cells)
    lines.sort(key=lambda l: (l[2], l[3], l[0][1] - l[1][1] == 0, l[0], l[1]))
    by_start = {}
    for l in lines:
        by_start.setdefault((l[0], l[2], l[3]), []).append(l)
    used = set()
    merged = []
    starts_with_pred = set()
    for l in lines:
        d = (l[1][0] - l[0][0], l[1][1] - l[0][1])
        for m in by_start.get((l[1], l[2], l[3]), []):
            if (m[1][0] - m[0][0], m[1][1] - m[0][1]) == d and l[1] not in L.jitter:
                starts_with_pred.add(m)
    for l in lines:
        if l in used or l in starts_with_pred:
            continue
        p, q = l[0], l[1]
        d = (q[0] - p[0], q[1] - p[1])
        used.add(l)
        length = 1
        while length < 4 and q not in L.jitter:
            nxt = [m for m in by_start.get((q, l[2], l[3]), [])
                   if m not in used and (m[1][0] - m[0][0], m[1][1] - m[0][1]) == d]
            if not nxt:
                break
            used.add(nxt[0])
            q = nxt[0][1]
            length += 1
        merged.append((p, q, l[2], l[3]))
    for l in lines:  # leftovers of long runs (run continued past max length)
        if l not in used:
            p, q = l[0], l[1]
            d = (q[0] - p[0], q[1] - p[1])
            used.add(l)
            length = 1
            while length < 4 and q not in L.jitter:
                nxt = [m for m in by_start.get((q, l[2], l[3]), [])
                       if m not in used and (m[1][0] - m[0][0], m[1][1] - m[0][1]) == d]
                if not nxt:
                    break
                used.add(nxt[0])
                q = nxt[0][1]
                length += 1
            merged.append((p, q, l[2], l[3]))

    vertices, vid = [], {}

    def V(g):
        c = L.vertex(*g)
        if c not in vid:
            vid[c] = len(vertices)
            vertices.append(c)
        return vid[c]

    linedefs, sidedefs = [], []
    lift_line_done = set()
    for (p, q, rs, ls) in merged:
        R = L.sectors[rs]
        special, tag, flags = 0, 0, 0
        xo = rng.choice([0, 0, 0, 8, 16, 24, 100])
        yo = rng.choice([0, 0, 0, 4, 16, -8])
        if ls < 0:
            flags = 1
            if rng.chance(0.2):
                flags |= 0x10  # lower unpegged one-sided
            mid = rng.choice(RICH_WALLS) if RICH_WALLS else R.wall
            if rs in L.scroll_sectors and rng.chance(0.5):
                special = 48
            if rng.chance(0.02):
                mid = 'NOSUCHTX'      # unknown texture: quad skipped with a warning
            if rng.chance(0.03):
                mid = 'SW1COMP'
            if rng.chance(0.03):
                mid = 'SLADRIP2'      # a non-first animation frame
            sidedefs.append((xo, yo, '-', '-', mid, rs))
            linedefs.append((V(p), V(q), flags, special, tag, len(sidedefs) - 1, -1))
            continue
        Ls = L.sectors[ls]
        flags = 4
        if rng.chance(0.3):
            flags |= 0x08
        if rng.chance(0.3):
            flags |= 0x10
        mid_r = mid_l = '-'
        if (min(rs, ls), max(rs, ls)) in L.mid_masked:
            mid_r = mid_l = 'MIDGRATE'
            flags |= 1
        if Ls.kind == 'door' and R.kind != 'door':
            special = 1  # manual door: tag 0, door sector is on the left side
        if Ls.kind == 'lift' or R.kind == 'lift':
            lift = Ls if Ls.kind == 'lift' else R
            special, tag = 88, lift.tag
        # textures needed on each side
        def side_tex(front, back):
            up = back.upper if back.kind in ('door', 'doorway', 'window') else front.upper
            lo = back.lower if back.kind in ('doorway', 'window', 'room') and back.lower != back.wall else front.lower
            return (up if back.ceil < front.ceil or back.kind == 'door' else '-',
                    lo if back.floor > front.floor or back.kind == 'lift' else '-')
        ur, lr = side_tex(R, Ls)
        ul, ll = side_tex(Ls, R)
        if RICH_WALLS:   # every upper / lower piece its own texture
            ur, lr, ul, ll = [rng.choice(RICH_WALLS) if t != '-' else t for t in (ur, lr, ul, ll)]
        if R.kind == 'door':
            mid_r = '-'
        sidedefs.append((xo, yo, ur, lr, mid_r, rs))
        sidedefs.append((0, 0, ul, ll, mid_l, ls))
        linedefs.append((V(p), V(q), flags, special, tag, len(sidedefs) - 2, len(sidedefs) - 1))
    if ('anytag',) in L.line_special and linedefs:
        sp, tg = L.line_special[('anytag',)]
        for i, ld in enumerate(linedefs):
            if ld[6] >= 0 and ld[3] == 0:
                linedefs[i] = ld[:3] + (sp, tg) + ld[5:]
                break
    # one linedef with an unknown special (error path in linedef_to_trigger) and one with special
    # whose tag matches no sector
    for i, ld in enumerate(linedefs):
        if ld[3] == 0 and ld[6] < 0 and i % 37 == 5:
            linedefs[i] = ld[:3] + (999, 0) + ld[5:]
            break
    for i, ld in enumerate(linedefs):
        if ld[3] == 0 and ld[6] < 0 and i % 41 == 7:
            linedefs[i] = ld[:3] + (63, 77) + ld[5:]
            break
    return vertices, linedefs, sidedefs


# --------------------------------------------------------------------------------------
# BSP node builder
# --------------------------------------------------------------------------------------
class Seg:
    __slots__ = ('x1', 'y1', 'x2', 'y2', 'v1', 'v2', 'line', 'side', 'offset', 'sector')

    def __init__(self, x1, y1, x2, y2, v1, v2, line, side, offset, sector):
        self.x1, self.y1, self.x2, self.y2 = x1, y1, x2, y2
        self.v1, self.v2, self.line, self.side, self.offset, self.sector = v1, v2, line, side, offset, sector


class NodeBuilder:
    ON_EPS = 0.51  # distance (map units) under which a point counts as on the partition line

    def __init__(self, vertices, linedefs, sidedefs):
        self.vertices = list(vertices)
        self.vid = {v: i for i, v in enumerate(self.vertices)}
        self.segs_out, self.ssectors, self.nodes = [], [], []
        segs = []
        for i, (v1, v2, flags, special, tag, right, left) in enumerate(linedefs):
            (x1, y1), (x2, y2) = self.vertices[v1], self.vertices[v2]
            segs.append(Seg(x1, y1, x2, y2, v1, v2, i, 0, 0, sidedefs[right][5]))
            if left >= 0:
                segs.append(Seg(x2, y2, x1, y1, v2, v1, i, 1, 0, sidedefs[left][5]))
        self.root = self.build(segs)

    @staticmethod
    def side_of(px, py, dx, dy, x, y):
        """>0: left of the partition direction, <0: right, 0: on (within ON_EPS)."""
        s = dx * (y - py) - dy * (x - px)
        if abs(s) <= NodeBuilder.ON_EPS * math.hypot(dx, dy):
            return 0
        return 1 if s > 0 else -1

    def classify(self, part, seg):
        px, py, dx, dy = part
        a = self.side_of(px, py, dx, dy, seg.x1, seg.y1)
        b = self.side_of(px, py, dx, dy, seg.x2, seg.y2)
        if a == 0 and b == 0:
            same = (seg.x2 - seg.x1) * dx + (seg.y2 - seg.y1) * dy > 0
            return 'R' if same else 'L'
        if a <= 0 and b <= 0:
            return 'R'
        if a >= 0 and b >= 0:
            return 'L'
        return 'S'

    def is_convex(self, segs):
        for a in segs:
            part = (a.x1, a.y1, a.x2 - a.x1, a.y2 - a.y1)
            for b in segs:
                if b is not a and self.classify(part, b) != 'R':
                    return False
        return True

    def choose(self, segs):
        best, best_cost = None, None
        seen = set()
        X1 = np.array([s.x1 for s in segs], np.float64)
        Y1 = np.array([s.y1 for s in segs], np.float64)
        X2 = np.array([s.x2 for s in segs], np.float64)
        Y2 = np.array([s.y2 for s in segs], np.float64)
        for s in segs:
            dx, dy = s.x2 - s.x1, s.y2 - s.y1
            g = math.gcd(abs(dx), abs(dy)) or 1
            key = (dx // g, dy // g, (dx // g) * s.y1 - (dy // g) * s.x1)
            if key in seen:
                continue
            seen.add(key)
            ln = math.hypot(dx, dy)
            s1 = (dx * (Y1 - s.y1) - dy * (X1 - s.x1)) / ln
            s2 = (dx * (Y2 - s.y1) - dy * (X2 - s.x1)) / ln
            a = np.where(np.abs(s1) <= self.ON_EPS, 0, np.sign(s1))
            b = np.where(np.abs(s2) <= self.ON_EPS, 0, np.sign(s2))
            both0 = (a == 0) & (b == 0)
            same = ((X2 - X1) * dx + (Y2 - Y1) * dy) > 0
            right = ((a <= 0) & (b <= 0) & ~both0) | (both0 & same)
            left = ((a >= 0) & (b >= 0) & ~both0) | (both0 & ~same)
            split = ~(right | left)
            nr, nl, ns = int(right.sum()), int(left.sum()), int(split.sum())
            if nl + ns == 0 or nr + ns == 0:
                continue
            cost = ns * 8 + abs(nl - nr) + (0 if (dx == 0 or dy == 0) else 4)
            if best_cost is None or cost < best_cost:
                best, best_cost = (s.x1, s.y1, dx, dy), cost
        return best

    def new_vertex(self, x, y):
        key = (x, y)
        if key not in self.vid:
            self.vid[key] = len(self.vertices)
            self.vertices.append(key)
        return self.vid[key]

    def split(self, part, seg):
        px, py, dx, dy = part
        s1 = dx * (seg.y1 - py) - dy * (seg.x1 - px)
        s2 = dx * (seg.y2 - py) - dy * (seg.x2 - px)
        t = s1 / (s1 - s2)
        x = int(math.floor(seg.x1 + t * (seg.x2 - seg.x1) + 0.5))
        y = int(math.floor(seg.y1 + t * (seg.y2 - seg.y1) + 0.5))
        first_side = 'L' if s1 > 0 else 'R'
        other_side = 'R' if first_side == 'L' else 'L'
        if (x, y) == (seg.x1, seg.y1):
            return {other_side: [seg]}
        if (x, y) == (seg.x2, seg.y2):
            return {first_side: [seg]}
        v = self.new_vertex(x, y)
        d = int(math.floor(math.hypot(x - seg.x1, y - seg.y1) + 0.5))
        a = Seg(seg.x1, seg.y1, x, y, seg.v1, v, seg.line, seg.side, seg.offset, seg.sector)
        b = Seg(x, y, seg.x2, seg.y2, v, seg.v2, seg.line, seg.side, seg.offset + d, seg.sector)
        return {first_side: [a], other_side: [b]}

    @staticmethod
    def bbox(segs):
        xs = [s.x1 for s in segs] + [s.x2 for s in segs]
        ys = [s.y1 for s in segs] + [s.y2 for s in segs]
        return (max(ys), min(ys), min(xs), max(xs))  # top, bottom, left, right

    def build(self, segs, depth=0):
        part = None if self.is_convex(segs) else self.choose(segs)
        if part is None or depth > 60:
            first = len(self.segs_out)
            self.segs_out.extend(segs)
            self.ssectors.append((len(segs), first))
            return 0x8000 | (len(self.ssectors) - 1)
        right, left = [], []
        for s in segs:
            c = self.classify(part, s)
            if c == 'R':
                right.append(s)
            elif c == 'L':
                left.append(s)
            else:
                pieces = self.split(part, s)
                right.extend(pieces.get('R', []))
                left.extend(pieces.get('L', []))
        if not right or not left:   # numerical corner: give up on this set
            first = len(self.segs_out)
            self.segs_out.extend(segs)
            self.ssectors.append((len(segs), first))
            return 0x8000 | (len(self.ssectors) - 1)
        rb, lb = self.bbox(right), self.bbox(left)
        r = self.build(right, depth + 1)
        l = self.build(left, depth + 1)
        self.nodes.append((part[0], part[1], part[2], part[3]) + rb + lb + (r, l))
        return len(self.nodes) - 1


# --------------------------------------------------------------------------------------
# lump packing
# --------------------------------------------------------------------------------------
def name8(s):
    b = s.encode('ascii')
    assert len(b) <= 8, s
    return b + b'\0' * (8 - len(b))


def level_lumps(L, rng):
    vertices, linedefs, sidedefs = build_lines(L, rng)
    nb = NodeBuilder(vertices, linedefs, sidedefs)
    things = b''.join(struct.pack('<hhhHH', *t) for t in L.things)
    ld = b''.join(struct.pack('<HHHHHhh', *l) for l in linedefs)
    sd = b''.join(struct.pack('<hh8s8s8sH', xo, yo, name8(u), name8(lo), name8(m), s)
                  for (xo, yo, u, lo, m, s) in sidedefs)
    vx = b''.join(struct.pack('<hh', x, y) for (x, y) in nb.vertices)
    sg = b''
    for s in nb.segs_out:
        ang = int(round(math.atan2(s.y2 - s.y1, s.x2 - s.x1) / (2 * math.pi) * 65536)) & 0xFFFF
        sg += struct.pack('<HHHHHH', s.v1, s.v2, ang, s.line, s.side, s.offset & 0xFFFF)
    ss = b''.join(struct.pack('<HH', n, f) for (n, f) in nb.ssectors)
    nd = b''.join(struct.pack('<hhhhhhhhhhhhHH', *n) for n in nb.nodes)
    sc = b''.join(struct.pack('<hh8s8shHH', s.floor, s.ceil, name8(s.ftex), name8(s.ctex), s.light, s.special, s.tag)
                  for s in L.sectors)
    assert len(nb.nodes) > 0
    stats = dict(things=len(L.things), linedefs=len(linedefs), sidedefs=len(sidedefs), vertices=len(nb.vertices),
                 segs=len(nb.segs_out), ssectors=len(nb.ssectors), nodes=len(nb.nodes), sectors=len(L.sectors))
    lumps = [('THINGS', things), ('LINEDEFS', ld), ('SIDEDEFS', sd), ('VERTEXES', vx), ('SEGS', sg),
             ('SSECTORS', ss), ('NODES', nd), ('SECTORS', sc), ('REJECT', b'\0' * ((len(L.sectors) ** 2 + 7) // 8)),
             ('BLOCKMAP', struct.pack('<hhHH', 0, 0, 1, 1) + struct.pack('<HHH', 5, 0, 0xFFFF))]
    return lumps, stats


def texture_lump(textures):
    body, offs = b'', []
    base = 4 + 4 * len(textures)
    for (name, w, h, prefs) in textures:
        offs.append(base + len(body))
        body += struct.pack('<8sIHHIH', name8(name), 0, w, h, 0, len(prefs))
        for (ox, oy, pi) in prefs:
            body += struct.pack('<hhHHH', ox, oy, pi, 1, 0)
    return struct.pack('<I', len(textures)) + b''.join(struct.pack('<I', o) for o in offs) + body


def build_wad(seed=1993, verbose=False, specs=None, shapes=False, rich=False):
    """specs (optional): list of (level name, ('gen', seed, grid, rooms) | ('kat',)) replacing the nine default levels.
    shapes: the lump shapes of real IWADs the default file lacks -- TEXTURE2, duplicated lump names (the LAST one is the
    one a name finds: wad/src/archive.rs:85), sprite lumps with paired rotations, textures of many overlapping patches;
    the levels then use the extra textures and things (the default IWAD, its levels and digests are unchanged).
    rich: the texture-rich stand-in (rich_graphics): 320 more wall textures and 192 more flats; every linedef side picks its own
    wall texture, every sector its own floor and ceiling flat."""
    global WALLS, DECOR_TYPES, FLOORS, CEILS, RICH_WALLS
    rng = Rng(seed)
    pals = make_playpal()
    cmaps = make_colormap(pals[0])
    patches, pnames, textures, flats, sprites, textures2 = make_graphics(rng, shapes)
    rich_names = None
    if rich:
        rp, rt, rf = rich_graphics(seed)
        missing = pnames.pop()          # ('MISSING1' stays the last name: it has no lump)
        patches.update(rp)
        pnames += list(rp.keys()) + [missing]
        textures = [(n, w, h, [(ox, oy, pnames.index(pn) if isinstance(pn, str) else (pn if pn < len(pnames) - 1 - len(rp) else len(pnames) - 1)) for ox, oy, pn in prefs])
                    for (n, w, h, prefs) in textures + rt]
        flats.update(rf)
        rich_names = ([t[0] for t in rt], list(rf.keys()))
    lumps = [('PLAYPAL', b''.join(p.tobytes() for p in pals)),
             ('COLORMAP', b''.join(c.tobytes() for c in cmaps))]
    lumps.append(('TEXTURE1', texture_lump(textures)))
    if textures2:
        lumps.append(('TEXTURE2', texture_lump(textures2)))
    lumps.append(('PNAMES', struct.pack('<I', len(pnames)) + b''.join(name8(n) for n in pnames)))
    saved = WALLS, DECOR_TYPES, FLOORS, CEILS
    if rich:
        # sectors take their flats in turn (gen_level draws rng.choice(FLOORS) / rng.choice(CEILS): a list per call would change
        # the number of draws and with it the geometry -- a sequence object whose indexing ignores the draw keeps E1M1's shape)
        class InTurn(list):
            def __init__(self, names):
                super().__init__(names)
                self.at = 0

            def __getitem__(self, _i):
                v = list.__getitem__(self, self.at % len(self))
                self.at += 1
                return v
        half = len(rich_names[1]) // 2
        FLOORS, CEILS = InTurn(rich_names[1][:half]), InTurn(rich_names[1][half:])
        WALLS = InTurn(rich_names[0])
        RICH_WALLS = rich_names[0]
    if shapes:
        WALLS = WALLS + ['OVERLAP3', 'GRATEMIX', 'T2ONLY']
        DECOR_TYPES = DECOR_TYPES + [3004, 3004]
    stats = {}
    specs = specs or [('E1M1', ('gen', seed * 7 + 1, 52, 14)), ('E1M2', ('kat',)), ('E1M3', ('gen', seed * 7 + 3, 40, 9)),
             ('E1M4', ('gen', seed * 7 + 4, 44, 11)), ('E1M5', ('gen', seed * 7 + 5, 48, 12)),
             ('E1M6', ('gen', seed * 7 + 6, 56, 16)), ('E1M7', ('gen', seed * 7 + 7, 36, 8)),
             ('E1M8', ('gen', seed * 7 + 8, 60, 18)), ('E1M9', ('gen', seed * 7 + 9, 42, 10))]
    try:
        for name, spec in specs:
            L = kat_level() if spec[0] == 'kat' else gen_level(spec[1], spec[2], spec[3])
            ll, st = level_lumps(L, Rng(seed + len(lumps)))
            stats[name] = st
            lumps.append((name, b''))
            lumps.extend(ll)
    finally:
        WALLS, DECOR_TYPES, FLOORS, CEILS = saved
        RICH_WALLS = None
    lumps.append(('P_START', b''))
    for n, pix in patches.items():
        if shapes and n in ('WALL02_1', 'STEP1'):   # a decoy under the same name FIRST: the name must find the later lump
            lumps.append((n, encode_picture(np.full_like(pix, 251))))
        lumps.append((n, encode_picture(pix)))
    lumps.append(('P_END', b''))
    lumps.append(('F_START', b''))
    for n, pix in flats.items():
        if shapes and n in ('FLAT14', 'FLOOR4_8'):  # likewise for flats (read in directory order into a map: the later one stays)
            lumps.append((n, np.full(4096, 176, np.uint8).tobytes()))
        lumps.append((n, pix.tobytes()))
    lumps.append(('F_END', b''))
    lumps.append(('S_START', b''))
    for n, (pix, xo, yo) in sprites.items():
        lumps.append((n, encode_picture(pix, xo, yo)))
    lumps.append(('S_END', b''))
    data = b''
    directory = b''
    pos = 12
    for n, d in lumps:
        directory += struct.pack('<ii8s', pos if d else 0, len(d), name8(n))
        data += d
        pos += len(d)
    wad = struct.pack('<4sii', b'IWAD', len(lumps), pos) + data + directory
    if verbose:
        for k, v in stats.items():
            print(k, v, file=sys.stderr)
    return wad, stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--seed', type=int, default=1993)
    ap.add_argument('-v', action='store_true')
    a = ap.parse_args()
    wad, _ = build_wad(a.seed, a.v)
    with open(a.out, 'wb') as f:
        f.write(wad)
    print(a.out, len(wad), hashlib.sha256(wad).hexdigest())


if __name__ == '__main__':
    main()
