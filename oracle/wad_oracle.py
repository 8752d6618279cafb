"""ORACLE (test infrastructure only) -- CPU restatement of rust-doom's level build path.

This file restates, function by function, what the reference does on the CPU once per level:
`wad::{archive,name,level,image,tex,light,meta,visitor}` and `game::{level::Builder,lights}`.
Every function cites the reference file:line it follows.  Arithmetic is IEEE binary32 through
numpy float32 scalars/arrays in the reference's evaluation order (Rust never contracts a*b+c).

It is NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it.  The product's own loader/builder is C++ (rust-doom_amd/csrc/host) and the
HIP kernels; this module exists to check them.

PARITY PIN STATUS: the reference ships no golden vectors for this path except the WadName
known-answer test (wad/src/name.rs:163-190, reproduced in tests/test_oracle_pins.py) and cannot
be compiled here (no rustc/cargo, no GL).  Everything beyond WadName is therefore
"parity unpinned" against the reference binary; it is pinned instead by analytic KATs authored
from the reference source (tests/test_kat_analytic.py) and by committed golden digests.
"""
import ctypes
import ctypes.util
import re
import struct

import numpy as np

try:  # python >= 3.11
    import tomllib as _toml
except ImportError:  # pragma: no cover
    import tomli as _toml

F = np.float32
EPS = F(np.finfo(np.float32).eps)
_libm = ctypes.CDLL(ctypes.util.find_library('m'))
_libm.sinf.restype = ctypes.c_float
_libm.sinf.argtypes = [ctypes.c_float]


class WadError(Exception):
    pass


# ------------------------------------------------------------------------------------------------
# wad/src/name.rs
# ------------------------------------------------------------------------------------------------
_VALID = set(b'ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_-[]%\\')


def wad_name(value):
    """WadName::from_bytes (wad/src/name.rs:41-75): upper-case, stop at NUL, reject other bytes,
    error if longer than 8 without an embedded NUL.  Returns 8 bytes."""
    if isinstance(value, str):
        value = value.encode('utf-8')
    name = bytearray(8)
    nulled = False
    for i, src in enumerate(value[:8]):
        if src >= 0x80:
            raise WadError('invalid byte in wad name')
        up = src - 32 if 97 <= src <= 122 else src
        if up == 0:
            nulled = True
            break
        if up not in _VALID:
            raise WadError('invalid byte in wad name')
        name[i] = up
    if not (nulled or len(value) <= 8):
        raise WadError('wad name too long')
    return bytes(name)


def name_push(name, byte):
    """WadName::push (wad/src/name.rs:17-39). Returns new name or None on error."""
    up = byte - 32 if 97 <= byte <= 122 else byte
    if up not in _VALID:
        return None
    b = bytearray(name)
    for i in range(8):
        if b[i] == 0:
            b[i] = up
            return bytes(b)
    return None


def is_untextured(name):  # wad/src/util.rs:4-6
    return name[0] == 0x2D and name[1] == 0


def is_sky_flat(name):  # wad/src/util.rs:8-10
    return name == b'F_SKY1\0\0'


def from_wad_height(x):  # wad/src/util.rs:12-14
    return F(x) / F(100.0)


def to_wad_height(x):  # wad/src/util.rs:16-18
    return x * F(100.0)


def from_wad_coords(x, y):  # wad/src/util.rs:20-22
    return (-from_wad_height(y), -from_wad_height(x))


def parse_child_id(cid):  # wad/src/util.rs:24-26
    return cid & 0x7FFF, (cid & 0x8000) != 0


# ------------------------------------------------------------------------------------------------
# wad/src/meta.rs
# ------------------------------------------------------------------------------------------------
class Metadata:
    """WadMetadata::from_file (wad/src/meta.rs:129-154)."""

    def __init__(self, path):
        with open(path, 'rb') as f:
            doc = _toml.load(f)
        self.sky = [dict(texture_name=wad_name(s['texture_name']), level_pattern=re.compile(s['level_pattern']),
                         tiled_band_size=F(s['tiled_band_size'])) for s in doc['sky']]
        self.anim_flats = [[wad_name(n) for n in a] for a in doc['animations']['flats']]
        self.anim_walls = [[wad_name(n) for n in a] for a in doc['animations']['walls']]
        self.things = []
        for cat in ('decorations', 'weapons', 'powerups', 'artifacts', 'ammo', 'keys', 'monsters'):
            for t in doc['things'][cat]:
                self.things.append(dict(thing_type=t['thing_type'], sprite=wad_name(t['sprite']),
                                        sequence=t['sequence'], hanging=t['hanging'], radius=t['radius']))
        self.linedef = {}
        for ld in doc.get('linedef', []):
            self.linedef[ld['special_type']] = ld  # IndexMap collect: last value wins (meta.rs:245-257)

    def find_thing(self, thing_type):  # meta.rs:173-206 (category order preserved in self.things)
        for t in self.things:
            if t['thing_type'] == thing_type:
                return t
        return None

    def sky_for(self, name):  # meta.rs:156-171
        text = name.decode('ascii')
        for s in self.sky:
            if s['level_pattern'].search(text):
                return s
        return self.sky[0] if self.sky else None


# ------------------------------------------------------------------------------------------------
# wad/src/archive.rs
# ------------------------------------------------------------------------------------------------
class Archive:
    def __init__(self, wad_path, meta_path):
        with open(wad_path, 'rb') as f:
            self.data = f.read()
        if len(self.data) < 12:
            raise WadError('bad wad header')
        ident, num_lumps, table = struct.unpack_from('<4sii', self.data, 0)
        if ident != b'IWAD':  # archive.rs:69-72
            raise WadError('bad wad header identifier')
        if num_lumps < 0 or table < 0 or table + 16 * num_lumps > len(self.data):
            raise WadError('bad lump info table')  # archive.rs:78-83: the reads of the table fail (bad_lump_info)
        self.lumps, self.index_map, self.levels = [], {}, []
        for i in range(num_lumps):  # archive.rs:80-98
            pos, size, raw = struct.unpack_from('<ii8s', self.data, table + 16 * i)
            name = wad_name(raw)
            self.index_map[name] = len(self.lumps)  # last duplicate wins
            self.lumps.append((name, pos, size))
            if name == b'THINGS\0\0':
                assert i > 0
                self.levels.append(i - 1)
        self.meta = Metadata(meta_path)

    def _range(self, index):
        """LumpReader::read (archive.rs:244-257): a lump is sought and read when asked for -- an entry that points outside
        the file is an I/O error then, and only then (file_pos / size are i32 in the file, taken as unsigned)"""
        name, pos, size = self.lumps[index]
        pos, size = pos & 0xFFFFFFFF, size & 0xFFFFFFFF
        if size > 0 and pos + size > len(self.data):
            raise WadError('reading lump %r failed: outside the file' % (name,))
        return (0 if size == 0 else pos), size

    def lump_bytes(self, index):
        pos, size = self._range(index)
        return self.data[pos:pos + size]

    def named(self, name):
        i = self.index_map.get(wad_name(name))
        return None if i is None else i

    def required(self, name):
        i = self.named(name)
        if i is None:
            raise WadError('missing required lump %r' % (name,))
        return i

    def decode_vec(self, index, fmt):  # archive.rs:172-190
        name = self.lumps[index][0]
        pos, size = self._range(index)
        st = struct.Struct(fmt)
        if not (size > 0 and size % st.size == 0):
            raise WadError('bad lump size %s %d %d' % (name, size, st.size))
        return [st.unpack_from(self.data, pos + k * st.size) for k in range(size // st.size)]

    def level_name(self, level_index):
        return self.lumps[self.levels[level_index]][0]


# ------------------------------------------------------------------------------------------------
# wad/src/level.rs + types.rs
# ------------------------------------------------------------------------------------------------
class Level:
    def __init__(self, wad, index):  # level.rs:34-81
        s = wad.levels[index]
        self.things = wad.decode_vec(s + 1, '<hhhHH')
        self.linedefs = wad.decode_vec(s + 2, '<HHHHHhh')
        self.vertices = wad.decode_vec(s + 4, '<hh')
        self.segs = wad.decode_vec(s + 5, '<HHHHHH')
        self.subsectors = wad.decode_vec(s + 6, '<HH')
        self.nodes = wad.decode_vec(s + 7, '<hhhhhhhhhhhhHH')
        sd = wad.decode_vec(s + 3, '<hh8s8s8sH')
        self.sidedefs = [(a, b, wad_name(u), wad_name(lo), wad_name(m), sec) for (a, b, u, lo, m, sec) in sd]
        sc = wad.decode_vec(s + 8, '<hh8s8shHH')
        self.sectors = [(f, c, wad_name(ft), wad_name(ct), li, ty, tg) for (f, c, ft, ct, li, ty, tg) in sc]

    # linedef tuple: (start, end, flags, special, tag, right, left)
    # sidedef tuple: (xoff, yoff, upper, lower, middle, sector)
    # sector tuple: (floor, ceil, ftex, ctex, light, type, tag)
    # seg tuple: (v1, v2, angle, linedef, direction, offset)
    def vertex(self, vid):  # level.rs:83-87
        if vid < len(self.vertices):
            x, y = self.vertices[vid]
            return from_wad_coords(x, y)
        return None

    def side(self, idx):  # level.rs:139-151
        if idx == -1 or idx < 0 or idx >= len(self.sidedefs):
            return None
        return idx

    def seg_linedef(self, seg):
        return seg[3] if seg[3] < len(self.linedefs) else None

    def seg_sidedef(self, seg):  # level.rs:101-109
        li = self.seg_linedef(seg)
        if li is None:
            return None
        ld = self.linedefs[li]
        return self.side(ld[5]) if seg[4] == 0 else self.side(ld[6])

    def seg_back_sidedef(self, seg):  # level.rs:111-119
        li = self.seg_linedef(seg)
        if li is None:
            return None
        ld = self.linedefs[li]
        return self.side(ld[5]) if seg[4] == 1 else self.side(ld[6])

    def side_sector(self, side):
        if side is None:
            return None
        s = self.sidedefs[side][5]
        return s if s < len(self.sectors) else None

    def seg_vertices(self, seg):
        a, b = self.vertex(seg[0]), self.vertex(seg[1])
        return None if a is None or b is None else (a, b)

    def adjacent_sectors(self, sector_id):  # level.rs:230-258
        for ld in self.linedefs:
            l, r = self.side(ld[6]), self.side(ld[5])
            if l is None or r is None:
                continue
            ls, rs = self.sidedefs[l][5], self.sidedefs[r][5]
            if ls == sector_id:
                adj = rs
            elif rs == sector_id:
                adj = ls
            else:
                continue
            if adj < len(self.sectors):
                yield adj

    def sector_min_light(self, sector_id):  # level.rs:178-182
        m = self.sectors[sector_id][4]
        for a in self.adjacent_sectors(sector_id):
            m = min(m, self.sectors[a][4])
        return m

    def neighbour_heights(self, sector_id):  # level.rs:184-212
        of_floor = self.sectors[sector_id][0]
        h = None
        for a in self.adjacent_sectors(sector_id):
            floor, ceil = self.sectors[a][0], self.sectors[a][1]
            if h is None:
                h = dict(lowest_floor=floor, highest_floor=floor, lowest_ceiling=ceil, highest_ceiling=ceil,
                         next_floor=floor if floor > of_floor else None)
            else:
                nf = h['next_floor']
                if floor > of_floor:
                    nf = floor if nf is None else min(nf, floor)
                h = dict(lowest_floor=min(h['lowest_floor'], floor), highest_floor=max(h['highest_floor'], floor),
                         lowest_ceiling=min(h['lowest_ceiling'], ceil), highest_ceiling=max(h['highest_ceiling'], ceil),
                         next_floor=nf)
        return h


# ------------------------------------------------------------------------------------------------
# wad/src/image.rs
# ------------------------------------------------------------------------------------------------
MAX_IMAGE_SIZE = 4096


class Image:
    def __init__(self, width, height, fill=0xFF00):  # image.rs:19-32
        if width > MAX_IMAGE_SIZE or height > MAX_IMAGE_SIZE:
            raise WadError('image too large')
        self.width, self.height = width, height
        self.x_offset = self.y_offset = 0
        self.pixels = np.full((height, width), fill, np.uint16)

    @staticmethod
    def from_buffer(buf):  # image.rs:39-169
        if len(buf) < 8:
            raise WadError('image header')
        w, h, xo, yo = struct.unpack_from('<HHhh', buf, 0)
        img = Image(w, h, 0xFFFF)
        img.x_offset, img.y_offset = xo, yo
        for col in range(w):
            if 8 + 4 * col + 4 > len(buf):
                raise WadError('unfinished image column')
            off = struct.unpack_from('<I', buf, 8 + 4 * col)[0]
            if off >= len(buf):
                raise WadError('invalid column offset')
            p = off
            while True:
                if p >= len(buf):
                    raise WadError('unfinished column')
                row = buf[p]
                p += 1
                if row == 255:
                    break
                if p >= len(buf):
                    raise WadError('missing run length')
                n = buf[p]
                p += 1
                if row + n > h:
                    raise WadError('run too big')
                if p >= len(buf):
                    raise WadError('pad 1')
                p += 1
                if len(buf) - p < n:
                    raise WadError('underrun')
                img.pixels[row:row + n, col] = np.frombuffer(buf, np.uint8, n, p)
                p += n
                if p >= len(buf):
                    raise WadError('pad 2')
                p += 1
        return img

    def blit(self, src, ox, oy, ignore_transparency):  # image.rs:171-252
        if ox >= self.width or oy >= self.height:
            return
        y_start = -oy if oy < 0 else 0
        x_start = -ox if ox < 0 else 0
        y_end = src.height if self.height > src.height + oy else self.height - oy
        x_end = src.width if self.width > src.width + ox else self.width - ox
        if x_end <= x_start or y_end <= y_start:
            return  # reference would underflow usize here; defined as "nothing copied"
        s = src.pixels[y_start:y_end, x_start:x_end]
        d = self.pixels[y_start + oy:y_end + oy, x_start + ox:x_end + ox]
        if ignore_transparency:
            d[...] = s
        else:  # image.rs:243-249: copy where the source's bit 15 is clear
            blend = (np.uint16(0) - (s >> 15)).astype(np.uint16)
            d[...] = (s & ~blend) | (d & blend)


# ------------------------------------------------------------------------------------------------
# wad/src/tex.rs
# ------------------------------------------------------------------------------------------------
def next_pow2(x):  # tex.rs:348-354
    p = 1
    while p < x:
        p *= 2
    return p


class Bounds:
    def __init__(self, pos, size, num_frames, row_height):
        self.pos, self.size, self.num_frames, self.row_height = pos, size, num_frames, row_height


class TextureDirectory:
    def __init__(self, wad):  # tex.rs:53-107
        pp = wad.lump_bytes(wad.required('PLAYPAL'))
        cm = wad.lump_bytes(wad.required('COLORMAP'))
        if not (len(pp) > 0 and len(pp) % 768 == 0 and len(cm) > 0 and len(cm) % 256 == 0):
            raise WadError('bad palette lumps')
        self.palettes = np.frombuffer(pp, np.uint8).reshape(-1, 768)
        self.colormaps = np.frombuffer(cm, np.uint8).reshape(-1, 256)
        self.patches = self._read_patches(wad)
        self.textures = {}
        for lump_name in ('TEXTURE1', 'TEXTURE2'):
            i = wad.named(lump_name)
            if i is not None:
                self._read_textures(wad.lump_bytes(i))
        # flats: tex.rs:594-606
        self.flats = {}
        for i in range(wad.required('F_START'), wad.required('F_END')):
            name, pos, size = wad.lumps[i]
            if size == 0:
                continue
            self.flats[name] = np.frombuffer(wad.lump_bytes(i), np.uint8)
        # sprites: tex.rs:475-497 (inserted into the same `textures` map)
        for i in range(wad.required('S_START') + 1, wad.required('S_END')):
            try:
                self.textures[wad.lumps[i][0]] = Image.from_buffer(wad.lump_bytes(i))
            except WadError:
                continue
        self.animated_walls = wad.meta.anim_walls
        self.animated_flats = wad.meta.anim_flats

    def _read_patches(self, wad):  # tex.rs:358-410
        buf = wad.lump_bytes(wad.required('PNAMES'))
        n = struct.unpack_from('<I', buf, 0)[0]
        out = []
        for i in range(n):
            if 4 + 8 * i + 8 > len(buf):
                continue
            try:
                name = wad_name(buf[4 + 8 * i:12 + 8 * i])
            except WadError:
                continue
            li = wad.index_map.get(name)
            if li is None:
                out.append((name, None))
                continue
            try:
                out.append((name, Image.from_buffer(wad.lump_bytes(li))))
            except WadError:
                out.append((name, None))
        return out

    def _read_textures(self, buf):  # tex.rs:499-592
        if len(buf) < 4:  # read_u32 fails: CorruptWad("Missing number of textures") (tex.rs:505-508) -- an empty TEXTURE2, say
            raise WadError('missing number of textures')
        n = struct.unpack_from('<I', buf, 0)[0]
        rest = buf[4:]
        if not (n * 4 < len(rest)):
            raise WadError('textures lump too small')
        for i in range(n):
            off = struct.unpack_from('<I', rest, 4 * i)[0]
            if off >= len(buf):
                raise WadError('texture offset')
            if off + 22 > len(buf):
                continue
            raw, masked, w, h, coldir, npatch = struct.unpack_from('<8sIHHIH', buf, off)
            try:
                name = wad_name(raw)
                img = Image(w, h)
            except WadError:
                continue
            p = off + 22
            for k in range(npatch):
                if p + 10 > len(buf):
                    continue
                ox, oy, pi, _, _ = struct.unpack_from('<hhHHH', buf, p)
                p += 10
                oy = 0 if oy <= 0 else oy  # tex.rs:560-567
                if pi < len(self.patches) and self.patches[pi][1] is not None:
                    img.blit(self.patches[pi][1], ox, oy, k == 0)  # tex.rs:570
            self.textures[name] = img

    def texture(self, name):
        return self.textures.get(name)

    def flat(self, name):
        return self.flats.get(name)

    def build_palette_texture(self, palette, cm_start, cm_end):  # tex.rs:137-166
        n = cm_end - cm_start
        mapped = np.zeros((n * 256, 3), np.uint8)
        pal = self.palettes[palette].reshape(256, 3)
        for i in range(cm_start, min(cm_end, len(self.colormaps))):
            mapped[i * 256:(i + 1) * 256] = pal[self.colormaps[i]]
        return mapped.reshape(-1)

    @staticmethod
    def _ordered_entries(animations, lookup, names):  # tex.rs:421-473
        first = {}
        for name in names:
            frames = None
            for anim in animations:
                if name in anim:
                    frames = anim
                    break
            first[frames[0] if frames else name] = frames  # IndexMap: position of first insert kept
        entries = []
        for name, frames in first.items():
            if frames is not None:
                for off, fname in enumerate(frames):
                    img = lookup(fname)
                    if img is not None:
                        entries.append((fname, img, off, len(frames)))
            else:
                img = lookup(name)
                if img is not None:
                    entries.append((name, img, 0, 1))
        return entries

    def build_texture_atlas(self, names):  # tex.rs:168-271
        entries = self._ordered_entries(self.animated_walls, self.texture, names)
        if not entries:
            return np.zeros((0, 0), np.uint16), {}
        max_w = max(e[1].width for e in entries)
        num_pixels = sum(e[1].width * e[1].height for e in entries)
        size = [min(128, next_pow2(max_w)), 128]

        def next_size(size):  # tex.rs:186-200
            while True:
                if size[0] <= size[1]:
                    if size[0] == 4096:
                        raise WadError('could not fit wall atlas')
                    size[0] *= 2
                    size[1] = 128
                else:
                    size[1] *= 2
                if size[0] * size[1] >= num_pixels:
                    break

        next_size(size)
        transposed = False
        while True:
            positions = []
            ox = oy = row_h = 0
            failed = False
            for (_, img, _, _) in entries:
                w, h = img.width, img.height
                if ox + w > size[0]:
                    ox = 0
                    oy += row_h
                    row_h = 0
                if h > row_h:
                    row_h = h
                if oy + h > size[1]:
                    failed = True
                    break
                positions.append((ox, oy, row_h))
                ox += w
            if not failed:
                break
            size = [size[1], size[0]]
            transposed = not transposed
            if transposed and size[0] != size[1]:
                continue
            transposed = False
            next_size(size)
        atlas = Image(size[0], size[1])
        bounds = {}
        for i, (name, img, frame_off, nframes) in enumerate(entries):
            atlas.blit(img, positions[i][0], positions[i][1], True)
            if frame_off > i:  # tex.rs:258-261 indexes with usize: the reference panics; a negative Python index must not wrap
                raise WadError('corrupt WAD: animated texture is missing its first frames')
            px, py, rh = positions[i - frame_off]  # tex.rs:258-261
            bounds[name] = Bounds((F(px), F(py)), (F(img.width), F(img.height)), nframes, rh)
        return atlas.pixels, bounds

    def build_flat_atlas(self, names):  # tex.rs:273-333
        entries = self._ordered_entries(self.animated_flats, self.flat, names)
        n = len(entries)
        width = next_pow2(int(np.ceil(np.sqrt(np.float64(n)))) * 64)
        per_row = width // 64
        rows = int(np.ceil(np.float64(n) / np.float64(per_row))) if per_row else 0
        height = next_pow2(rows * 64)
        data = np.full((height, width), 255, np.uint8)
        bounds = {}
        row = col = 0
        anim_start = (F(0), F(0))
        for (name, img, frame_off, nframes) in entries:
            ox, oy = col * 64, row * 64
            if frame_off == 0:
                anim_start = (F(ox), F(oy))
            bounds[name] = Bounds(anim_start, (F(64), F(64)), nframes, 64)
            data[oy:oy + 64, ox:ox + 64] = img[:4096].reshape(64, 64)
            col += 1
            if col == per_row:
                col = 0
                row += 1
        return data, bounds


# ------------------------------------------------------------------------------------------------
# wad/src/light.rs
# ------------------------------------------------------------------------------------------------
GLOW, RANDOM, ALTERNATE = 0, 1, 2


class LightInfo:
    __slots__ = ('level', 'effect')

    def __init__(self, level, effect=None):
        self.level, self.effect = level, effect  # effect = (alt_level, speed, duration, sync, kind)

    def __eq__(self, o):
        return self.level == o.level and self.effect == o.effect


def light_to_f32(level):  # light.rs:113-115
    return F(level >> 3) / F(31.0)


def new_light(level, sector_id):  # light.rs:27-79
    sec = level.sectors[sector_id]
    base = light_to_f32(sec[4])
    st = sec[5]
    if st not in (1, 2, 4, 13, 3, 12, 8, 17):
        return LightInfo(base)
    alt = light_to_f32(level.sector_min_light(sector_id))
    if abs(alt - base) < EPS:
        return LightInfo(base)
    if st in (12, 13, 8):
        sync = F(0.0)
    else:  # light.rs:109-111
        sync = F(float((sector_id * 1664525 + 1013904223) & 0xFFFF)) / F(15.0)
    if st == 1:
        kind, speed, dur = RANDOM, F(20.0), F(0.06)
    elif st == 17:
        kind, speed, dur = RANDOM, F(8.0), F(0.5)
    elif st in (3, 12):
        kind, speed, dur = ALTERNATE, F(1.0), F(0.85)
    elif st in (2, 4, 13):
        kind, speed, dur = ALTERNATE, F(2.0), F(0.7)
    else:
        kind, speed, dur = GLOW, F(0.5), F(0.0)
    return LightInfo(base, (alt, speed, dur, sync, kind))


def clamp01(x):
    return F(1.0) if x > F(1.0) else (F(0.0) if x < F(0.0) else x)


def with_contrast(info, brighten):  # light.rs:81-91
    c = F(2.0) / F(31.0) if brighten else F(-2.0) / F(31.0)
    return LightInfo(clamp01(info.level + c), info.effect)


# ------------------------------------------------------------------------------------------------
# math/src/line.rs
# ------------------------------------------------------------------------------------------------
def magnitude(x, y):
    return np.sqrt(x * x + y * y)


def normalize_or_zero(x, y):  # math/src/lib.rs:40-42
    m = max(magnitude(x, y), EPS)
    return x / m, y / m


class Line2f:
    __slots__ = ('ox', 'oy', 'dx', 'dy', 'length')

    @staticmethod
    def from_two_points(o, t):  # line.rs:12-33
        ln = Line2f()
        dx, dy = t[0] - o[0], t[1] - o[1]
        length = magnitude(dx, dy)
        ln.ox, ln.oy = o
        if abs(length) >= F(1e-16):
            ln.dx, ln.dy, ln.length = dx / length, dy / length, length
        else:
            ln.dx, ln.dy, ln.length = F(0.0), F(0.0), F(0.0)
        return ln

    def inverted(self):  # line.rs:35-41
        ln = Line2f()
        ln.ox, ln.oy, ln.dx, ln.dy, ln.length = self.ox, self.oy, -self.dx, -self.dy, self.length
        return ln

    def signed_distance(self, p):  # line.rs:43-45
        return (p[0] * self.dy - p[1] * self.dx) + (self.dx * self.oy - self.dy * self.ox)

    def intersect_point(self, other):  # line.rs:68-84
        den = self.dx * other.dy - self.dy * other.dx
        if abs(den) < F(1e-16):
            return None
        off = ((other.ox - self.ox) * other.dy - (other.oy - self.oy) * other.dx) / den
        return (self.ox + self.dx * off, self.oy + self.dy * off)


def _sd_all(lines, p):
    """signed_distance of p to every line of an (n,4) float32 array [ox,oy,dx,dy] (same op order)."""
    ox, oy, dx, dy = lines[:, 0], lines[:, 1], lines[:, 2], lines[:, 3]
    return (p[0] * dy - p[1] * dx) + (dx * oy - dy * ox)


# ------------------------------------------------------------------------------------------------
# wad/src/visitor.rs : LevelAnalysis
# ------------------------------------------------------------------------------------------------
HEIGHT_REFS = ('LowestFloor', 'NextFloor', 'HighestFloor', 'LowestCeiling', 'HighestCeiling', 'Floor', 'Ceiling')


class DynamicSectorInfo:
    def __init__(self):
        self.floor_id = self.ceiling_id = 0
        self.neighbour_heights = None
        self.floor_range = self.ceiling_range = None


def _to_height(hdef, sector, heights):  # visitor.rs:273-286
    to = hdef['to']
    if to == 'LowestFloor':
        base = heights['lowest_floor']
    elif to == 'NextFloor':
        base = heights['next_floor']
        if base is None:
            return None
    elif to == 'HighestFloor':
        base = heights['highest_floor']
    elif to == 'LowestCeiling':
        base = heights['lowest_ceiling']
    elif to == 'HighestCeiling':
        base = heights['highest_ceiling']
    elif to == 'Floor':
        base = sector[0]
    elif to == 'Ceiling':
        base = sector[1]
    else:
        raise WadError('bad height ref')
    return i16(base + hdef.get('off', 0))


def _option_to_heights(edef, sector, heights):  # visitor.rs:288-301
    if edef is None:
        return None, None
    first = _to_height(edef['first'], sector, heights)
    second = _to_height(edef['second'], sector, heights) if edef.get('second') is not None else None
    return first, second


def _merge_range(rng, current, coords):  # visitor.rs:247-261
    for c in coords:
        rng = (c, c) if rng is None else (min(rng[0], c), max(rng[1], c))
    if rng is not None:
        rng = (min(rng[0], current), max(rng[1], current))
    return rng


class LevelAnalysis:
    def __init__(self, level, meta):  # visitor.rs:323-444
        self.dynamic_info = {}
        self.num_objects = 0
        self.num_triggers = 0
        tags = sorted((s[6], i) for i, s in enumerate(level.sectors) if s[6] > 0)
        if not tags:
            return
        first_index = {}
        for i, (tag, _) in enumerate(tags):
            first_index.setdefault(tag, i)
        next_id = [1]
        for ld in level.linedefs:
            special = ld[3]
            if special == 0:
                continue
            if level.vertex(ld[0]) is None or level.vertex(ld[1]) is None:
                continue
            m = meta.linedef.get(special)
            move = m.get('move') if m is not None else None
            self.num_triggers += 1
            tag = ld[4]
            if tag == 0:
                left = level.side(ld[6])
                if left is not None:
                    sid = level.sidedefs[left][5]
                    if sid >= len(level.sectors):
                        continue  # the reference indexes level.sectors[sid] (visitor.rs:175, a panic): defined as warn-and-skip
                    self._update(self.dynamic_info.setdefault(sid, DynamicSectorInfo()), next_id, level, sid, move)
                continue
            if tag in first_index:
                for (t, sid) in tags[first_index[tag]:]:
                    if t != tag:
                        break
                    self._update(self.dynamic_info.setdefault(sid, DynamicSectorInfo()), next_id, level, sid, move)
        self.num_objects = next_id[0]

    @staticmethod
    def _update(info, next_id, level, sid, move):  # visitor.rs:168-244
        if move is None:
            return
        sector = level.sectors[sid]
        heights = info.neighbour_heights
        if heights is None:
            heights = level.neighbour_heights(sid)
            if heights is None:
                return
            info.neighbour_heights = heights
        ff, sf = _option_to_heights(move.get('floor'), sector, heights)
        fc, sc = _option_to_heights(move.get('ceiling'), sector, heights)
        info.floor_range = _merge_range(info.floor_range, sector[0], [c for c in (ff, sf) if c is not None])
        info.ceiling_range = _merge_range(info.ceiling_range, sector[1], [c for c in (fc, sc) if c is not None])
        if info.ceiling_range is not None and info.ceiling_id == 0:
            info.ceiling_id = next_id[0]
            next_id[0] += 1
        if info.floor_range is not None and info.floor_id == 0:
            info.floor_id = next_id[0]
            next_id[0] += 1


# ------------------------------------------------------------------------------------------------
# wad/src/visitor.rs : LevelWalker
# ------------------------------------------------------------------------------------------------
BSP_TOLERANCE = F(1e-3)
SEG_TOLERANCE = F(0.1)
POLY_BIAS = F(0.64) * F(3e-4)
PEG_TOP, PEG_BOTTOM, PEG_BOTTOM_LOWER, PEG_TOP_FLOAT, PEG_BOTTOM_FLOAT = range(5)


class LevelVisitor:
    """Mirror of `trait LevelVisitor` (wad/src/visitor.rs:65-127): every callback defaults to a no-op."""

    def visit_wall_quad(self, quad): pass
    def visit_floor_poly(self, poly): pass
    def visit_ceil_poly(self, poly): pass
    def visit_floor_sky_poly(self, poly): pass
    def visit_ceil_sky_poly(self, poly): pass
    def visit_sky_quad(self, quad): pass
    def visit_marker(self, pos, yaw, marker): pass
    def visit_decor(self, decor): pass
    def visit_bsp_root(self, line): pass
    def visit_bsp_node(self, line, branch): pass
    def visit_bsp_leaf(self, branch): pass
    def visit_bsp_leaf_end(self): pass
    def visit_bsp_node_end(self): pass


def polygon_center(pts):  # visitor.rs:1184-1190
    cx, cy = F(0.0), F(0.0)
    for p in pts:
        cx, cy = cx + p[0], cy + p[1]
    n = F(len(pts))
    return cx / n, cy / n


def _poly_less(a, b, c):
    """The comparator of visitor.rs:1195-1224; True iff it returns Ordering::Less."""
    acx, acy = a[0] - c[0], a[1] - c[1]
    bcx, bcy = b[0] - c[0], b[1] - c[1]
    if acx >= 0 and bcx < 0:
        return True
    if acx < 0 and bcx >= 0:
        return False
    if acx == 0 and bcx == 0:
        if acy >= 0 or bcy >= 0:
            return bool(a[1] > b[1])
        return bool(b[1] > a[1])
    return bool(acx * bcy - acy * bcx < 0)


def points_to_polygon(points):  # visitor.rs:1192-1259
    """Returns the canonical polygon (possibly empty).  The sort is a left-to-right linear insertion
    sort: the comparator is not a strict weak order (never Equal), so the algorithm is pinned here
    and in the product (DESIGN.md 'polygon sort'); it equals Rust's sort_unstable_by whenever no two
    points are comparator-ambiguous."""
    pts = list(points)
    if len(pts) < 2:
        return []
    c = polygon_center(pts)
    for i in range(1, len(pts)):
        j = i
        while j > 0 and _poly_less(pts[j], pts[j - 1], c):
            pts[j], pts[j - 1] = pts[j - 1], pts[j]
            j -= 1
    simplified = [pts[0]]
    cur = pts[1]
    area = F(0.0)
    for i in range(2, len(pts)):
        nxt = pts[i]
        prev = simplified[-1]
        new_area = ((nxt[0] - cur[0]) * (cur[1] - prev[1]) - (nxt[1] - cur[1]) * (cur[0] - prev[0])) * F(0.5)
        if new_area >= 0:
            if area + new_area > F(1.024e-5):
                area = F(0.0)
                simplified.append(cur)
            else:
                area = area + new_area
        cur = nxt
    simplified.append(pts[-1])
    if len(simplified) < 3:
        return []
    while len(simplified) > 1 and magnitude(simplified[0][0] - simplified[-1][0],
                                            simplified[0][1] - simplified[-1][1]) < F(0.0032):
        simplified.pop()
    c = polygon_center(simplified)
    out = []
    for p in simplified:
        nx, ny = normalize_or_zero(p[0] - c[0], p[1] - c[1])
        out.append((p[0] + nx * POLY_BIAS, p[1] + ny * POLY_BIAS))
    return out


def i16(x):
    """WadCoord is i16 in the reference and its integer height arithmetic is plain `+` / `-`: an overflow wraps in a
    release build (and panics in a debug build).  Defined here, as in the product, as the release behaviour."""
    return ((int(x) + 32768) & 0xFFFF) - 32768


def partition_line(node):  # visitor.rs:1150-1155
    return Line2f.from_two_points(from_wad_coords(node[0], node[1]),
                                  from_wad_coords(i16(node[0] + node[2]), i16(node[1] + node[3])))


class LevelWalker:
    def __init__(self, level, analysis, tex, meta, visitor):  # visitor.rs:519-539
        self.level, self.tex, self.meta, self.visitor = level, tex, meta, visitor
        self.dynamic_info = analysis.dynamic_info
        mn, mx = 32767, -32768  # visitor.rs:1173-1182
        for s in level.sectors:
            mn, mx = min(mn, s[0]), max(mx, s[1])
        self.height_range = (i16(mn - 512), i16(mx + 512))
        self.bsp_lines = []
        self.light_cache = {}

    def walk(self):  # visitor.rs:541-555
        if not self.level.nodes:
            return
        root = self.level.nodes[-1]
        part = partition_line(root)
        self.visitor.visit_bsp_root(part)
        self.children(root, part)
        self.visitor.visit_bsp_node_end()
        self.things()

    def sector_info(self, sid):  # visitor.rs:569-588
        s = self.level.sectors[sid]
        fr, cr = (s[0], s[0]), (s[1], s[1])
        d = self.dynamic_info.get(sid)
        if d is None:
            return (0, 0, fr, cr)
        return (d.floor_id, d.ceiling_id, d.floor_range or fr, d.ceiling_range or cr)

    def node(self, cid, branch):  # visitor.rs:590-609
        idx, leaf = parse_child_id(cid)
        if leaf:
            self.visitor.visit_bsp_leaf(branch)
            self.subsector(idx)
            self.visitor.visit_bsp_leaf_end()
            return
        if idx >= len(self.level.nodes):
            return
        node = self.level.nodes[idx]
        part = partition_line(node)
        self.visitor.visit_bsp_node(part, branch)
        self.children(node, part)
        self.visitor.visit_bsp_node_end()

    def children(self, node, part):  # visitor.rs:611-619
        self.bsp_lines.append(part)
        self.node(node[13], 'Positive')  # left
        self.bsp_lines.pop()
        self.bsp_lines.append(part.inverted())
        self.node(node[12], 'Negative')  # right
        self.bsp_lines.pop()

    def subsector(self, idx):  # visitor.rs:621-709
        lv = self.level
        if idx >= len(lv.subsectors):
            return
        nsegs, first = lv.subsectors[idx]
        if first + nsegs > len(lv.segs):
            return
        segs = lv.segs[first:first + nsegs]
        if not segs:
            return
        sid = lv.side_sector(lv.seg_sidedef(segs[0]))
        if sid is None:
            return
        info = self.sector_info(sid)
        points, seg_lines = [], []
        for seg in segs:
            vs = lv.seg_vertices(seg)
            if vs is None:
                return
            points.append(vs[0])
            points.append(vs[1])
            seg_lines.append(Line2f.from_two_points(vs[0], vs[1]))
            self.seg(sid, info, seg, vs)
        bl = self.bsp_lines
        bsp_arr = np.array([[l.ox, l.oy, l.dx, l.dy] for l in bl], np.float32).reshape(-1, 4)
        seg_arr = np.array([[l.ox, l.oy, l.dx, l.dy] for l in seg_lines], np.float32).reshape(-1, 4)
        for i in range(len(bl) - 1):
            for j in range(i + 1, len(bl)):
                p = bl[i].intersect_point(bl[j])
                if p is None:
                    continue
                if np.all(_sd_all(bsp_arr, p) >= -BSP_TOLERANCE) and np.all(_sd_all(seg_arr, p) <= SEG_TOLERANCE):
                    points.append(p)
        poly = points_to_polygon(points)
        if len(poly) >= 3:
            self.flat_poly(sid, info, poly)

    def light_info(self, sid):  # visitor.rs:1140-1148
        li = self.light_cache.get(sid)
        if li is None:
            li = self.light_cache[sid] = new_light(self.level, sid)
        return li

    def seg(self, sid, info, seg, vertices):  # visitor.rs:711-837
        lv = self.level
        li = lv.seg_linedef(seg)
        if li is None:
            return
        line = lv.linedefs[li]
        side_i = lv.seg_sidedef(seg)
        if side_i is None:
            return
        sidedef = lv.sidedefs[side_i]
        sector = lv.sectors[sid]
        mn, mx = self.height_range
        floor, ceiling = sector[0], sector[1]
        unpeg_lower = (line[2] & 0x10) != 0
        floor_id, ceiling_id, floor_range, ceiling_range = info
        max_height = i16(ceiling_range[1] - floor_range[0])
        back_sid = lv.side_sector(lv.seg_back_sidedef(seg))
        if back_sid is None:
            self.wall_quad(sid, seg, vertices, floor_id if unpeg_lower else ceiling_id,
                           (floor, i16(floor + max_height)) if unpeg_lower else (i16(ceiling - max_height), ceiling),
                           sidedef[4], PEG_BOTTOM if unpeg_lower else PEG_TOP, True)
            if is_sky_flat(sector[3]):
                self.sky_quad(ceiling_id, vertices, (ceiling, mx))
            if is_sky_flat(sector[2]):
                self.sky_quad(floor_id, vertices, (mn, floor))
            return
        back = lv.sectors[back_sid]
        back_floor, back_ceiling = back[0], back[1]
        binfo = self.sector_info(back_sid)
        if is_sky_flat(sector[3]) and not is_sky_flat(back[3]):
            self.sky_quad(ceiling_id, vertices, (ceiling, mx))
        if is_sky_flat(sector[2]) and not is_sky_flat(back[2]):
            self.sky_quad(floor_id, vertices, (mn, floor))
        unpeg_upper = (line[2] & 0x08) != 0
        if binfo[2][1] > floor_range[0]:
            self.wall_quad(sid, seg, vertices, binfo[0], (i16(back_floor - binfo[2][1] + floor_range[0]), back_floor),
                           sidedef[3], PEG_BOTTOM_LOWER if unpeg_lower else PEG_TOP, True)
            fl = back_floor
        else:
            fl = floor
        if back_ceiling < ceiling:
            if not is_sky_flat(back[3]):
                self.wall_quad(sid, seg, vertices, binfo[1], (back_ceiling, ceiling), sidedef[2],
                               PEG_TOP if unpeg_upper else PEG_BOTTOM, True)
            ce = back_ceiling
        else:
            ce = ceiling
        if unpeg_lower:
            peg = PEG_TOP_FLOAT if is_untextured(sidedef[2]) else PEG_BOTTOM
        else:
            peg = PEG_BOTTOM_FLOAT if is_untextured(sidedef[3]) else PEG_TOP
        self.wall_quad(sid, seg, vertices, floor_id if unpeg_lower else ceiling_id, (fl, ce), sidedef[4], peg,
                       (line[2] & 1) != 0)

    def wall_quad(self, sid, seg, vertices, object_id, height_range, texture_name, peg, blocker):  # visitor.rs:839-937
        lv = self.level
        low, high = height_range
        if low >= high:
            return
        if is_untextured(texture_name):
            size = None
        else:
            img = self.tex.texture(texture_name)
            if img is None:
                return
            size = (F(img.width), F(img.height))
        line = lv.linedefs[lv.seg_linedef(seg)]
        sidedef = lv.sidedefs[lv.seg_sidedef(seg)]
        sector = lv.sectors[sid]
        v1, v2 = vertices
        nx, ny = normalize_or_zero(v2[0] - v1[0], v2[1] - v1[1])
        bx, by = nx * POLY_BIAS, ny * POLY_BIAS
        v1 = (v1[0] + (-bx), v1[1] + (-by))
        v2 = (v2[0] + bx, v2[1] + by)
        y_off = sidedef[1]
        if size is not None and peg == PEG_TOP_FLOAT:
            lo, hi = from_wad_height(i16(low + y_off)), from_wad_height(i16(low + i16(int(size[1])) + y_off))
        elif size is not None and peg == PEG_BOTTOM_FLOAT:
            lo, hi = from_wad_height(i16(high + y_off - i16(int(size[1])))), from_wad_height(i16(high + y_off))
        else:
            lo, hi = from_wad_height(low), from_wad_height(high)
        light = self.light_info(sid)
        if light.effect is None:  # visitor.rs:889-901
            if abs(v1[0] - v2[0]) < EPS:
                light = with_contrast(light, True)
            elif abs(v1[1] - v2[1]) < EPS:
                light = with_contrast(light, False)
        height = to_wad_height(hi - lo)
        s1 = F(seg[5]) + F(sidedef[0])
        s2 = s1 + to_wad_height(magnitude(v2[0] - v1[0], v2[1] - v1[1]))
        if size is None or peg == PEG_TOP:
            t1, t2 = height, F(0.0)
        elif peg == PEG_BOTTOM:
            t1, t2 = size[1], size[1] - height
        elif peg == PEG_BOTTOM_LOWER:
            sh = F(i16(sector[1] - sector[0]))
            t1, t2 = size[1] + sh, size[1] - height + sh
        else:
            t1, t2 = size[1], F(0.0)
        t1, t2 = t1 + F(y_off), t2 + F(y_off)
        scroll = F(35.0) if line[3] == 0x30 else F(0.0)
        lo, hi = lo - POLY_BIAS, hi + POLY_BIAS
        self.visitor.visit_wall_quad(dict(object_id=object_id, vertices=(v1, v2), tex_start=(s1, t1), tex_end=(s2, t2),
                                          height_range=(lo, hi), light_info=light, scroll=scroll,
                                          tex_name=texture_name if size is not None else None, blocker=blocker))

    def flat_poly(self, sid, info, poly):  # visitor.rs:939-985
        sector = self.level.sectors[sid]
        light = self.light_info(sid)
        floor_sky, ceil_sky = is_sky_flat(sector[2]), is_sky_flat(sector[3])
        floor_y = from_wad_height(self.height_range[0] if floor_sky else sector[0])
        ceil_y = from_wad_height(self.height_range[1] if ceil_sky else sector[1])
        if floor_sky:
            self.visitor.visit_floor_sky_poly(dict(object_id=info[0], vertices=poly, height=floor_y))
        else:
            self.visitor.visit_floor_poly(dict(object_id=info[0], vertices=poly, height=floor_y, light_info=light,
                                               tex_name=sector[2]))
        if ceil_sky:
            self.visitor.visit_ceil_sky_poly(dict(object_id=info[1], vertices=poly, height=ceil_y))
        else:
            self.visitor.visit_ceil_poly(dict(object_id=info[1], vertices=poly, height=ceil_y, light_info=light,
                                              tex_name=sector[3]))

    def sky_quad(self, object_id, vertices, height_range):  # visitor.rs:987-1008
        low, high = height_range
        if low >= high:
            return
        v1, v2 = vertices
        ex, ey = normalize_or_zero(v2[0] - v1[0], v2[1] - v1[1])
        bx, by = ex * POLY_BIAS * F(16.0), ey * POLY_BIAS * F(16.0)
        nx, ny = -ey, ex
        nbx, nby = nx * POLY_BIAS * F(16.0), ny * POLY_BIAS * F(16.0)
        v1 = (v1[0] + (nbx - bx), v1[1] + (nby - by))
        v2 = (v2[0] + (nbx + bx), v2[1] + (nby + by))
        self.visitor.visit_sky_quad(dict(object_id=object_id, vertices=(v1, v2),
                                         height_range=(from_wad_height(low), from_wad_height(high))))

    def sector_at(self, pos):  # visitor.rs:1028-1060
        lv = self.level
        cid = (len(lv.nodes) - 1) & 0xFFFF
        while True:
            idx, leaf = parse_child_id(cid)
            if leaf:
                if idx >= len(lv.subsectors):
                    return None
                n, first = lv.subsectors[idx]
                if first + n > len(lv.segs) or n == 0:
                    return None
                segs = lv.segs[first:first + n]
                sid = lv.side_sector(lv.seg_sidedef(segs[0]))
                if sid is None:
                    return None
                for seg in segs:
                    vs = lv.seg_vertices(seg)
                    if vs is None:
                        continue
                    if not (Line2f.from_two_points(vs[0], vs[1]).signed_distance(pos) <= SEG_TOLERANCE):
                        return None
                return sid
            if idx >= len(lv.nodes):
                return None
            node = lv.nodes[idx]
            cid = node[13] if partition_line(node).signed_distance(pos) > F(0.0) else node[12]

    def things(self):  # visitor.rs:1010-1026
        for th in self.level.things:
            pos = from_wad_coords(th[0], th[1])
            q = F(th[2]) / F(45.0)
            yaw_deg = F(np.floor(abs(q) + F(0.5))) * (F(1.0) if q >= 0 else F(-1.0)) * F(45.0)  # f32::round
            sid = self.sector_at(pos)
            if sid is None:
                continue
            marker = {1: ('StartPos', 0), 2: ('StartPos', 1), 3: ('StartPos', 2), 4: ('StartPos', 3),
                      11: ('TeleportStart', 0), 14: ('TeleportEnd', 0)}.get(th[3])
            if marker is not None:
                p3 = (pos[0], from_wad_height(self.level.sectors[sid][0]), pos[1])
                self.visitor.visit_marker(p3, yaw_deg * F(np.pi / 180.0), marker)
            else:
                self.decor(th, pos, sid)

    def decor(self, thing, pos, sid):  # visitor.rs:1062-1137
        meta = self.meta.find_thing(thing[3])
        if meta is None:
            return
        s0 = name_push(meta['sprite'], meta['sequence'].encode()[0])
        base = s0 if s0 is not None else meta['sprite']
        sprite0 = name_push(base, 0x30)
        sprite1 = name_push(base, 0x31)
        if sprite0 is None or sprite1 is None:
            return
        img = self.tex.texture(sprite0)
        name = sprite0
        if img is None:
            img = self.tex.texture(sprite1)
            name = sprite1
            if img is None:
                return
        sector = self.level.sectors[sid]
        sx, sy = from_wad_height(img.width), from_wad_height(img.height)
        d = self.dynamic_info.get(sid)
        if meta['hanging']:
            oid = d.ceiling_id if d is not None else 0
            low = (pos[0], from_wad_height(sector[1]) - sy, pos[1])
            high = (pos[0], from_wad_height(sector[1]), pos[1])
        else:
            oid = d.floor_id if d is not None else 0
            low = (pos[0], from_wad_height(sector[0]), pos[1])
            high = (pos[0], from_wad_height(sector[0]) + sy, pos[1])
        self.visitor.visit_decor(dict(object_id=oid, low=low, high=high, half_width=sx * F(0.5),
                                      light_info=self.light_info(sid), tex_name=name))


# ------------------------------------------------------------------------------------------------
# game/src/lights.rs
# ------------------------------------------------------------------------------------------------
def _fract(x):
    return x - np.floor(x)


class Lights:
    def __init__(self):
        self.lights = []

    def push(self, info):  # lights.rs:14-24
        for i, x in enumerate(self.lights):
            if x == info:
                return i
        assert len(self.lights) < 255
        self.lights.append(LightInfo(info.level, info.effect))
        return len(self.lights) - 1

    def fill_buffer_at(self, time):  # lights.rs:26-30
        out = np.zeros(256, np.uint8)
        for i, info in enumerate(self.lights):
            out[i] = int(clamp01(light_level_at(info, F(time))) * F(255.0))
        return out


def light_level_at(info, time):  # lights.rs:33-59
    if info.effect is None:
        return info.level
    alt, speed, dur, sync, kind = info.effect
    if kind == GLOW:
        scale = info.level - alt
        phase = time * speed / scale
        return abs(F(0.5) - _fract(phase)) * F(2.0) * scale + alt
    if kind == RANDOM:
        t = np.floor(time * speed)
        arg = (sync + t / F(1000.0)) * F(12.9898) + sync * F(78.233)
        n = _fract(F(1.0) + F(_libm.sinf(float(arg))) * F(43758.547))  # lights.rs:62-64 (libm sinf)
        return alt if n < dur else info.level
    return alt if _fract(time * speed + sync * F(3.5435)) < dur else info.level


# ------------------------------------------------------------------------------------------------
# game/src/level.rs : Builder  (+ game_shaders.rs atlas name selection)
# ------------------------------------------------------------------------------------------------
STATIC_VERTEX = np.dtype([('a_pos', '<f4', 3), ('a_atlas_uv', '<f4', 2), ('a_tile_uv', '<f4', 2),
                          ('a_tile_size', '<f4', 2), ('a_scroll_rate', '<f4'), ('a_row_height', '<f4'),
                          ('a_num_frames', 'u1'), ('a_light', 'u1'), ('_pad', 'u1', 2)])  # vertex.rs:5-16 (48 B)
SPRITE_VERTEX = np.dtype([('a_pos', '<f4', 3), ('a_atlas_uv', '<f4', 2), ('a_tile_uv', '<f4', 2),
                          ('a_tile_size', '<f4', 2), ('a_local_x', '<f4'), ('a_num_frames', 'u1'),
                          ('a_light', 'u1'), ('_pad', 'u1', 2)])  # vertex.rs:30-40 (44 B)
assert STATIC_VERTEX.itemsize == 48 and SPRITE_VERTEX.itemsize == 44
KIND_FLAT, KIND_WALL, KIND_DECOR, KIND_SKY = 0, 1, 2, 3


class Builder(LevelVisitor):
    """game::level::Builder (game/src/level.rs:307-327, 513-794)."""

    def __init__(self, flat_bounds, wall_bounds, decor_bounds):
        self.flat_bounds, self.wall_bounds, self.decor_bounds = flat_bounds, wall_bounds, decor_bounds
        self.lights = Lights()
        self.start_pos = (F(0), F(0), F(0))
        self.start_yaw = F(0)
        self.static_vertices, self.sky_vertices, self.decor_vertices = [], [], []
        self.object_indices = {}
        self.counters = dict(num_wall_quads=0, num_floor_polys=0, num_ceil_polys=0, num_sky_wall_quads=0,
                             num_sky_floor_polys=0, num_sky_ceil_polys=0, num_decors=0)

    def _indices(self, oid):
        return self.object_indices.setdefault(oid, dict(wall=[], flat=[], sky=[], decor=[]))

    @staticmethod
    def _any_quad(n, out):  # level.rs:620-634
        v0 = n - 4
        out.extend([v0, v0 + 1, v0 + 3, v0 + 1, v0 + 2, v0 + 3])

    @staticmethod
    def _any_poly(n, plen, out):  # level.rs:636-645
        v0 = n - plen
        for v1, v2 in zip(range(v0, n), range(v0 + 1, n)):
            out.extend([v0, v1, v2])

    def _static(self, xz, y, tu, tv, light, scroll, b):
        self.static_vertices.append(((xz[0], y, xz[1]), (b.pos[0], b.pos[1]), (tu, tv), (b.size[0], b.size[1]),
                                     scroll, F(b.row_height), b.num_frames & 0xFF, light, (0, 0)))

    def visit_wall_quad(self, q):  # level.rs:650-681
        self.counters['num_wall_quads'] += 1
        if q['tex_name'] is None:
            return
        b = self.wall_bounds.get(q['tex_name'])
        if b is None:
            return
        light = self.lights.push(q['light_info'])
        (v1, v2), (low, high) = q['vertices'], q['height_range']
        (s1, t1), (s2, t2) = q['tex_start'], q['tex_end']
        self._static(v1, low, s1, t1, light, q['scroll'], b)
        self._static(v2, low, s2, t1, light, q['scroll'], b)
        self._static(v2, high, s2, t2, light, q['scroll'], b)
        self._static(v1, high, s1, t2, light, q['scroll'], b)
        self._any_quad(len(self.static_vertices), self._indices(q['object_id'])['wall'])

    def _flat(self, p, reverse):  # level.rs:683-741
        b = self.flat_bounds.get(p['tex_name'])
        if b is None:
            return
        light = self.lights.push(p['light_info'])
        vs = p['vertices'][::-1] if reverse else p['vertices']
        for v in vs:  # level.rs:536-549: tile_uv = (-x*100, -z*100)
            self._static(v, p['height'], -v[0] * F(100.0), -v[1] * F(100.0), light, F(0.0), b)
        self._any_poly(len(self.static_vertices), len(vs), self._indices(p['object_id'])['flat'])

    def visit_floor_poly(self, p):
        self.counters['num_floor_polys'] += 1
        self._flat(p, False)

    def visit_ceil_poly(self, p):
        self.counters['num_ceil_polys'] += 1
        self._flat(p, True)

    def _sky_poly(self, p, reverse):  # level.rs:727-755
        vs = p['vertices'][::-1] if reverse else p['vertices']
        for v in vs:
            self.sky_vertices.append((v[0], p['height'], v[1]))
        self._any_poly(len(self.sky_vertices), len(vs), self._indices(p['object_id'])['sky'])

    def visit_floor_sky_poly(self, p):
        self.counters['num_sky_floor_polys'] += 1
        self._sky_poly(p, False)

    def visit_ceil_sky_poly(self, p):
        self.counters['num_sky_ceil_polys'] += 1
        self._sky_poly(p, True)

    def visit_sky_quad(self, q):  # level.rs:743-755
        self.counters['num_sky_wall_quads'] += 1
        (v1, v2), (low, high) = q['vertices'], q['height_range']
        for (v, y) in ((v1, low), (v2, low), (v2, high), (v1, high)):
            self.sky_vertices.append((v[0], y, v[1]))
        self._any_quad(len(self.sky_vertices), self._indices(q['object_id'])['sky'])

    def visit_marker(self, pos, yaw, marker):  # level.rs:757-762
        if marker == ('StartPos', 0):
            self.start_pos = (pos[0] + F(0.0), pos[1] + F(0.5), pos[2] + F(32.0) / F(100.0))
            self.start_yaw = yaw

    def visit_decor(self, d):  # level.rs:764-793
        self.counters['num_decors'] += 1
        light = self.lights.push(d['light_info'])
        b = self.decor_bounds.get(d['tex_name'])
        if b is None:
            return
        hw = d['half_width']
        for (pos, lx, tu, tv) in ((d['low'], -hw, F(0.0), b.size[1]), (d['low'], hw, b.size[0], b.size[1]),
                                  (d['high'], hw, b.size[0], F(0.0)), (d['high'], -hw, F(0.0), F(0.0))):
            self.decor_vertices.append((pos, (b.pos[0], b.pos[1]), (tu, tv), (b.size[0], b.size[1]), lx, 1, light,
                                        (0, 0)))
        self._any_quad(len(self.decor_vertices), self._indices(d['object_id'])['decor'])


class BuiltLevel:
    """Everything the reference hands to glium for one level (SURVEY section 8(b) 'downstream hand-off')."""
    pass


def build_level(wad_path, meta_path, level_index):
    """game::create's level half: WadSystem::create -> GameShaders::load_level ->
    Builder::build, returning the arrays in reference draw order."""
    wad = Archive(wad_path, meta_path)
    tex = TextureDirectory(wad)
    level = Level(wad, level_index)
    analysis = LevelAnalysis(level, wad.meta)
    out = BuiltLevel()
    out.wad, out.tex, out.level, out.analysis = wad, tex, level, analysis
    # game_shaders.rs:282-356 : which names feed which atlas
    flat_names = [n for s in level.sectors for n in (s[2], s[3]) if not is_untextured(n) and not is_sky_flat(n)]
    wall_names = [n for sd in level.sidedefs for n in (sd[2], sd[3], sd[4]) if not is_untextured(n)]
    decor_names = []
    for th in level.things:
        m = wad.meta.find_thing(th[3])
        if m is None:
            continue
        s0 = name_push(m['sprite'], m['sequence'].encode()[0])
        base = s0 if s0 is not None else m['sprite']
        for d in (0x30, 0x31):
            nm = name_push(base, d)
            if nm is not None:
                decor_names.append(nm)
    out.flat_atlas, flat_bounds = tex.build_flat_atlas(flat_names)
    out.wall_atlas, wall_bounds = tex.build_texture_atlas(wall_names)
    out.decor_atlas, decor_bounds = tex.build_texture_atlas(decor_names)
    out.flat_bounds, out.wall_bounds, out.decor_bounds = flat_bounds, wall_bounds, decor_bounds
    sky = wad.meta.sky_for(wad.level_name(level_index))
    out.sky_band = sky['tiled_band_size'] if sky else F(0.0)
    sky_img = tex.texture(sky['texture_name']) if sky else None
    out.sky_texture = sky_img.pixels if sky_img is not None else np.zeros((1, 1), np.uint16)
    out.palette = tex.palettes[0].copy()
    if len(tex.colormaps) < 32:  # game_shaders.rs:123-131 asks build_palette_texture for maps 0..=31: the reference indexes past the end (a panic) -- defined here as a corrupt WAD
        raise WadError('COLORMAP has fewer than 32 maps')
    out.colormap = tex.colormaps[:32].reshape(-1).copy()
    out.palette_texture = tex.build_palette_texture(0, 0, 32)
    b = Builder(flat_bounds, wall_bounds, decor_bounds)
    LevelWalker(level, analysis, tex, wad.meta, b).walk()
    out.builder = b
    out.static_vertices = np.array(b.static_vertices, STATIC_VERTEX) if b.static_vertices else np.zeros(0, STATIC_VERTEX)
    out.sky_vertices = np.array(b.sky_vertices, np.float32).reshape(-1, 3)
    out.decor_vertices = np.array(b.decor_vertices, SPRITE_VERTEX) if b.decor_vertices else np.zeros(0, SPRITE_VERTEX)
    # draw order: for each object id ascending: flats, walls, decor, sky (level.rs:443-496)
    draws = []
    idx = {KIND_FLAT: [], KIND_WALL: [], KIND_DECOR: [], KIND_SKY: []}
    static_idx, decor_idx, sky_idx = [], [], []
    for oid in sorted(b.object_indices):
        ind = b.object_indices[oid]
        for kind, key, dst in ((KIND_FLAT, 'flat', static_idx), (KIND_WALL, 'wall', static_idx),
                               (KIND_DECOR, 'decor', decor_idx), (KIND_SKY, 'sky', sky_idx)):
            if ind[key]:
                draws.append((kind, oid, len(dst), len(ind[key])))
                dst.extend(ind[key])
    out.draws = np.array(draws, np.uint32).reshape(-1, 4)
    out.static_indices = np.array(static_idx, np.uint32)
    out.decor_indices = np.array(decor_idx, np.uint32)
    out.sky_indices = np.array(sky_idx, np.uint32)
    out.num_objects = max(1, analysis.num_objects)  # defined, not replicated
    out.counters = dict(b.counters)
    out.counters['num_static_tris'] = len(static_idx) // 3
    out.counters['num_sky_tris'] = len(sky_idx) // 3
    out.counters['num_sprite_tris'] = len(decor_idx) // 3
    out.lights = b.lights
    out.start_pos, out.start_yaw = b.start_pos, b.start_yaw
    return out
