"""The seeded synthetic IWADs the benchmark, smoke() and the tests run on (no DOOM1.WAD / DOOM2.WAD exists in the image
or on the GPU box): generated on demand by tools/mkwad.py, never committed; their digests are (tests/golden)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
WAD_PATH = os.path.join(GOLDEN, 'synth.wad')
BIG_WAD_PATH = os.path.join(GOLDEN, 'synth_big.wad')
RICH_WAD_PATH = os.path.join(GOLDEN, 'synth_rich.wad')
META_PATH = os.path.join(ROOT, 'assets', 'meta', 'synth.toml')   # metadata in the reference's schema (assets/meta/doom.toml)


def _generate(path, **kw):
    tools = os.path.join(ROOT, 'tools')
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import mkwad
    wad, _ = mkwad.build_wad(1993, **kw)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = '%s.%d.tmp' % (path, os.getpid())  # several ranks may get here at once: write aside, then rename
    with open(tmp, 'wb') as f:
        f.write(wad)
    os.replace(tmp, path)


def ensure_wad():
    """E1M1..E1M9; E1M1 sized like the shareware E1M1 (713 linedefs, 284 sub-sectors, 3.6 k static triangles)."""
    if not os.path.exists(WAD_PATH):
        _generate(WAD_PATH)
    return WAD_PATH


def ensure_big_wad():
    """A second IWAD with ONE level ten times the size of E1M1 (7.2 k linedefs, 3 k sub-sectors, ~36 k triangles: larger
    than any level of DOOM / DOOM2) -- the stand-in for BASELINE config 5's MAP29."""
    if not os.path.exists(BIG_WAD_PATH):
        _generate(BIG_WAD_PATH, specs=[('E1M1', ('gen', 424242, 128, 90))])
    return BIG_WAD_PATH


def ensure_rich_wad():
    """A third IWAD with ONE level: E1M1's geometry (same generator seed), but every linedef side with a wall texture of its own out
    of 320 and every sector its own flats out of 192 -- a wall atlas of 2048 x 2048 and more, a texel store several times one
    XCD's L2 (tools/mkwad.py build_wad(rich=True)).  The nine default levels keep theirs at 1.1 MB, which a real IWAD does not."""
    if not os.path.exists(RICH_WAD_PATH):
        _generate(RICH_WAD_PATH, specs=[('E1M1', ('gen', 1993 * 7 + 1, 52, 14))], rich=True)
    return RICH_WAD_PATH


def wad_digest():
    with open(ensure_wad(), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()
