#!/usr/bin/env python3
"""Prints, for every level of an IWAD, the counters game::level::Builder logs after a build
(game/src/level.rs:384-422: "Level built in ... ms:" followed by the counts), computed by THIS repository's loader and
builder (rdoom_wad_open / rdoom_wad_build_level / rdoom_built_counters) -- to be put next to the reference's own log
(docs/RUST_CROSSCHECK.md).  No GPU needed.

    python tools/print_counters.py DOOM1.WAD /root/reference/assets/meta/doom.toml [level index ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_doom_amd as rd  # noqa: E402


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    wad = rd.Wad(sys.argv[1], sys.argv[2])
    levels = [int(a) for a in sys.argv[3:]] or range(wad.num_levels())
    for index in levels:
        c = wad.build_level(index).counters()
        print('%s:' % wad.level_name(index))
        # the reference's wording and order (game/src/level.rs:384-396): "Level built in ..ms:" + one line per counter
        print('Level built:')
        for key in ('num_wall_quads', 'num_floor_polys', 'num_ceil_polys', 'num_sky_wall_quads', 'num_sky_floor_polys',
                    'num_sky_ceil_polys', 'num_decors', 'num_static_tris', 'num_sky_tris', 'num_sprite_tris'):
            print('\t%s = %d' % (key, c[key]))

if __name__ == '__main__':
    main()
