"""GPU: the rarely taken paths, forced through the library's test hooks (rdoom_debug_set, include/rdoom.h; each child
process sets its own, they are process-wide).  Every hook selects an equivalent path: the image must not change.
  leak_mod=n     every n-th pixel is treated as an alpha leak, so fixup_kernel's general per-pixel rule (all candidates of
                 the tile, lexicographic (depth, primitive) minimum) re-resolves ordinary pixels;
  no_bins=1      the rasteriser's fallback scan (no per-tile bins), as used when a pose overflows them;
  entry_cap=n    tile-list entries per pose the binning kernel may emit, to force that overflow;
  frag_nq=1      one quad per lane in the fragment kernel (default two when the width is a multiple of 8);
  vis32=1        32-bit visibility words (levels with >= 65535 triangles) instead of 16-bit ones;
  no_cover=1     the rasteriser without its depth-only body for quadrant-covering triangles;
  frag_bw=k      the fragment kernel's wave block is 2^k units wide (default: 2 = 32 x 16 pixels for frames of 1280 x 720 and up,
                 3 = 64 x 8 below);
  no_qtab=1      the fragment kernel ignores the rasteriser's quadrant table ("every pixel of this 32 x 32 quadrant shows
                 record r") and reads the visibility words of every block, as it did before the table existed;
  keep_vis=1     the rasteriser writes the visibility words of every quadrant, also of those its table describes (by default it
                 leaves them out and every reader asks the table first);
  qpath=1        the whole-quadrant fragment kernel runs first (fragment_quadrant_kernel shades the described quadrants whose record
                 qualifies -- queueing uncertified-mod runs and transparent texels for fixup_kernel -- and fragment_kernel skips
                 the blocks that lie in them; off by default: measured slower than fragment_kernel alone).
  no_split=1     no list per quadrant for the tiles whose list holds more than 64 entries (bin.hip "split lists": the rasteriser
                 takes such a tile as four parts, each quadrant with its own list; without them it re-gathers the whole list
                 for every quadrant).
Each child renders another set of poses into its batch before the checked render: whatever the checked render does not write
holds another frame's values.  The image is checked with and without primitive ids."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_child(hooks, args=('0', '320', '200', '6')):
    p = subprocess.run([sys.executable, os.path.join(HERE, 'gpu_child_case.py'), *args, *['%s=%s' % kv for kv in hooks.items()]],
                       cwd=HERE, capture_output=True, text=True, timeout=600)
    m = re.search(r'RESULT bad=(\d+) fixups=(\d+)', p.stdout)
    assert m, p.stdout + p.stderr
    return int(m.group(1)), int(m.group(2))


def test_forced_alpha_leak_fixups_do_not_change_the_image():
    bad, fixups = run_child({'leak_mod': 97})
    assert bad == 0
    assert fixups > 3000  # about 1/97 of 6 x 320 x 200 covered pixels went through fixup_kernel


def test_fallback_scan_without_bins():
    bad, _ = run_child({'no_bins': 1})
    assert bad == 0


def test_tile_list_overflow_falls_back_to_the_scan():
    """entry_cap=300: nearly every pose needs more tile-list entries than that, sets its overflow flag and is rasterised by the scan"""
    bad, _ = run_child({'entry_cap': 300})
    assert bad == 0


def test_32_bit_visibility_words():
    """vis32=1: the visibility buffer keeps 32-bit record indices (the format used when a level has 65535 or
    more triangles) instead of the 16-bit words the synthetic levels qualify for"""
    bad, _ = run_child({'vis32': 1})
    assert bad == 0
    bad, fixups = run_child({'vis32': 1, 'leak_mod': 101})
    assert bad == 0 and fixups > 3000


def test_one_quad_per_lane_fragment_kernel():
    """frag_nq=1: the fragment kernel variant used for frame widths that are not a multiple of 8"""
    bad, _ = run_child({'frag_nq': 1})
    assert bad == 0


def test_rasteriser_without_the_quadrant_cover_body():
    bad, _ = run_child({'no_cover': 1})
    assert bad == 0
    bad, _ = run_child({'no_cover': 1, 'no_bins': 1})
    assert bad == 0


@pytest.mark.parametrize('hooks', [{'no_pair': 1}, {'no_pair': 1, 'no_split': 1}, {'no_pair': 1, 'keep_vis': 1}, {'no_pair': 1, 'no_bins': 1}])
def test_rasteriser_without_the_two_entry_shortcut(hooks):
    """no_pair=1: a quadrant shared by two triangles along a common edge (7 % of the passes at 1080p) is settled by the sign of
    that edge function, without a depth compare -- a default-on exactness path that no_cover switched off only TOGETHER with the
    one-entry shortcut.  With the hook it alone is off: the general pass must produce the same bytes."""
    for args in (('0', '320', '200', '6'), ('0', '1920', '1080', '3'), ('4', '1280', '720', '3'), ('6', '1000', '520', '3')):
        bad, _ = run_child(hooks, args)
        assert bad == 0, (hooks, args)


@pytest.mark.parametrize('hooks', [{'no_settle': 1}, {'settle_max': 1}, {'settle_max': 4}, {'settle_max': 64}, {'settle_max': 64, 'no_split': 1},
                                   {'no_settle': 1, 'no_pair': 1}, {'settle_max': 8, 'no_pair': 1}, {'settle_max': 16, 'bin_threads': 512}])
def test_settle_kernel_on_off_and_list_limits(hooks):
    """settle_kernel (round 5) enters the one-triangle quadrants in the table before the rasteriser's waves start; tiles with
    nothing left to do end at once, others pass only their TODO quadrants.  Off (no_settle), with list limits of 1 (only
    one-entry lists), 4, 8, 16, 64 (every list that is stored whole), frames with partial tiles, many described quadrants (1080p)
    and long lists (320 x 200): the rasteriser's own shortcut must agree with it everywhere -- same bytes."""
    for args in (('0', '320', '200', '6'), ('0', '1920', '1080', '3'), ('5', '1000', '520', '3'), ('2', '1366', '768', '2')):
        bad, _ = run_child(hooks, args)
        assert bad == 0, (hooks, args)


@pytest.mark.parametrize('args', [('0', '1920', '1080', '3'), ('3', '1280', '720', '3'), ('6', '640', '400', '3')])
def test_both_default_block_shapes_at_every_size(args):
    """the fragment kernel's default wave block is 32 x 16 pixels from 1280 x 720 up and 64 x 8 below (round 5): each shape at the
    sizes where the other is the default"""
    for bw in (2, 3):
        bad, _ = run_child({'frag_bw': bw}, args)
        assert bad == 0, (bw, args)


@pytest.mark.parametrize('size', [(322, 200), (1366, 768), (323, 131)])
def test_padded_row_pitch_under_the_hooks(size):
    """widths that are not a multiple of 4 (padded row pitch) through the paths that address pixels on their own: the alpha-leak
    queue (leak_mod: fixup_kernel divides by the pitch), 32-bit visibility words, the sorted-list fallback, every block shape"""
    for hooks in ({'leak_mod': 5}, {'vis32': 1, 'leak_mod': 11}, {'no_bins': 1}, {'frag_bw': 2}, {'frag_nq': 1}, {'frag_bw': 5}, {'no_qtab': 1},
                  {'keep_vis': 1}, {'qpath': 1}):
        bad, _ = run_child(hooks, ('0', str(size[0]), str(size[1]), '3'))
        assert bad == 0, (hooks, size)


@pytest.mark.parametrize('bw', [0, 2, 4, 6])
def test_fragment_wave_block_shapes(bw):
    """1 x 64, 4 x 16, 16 x 4 and 64 x 1 units per wave (frame 320 x 200: 40 units per row, partial blocks in both
    directions for most shapes)"""
    bad, _ = run_child({'frag_bw': bw})
    assert bad == 0
    bad, _ = run_child({'frag_bw': bw, 'frag_nq': 1})
    assert bad == 0


@pytest.mark.parametrize('hooks', [{'no_qtab': 1}, {'frag_bw': 2}, {'frag_bw': 2, 'no_qtab': 1}, {'frag_bw': 3, 'frag_nq': 1},
                                   {'frag_bw': 4, 'frag_nq': 1}, {'frag_bw': 2, 'no_bins': 1}, {'frag_bw': 3, 'vis32': 1},
                                   {'keep_vis': 1}, {'keep_vis': 1, 'frag_bw': 2}, {'leak_mod': 5}, {'leak_mod': 3, 'frag_bw': 2},
                                   {'frag_bw': 5}, {'frag_bw': 1}, {'qpath': 1}, {'qpath': 1, 'frag_bw': 2},
                                   {'qpath': 1, 'keep_vis': 1}, {'qpath': 1, 'vis32': 1, 'no_bins': 1}, {'frag_nq': 1},
                                   {'frag_bw': 2, 'keep_vis': 1, 'vis32': 1}])
def test_quadrant_table_paths(hooks):
    """the table serves 32-pixel-wide blocks (one quadrant) and 64-pixel-wide ones (two quadrants side by side), with 8- and
    4-pixel runs per lane; frames whose right / top quadrants are partly outside (1000 x 520 = 15.6 x 8.1 tiles).  Where the
    table is in use the rasteriser leaves out the visibility words of the quadrants it describes (keep_vis: writes them all the
    same; leak_mod: the fragment kernel wants them all; block shapes the table does not serve, frag_bw 1 / 5: no table at all)"""
    for args in (('0', '320', '200', '6'), ('0', '1000', '520', '3'), ('3', '712', '296', '3')):
        bad, _ = run_child(hooks, args)
        assert bad == 0, (hooks, args)


@pytest.mark.parametrize('size', [(8, 8), (64, 64), (128, 64), (1024, 512), (1920, 1088), (12, 300), (2052, 36)])
def test_frame_sizes_around_the_tile_grid(size):
    """frames that are one tile, whole tiles only (the tile-level shortcut can fire on every tile), narrower than a quadrant,
    or a strip one tile high; width a multiple of 4 but not of 8 takes the one-quad-per-lane fragment variant"""
    bad, _ = run_child({}, ('0', str(size[0]), str(size[1]), '3'))
    assert bad == 0, size
    bad, _ = run_child({'no_qtab': 1}, ('4', str(size[0]), str(size[1]), '2'))
    assert bad == 0, size


@pytest.mark.parametrize('args', [('0', '1920', '1080', '3'), ('4', '1280', '720', '3'), ('7', '640', '400', '4')])
def test_quadrant_path_at_larger_frames(args):
    """qpath=1 (the whole-quadrant fragment kernel + fragment_kernel skipping what it shaded) on frames with many described
    quadrants: masked wall textures (opacity test, leaks queued for fixup_kernel), integer tile sizes (certified mod, failures
    queued), partial quadrants at the frame's bottom (1080 = 33.75 x 32)"""
    bad, _ = run_child({'qpath': 1}, args)
    assert bad == 0, args


@pytest.mark.parametrize('hooks', [{}, {'no_split': 1}, {'no_split': 1, 'no_cover': 1}, {'no_cover': 1}, {'keep_vis': 1, 'no_split': 1}, {'vis32': 1},
                                   {'leak_mod': 7}, {'leak_mod': 7, 'no_split': 1}, {'bin_threads': 128}, {'bin_threads': 512, 'no_qtab': 1}])
def test_long_tile_lists_split_by_quadrant_or_not(hooks):
    """frames whose far tiles hold long lists (320 x 200: a quarter of the tiles hold more than 64 entries, some several
    hundred -- quadrant lists that fit one batch and take the shortcuts, quadrant lists of several batches, quadrants outside
    the frame, entries that touch several quadrants and are in each of their lists) with the per-quadrant lists (default)
    and without (no_split); leak_mod sends pixels of such tiles through fixup_kernel, which reads the pixel's quadrant's list"""
    for args in (('0', '320', '200', '6'), ('2', '200', '136', '4'), ('5', '712', '296', '3')):
        bad, _ = run_child(hooks, args)
        assert bad == 0, (hooks, args)


@pytest.mark.parametrize('size', [(5120, 2880), (6144, 3456)])
def test_frames_around_the_split_lists_tile_limit(size):
    """the binning kernel keeps three LDS counters per tile for the per-quadrant lists: a 5K frame (3 600 tiles) still has
    them, a 6K frame (5 184 tiles) falls back to whole-tile lists and the rasteriser's instantiation without them"""
    bad, _ = run_child({}, ('0', str(size[0]), str(size[1]), '1'))
    assert bad == 0, size


def test_child_case_plain():
    bad, fixups = run_child({})
    assert bad == 0 and fixups < 1000
