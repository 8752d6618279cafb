// Launchers of the renderer's kernels, one translation unit per kernel (setup.hip, bin.hip, raster.hip,
// fragment.hip); renderer.hip (the C ABI) strings them together on the caller's stream.
#pragma once
#include "records.hpp"

#define HIP_TRY(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess)                                                                                 \
      return rdoom::fail(_e == hipErrorOutOfMemory ? RDOOM_OOM : RDOOM_HIP_ERROR, "%s failed: %s", #expr, \
                         hipGetErrorString(_e));                                                          \
  } while (0)

namespace rdoom_dev {

constexpr uint32_t MAX_TILES = 8192;     // tiles per frame the binning kernel keeps counters for
// A tile whose list is longer than one batch of the rasteriser (one entry per lane) gets a list per 32 x 32 quadrant (bin.hip):
// tile header word y carries TILE_SPLIT, word x points at (first entry, count) x 4 in the pose's entry array
#ifndef RDOOM_LONG_LIST
#define RDOOM_LONG_LIST 64
#endif
constexpr uint32_t LONG_LIST = RDOOM_LONG_LIST, TILE_SPLIT = 0x80000000u;

// Kernel 1: vertex stage, triangle setup, near-to-far record order (setup.hip)
rdoom_status launch_setup(hipStream_t st, uint32_t n_poses, const DeviceLevelView &lv, const PoseConst *poses,
                          const ObjectConst *objects, uint32_t n_objects, int width, int height, uint32_t kinds_mask,
                          TriRec *recs, uint32_t *visible, uint32_t *counts, uint32_t *ghist, uint32_t cap,
                          uint32_t *mismatch_flag);  // set when the set-up kernel rejects a triangle the cull kernel kept
size_t setup_histogram_bytes(uint32_t max_poses);  // scratch of the counting sort (per pose: one counter per depth bucket)
// Kernel 1b: per-tile triangle lists (bin.hip).  *launched = false: the frame has too many tiles for the kernel's LDS counters --
// nothing was launched and the caller must flag every pose as "bins incomplete".  A launch that FAILS is an error.
rdoom_status launch_bin(hipStream_t st, uint32_t n_poses, const TriRec *recs, const uint32_t *counts,
                        uint32_t cap, int tiles_x, int tiles_y, uint2 *tile_hdr, uint32_t *entries, uint32_t entry_cap,
                        uint2 *hits, uint32_t *overflow,
                        bool want_split, bool *launched, bool *used_split);  // lists of more than LONG_LIST entries per quadrant, if the counters fit (bin.hip)
// Kernel 2: tiled rasteriser -> visibility words (raster.hip)
rdoom_status launch_raster(hipStream_t st, uint32_t n_poses, const DeviceLevelView &lv, const TriRec *recs,
                           const uint32_t *counts, uint32_t cap, int width, int height, int tiles_x,
                           int tiles_y, const uint2 *tile_hdr, const uint32_t *entries, uint32_t entry_cap,
                           const uint32_t *overflow, uint32_t *vis, bool vis16, uint32_t *prim_out,
                           uint32_t *qtab,  // qtab (optional): per (pose, tile, quadrant) the record all its pixels show, or NONE
                           bool skip_described_vis,  // no visibility words for quadrants the table describes (FragmentPlan)
                           bool split_lists,  // the binning kernel stored long lists per quadrant (launch_bin's answer for this render)
                           bool bins_launched);  // the binning kernel ran (its overflow flags are this render's): settle_kernel may read the lists
// How the fragment kernel will walk a frame of this size, decided ONCE per render from the debug hooks (rasteriser and
// fragment kernel must agree on who reads the quadrant table): quads per lane, log2(units per block row), blocks per
// workgroup wave, the test hook leak_mod, and qtab_mode (0: table unused; 1 / 2: a wave block lies in one / two quadrants).
struct FragmentPlan {
  int nq;
  uint32_t bwl, chunk, leak_mod, qtab_mode;
  // the rasteriser may leave out the visibility words of quadrants the table describes: every reader consults the table first
  bool skip_described_vis;
  // fragment_quadrant_kernel shades the described quadrants whose record qualifies before fragment_kernel runs
  bool quadrant_path;
};
FragmentPlan plan_fragment(int width, int pitch, int height, bool have_qtab);
// Kernels 3 + 4: fragment kernel -> palette indices, then the alpha-leak fixup (fragment.hip)
rdoom_status launch_fragment(hipStream_t st, uint32_t n_poses, const DeviceLevelView &lv, const TriRec *recs,
                             const uint32_t *counts, uint32_t cap, const PoseConst *poses,
                             int width, int pitch,  // the frame's width; pixels between rows of visibility words / framebuffer bytes
                             int height, int tiles_x, int tiles_y, const uint2 *tile_hdr,
                             const uint32_t *entries, uint32_t entry_cap, const uint32_t *overflow, uint32_t *vis,
                             bool vis16, uint32_t *prim_out, const float *ndc_tab, uint8_t *fb, uint32_t *fix_count,
                             uint2 *fix_list, uint32_t fix_cap, uint32_t *qtab, void *d_frag_const,
                             bool *frag_const_ready, const FragmentPlan &plan);  // d_frag_const: fragment_const_bytes() of device memory owned by the batch
size_t fragment_const_bytes();

}  // namespace rdoom_dev
