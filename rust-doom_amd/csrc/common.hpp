// Shared by the host library and the HIP translation units: error reporting across the C ABI.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <string>

#include "rdoom.h"

namespace rdoom {
std::string &last_error_ref();

// Test hooks set through rdoom_debug_set (include/rdoom.h): every one of them selects a differently shaped but
// EQUIVALENT path -- the image must not change (tests/test_gpu_debug_paths.py).  Nothing reads the environment.
struct DebugOptions {
  int no_bins = 0;       // rasterise from the sorted list (the fallback used when a pose overflows its tile lists)
  int entry_cap = 0;     // tile-list entries per pose (0 = default), to force that overflow
  int vis32 = 0;         // 32-bit visibility words (the format of levels with >= 65535 triangles)
  int leak_mod = 0;      // every n-th pixel is queued as an alpha leak: fixup_kernel re-resolves ordinary pixels
  int frag_nq = 2;       // quads per lane in the fragment kernel (1 = the variant for widths that are not a multiple of 8)
  int frag_bw = -1;      // log2(units per row of the fragment kernel's wave block); -1 = by frame size (fragment.hip: plan_fragment)
  int frag_chunk = 0;    // wave blocks per wave in the fragment kernel (0 = default)
  int bin_threads = 0;   // workgroup size of the binning kernel (0 = by frame size: bin.hip)
  int no_cover = 0;      // no depth-only body for quadrant-covering triangles
  int no_qtab = 0;       // the fragment kernel ignores the rasteriser's quadrant table (every wave reads its visibility words)
  int qpath = 0;         // the whole-quadrant fragment kernel runs first (off by default: measured slower)
  int keep_vis = 0;      // the rasteriser writes the visibility words of every quadrant, also of those the table describes
  int no_pair = 0;       // no two-entry shortcut in the rasteriser (a quadrant shared by two triangles along a common edge takes the general pass)
  int no_settle = 0;     // no settle_kernel: the rasteriser's waves find the one-triangle quadrants themselves, as before round 5
  int settle_max = 0;    // longest tile list settle_kernel examines (0 = default, raster.hip RDOOM_SETTLE_MAX)
  int no_split = 0;      // no per-quadrant lists for tiles with more than 64 entries: the rasteriser re-gathers such a tile's whole list for every quadrant
  int raster_stats = 0;  // census of the rasteriser's paths on stderr (instrumented instantiation: slower)
};
DebugOptions debug_options();  // a snapshot taken under the lock rdoom_debug_set writes under (one host thread per GPU may render while a test thread sets hooks)
inline rdoom_status fail(rdoom_status code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return code;
}
}  // namespace rdoom
