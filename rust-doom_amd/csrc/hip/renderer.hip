// C ABI of the device renderer (include/rdoom.h "device renderer"): level upload, batch scratch, and
// rdoom_batch_render = setup -> bin -> raster -> fragment -> fixup on the caller's stream.
//
// Replaces the reference's GL draw path: VertexBuffer / IndexBuffer / Texture2d uploads (engine/src/meshes.rs:126-201,
// engine/src/uniforms.rs:146-221) and the frame.draw loop of Renderer::update (engine/src/renderer.rs:98-157).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

#include "kernels.hpp"

#pragma clang fp contract(off)

using namespace rdoom_dev;

namespace {
bool is_pow2(uint32_t x) { return x != 0 && (x & (x - 1)) == 0; }
}  // namespace

// Host-side section timers of rdoom_batch_render (tools/variant.sh NAME renderer -DRDOOM_HOST_TIMERS; never in the shipped
// library): where does the host's time per render go?  Sums are printed when a batch is destroyed.
#ifdef RDOOM_HOST_TIMERS
#include <chrono>
static double g_host_t[12];
static unsigned long long g_host_n;
#define HT_DECL auto ht_last = std::chrono::steady_clock::now();
#define HT_MARK(i) do { const auto n_ = std::chrono::steady_clock::now(); g_host_t[i] += std::chrono::duration<double, std::micro>(n_ - ht_last).count(); ht_last = n_; } while (0)
#else
#define HT_DECL
#define HT_MARK(i) do { } while (0)
#endif

// A resident SET of levels (rdoom_levelset_create); rdoom_level_create makes a set of one.
struct rdoom_level {
  int device = 0;
  DeviceLevelView view{};
  void *d_clusters = nullptr;
  void *d_tris = nullptr, *d_texels = nullptr, *d_sky = nullptr, *d_cmap = nullptr, *d_slices = nullptr;
  std::vector<LevelSlice> slices;      // host copy of the slice table
  std::vector<uint32_t> slice_objects; // 1 + the largest object id each level draws
  uint32_t ntri = 0;        // the LARGEST level's triangle count: a pose's records and visible list have this stride
  uint32_t n_objects = 1;   // 1 + the largest object id any level of the set draws
};

struct rdoom_batch {
  const rdoom_level *level = nullptr;
  uint32_t width = 0, height = 0, max_poses = 0, cap = 0, last_n = 0;
  // Row pitch, in pixels, of the visibility words, primitive ids and framebuffers: the width itself when it is a multiple of 4
  // (every size the kernels were written for), else the next multiple of 8 -- the reference takes any --resolution WxH
  // (src/main.rs:41).  The frame's geometry (viewport, bounding boxes, sky ndc) uses `width`; the padding columns are never
  // covered (S6 clamps every bounding box to the frame) and never read back.
  uint32_t pitch = 0;
  PoseConst *d_poses = nullptr;
  TriRec *d_recs = nullptr;   // max_poses x cap records in near-to-far order (setup -> bin, raster, fragment)
  uint32_t *d_visible = nullptr;  // max_poses x cap: the visible triangles of each pose (cull kernel -> set-up kernel)
  uint2 *d_tile_hdr = nullptr;     // per (pose, tile): (first entry, entry count)
  uint32_t *d_entries = nullptr;   // per pose: entry_cap tile-list entries (record index | quadrant mask << 28)
  uint2 *d_hits = nullptr;         // per pose: entry_cap (entry, tile) pairs: the binning kernel's count pass -> its fill pass
  uint32_t *d_overflow = nullptr;  // per pose: 1 = bins incomplete, rasteriser scans the sorted list
  uint32_t entry_cap = 0, n_tiles = 0;
  // The words every render starts from zero -- d_fix_count (4), d_counts (max_poses), d_ghist (2048 per pose) -- are ONE
  // allocation, d_zeroed, cleared by one hipMemsetAsync: four separate fills per render were four more packets between the
  // kernels of a stream (the 1/8 share of config 4 queues 9 x 13 packets per step for a few hundred microseconds of work each).
  uint32_t *d_zeroed = nullptr;
  uint32_t *d_fix_count = nullptr;  // [0] = queued pixels, [1] = error flag (fixup list overflow), [2] = error flag (the set-up kernel
                                    // disagreed with the cull kernel about a triangle: the counting sort's buckets would not add up)
  uint2 *d_fix_list = nullptr;
  uint32_t fix_cap = 1u << 20;
  uint32_t *d_counts = nullptr, *d_vis = nullptr, *d_prim = nullptr;
  void *d_frag_const = nullptr;  // the fragment kernel's rarely read constants (fragment.hip: FragConst), written on first use
  bool frag_const_ready = false;
  uint32_t *d_qtab = nullptr;  // per (pose, tile, quadrant): the record every pixel of the quadrant shows, or NONE (rasteriser -> fragment kernel)
  uint32_t *d_ghist = nullptr;  // counting sort of the set-up: per pose, one counter per depth bucket
  uint8_t *d_fb = nullptr;
  // Pinned staging for the per-pose constants, TWO deep: a render waits for the H2D copy of the render before the previous
  // one, not for the previous one's -- with one buffer the host could not queue a render before the stream had reached the last
  // render's copy, and a host thread that feeds several batches on several streams stalled on each in turn (the 1/8 share of
  // BASELINE config 4, nine levels x small batches, ran SLOWER on two streams than on one).
  static constexpr uint32_t STAGES = 2;
  PoseConst *h_poses[STAGES] = {nullptr, nullptr};
  ObjectConst *d_objects = nullptr, *h_objects[STAGES] = {nullptr, nullptr};  // max_poses x n_objects, allocated on first use
  uint32_t stage = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  static constexpr uint32_t RING = 64;  // renders whose per-kernel events may be pending (rdoom_batch_render_profiled)
  hipEvent_t ring[RING][4] = {};
  uint32_t ring_n = 0, ring_poses = 0;
  hipEvent_t ev_copy[STAGES] = {nullptr, nullptr};  // H2D of h_poses[i] finished: that staging buffer may be rewritten
  // The end of the last render on ITS stream, and a stream of the batch's own for read-backs: rdoom_batch_finish and the
  // rdoom_batch_read_* wait for this batch's work only, never for the device -- another host thread's render on another
  // stream (one host thread per GPU or stream) does not delay them.
  hipEvent_t ev_done = nullptr;
  hipStream_t copy_stream = nullptr;
  bool want_prim = false;
  bool vis16 = false;  // record indices fit 16 bits: visibility words are u16
  float *d_ndc = nullptr;  // (ix + 0.5) / (width / 2) - 1 for every column, then (iy + 0.5) / (height / 2) - 1 for every row
};

// Every entry point that touches a batch's memory or waits for its work first makes the level's device current: a host
// thread that drives several GPUs (or read another batch in between) would otherwise synchronise / copy on the wrong one.
static hipError_t bind_device(const rdoom_batch *b) { return hipSetDevice(b->level->device); }

// What only the device finds out about a render: read after a synchronisation, reported as a status.
// Device -> host over the batch's own stream, after the batch's last render (and nothing else) has finished.  rows > 1: a
// strided source (pitch_bytes between rows), tightly packed at the destination.
static hipError_t read_back(const rdoom_batch *b, void *dst, const void *src, size_t row_bytes, size_t rows = 1, size_t pitch_bytes = 0) {
  hipError_t e = hipStreamWaitEvent(b->copy_stream, b->ev_done, 0);  // (an event never recorded counts as complete)
  if (e != hipSuccess) return e;
  if (rows <= 1 || pitch_bytes == row_bytes)
    e = hipMemcpyAsync(dst, src, row_bytes * (rows ? rows : 1), hipMemcpyDeviceToHost, b->copy_stream);
  else
    e = hipMemcpy2DAsync(dst, row_bytes, src, pitch_bytes, row_bytes, rows, hipMemcpyDeviceToHost, b->copy_stream);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(b->copy_stream);
}

static rdoom_status device_flags(const rdoom_batch *b, uint32_t *out_fixups = nullptr) {
  uint32_t fix[3] = {0, 0, 0};
  HIP_TRY(read_back(b, fix, b->d_fix_count, sizeof fix));
  if (out_fixups) *out_fixups = fix[0];
  if (fix[1]) return rdoom::fail(RDOOM_BAD_LEVEL, "alpha-leak fixup list overflow (%u pixels)", fix[0]);
  if (fix[2]) return rdoom::fail(RDOOM_HIP_ERROR, "internal: set-up and cull kernels disagree about a triangle (build flags changed?)");
  return RDOOM_OK;
}

extern "C" {

rdoom_status rdoom_device_count(int32_t *out_count) {
  if (!out_count) return rdoom::fail(RDOOM_BAD_ARG, "out_count is null");
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  *out_count = n;
  return RDOOM_OK;
}

rdoom_status rdoom_set_device(int32_t device) {
  HIP_TRY(hipSetDevice(device));
  return RDOOM_OK;
}

void rdoom_level_destroy(rdoom_level *level) {
  if (!level) return;
  for (void *p : {level->d_clusters, level->d_tris, level->d_texels, level->d_sky, level->d_cmap, level->d_slices})
    if (p) (void)hipFree(p);
  delete level;
}

// One level of a set, flattened on the host: what rdoom_levelset_create concatenates and uploads.
namespace {
struct HostLevel {
  std::vector<LevelTri> tris;
  std::vector<Cluster> clusters;     // Cluster::first relative to this level's first triangle
  std::vector<uint16_t> texels;      // wall atlas, then (at multiples of 1024) the flat atlas promoted to u16, the decor atlas
  size_t flat_base = 0, decor_base = 0;
  std::vector<uint16_t> sky;
  LevelSlice dims{};                 // the atlas / sky sizes (bases and ranges are filled in when the set is assembled)
  uint32_t n_objects = 1;
};
}  // namespace

static rdoom_status flatten_level(const rdoom_level_desc *d, HostLevel &out);
static rdoom_status levelset_create_impl(const rdoom_level_desc *const *descs, uint32_t n_levels, rdoom_level **out_level);

rdoom_status rdoom_levelset_create(const rdoom_level_desc *const *descs, uint32_t n_levels, rdoom_level **out_level) {
  try {  // std::vector / std::map below may throw: nothing unwinds across the C ABI
    return levelset_create_impl(descs, n_levels, out_level);
  } catch (const std::bad_alloc &) {
    return rdoom::fail(RDOOM_OOM, "out of host memory");
  } catch (const std::exception &e) {
    return rdoom::fail(RDOOM_BAD_LEVEL, "%s", e.what());
  }
}

rdoom_status rdoom_level_create(const rdoom_level_desc *d, rdoom_level **out_level) {
  return rdoom_levelset_create(&d, 1u, out_level);
}

static rdoom_status flatten_level(const rdoom_level_desc *d, HostLevel &out) {
  if (!d) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  if (!d->colormap) return rdoom::fail(RDOOM_BAD_ARG, "colormap is null");
  if ((d->flat_atlas && !(d->flat_w && d->flat_h)) || (d->wall_atlas && !(d->wall_w && d->wall_h)) ||
      (d->decor_atlas && !(d->decor_w && d->decor_h)))
    return rdoom::fail(RDOOM_BAD_ARG, "an atlas pointer is set but its size is zero");
  if ((d->flat_w | d->flat_h) && !(is_pow2(d->flat_w) && is_pow2(d->flat_h)))
    return rdoom::fail(RDOOM_BAD_ARG, "flat atlas %ux%u is not a power of two", d->flat_w, d->flat_h);
  if ((d->wall_w | d->wall_h) && !(is_pow2(d->wall_w) && is_pow2(d->wall_h)))
    return rdoom::fail(RDOOM_BAD_ARG, "wall atlas %ux%u is not a power of two", d->wall_w, d->wall_h);
  if ((d->decor_w | d->decor_h) && !(is_pow2(d->decor_w) && is_pow2(d->decor_h)))
    return rdoom::fail(RDOOM_BAD_ARG, "decor atlas %ux%u is not a power of two", d->decor_w, d->decor_h);
  if (d->flat_w > 32768 || d->flat_h > 32768 || d->wall_w > 32768 || d->wall_h > 32768 || d->decor_w > 32768 ||
      d->decor_h > 32768)
    return rdoom::fail(RDOOM_BAD_ARG, "atlas larger than 32768 texels on a side");
  // flatten the draws into one primitive list in draw order (primitive id == position)
  std::vector<LevelTri> &tris = out.tris;
  uint32_t n_objects = 1;
  // Alpha-test classification of a wall texture (all its animation frames): bit 0 = a texel in the
  // one-texel ring AROUND the rectangle is transparent (the float mod of F2 can land there), bit 1 =
  // the rectangle itself contains transparent texels (a genuinely masked texture).
  std::map<std::tuple<float, float, float, float, uint32_t, float>, uint32_t> masked_cache;
  auto region_masked = [&](const rdoom_static_vertex &v) -> uint32_t {
    auto key = std::make_tuple(v.a_atlas_uv[0], v.a_atlas_uv[1], v.a_tile_size[0], v.a_tile_size[1],
                               (uint32_t)v.a_num_frames, v.a_row_height);
    auto it = masked_cache.find(key);
    if (it != masked_cache.end()) return it->second;
    uint32_t m = 0;
    const double W = d->wall_w, au = v.a_atlas_uv[0], av = v.a_atlas_uv[1], sx = v.a_tile_size[0],
                 sy = v.a_tile_size[1];
    const uint32_t nf = v.a_num_frames == 0 ? 1u : v.a_num_frames;
    for (uint32_t f = 0; f < nf; f++) {
      double u0 = au + f * sx;
      double rows = std::ceil((u0 + sx) / W) - 1.0;
      if (nf == 1) rows = 0;
      const double md = sx > 0 ? (W - au) - sx * std::floor((W - au) / sx) : 0;
      u0 += md * rows;
      const double v0 = av + rows * v.a_row_height;
      const long xlo = (long)std::floor(u0), xhi = (long)std::ceil(u0 + sx), ylo = (long)std::floor(v0),
                 yhi = (long)std::ceil(v0 + sy);  // interior = [xlo, xhi) x [ylo, yhi)
      for (long y = ylo - 1; y <= yhi; y++)
        for (long x = xlo - 1; x <= xhi; x++) {
          const uint32_t xx = (uint32_t)x & (d->wall_w - 1), yy = (uint32_t)y & (d->wall_h - 1);
          if (d->wall_atlas[(size_t)yy * d->wall_w + xx] & 0x8000u)
            m |= (x >= xlo && x < xhi && y >= ylo && y < yhi) ? 2u : 1u;
        }
    }
    masked_cache[key] = m;
    return m;
  };
  for (uint32_t di = 0; di < d->n_draws; di++) {
    const rdoom_draw &dr = d->draws[di];
    if (dr.index_count % 3 != 0) return rdoom::fail(RDOOM_BAD_ARG, "draw %u: index_count not a multiple of 3", di);
    if (dr.object_id >= 4096u) return rdoom::fail(RDOOM_BAD_LEVEL, "draw %u: object id %u (at most 4095)", di, dr.object_id);
    n_objects = std::max(n_objects, dr.object_id + 1u);
    for (uint32_t t = 0; t < dr.index_count / 3; t++) {
      LevelTri lt;
      std::memset(&lt, 0, sizeof lt);
      lt.packed = 1u | (dr.kind << 16);
      if (dr.kind == RDOOM_KIND_FLAT || dr.kind == RDOOM_KIND_WALL) {
        if ((uint64_t)dr.first_index + dr.index_count > d->n_static_indices)
          return rdoom::fail(RDOOM_BAD_ARG, "draw %u: static index range out of bounds", di);
        const rdoom_static_vertex *vv[3];
        for (int i = 0; i < 3; i++) {
          const uint32_t idx = d->static_indices[dr.first_index + 3 * t + i];
          if (idx >= d->n_static_verts) return rdoom::fail(RDOOM_BAD_ARG, "draw %u: vertex index out of bounds", di);
          vv[i] = &d->static_verts[idx];
          std::memcpy(&lt.pos[3 * i], vv[i]->a_pos, 12);
          lt.uv[2 * i] = vv[i]->a_tile_uv[0];
          lt.uv[2 * i + 1] = vv[i]->a_tile_uv[1];
          lt.scroll[i] = vv[i]->a_scroll_rate;
        }
        const rdoom_static_vertex &pv = *vv[2];  // flat varyings: provoking (last) vertex
        lt.atlas_u = pv.a_atlas_uv[0];
        lt.atlas_v = pv.a_atlas_uv[1];
        lt.size_x = pv.a_tile_size[0];
        lt.size_y = pv.a_tile_size[1];
        lt.row_height = pv.a_row_height;
        uint32_t masked = 0;
        if (dr.kind == RDOOM_KIND_WALL) {
          if (!d->wall_atlas) return rdoom::fail(RDOOM_BAD_ARG, "wall draw without a wall atlas");
          masked = region_masked(pv);
        } else if (!d->flat_atlas) {
          return rdoom::fail(RDOOM_BAD_ARG, "flat draw without a flat atlas");
        }
        lt.packed = (uint32_t)pv.a_num_frames | ((uint32_t)pv.a_light << 8) | (dr.kind << 16) |
                    (masked << 18);
      } else if (dr.kind == RDOOM_KIND_SKY) {
        if (!d->sky_texture || !d->sky_w || !d->sky_h) return rdoom::fail(RDOOM_BAD_ARG, "sky draw without a sky texture");
        if ((uint64_t)dr.first_index + dr.index_count > d->n_sky_indices)
          return rdoom::fail(RDOOM_BAD_ARG, "draw %u: sky index range out of bounds", di);
        for (int i = 0; i < 3; i++) {
          const uint32_t idx = d->sky_indices[dr.first_index + 3 * t + i];
          if (idx >= d->n_sky_verts) return rdoom::fail(RDOOM_BAD_ARG, "draw %u: sky vertex out of bounds", di);
          std::memcpy(&lt.pos[3 * i], &d->sky_verts[3 * idx], 12);
        }
      } else if (dr.kind == RDOOM_KIND_DECOR) {
        if ((uint64_t)dr.first_index + dr.index_count > d->n_decor_indices)
          return rdoom::fail(RDOOM_BAD_ARG, "draw %u: decor index range out of bounds", di);
        if (!d->decor_atlas) return rdoom::fail(RDOOM_BAD_ARG, "decor draw without a decor atlas");
        const rdoom_sprite_vertex *vv[3];
        for (int i = 0; i < 3; i++) {
          const uint32_t idx = d->decor_indices[dr.first_index + 3 * t + i];
          if (idx >= d->n_decor_verts) return rdoom::fail(RDOOM_BAD_ARG, "draw %u: decor vertex out of bounds", di);
          vv[i] = &d->decor_verts[idx];
          std::memcpy(&lt.pos[3 * i], vv[i]->a_pos, 12);
          lt.uv[2 * i] = vv[i]->a_tile_uv[0];
          lt.uv[2 * i + 1] = vv[i]->a_tile_uv[1];
          lt.scroll[i] = vv[i]->a_local_x;  // decor triangles carry a_local_x here (sprite.vert:41-42)
        }
        const rdoom_sprite_vertex &pv = *vv[2];
        lt.atlas_u = pv.a_atlas_uv[0];
        lt.atlas_v = pv.a_atlas_uv[1];
        lt.size_x = pv.a_tile_size[0];
        lt.size_y = pv.a_tile_size[1];
        lt.row_height = pv.a_tile_size[1];  // sprite.vert:37 advances animation rows by the tile height
        // sprites are alpha tested per pixel (sprite.frag:20): always the exact path of the rasteriser
        lt.packed = (uint32_t)pv.a_num_frames | ((uint32_t)pv.a_light << 8) | (dr.kind << 16) | (3u << 18);
      } else {
        return rdoom::fail(RDOOM_BAD_ARG, "draw %u: unknown kind %u", di, dr.kind);
      }
      lt.packed |= dr.object_id << 20;
      tris.push_back(lt);
    }
  }
  if (tris.size() >= (1u << 24)) return rdoom::fail(RDOOM_BAD_LEVEL, "too many triangles (%zu)", tris.size());
  out.n_objects = n_objects;
  // clusters for the set-up kernel's coarse cull: runs of at most CLUSTER_TRIS consecutive triangles of one object
  std::vector<Cluster> &clusters = out.clusters;
  for (size_t t = 0; t < tris.size();) {
    const uint32_t obj = tris[t].packed >> 20;
    const bool decor = ((tris[t].packed >> 16) & 3u) == RDOOM_KIND_DECOR;
    Cluster c;
    for (int k = 0; k < 3; k++) c.lo[k] = INFINITY, c.hi[k] = -INFINITY;
    c.first = (uint32_t)t;
    uint32_t n = 0;
    while (t < tris.size() && n < CLUSTER_TRIS && (tris[t].packed >> 20) == obj &&
           (((tris[t].packed >> 16) & 3u) == RDOOM_KIND_DECOR) == decor) {
      for (int v = 0; v < 3; v++)
        for (int k = 0; k < 3; k++) {
          c.lo[k] = std::min(c.lo[k], tris[t].pos[3 * v + k]);
          c.hi[k] = std::max(c.hi[k], tris[t].pos[3 * v + k]);
        }
      t++, n++;
    }
    c.count_object = n | (obj << 8) | (decor ? 0x80000000u : 0u);
    clusters.push_back(c);
  }
  // this level's u16 texels: wall atlas, then (at a multiple of 1024 elements) the flat atlas promoted to u16, the decor atlas
  const size_t wall_n = d->wall_atlas ? (size_t)d->wall_w * d->wall_h : 0;
  const size_t flat_n = d->flat_atlas ? (size_t)d->flat_w * d->flat_h : 0;
  const size_t decor_n = d->decor_atlas ? (size_t)d->decor_w * d->decor_h : 0;
  out.flat_base = (wall_n + 1023) / 1024 * 1024;
  out.decor_base = (out.flat_base + flat_n + 1023) / 1024 * 1024;
  out.texels.assign((out.decor_base + decor_n + 1023) / 1024 * 1024, 0);
  if (wall_n) std::memcpy(out.texels.data(), d->wall_atlas, wall_n * 2);
  for (size_t i = 0; i < flat_n; i++) out.texels[out.flat_base + i] = d->flat_atlas[i];
  if (decor_n) std::memcpy(out.texels.data() + out.decor_base, d->decor_atlas, decor_n * 2);
  if (d->sky_texture && d->sky_w && d->sky_h) out.sky.assign(d->sky_texture, d->sky_texture + (size_t)d->sky_w * d->sky_h);
  out.dims.wall_w = d->wall_w, out.dims.wall_h = d->wall_h;
  out.dims.flat_w = d->flat_w, out.dims.flat_h = d->flat_h;
  out.dims.decor_w = d->decor_atlas ? d->decor_w : 0, out.dims.decor_h = d->decor_atlas ? d->decor_h : 0;
  out.dims.sky_w = out.sky.empty() ? 0 : d->sky_w, out.dims.sky_h = out.sky.empty() ? 0 : d->sky_h;
  out.dims.sky_band = d->sky_tiled_band_size;
  return RDOOM_OK;
}

static rdoom_status levelset_create_impl(const rdoom_level_desc *const *descs, uint32_t n_levels, rdoom_level **out_level) {
  if (!descs || !out_level) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out_level = nullptr;
  if (n_levels == 0 || n_levels > 4096u) return rdoom::fail(RDOOM_BAD_ARG, "n_levels %u outside 1..4096", n_levels);
  std::vector<HostLevel> host(n_levels);
  for (uint32_t k = 0; k < n_levels; k++) {
    if (!descs[k]) return rdoom::fail(RDOOM_BAD_ARG, "level %u: null descriptor", k);
    if (rdoom_status rs = flatten_level(descs[k], host[k])) return rs;
    // the fragment kernel stages ONE COLORMAP in LDS per workgroup: the levels of a set come from one IWAD
    // (build_palette_texture reads the archive's COLORMAP lump, wad/src/tex.rs:137-166)
    if (k && std::memcmp(descs[k]->colormap, descs[0]->colormap, 32 * 256) != 0)
      return rdoom::fail(RDOOM_BAD_ARG, "level %u: its COLORMAP differs from level 0's (the levels of a set share one)", k);
  }
  rdoom_level *lv = new rdoom_level;
  (void)hipGetDevice(&lv->device);
  // the set's arrays: triangles, clusters, texels, skies, level after level
  std::vector<LevelTri> tris;
  std::vector<Cluster> clusters;
  std::vector<uint16_t> texels, sky;
  lv->slices.resize(n_levels);
  uint32_t max_clusters = 0;
  for (uint32_t k = 0; k < n_levels; k++) {
    HostLevel &h = host[k];
    LevelSlice &sl = lv->slices[k];
    sl = h.dims;
    sl.first_tri = (uint32_t)tris.size(), sl.ntri = (uint32_t)h.tris.size();
    sl.first_cluster = (uint32_t)clusters.size(), sl.n_clusters = (uint32_t)h.clusters.size();
    sl.wall_base = (uint32_t)texels.size();
    sl.flat_base = (uint32_t)(texels.size() + h.flat_base), sl.decor_base = (uint32_t)(texels.size() + h.decor_base);
    sl.sky_base = (uint32_t)sky.size();
    for (Cluster c : h.clusters) {
      c.first += sl.first_tri;
      clusters.push_back(c);
    }
    tris.insert(tris.end(), h.tris.begin(), h.tris.end());
    texels.insert(texels.end(), h.texels.begin(), h.texels.end());
    sky.insert(sky.end(), h.sky.begin(), h.sky.end());
    lv->ntri = std::max(lv->ntri, sl.ntri);
    lv->n_objects = std::max(lv->n_objects, h.n_objects);
    lv->slice_objects.push_back(h.n_objects);
    max_clusters = std::max(max_clusters, sl.n_clusters);
    h = HostLevel{};  // (released: a set of many levels would otherwise hold every atlas twice)
  }
  // a record addresses the store with (base >> 10) in 16 bits (ShadeRec::flags), a cluster index fits 32 bits
  if (texels.size() >= ((size_t)1 << 26)) {
    const size_t n = texels.size();
    rdoom_level_destroy(lv);
    return rdoom::fail(RDOOM_BAD_LEVEL, "atlases too large (%zu texels in the set; at most 2^26)", n);
  }
  if (tris.size() >= ((size_t)1 << 31)) {
    rdoom_level_destroy(lv);
    return rdoom::fail(RDOOM_BAD_LEVEL, "too many triangles in the set");
  }
  texels.push_back(0);  // never empty; the fragment kernel's 32-bit load at the last texel's 2-byte-aligned address reads one past it
  texels.push_back(0);
  auto upload = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
    if (bytes == 0 || !src) {
      *dst = nullptr;
      return hipSuccess;
    }
    hipError_t e = hipMalloc(dst, bytes);
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
  };
  hipError_t e = upload(&lv->d_tris, tris.data(), tris.size() * sizeof(LevelTri));
  if (e == hipSuccess) e = upload(&lv->d_clusters, clusters.data(), clusters.size() * sizeof(Cluster));
  if (e == hipSuccess) e = upload(&lv->d_texels, texels.data(), texels.size() * 2);
  if (e == hipSuccess) e = upload(&lv->d_sky, sky.data(), sky.size() * 2);
  if (e == hipSuccess) e = upload(&lv->d_cmap, descs[0]->colormap, 32 * 256);
  if (e == hipSuccess) e = upload(&lv->d_slices, lv->slices.data(), lv->slices.size() * sizeof(LevelSlice));
  if (e != hipSuccess) {
    rdoom_level_destroy(lv);
    return rdoom::fail(e == hipErrorOutOfMemory ? RDOOM_OOM : RDOOM_HIP_ERROR, "level upload failed: %s",
                       hipGetErrorString(e));
  }
  lv->view.tris = (const LevelTri *)lv->d_tris;
  lv->view.clusters = (const Cluster *)lv->d_clusters;
  lv->view.texels = (const uint16_t *)lv->d_texels;
  lv->view.sky_texels = (const uint16_t *)lv->d_sky;
  lv->view.colormap = (const uint8_t *)lv->d_cmap;
  lv->view.slices = (const LevelSlice *)lv->d_slices;
  lv->view.n_slices = n_levels;
  lv->view.max_clusters = max_clusters;
  lv->view.max_ntri = lv->ntri;
  *out_level = lv;
  return RDOOM_OK;
}

void rdoom_batch_destroy(rdoom_batch *b) {
  if (!b) return;
#ifdef RDOOM_HOST_TIMERS
  if (g_host_n) {
    static const char *names[12] = {"wait for the staging buffer", "per-pose constants", "H2D + event", "flag memset + set-up launches", "binning launch", "rasteriser launch",
                                    "fragment + fixup launches", "closing event", "", "", "", ""};
    fprintf(stderr, "[host timers] %llu renders, microseconds per render:", g_host_n);
    for (int k = 0; k < 8; k++) fprintf(stderr, "  %s %.1f", names[k], g_host_t[k] / (double)g_host_n);
    fprintf(stderr, "\n");
    g_host_n = 0;
    for (auto &t : g_host_t) t = 0;
  }
#endif
  for (void *p : {(void *)b->d_poses, (void *)b->d_recs, (void *)b->d_visible, (void *)b->d_tile_hdr, (void *)b->d_entries, (void *)b->d_hits,
                  (void *)b->d_overflow, (void *)b->d_zeroed, (void *)b->d_fix_list, (void *)b->d_vis,
                  (void *)b->d_prim, (void *)b->d_fb, (void *)b->d_qtab, b->d_frag_const})
    if (p) (void)hipFree(p);
  for (auto &e : b->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto &slot : b->ring)
    for (auto &e : slot)
      if (e) (void)hipEventDestroy(e);
  for (auto &e : b->ev_copy)
    if (e) (void)hipEventDestroy(e);
  if (b->ev_done) (void)hipEventDestroy(b->ev_done);
  if (b->copy_stream) (void)hipStreamDestroy(b->copy_stream);
  for (auto &h : b->h_poses)
    if (h) (void)hipHostFree(h);
  for (auto &h : b->h_objects)
    if (h) (void)hipHostFree(h);
  if (b->d_ndc) (void)hipFree(b->d_ndc);
  if (b->d_objects) (void)hipFree(b->d_objects);
  delete b;
}

rdoom_status rdoom_batch_create(const rdoom_level *level, uint32_t width, uint32_t height, uint32_t max_poses,
                                rdoom_batch **out_batch) {
  if (!level || !out_batch) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out_batch = nullptr;
  if (width == 0 || height == 0 || max_poses == 0 || width > 16384 || height > 16384)
    return rdoom::fail(RDOOM_BAD_ARG, "bad frame size %ux%u (1..16384 on a side) or max_poses %u", width, height, max_poses);
  HIP_TRY(hipSetDevice(level->device));  // the batch lives on the level's device, whatever the caller's current one is
  rdoom_batch *b = new rdoom_batch;
  b->level = level;
  b->width = width;
  b->height = height;
  b->pitch = std::max(8u, width % 4u == 0u ? width : ((width + 7u) & ~7u));
  b->max_poses = max_poses;
  b->cap = level->ntri ? level->ntri : 1;
  const rdoom::DebugOptions &dbg = rdoom::debug_options();
  b->vis16 = b->cap < 0xFFFFu && !dbg.vis32;
  const size_t npx = (size_t)b->pitch * height * max_poses;
  hipError_t e = hipMalloc((void **)&b->d_poses, sizeof(PoseConst) * max_poses);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_recs, sizeof(TriRec) * (size_t)b->cap * max_poses);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_visible, sizeof(uint32_t) * (size_t)b->cap * max_poses);
  b->n_tiles = ((width + TILE_W - 1) / TILE_W) * ((height + TILE_H - 1) / TILE_H);
  // tile-list entries per pose; beyond it the pose is rasterised from its sorted list (slow: every tile scans every visible
  // triangle).  A long tile's list is stored per quadrant (bin.hip), an entry once per quadrant it touches, so the array is
  // sized generously -- never beyond what the level could ever need: every triangle in every quadrant of every tile
  // (the floor scales with the frame: 256 entries per tile, at least 16 384 -- 131 072 for the 510 tiles of 1080p as before,
  // 16 384 instead of 131 072 for the 20 tiles of 320 x 200, whose 8 192-pose batches had 12.9 GB of binning scratch)
  // -- and with the level: a split list stores an entry once per quadrant it touches, so a large level at a small frame (the 10x
  // level, 36 k triangles, at 320 x 200) needs more than 16 384 per pose or every such pose silently takes the overflow path)
  const uint64_t floor_entries = std::max<uint64_t>(std::max<uint64_t>(16384u, 256u * (uint64_t)b->n_tiles), std::min<uint64_t>(2u * (uint64_t)b->cap, 1u << 20));
  b->entry_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(floor_entries, 64u * (uint64_t)b->n_tiles),
                                              (uint64_t)b->cap * b->n_tiles * 4u + 8u * (uint64_t)b->n_tiles);
  b->entry_cap = (b->entry_cap + 3u) & ~3u;
  if (dbg.entry_cap > 0) b->entry_cap = (uint32_t)dbg.entry_cap;  // tests: force that fallback
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_tile_hdr, sizeof(uint2) * (size_t)b->n_tiles * max_poses);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_entries, sizeof(uint32_t) * (size_t)b->entry_cap * max_poses);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_hits, sizeof(uint2) * (size_t)b->entry_cap * max_poses);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_overflow, sizeof(uint32_t) * max_poses);
  const size_t counts_words = ((size_t)max_poses + 3u) & ~(size_t)3u;
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_zeroed, sizeof(uint32_t) * (4u + counts_words) + setup_histogram_bytes(max_poses));
  if (e == hipSuccess) b->d_fix_count = b->d_zeroed, b->d_counts = b->d_zeroed + 4, b->d_ghist = b->d_zeroed + 4 + counts_words;
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_fix_list, sizeof(uint2) * (size_t)b->fix_cap);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_vis, (b->vis16 ? sizeof(uint16_t) : sizeof(uint32_t)) * npx);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_fb, npx);
  if (e == hipSuccess) e = hipMalloc(&b->d_frag_const, fragment_const_bytes());
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_qtab, sizeof(uint32_t) * 4u * (size_t)b->n_tiles * max_poses);
  // (the rasteriser never writes the entries of quadrants that lie outside the frame: NONE from the start, so that the
  // table is self-consistent for every reader)
  if (e == hipSuccess) e = hipMemset(b->d_qtab, 0xFF, sizeof(uint32_t) * 4u * (size_t)b->n_tiles * max_poses);
  if (e == hipSuccess) {  // sky.frag:13's ndc per column / row, same two operations as the per-pixel form
    // (a run of sky that ends in the padding columns of a padded row pitch reads up to 7 values past the columns: row values, or
    // the zeros the table is extended by -- colours of pixels nobody reads)
    std::vector<float> ndc(std::max(width + height, b->pitch), 0.0f);
    for (uint32_t i = 0; i < width; i++) ndc[i] = ((float)i + 0.5f) / (0.5f * (float)width) - 1.0f;
    for (uint32_t i = 0; i < height; i++) ndc[width + i] = ((float)i + 0.5f) / (0.5f * (float)height) - 1.0f;
    e = hipMalloc((void **)&b->d_ndc, sizeof(float) * ndc.size());
    if (e == hipSuccess) e = hipMemcpy(b->d_ndc, ndc.data(), sizeof(float) * ndc.size(), hipMemcpyHostToDevice);
  }
  for (auto &ev : b->ev)
    if (e == hipSuccess) e = hipEventCreate(&ev);
  for (auto &ev : b->ev_copy)
    if (e == hipSuccess) e = hipEventCreate(&ev);
  if (e == hipSuccess) e = hipEventCreate(&b->ev_done);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking);
  for (auto &h : b->h_poses)
    if (e == hipSuccess) e = hipHostMalloc((void **)&h, sizeof(PoseConst) * max_poses, hipHostMallocDefault);
  if (e != hipSuccess) {
    rdoom_batch_destroy(b);
    return rdoom::fail(e == hipErrorOutOfMemory ? RDOOM_OOM : RDOOM_HIP_ERROR, "batch allocation failed: %s",
                       hipGetErrorString(e));
  }
  *out_batch = b;
  return RDOOM_OK;
}

static void mat_mul_v1(const float *P, const float *M, float *pm) {  // V1: PM = P * M, plain multiply/add, left to right
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++)
      pm[c * 4 + r] = ((P[0 * 4 + r] * M[c * 4 + 0] + P[1 * 4 + r] * M[c * 4 + 1]) + P[2 * 4 + r] * M[c * 4 + 2]) +
                      P[3 * 4 + r] * M[c * 4 + 3];
}

// Records ev_done on every way out of render_impl once kernels may have been queued: rdoom_batch_finish and the read functions
// wait for THAT event, and a render that failed half-way must not leave them waiting for the render before it while its
// own kernels still write the batch's scratch.
namespace {
struct DoneGuard {
  rdoom_batch *b;
  hipStream_t st;
  bool armed = false;
  ~DoneGuard() {
    if (armed) (void)hipEventRecord(b->ev_done, st);
  }
};
}  // namespace

static rdoom_status render_impl(rdoom_batch *b, const rdoom_pose *poses, const uint8_t *lights, uint32_t lights_stride,
                                uint32_t n, uint32_t kinds_mask, hipStream_t st, rdoom_timings *tm,
                                const float *object_modelviews = nullptr, uint32_t n_objects = 0, bool profiled = false,
                                const uint32_t *level_of_pose = nullptr) {
  if (!b || !poses || !lights) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  if (n == 0 || n > b->max_poses) return rdoom::fail(RDOOM_BAD_ARG, "n_poses %u outside 1..%u", n, b->max_poses);
  const rdoom_level *lv = b->level;
  if (level_of_pose)
    for (uint32_t p = 0; p < n; p++)
      if (level_of_pose[p] >= lv->view.n_slices)
        return rdoom::fail(RDOOM_BAD_ARG, "pose %u names level %u of a set of %u", p, level_of_pose[p], lv->view.n_slices);
  HIP_TRY(hipSetDevice(lv->device));  // level, scratch and kernels on one device (several GPUs driven from one process)
  if (object_modelviews && n_objects < lv->n_objects)
    return rdoom::fail(RDOOM_BAD_ARG, "n_objects %u but the level draws objects 0..%u", n_objects, lv->n_objects - 1);
  HT_DECL
  const uint32_t sg = b->stage;  // this render's staging buffers: the H2D copy that last read them is two renders back
  b->stage = (sg + 1u) % rdoom_batch::STAGES;
  HIP_TRY(hipEventSynchronize(b->ev_copy[sg]));
  HT_MARK(0);
  PoseConst *h_poses = b->h_poses[sg];
  if (object_modelviews) {
    const size_t count = (size_t)b->max_poses * lv->n_objects;
    if (!b->d_objects) HIP_TRY(hipMalloc((void **)&b->d_objects, sizeof(ObjectConst) * count));
    for (auto &h : b->h_objects)
      if (!h) HIP_TRY(hipHostMalloc((void **)&h, sizeof(ObjectConst) * count, hipHostMallocDefault));
    ObjectConst *h_objects = b->h_objects[sg];
    for (uint32_t p = 0; p < n; p++)
      for (uint32_t o = 0; o < lv->n_objects; o++) {
        ObjectConst &oc = h_objects[(size_t)p * lv->n_objects + o];
        const float *M = object_modelviews + ((size_t)p * n_objects + o) * 16;
        mat_mul_v1(poses[p].projection, M, oc.pm);
        std::memcpy(oc.mv, M, sizeof oc.mv);
        oc.vr0 = atan2f(oc.pm[8], oc.pm[10]);  // sky.vert:10-12
        oc.vr1 = oc.pm[9] / oc.pm[11];
        oc.pad0 = oc.pad1 = 0;
      }
  }
  for (uint32_t p = 0; p < n; p++) {  // V1: PM = P * M, plain multiply/add, left to right
    PoseConst &pc = h_poses[p];
    const float *P = poses[p].projection, *M = poses[p].modelview;
    mat_mul_v1(P, M, pc.pm);
    std::memcpy(pc.mv, M, sizeof pc.mv);
    std::memcpy(pc.proj, P, sizeof pc.proj);
    pc.time = poses[p].time;
    pc.vr0 = atan2f(pc.pm[8], pc.pm[10]);  // sky.vert:10-12
    pc.vr1 = pc.pm[9] / pc.pm[11];
    pc.zk = pc.proj[11] != 0.0f ? pc.proj[10] / pc.proj[11] : 0.0f;  // S5: Z - zk * W is small for a perspective matrix
    std::memcpy(pc.lights, lights + (size_t)p * lights_stride, 256);
    pc.level = level_of_pose ? level_of_pose[p] : 0u;
    pc.pad0 = pc.pad1 = pc.pad2 = 0u;
  }
  b->last_n = n;
  HT_MARK(1);
  hipEvent_t *ev = b->ev;  // the four marks of this render: the batch's own, or a slot of the ring when nobody waits
  if (profiled) {
    if (b->ring_n == rdoom_batch::RING) return rdoom::fail(RDOOM_BAD_ARG, "%u profiled renders pending: collect the timings first", b->ring_n);
    ev = b->ring[b->ring_n];
    for (int k = 0; k < 4; k++)
      if (!ev[k]) HIP_TRY(hipEventCreate(&ev[k]));
  }
  const bool marks = tm || profiled;
  if (marks) HIP_TRY(hipEventRecord(ev[0], st));
  DoneGuard done{b, st};
  done.armed = true;  // from here on work of this render may be queued: whatever happens, ev_done is recorded after it
  HIP_TRY(hipMemcpyAsync(b->d_poses, h_poses, sizeof(PoseConst) * n, hipMemcpyHostToDevice, st));
  if (object_modelviews)
    HIP_TRY(hipMemcpyAsync(b->d_objects, b->h_objects[sg], sizeof(ObjectConst) * (size_t)n * lv->n_objects,
                           hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(b->ev_copy[sg], st));
  HT_MARK(2);
  const int W = (int)b->width, H = (int)b->height, PITCH = (int)b->pitch;
  // fix_count[0..2], the poses' visible-triangle counts and depth-bucket histograms (of the first n poses): one fill
  HIP_TRY(hipMemsetAsync(b->d_zeroed, 0, (size_t)((const char *)b->d_ghist - (const char *)b->d_zeroed) + setup_histogram_bytes(n), st));
  if (lv->ntri)
    if (rdoom_status rs = launch_setup(st, n, lv->view, b->d_poses, object_modelviews ? (const ObjectConst *)b->d_objects : nullptr,
                                       lv->n_objects, W, H, kinds_mask, b->d_recs, b->d_visible, b->d_counts,
                                       b->d_ghist, b->cap, b->d_fix_count + 2))
      return rs;
  HT_MARK(3);
  const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
  bool split_lists = false;  // long tile lists stored per quadrant this render (binning kernel and rasteriser must agree)
  bool bins = false;
  if (lv->ntri && !rdoom::debug_options().no_bins)
    if (rdoom_status rs = launch_bin(st, n, b->d_recs, b->d_counts, b->cap, tiles_x, tiles_y, b->d_tile_hdr, b->d_entries,
                                     b->entry_cap, b->d_hits, b->d_overflow, !rdoom::debug_options().no_split, &bins, &split_lists))
      return rs;
  if (!bins) {
    HIP_TRY(hipMemsetAsync(b->d_overflow, 0xFF, sizeof(uint32_t) * n, st));
  }
  if (marks) HIP_TRY(hipEventRecord(ev[1], st));
  HT_MARK(4);
  uint32_t *prim_out = b->want_prim ? b->d_prim : nullptr;
  // one reading of the debug hooks for both kernels: who writes and who reads visibility words must not change in between
  const FragmentPlan plan = plan_fragment(W, PITCH, H, b->d_qtab != nullptr);
  // (the rasteriser's frame is PITCH pixels wide: the padding columns of a width that is not a multiple of 4 are pixels no
  // bounding box reaches -- they stay uncovered, and the quadrants they lie in simply never count as covered)
  if (rdoom_status rs = launch_raster(st, n, lv->view, b->d_recs, b->d_counts, b->cap, PITCH, H, tiles_x, tiles_y,
                                      b->d_tile_hdr, b->d_entries, b->entry_cap, b->d_overflow, b->d_vis, b->vis16, prim_out, b->d_qtab,
                                      plan.skip_described_vis, split_lists, bins))
    return rs;
  if (marks) HIP_TRY(hipEventRecord(ev[2], st));
  HT_MARK(5);
  if (rdoom_status rs = launch_fragment(st, n, lv->view, b->d_recs, b->d_counts, b->cap, b->d_poses, W, PITCH, H, tiles_x,
                                        tiles_y, b->d_tile_hdr, b->d_entries, b->entry_cap, b->d_overflow, b->d_vis, b->vis16,
                                        prim_out, b->d_ndc, b->d_fb, b->d_fix_count, b->d_fix_list, b->fix_cap, b->d_qtab, b->d_frag_const,
                                        &b->frag_const_ready, plan))
    return rs;
  HIP_TRY(hipGetLastError());
  HT_MARK(6);
  done.armed = false;
  HIP_TRY(hipEventRecord(b->ev_done, st));
  HT_MARK(7);
#ifdef RDOOM_HOST_TIMERS
  g_host_n++;
#endif
  if (profiled) {
    HIP_TRY(hipEventRecord(ev[3], st));
    b->ring_n++;
    b->ring_poses += n;
  }
  if (tm) {
    HIP_TRY(hipEventRecord(b->ev[3], st));
    HIP_TRY(hipEventSynchronize(b->ev[3]));
    HIP_TRY(hipEventElapsedTime(&tm->setup_ms, b->ev[0], b->ev[1]));
    HIP_TRY(hipEventElapsedTime(&tm->raster_ms, b->ev[1], b->ev[2]));
    HIP_TRY(hipEventElapsedTime(&tm->fragment_ms, b->ev[2], b->ev[3]));
    HIP_TRY(hipEventElapsedTime(&tm->total_ms, b->ev[0], b->ev[3]));
    tm->pixels = (uint64_t)n * W * H;
    std::vector<uint32_t> counts(n);
    HIP_TRY(read_back(b, counts.data(), b->d_counts, sizeof(uint32_t) * n));
    tm->visible_triangles = 0;
    for (uint32_t c : counts) tm->visible_triangles += c;
    uint32_t fixups = 0;
    const rdoom_status fs = device_flags(b, &fixups);
    tm->fixup_pixels = fixups;
    if (fs) return fs;
  }
  return RDOOM_OK;
}

rdoom_status rdoom_batch_render(rdoom_batch *batch, const rdoom_pose *poses, const uint8_t *lights,
                                uint32_t lights_stride, uint32_t n_poses, uint32_t kinds_mask, void *stream) {
  return render_impl(batch, poses, lights, lights_stride, n_poses, kinds_mask, (hipStream_t)stream, nullptr);
}

rdoom_status rdoom_batch_render_timed(rdoom_batch *batch, const rdoom_pose *poses, const uint8_t *lights,
                                      uint32_t lights_stride, uint32_t n_poses, uint32_t kinds_mask, void *stream,
                                      rdoom_timings *out) {
  if (!out) return rdoom::fail(RDOOM_BAD_ARG, "out is null");
  return render_impl(batch, poses, lights, lights_stride, n_poses, kinds_mask, (hipStream_t)stream, out);
}

rdoom_status rdoom_batch_render_profiled(rdoom_batch *batch, const rdoom_pose *poses, const uint8_t *lights,
                                         uint32_t lights_stride, uint32_t n_poses, uint32_t kinds_mask, void *stream) {
  return render_impl(batch, poses, lights, lights_stride, n_poses, kinds_mask, (hipStream_t)stream, nullptr, nullptr, 0, true);
}

rdoom_status rdoom_batch_collect_timings(rdoom_batch *b, rdoom_timings *out, uint32_t *out_renders) {
  if (!b || !out) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out = rdoom_timings{};
  if (out_renders) *out_renders = b->ring_n;
  if (b->ring_n == 0) return RDOOM_OK;
  HIP_TRY(bind_device(b));
  // profiled renders may have been queued on several streams (bench.py --streams): nothing orders an earlier slot's
  // events against the last one's, so every slot's closing event is waited for
  for (uint32_t i = 0; i < b->ring_n; i++) HIP_TRY(hipEventSynchronize(b->ring[i][3]));
  for (uint32_t i = 0; i < b->ring_n; i++) {
    float a = 0, r = 0, f = 0, t = 0;
    HIP_TRY(hipEventElapsedTime(&a, b->ring[i][0], b->ring[i][1]));
    HIP_TRY(hipEventElapsedTime(&r, b->ring[i][1], b->ring[i][2]));
    HIP_TRY(hipEventElapsedTime(&f, b->ring[i][2], b->ring[i][3]));
    HIP_TRY(hipEventElapsedTime(&t, b->ring[i][0], b->ring[i][3]));
    out->setup_ms += a, out->raster_ms += r, out->fragment_ms += f, out->total_ms += t;
  }
  out->pixels = (uint64_t)b->ring_poses * b->width * b->height;
  std::vector<uint32_t> counts(b->last_n);  // of the last render
  HIP_TRY(read_back(b, counts.data(), b->d_counts, sizeof(uint32_t) * b->last_n));
  for (uint32_t c : counts) out->visible_triangles += c;
  uint32_t fixups = 0;
  const rdoom_status fs = device_flags(b, &fixups);
  out->fixup_pixels = fixups;
  b->ring_n = 0, b->ring_poses = 0;
  return fs;
}

rdoom_status rdoom_batch_render_objects(rdoom_batch *batch, const rdoom_pose *poses, const uint8_t *lights,
                                        uint32_t lights_stride, uint32_t n_poses, uint32_t kinds_mask, void *stream,
                                        const float *object_modelviews, uint32_t n_objects) {
  if (!object_modelviews) return rdoom::fail(RDOOM_BAD_ARG, "object_modelviews is null");
  return render_impl(batch, poses, lights, lights_stride, n_poses, kinds_mask, (hipStream_t)stream, nullptr,
                     object_modelviews, n_objects);
}

rdoom_status rdoom_batch_render_levels(rdoom_batch *batch, const rdoom_pose *poses, const uint32_t *level_of_pose, const uint8_t *lights,
                                       uint32_t lights_stride, uint32_t n_poses, uint32_t kinds_mask, void *stream,
                                       const float *object_modelviews, uint32_t n_objects, uint32_t flags) {
  if (!level_of_pose) return rdoom::fail(RDOOM_BAD_ARG, "level_of_pose is null");
  if (flags & ~(uint32_t)RDOOM_RENDER_PROFILED) return rdoom::fail(RDOOM_BAD_ARG, "unknown flags 0x%x", flags);
  return render_impl(batch, poses, lights, lights_stride, n_poses, kinds_mask, (hipStream_t)stream, nullptr, object_modelviews,
                     n_objects, (flags & RDOOM_RENDER_PROFILED) != 0u, level_of_pose);
}

rdoom_status rdoom_level_num_levels(const rdoom_level *level, uint32_t *out) {
  if (!level || !out) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out = level->view.n_slices;
  return RDOOM_OK;
}

rdoom_status rdoom_level_num_objects(const rdoom_level *level, uint32_t *out) {
  if (!level || !out) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out = level->n_objects;
  return RDOOM_OK;
}

rdoom_status rdoom_batch_finish(rdoom_batch *b) {
  if (!b) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  HIP_TRY(bind_device(b));
  return device_flags(b);  // (waits for the batch's last render on its own stream: read_back)
}

rdoom_status rdoom_batch_path_stats(rdoom_batch *b, rdoom_path_stats *out) {
  if (!b || !out) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out = rdoom_path_stats{};
  HIP_TRY(bind_device(b));
  if (rdoom_status fs = device_flags(b)) return fs;
  const uint32_t n = b->last_n, T = b->n_tiles;
  out->poses = n;
  if (n == 0) return RDOOM_OK;
  try {
    std::vector<uint32_t> over(n), qtab((size_t)n * T * 4u);
    std::vector<uint2> hdr((size_t)n * T);
    HIP_TRY(read_back(b, over.data(), b->d_overflow, sizeof(uint32_t) * n));
    HIP_TRY(read_back(b, hdr.data(), b->d_tile_hdr, sizeof(uint2) * hdr.size()));
    HIP_TRY(read_back(b, qtab.data(), b->d_qtab, sizeof(uint32_t) * qtab.size()));
    const uint32_t tiles_x = (b->width + TILE_W - 1) / TILE_W;
    for (uint32_t p = 0; p < n; p++) {
      const bool binned = over[p] == 0u;
      out->bins_overflowed_poses += binned ? 0u : 1u;
      for (uint32_t t = 0; t < T; t++) {
        if (binned) {  // (the headers of a pose that overflowed are stale)
          const uint2 h = hdr[(size_t)p * T + t];
          out->split_tiles += (h.y & TILE_SPLIT) ? 1u : 0u;
          out->tile_entries += h.y & ~TILE_SPLIT;
        }
        const uint32_t tx = (t % tiles_x) * TILE_W, ty = (t / tiles_x) * TILE_H;
        for (uint32_t q = 0; q < 4u; q++) {
          if (tx + (q & 1u) * 32u >= b->pitch || ty + (q >> 1) * 32u >= b->height) continue;  // outside the frame: never written
          out->quadrants++;
          out->described_quadrants += qtab[((size_t)p * T + t) * 4u + q] != NONE ? 1u : 0u;
        }
      }
    }
    out->tiles = (uint64_t)n * T;
  } catch (const std::bad_alloc &) {
    return rdoom::fail(RDOOM_OOM, "out of host memory");
  }
  return RDOOM_OK;
}

rdoom_status rdoom_batch_framebuffer_device(const rdoom_batch *batch, uint8_t **out_device_ptr) {
  if (!batch || !out_device_ptr) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out_device_ptr = batch->d_fb;
  return RDOOM_OK;
}

rdoom_status rdoom_batch_framebuffer_pitch(const rdoom_batch *batch, uint32_t *out_pitch) {
  if (!batch || !out_pitch) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out_pitch = batch->pitch;
  return RDOOM_OK;
}

rdoom_status rdoom_batch_read_framebuffer(rdoom_batch *b, uint32_t first, uint32_t count, uint8_t *host_out) {
  if (!b || !host_out) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  if ((uint64_t)first + count > b->last_n) return rdoom::fail(RDOOM_BAD_ARG, "frame range outside the last render");
  const size_t frame = (size_t)b->pitch * b->height;
  HIP_TRY(bind_device(b));
  if (rdoom_status fs = device_flags(b)) return fs;
  if (count) HIP_TRY(read_back(b, host_out, b->d_fb + frame * first, b->width, (size_t)b->height * count, b->pitch));
  return RDOOM_OK;
}

rdoom_status rdoom_batch_enable_primitive_ids(rdoom_batch *b) {
  if (!b) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  HIP_TRY(bind_device(b));  // the ids live next to the batch's other scratch, on the level's device
  if (!b->d_prim) {
    const size_t npx = (size_t)b->pitch * b->height * b->max_poses;
    HIP_TRY(hipMalloc((void **)&b->d_prim, sizeof(uint32_t) * npx));
  }
  b->want_prim = true;
  b->last_n = 0;  // nothing captured yet: render first
  return RDOOM_OK;
}

rdoom_status rdoom_batch_read_primitive_ids(rdoom_batch *b, uint32_t first, uint32_t count, uint32_t *host_out) {
  if (!b || !host_out) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  if (!b->want_prim || !b->d_prim)
    return rdoom::fail(RDOOM_BAD_ARG, "primitive ids are not captured: call rdoom_batch_enable_primitive_ids, then render");
  if ((uint64_t)first + count > b->last_n) return rdoom::fail(RDOOM_BAD_ARG, "frame range outside the last render");
  const size_t frame = (size_t)b->pitch * b->height;
  HIP_TRY(bind_device(b));
  if (count)
    HIP_TRY(read_back(b, host_out, b->d_prim + frame * first, sizeof(uint32_t) * b->width, (size_t)b->height * count, sizeof(uint32_t) * b->pitch));
  return RDOOM_OK;
}

}  // extern "C"
