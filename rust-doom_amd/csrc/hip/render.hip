// Pose-batch renderer for gfx950 (MI355X): per-pose triangle setup, tiled rasteriser with an LDS
// triangle queue and register-resident depth/coverage, and the PLAYPAL/COLORMAP fragment kernel.
//
// Replaces the reference's GL draw path: assets/shaders/static.{vert,frag}, sky.{vert,frag} and the
// fixed-function state of engine/src/renderer.rs:49-57 + engine/src/window.rs:12,40-44.  The
// arithmetic every kernel must reproduce is specified in DESIGN.md "Raster arithmetic" (steps
// V1.., S1.., R1.., F1..); operation order below follows that text, not the oracle's source.
// Built with -ffp-contract=off: a*b+c is never fused unless written as fmaf.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <type_traits>
#include <vector>

#include "../common.hpp"
#include "fastmath.hpp"

#pragma clang fp contract(off)

namespace {
using namespace rdoom_fm;

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int TILE_W = 64, TILE_H = 64;  // one 256-thread workgroup: 4 waves x 32x32 quadrant, 4x4 pixels per lane

// ---- level-constant triangle record (built once per level on the host) -----------------------
struct alignas(16) LevelTri {  // 96 bytes
  float pos[9];
  float uv[6];
  float scroll[3];              // a_scroll_rate per vertex; for decor triangles: a_local_x per vertex
  float atlas_u, atlas_v, size_x, size_y, row_height;
  uint32_t packed;  // num_frames | light << 8 | kind << 16 | masked border << 18 | masked interior << 19 | object id << 20
};
static_assert(sizeof(LevelTri) == 96, "LevelTri layout");

struct alignas(16) PoseConst {  // 464 bytes
  float pm[16];                 // projection * modelview (V1)
  float time, vr0, vr1, pad;
  uint8_t lights[256];
  float mv[16], proj[16];       // the two uniforms themselves: sprite.vert transforms in two steps (D1..D3)
};
static_assert(sizeof(PoseConst) == 464, "PoseConst layout");

// Per (pose, object) uniforms when objects move (doors, lifts): the reference sets u_modelview = view o model
// transform for the draws of each object (engine/src/renderer.rs:120-132, game/src/level.rs:203-255).
struct alignas(16) ObjectConst {  // 144 bytes
  float pm[16];                   // projection * object modelview (V1)
  float mv[16];                   // object modelview
  float vr0, vr1, pad0, pad1;     // sky.vert:10-12 from this object's transform
};
static_assert(sizeof(ObjectConst) == 144, "ObjectConst layout");

// ---- per (pose, visible triangle) records -----------------------------------------------------
struct alignas(16) RasterRec {  // 80 bytes
  float e[9];                   // edge functions A,B,C x3
  float zp[3];                  // window-depth plane
  float wp[3];                  // 1/w plane
  uint32_t bb0, bb1;            // x0 | y0 << 16, x1 | y1 << 16 (inclusive)
  uint32_t flags;               // prim id (24 bits) | tl << 24 | kind << 27 | RASTER_MASKED_*
  uint32_t pad[2];
};
static_assert(sizeof(RasterRec) == 80 && offsetof(RasterRec, zp) == 36 && offsetof(RasterRec, bb0) == 60, "RasterRec layout");

constexpr uint32_t RASTER_MASKED_BORDER = 1u << 29;    // a texel bordering the texture rectangle is transparent
constexpr uint32_t RASTER_MASKED_INTERIOR = 1u << 30;  // the texture rectangle itself has transparent texels
constexpr uint32_t RASTER_MASKED_ANY = RASTER_MASKED_BORDER | RASTER_MASKED_INTERIOR;
constexpr uint32_t SHADE_POW2_X = 1u << 2, SHADE_POW2_Y = 1u << 3;
// eligible for the fragment kernel's packed path: flat or wall whose tile sizes are each a power of two in
// [2^-20, 2^20] or an integer in [1, 4096]; SHADE_NP2 = at least one of them is not a power of two
constexpr uint32_t SHADE_FAST = 1u << 4, SHADE_NP2 = 1u << 5;

struct alignas(16) ShadeRec {  // 64 bytes
  float wp[3];
  float up[3];
  float vp[3];
  float atlas_u, atlas_v, size_x, size_y, light;
  uint32_t flags;  // kind (2 bits) | SHADE_POW2_X | SHADE_POW2_Y | SHADE_FAST | log2(atlas width) << 8 | (texel base >> 10) << 16
  uint32_t tex;    // (atlas width - 1) | (atlas height - 1) << 16
};
static_assert(sizeof(ShadeRec) == 64, "ShadeRec layout");

struct alignas(16) TriRec {  // 144 bytes = 9 x 16 B: what one (pose, visible triangle) carries
  RasterRec r;
  ShadeRec s;
};
static_assert(sizeof(TriRec) == 144, "TriRec layout");

struct DeviceLevelView {
  const LevelTri *tris;
  uint32_t ntri;
  // one u16 texel store: the wall atlas (lo = palette index, bit 15 = transparent) followed, at element
  // flat_base (a multiple of 1024), by the flat atlas promoted to u16 (hi byte 0: never transparent)
  const uint16_t *texels;
  uint32_t flat_base, decor_base;  // the decor (sprite) atlas follows the flats, also at a multiple of 1024
  uint32_t flat_w, flat_h;
  uint32_t wall_w, wall_h;
  uint32_t decor_w, decor_h;
  const uint16_t *sky_tex;
  uint32_t sky_w, sky_h;
  float sky_band;
  const uint8_t *colormap;
};

__device__ __forceinline__ float plane3(const float *p, float px, float py) {
  return fmaf(p[0], px, fmaf(p[1], py, p[2]));
}
__device__ __forceinline__ float dop(float a, float b, float c, float d) {
  float p = a * b;
  float q = c * d;
  return p - q;
}
__device__ __forceinline__ float glsl_mod(float x, float y) { return x - y * floorf(x / y); }

// =================================================================================================
// Kernel 1: vertex stage + triangle setup (V2..V5, S1..S6) and front-to-back ordering.
// One 256-thread workgroup per pose walks the level's triangle list in chunks; visible triangles are
// compacted in order (ballot + prefix) into the pose's record array, then a counting sort over a
// log-depth bucket (exponent + top mantissa bits of the nearest vertex's w) produces the `sorted`
// list the rasteriser consumes: near geometry first, so its exact early-z test rejects most occluded
// triangles.  The order only affects speed: the winner is the lexicographic min of (d24, primitive).
// =================================================================================================
constexpr uint32_t SORT_BUCKETS = 2048;
constexpr uint32_t SORT_KEY_CAP = 8192;  // visible triangles per pose whose keys fit the LDS key array (more: unsorted, still correct)

__device__ __forceinline__ bool setup_triangle(const DeviceLevelView &lv, const PoseConst &pc,
                                               const ObjectConst *__restrict__ objs, uint32_t t, int width,
                                               int height, uint32_t kinds_mask, RasterRec &rr, ShadeRec &sr,
                                               float &wkey) {
  wkey = 0.0f;
  bool ok = t < lv.ntri;
  if (ok) {
    const LevelTri tri = lv.tris[t];
    const uint32_t kind = (tri.packed >> 16) & 3u;
    ok = ((kinds_mask >> kind) & 1u) != 0u;
    if (ok) {
      // uniforms of this triangle's object: the pose's own unless objects move
      const float *pm = pc.pm, *mv = pc.mv;
      float vr0 = pc.vr0, vr1 = pc.vr1;
      if (objs) {
        const ObjectConst &oc = objs[tri.packed >> 20];
        pm = oc.pm, mv = oc.mv, vr0 = oc.vr0, vr1 = oc.vr1;
      }
      float clip[3][4], u[3], v[3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const float x = tri.pos[3 * i], y = tri.pos[3 * i + 1], z = tri.pos[3 * i + 2];
        if (kind == RDOOM_KIND_DECOR) {
          // D1..D3 (sprite.vert:41-46): camera-facing expansion along row 0 of the modelview, then
          // projection * (modelview * pos) in two steps
          const float lx = tri.scroll[i];
          const float px = fmaf(mv[0], lx, x), py = fmaf(mv[4], lx, y), pz = fmaf(mv[8], lx, z);
          float eye[4];
#pragma unroll
          for (int r = 0; r < 4; r++) eye[r] = fmaf(mv[8 + r], pz, fmaf(mv[4 + r], py, fmaf(mv[r], px, mv[12 + r])));
#pragma unroll
          for (int r = 0; r < 4; r++)
            clip[i][r] = fmaf(pc.proj[12 + r], eye[3], fmaf(pc.proj[8 + r], eye[2], fmaf(pc.proj[4 + r], eye[1], pc.proj[r] * eye[0])));
          u[i] = tri.uv[2 * i];  // sprite.vert:24: no scroll
        } else {
#pragma unroll
          for (int r = 0; r < 4; r++)
            clip[i][r] = fmaf(pm[8 + r], z, fmaf(pm[4 + r], y, fmaf(pm[r], x, pm[12 + r])));
          u[i] = tri.uv[2 * i] + pc.time * tri.scroll[i];
        }
        v[i] = tri.uv[2 * i + 1];
      }
      // flat varyings (provoking vertex data were folded into LevelTri on the host)
      const uint32_t nframes = tri.packed & 0xFFu;
      float au = tri.atlas_u, av = tri.atlas_v;
      if (kind == RDOOM_KIND_SKY) au = vr0, av = vr1;  // sky records carry v_r (flat varying of sky.vert) here
      if (nframes != 1u && kind != RDOOM_KIND_SKY) {
        const float aw = kind == RDOOM_KIND_FLAT ? (float)lv.flat_w : (kind == RDOOM_KIND_DECOR ? (float)lv.decor_w : (float)lv.wall_w);
        const float anim_fps = 8.0f / 35.0f;
        float fi = pc.time / anim_fps;
        fi = floorf(glsl_mod(fi, (float)nframes));
        float atlas_u = tri.atlas_u + fi * tri.size_x;
        const float rows_down = ceilf((atlas_u + tri.size_x) / aw) - 1.0f;
        atlas_u = atlas_u + glsl_mod(aw - tri.atlas_u, tri.size_x) * rows_down;
        au = atlas_u;
        av = tri.atlas_v + rows_down * tri.row_height;
      }
      ok = !(clip[0][3] <= 0.0f && clip[1][3] <= 0.0f && clip[2][3] <= 0.0f);
      if (ok) {
        const float hw = 0.5f * (float)width, hh = 0.5f * (float)height;
        float xw[3], yw[3], w[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
          xw[i] = (clip[i][0] + clip[i][3]) * hw;
          yw[i] = (clip[i][1] + clip[i][3]) * hh;
          w[i] = clip[i][3];
        }
        uint32_t tl = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const int j = (i + 1) % 3, k = (i + 2) % 3;
          const float A = dop(yw[j], w[k], yw[k], w[j]);
          const float B = dop(xw[k], w[j], xw[j], w[k]);
          const float C = dop(xw[j], yw[k], xw[k], yw[j]);
          rr.e[3 * i] = A;
          rr.e[3 * i + 1] = B;
          rr.e[3 * i + 2] = C;
          if ((A > 0.0f) || (A == 0.0f && B > 0.0f)) tl |= 1u << i;
        }
        const float det = fmaf(w[0], rr.e[2], fmaf(yw[0], rr.e[1], xw[0] * rr.e[0]));
        ok = det > 0.0f;
        if (ok) {
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const float e0 = rr.e[c], e1 = rr.e[3 + c], e2 = rr.e[6 + c];
            const float nz = fmaf(clip[2][2], e2, fmaf(clip[1][2], e1, clip[0][2] * e0));
            const float n1 = (e0 + e1) + e2;
            const float nu = fmaf(u[2], e2, fmaf(u[1], e1, u[0] * e0));
            const float nv = fmaf(v[2], e2, fmaf(v[1], e1, v[0] * e0));
            rr.zp[c] = 0.5f * (nz / det);
            rr.wp[c] = n1 / det;
            sr.up[c] = nu / det;
            sr.vp[c] = nv / det;
          }
          rr.zp[2] = rr.zp[2] + 0.5f;
          int x0 = 0, y0 = 0, x1 = width - 1, y1 = height - 1;
          const float wmin = fminf(w[0], fminf(w[1], w[2]));
          wkey = wmin;
          if (wmin >= 1e-5f) {
            float sx[3], sy[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
              sx[i] = xw[i] / w[i];
              sy[i] = yw[i] / w[i];
            }
            const float fx0 = floorf(fminf(sx[0], fminf(sx[1], sx[2]))) - 1.0f;
            const float fx1 = ceilf(fmaxf(sx[0], fmaxf(sx[1], sx[2]))) + 1.0f;
            const float fy0 = floorf(fminf(sy[0], fminf(sy[1], sy[2]))) - 1.0f;
            const float fy1 = ceilf(fmaxf(sy[0], fmaxf(sy[1], sy[2]))) + 1.0f;
            ok = fx0 <= (float)(width - 1) && fx1 >= 0.0f && fy0 <= (float)(height - 1) && fy1 >= 0.0f;
            if (ok) {
              x0 = (int)fmaxf(fx0, 0.0f);
              y0 = (int)fmaxf(fy0, 0.0f);
              x1 = (int)fminf(fx1, (float)(width - 1));
              y1 = (int)fminf(fy1, (float)(height - 1));
            }
          }
          rr.bb0 = (uint32_t)x0 | ((uint32_t)y0 << 16);
          rr.bb1 = (uint32_t)x1 | ((uint32_t)y1 << 16);
          const uint32_t masked = (tri.packed >> 18) & 3u;  // border, interior
          rr.flags = (t & 0xFFFFFFu) | (tl << 24) | (kind << 27) | (masked << 29);
          rr.pad[0] = rr.pad[1] = 0;
          sr.wp[0] = rr.wp[0];
          sr.wp[1] = rr.wp[1];
          sr.wp[2] = rr.wp[2];
          sr.atlas_u = au;
          sr.atlas_v = av;
          sr.size_x = kind == RDOOM_KIND_SKY ? 4.0f * au / 3.14159265358f : tri.size_x;  // sky: the u shift of sky.frag:15
          sr.size_y = tri.size_y;
          sr.light = (float)pc.lights[(tri.packed >> 8) & 0xFFu] / 255.0f;
          const uint32_t bx = __float_as_uint(tri.size_x), by = __float_as_uint(tri.size_y);
          const bool p2x = (bx & 0x7FFFFFu) == 0u && tri.size_x > 0.0f, p2y = (by & 0x7FFFFFu) == 0u && tri.size_y > 0.0f;
          const bool is_flat = kind == RDOOM_KIND_FLAT, is_decor = kind == RDOOM_KIND_DECOR;
          const uint32_t aw = is_flat ? lv.flat_w : (is_decor ? lv.decor_w : lv.wall_w);
          const uint32_t ah = is_flat ? lv.flat_h : (is_decor ? lv.decor_h : lv.wall_h);
          const uint32_t tbase = is_flat ? lv.flat_base : (is_decor ? lv.decor_base : 0u);
          const uint32_t lw = aw ? 31u - (uint32_t)__clz(aw) : 0u;
          auto packed_ok = [](float sz, bool p2) {
            return p2 ? (sz >= 0x1p-20f && sz <= 0x1p20f) : (sz >= 1.0f && sz <= 4096.0f && floorf(sz) == sz);
          };
          const bool fast_ok = packed_ok(tri.size_x, p2x) && packed_ok(tri.size_y, p2y) && kind <= RDOOM_KIND_WALL;
          sr.flags = kind | (p2x ? SHADE_POW2_X : 0u) | (p2y ? SHADE_POW2_Y : 0u) | (fast_ok ? SHADE_FAST : 0u) |
                     ((p2x && p2y) ? 0u : SHADE_NP2) | (lw << 8) | ((tbase >> 10) << 16);
          sr.tex = kind == RDOOM_KIND_SKY ? 0u : (((aw - 1u) & 0xFFFFu) | (((ah - 1u) & 0xFFFFu) << 16));
        }
      }
    }
  }
  return ok;
}

__device__ __forceinline__ uint32_t depth_bucket(float wmin) {
  // monotone in wmin: 16 binades [2^-8, 2^8) x 128 steps; anything nearer (or behind the eye) -> 0
  if (!(wmin > 0.00390625f)) return 0u;
  const uint32_t b = (__float_as_uint(wmin) >> 16) - (0x3B80u);  // 0x3B800000 = 2^-8
  return min(b, SORT_BUCKETS - 1u);
}

__global__ __launch_bounds__(256) void setup_kernel(DeviceLevelView lv, const PoseConst *__restrict__ poses,
                                                    const ObjectConst *__restrict__ objects, uint32_t n_objects,
                                                    int width, int height, uint32_t kinds_mask,
                                                    TriRec *__restrict__ recs, TriRec *__restrict__ tmp_recs,
                                                    uint4 *__restrict__ sorted, uint32_t *__restrict__ counts,
                                                    uint32_t cap) {
  __shared__ uint32_t hist[SORT_BUCKETS];
  __shared__ uint16_t keys[SORT_KEY_CAP];
  __shared__ uint32_t wcnt[4];
  __shared__ uint32_t scan_tmp[256];
  const uint32_t pose = blockIdx.x;
  const PoseConst &pc = poses[pose];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  TriRec *prec = recs + (size_t)pose * cap;
  TriRec *ptmp = tmp_recs + (size_t)pose * cap;  // records in compaction (= primitive) order, before the sort
  uint4 *psorted = sorted + (size_t)pose * cap;
  for (uint32_t i = tid; i < SORT_BUCKETS; i += 256) hist[i] = 0;
  __syncthreads();
  uint32_t n = 0;  // visible so far (uniform)
  for (uint32_t base = 0; base < lv.ntri; base += 256u) {
    const uint32_t t = base + (uint32_t)tid;
    RasterRec rr;
    ShadeRec sr;
    float wkey;
    const bool ok = t < lv.ntri && setup_triangle(lv, pc, objects ? objects + (size_t)pose * n_objects : nullptr, t,
                                                  width, height, kinds_mask, rr, sr, wkey);
    const unsigned long long m = __ballot(ok);
    if (lane == 0) wcnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint32_t c = wcnt[w];
      if (w < wave) off += c;
      total += c;
    }
    if (ok) {
      ptmp[off].r = rr;
      ptmp[off].s = sr;
      const uint32_t bucket = depth_bucket(wkey);
      if (off < SORT_KEY_CAP) {
        keys[off] = (uint16_t)bucket;
        atomicAdd(&hist[bucket], 1u);
      }
    }
    n += total;
    __syncthreads();
  }
  if (tid == 0) counts[pose] = n;
  const bool sortable = n <= SORT_KEY_CAP;  // else too many for the LDS key array: primitive order (still correct)
  if (sortable) {
    // exclusive scan of the histogram: 8 buckets per thread + a 256-wide block scan
    uint32_t local[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      local[k] = sum;
      sum += hist[tid * 8 + k];
    }
    scan_tmp[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const uint32_t v = tid >= d ? scan_tmp[tid - d] : 0u;
      __syncthreads();
      scan_tmp[tid] += v;
      __syncthreads();
    }
    const uint32_t before = scan_tmp[tid] - sum;
#pragma unroll
    for (int k = 0; k < 8; k++) hist[tid * 8 + k] = before + local[k];
    __syncthreads();
  }
  // move every record to its near-to-far position: record index == position in the sorted list from here on
  // (bin/raster/fragment gather records by that index; ptmp was written by this workgroup, same CU, after a barrier)
  for (uint32_t i = tid; i < n; i += 256) {
    const uint32_t bucket = sortable ? (uint32_t)keys[i] : 0u;
    const uint32_t pos = sortable ? atomicAdd(&hist[bucket], 1u) : i;
    const uint4 *src = reinterpret_cast<const uint4 *>(&ptmp[i]);
    uint4 *dst = reinterpret_cast<uint4 *>(&prec[pos]);
    uint4 v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = src[k];
#pragma unroll
    for (int k = 0; k < 9; k++) dst[k] = v[k];
    psorted[pos] = make_uint4(v[3].w, v[4].x, pos, bucket);  // RasterRec::bb0, bb1 (dwords 15, 16)
  }
}

// Exact rejection of a triangle against the four 32x32 quadrants of the 64x64 tile at (tx0, ty0):
// fmaf is monotone in each argument, so the extreme of a *computed* edge function / depth plane over a
// rectangle of pixel centres is attained at a corner.  Bit q of the result = "may touch quadrant q".
__device__ __forceinline__ uint32_t tile_quadrant_mask(const uint4 c0, const uint4 c1, const uint4 c2, int x0, int y0,
                                                       int x1, int y1, int tx0, int ty0) {
  const float e0a = __uint_as_float(c0.x), e0b = __uint_as_float(c0.y), e0c = __uint_as_float(c0.z),
              e1a = __uint_as_float(c0.w), e1b = __uint_as_float(c1.x), e1c = __uint_as_float(c1.y),
              e2a = __uint_as_float(c1.z), e2b = __uint_as_float(c1.w), e2c = __uint_as_float(c2.x),
              za = __uint_as_float(c2.y), zb = __uint_as_float(c2.z), zc = __uint_as_float(c2.w);
  uint32_t qm = 0;
#pragma unroll
  for (int qd = 0; qd < 4; qd++) {
    const int rx0 = tx0 + (qd & 1) * 32, ry0 = ty0 + (qd >> 1) * 32;
    const float xl = (float)rx0 + 0.5f, xh = (float)rx0 + 31.5f, yl = (float)ry0 + 0.5f, yh = (float)ry0 + 31.5f;
    const float m0 = fmaf(e0a, e0a > 0.0f ? xh : xl, fmaf(e0b, e0b > 0.0f ? yh : yl, e0c));
    const float m1 = fmaf(e1a, e1a > 0.0f ? xh : xl, fmaf(e1b, e1b > 0.0f ? yh : yl, e1c));
    const float m2 = fmaf(e2a, e2a > 0.0f ? xh : xl, fmaf(e2b, e2b > 0.0f ? yh : yl, e2c));
    const float zn = fmaf(za, za > 0.0f ? xl : xh, fmaf(zb, zb > 0.0f ? yl : yh, zc));
    const float zf = fmaf(za, za > 0.0f ? xh : xl, fmaf(zb, zb > 0.0f ? yh : yl, zc));
    const bool ok = (x0 <= rx0 + 31) & (x1 >= rx0) & (y0 <= ry0 + 31) & (y1 >= ry0) & (m0 >= 0.0f) & (m1 >= 0.0f) &
                    (m2 >= 0.0f) & (zn <= 1.0f) & (zf >= 0.0f);
    qm |= ok ? (1u << qd) : 0u;
  }
  return qm;
}

// Same corner argument for any rectangle of pixel centres [xl, xh] x [yl, yh]: false = no pixel in it can be
// covered.  Monotonicity makes it hierarchical: a rectangle that fails rules out every rectangle inside it.
__device__ __forceinline__ bool rect_may_touch(const uint4 c0, const uint4 c1, const uint4 c2, float xl, float xh,
                                               float yl, float yh) {
  const float e0a = __uint_as_float(c0.x), e0b = __uint_as_float(c0.y), e0c = __uint_as_float(c0.z),
              e1a = __uint_as_float(c0.w), e1b = __uint_as_float(c1.x), e1c = __uint_as_float(c1.y),
              e2a = __uint_as_float(c1.z), e2b = __uint_as_float(c1.w), e2c = __uint_as_float(c2.x),
              za = __uint_as_float(c2.y), zb = __uint_as_float(c2.z), zc = __uint_as_float(c2.w);
  const float m0 = fmaf(e0a, e0a > 0.0f ? xh : xl, fmaf(e0b, e0b > 0.0f ? yh : yl, e0c));
  const float m1 = fmaf(e1a, e1a > 0.0f ? xh : xl, fmaf(e1b, e1b > 0.0f ? yh : yl, e1c));
  const float m2 = fmaf(e2a, e2a > 0.0f ? xh : xl, fmaf(e2b, e2b > 0.0f ? yh : yl, e2c));
  const float zn = fmaf(za, za > 0.0f ? xl : xh, fmaf(zb, zb > 0.0f ? yl : yh, zc));
  const float zf = fmaf(za, za > 0.0f ? xh : xl, fmaf(zb, zb > 0.0f ? yh : yl, zc));
  return (m0 >= 0.0f) & (m1 >= 0.0f) & (m2 >= 0.0f) & (zn <= 1.0f) & (zf >= 0.0f);
}
__device__ __forceinline__ bool tile_may_touch(const uint4 c0, const uint4 c1, const uint4 c2, int tx0, int ty0) {
  return rect_may_touch(c0, c1, c2, (float)tx0 + 0.5f, (float)tx0 + 63.5f, (float)ty0 + 0.5f, (float)ty0 + 63.5f);
}

// =================================================================================================
// Kernel 1b: binning.  One workgroup per pose turns the near-to-far record list into
// per-tile lists: count -> scan -> fill.  The unit of work is a (triangle, tile of its bbox) pair: the
// raster coefficients of BIN_CHUNK triangles are staged in LDS together with an exclusive prefix sum of
// their bbox tile counts, and every lane finds its pair by binary search in that prefix -- lanes stay busy
// whatever the mix of one-tile and whole-frame triangles, and the exact tile/quadrant test runs from LDS.
// entry = record index | quadrant mask << 28.  Pairs are visited in list order one workgroup-full at a time, so a tile's
// list is near-to-far up to that window; the rasteriser re-sorts each list chunk by record index (= depth
// rank).  Order only affects early-z efficiency: the winner is order-independent.  If a pose needs more than
// entry_cap entries (or the frame has more than MAX_TILES tiles) its overflow flag is set and the
// rasteriser scans the sorted list instead.
// =================================================================================================
constexpr uint32_t MAX_TILES = 8192;

template <int BIN_THREADS, int BIN_LOG2>  // threads per workgroup = triangles staged per round (one per thread)
__global__ __launch_bounds__(BIN_THREADS) void bin_kernel(const TriRec *__restrict__ recs,
                                                          const uint4 *__restrict__ sorted,
                                                          const uint32_t *__restrict__ counts, uint32_t cap,
                                                          int tiles_x, int tiles_y, uint2 *__restrict__ tile_hdr,
                                                          uint32_t *__restrict__ entries, uint32_t entry_cap,
                                                          uint32_t *__restrict__ overflow) {
  constexpr uint32_t BIN_CHUNK = BIN_THREADS;
  static_assert((1 << BIN_LOG2) == BIN_THREADS, "bin_kernel: BIN_LOG2");
  extern __shared__ uint32_t bin_dyn[];  // tile_cnt[T], tile_off[T]
  __shared__ uint4 coef[BIN_CHUNK][3];   // e[9], zp[3] of the staged triangles
  __shared__ uint2 bbox[BIN_CHUNK];
  __shared__ uint32_t trange[BIN_CHUNK];  // tile rectangle to visit: tx0 | ty0 << 8 | width << 16
  __shared__ uint32_t pref[BIN_CHUNK + 1];
  __shared__ uint32_t scan_tmp[BIN_THREADS];
  const uint32_t pose = blockIdx.x;
  const int tid = threadIdx.x;
  const uint32_t T = (uint32_t)(tiles_x * tiles_y);
  if (T > MAX_TILES) {
    if (tid == 0) overflow[pose] = 1u;
    return;
  }
  uint32_t *tile_cnt = bin_dyn, *tile_off = bin_dyn + T;
  const TriRec *prec = recs + (size_t)pose * cap;
  const uint4 *psorted = sorted + (size_t)pose * cap;
  uint2 *hdr = tile_hdr + (size_t)pose * T;
  uint32_t *pent = entries + (size_t)pose * entry_cap;
  const uint32_t n = counts[pose];
  for (uint32_t i = tid; i < T; i += BIN_THREADS) tile_cnt[i] = 0;
  for (int pass = 0; pass < 2; pass++) {
    for (uint32_t cbase = 0; cbase < n; cbase += BIN_CHUNK) {
      const uint32_t cn = min(BIN_CHUNK, n - cbase);
      __syncthreads();  // previous round's readers of coef/pref are done (and tile_cnt / tile_off are ready)
      uint32_t nt = 0;
      if ((uint32_t)tid < cn) {
        const uint4 ent = psorted[cbase + tid];  // (bb0, bb1, record index == cbase + tid, depth bucket)
        const uint4 *rp = reinterpret_cast<const uint4 *>(&prec[cbase + tid]);
        const uint4 c0 = rp[0], c1 = rp[1], c2 = rp[2];
        coef[tid][0] = c0;
        coef[tid][1] = c1;
        coef[tid][2] = c2;
        bbox[tid] = make_uint2(ent.x, ent.y);
        int tx0 = (int)((ent.x & 0xFFFFu) >> 6), ty0 = (int)((ent.x >> 16) >> 6), tx1 = (int)((ent.y & 0xFFFFu) >> 6),
            ty1 = (int)((ent.y >> 16) >> 6);
        if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > 32) {
          // a big tile rectangle (typically a triangle that crosses the eye plane: bbox = whole frame, S6): shrink
          // it to the tile rows / columns whose full-width / full-height strip can be touched at all.  Exact
          // (rect_may_touch is conservative and hierarchical), so no tile that passes the per-tile test is lost.
          const float fx0 = (float)(tx0 * 64) + 0.5f, fx1 = (float)(tx1 * 64) + 63.5f;
          const float fy0 = (float)(ty0 * 64) + 0.5f, fy1 = (float)(ty1 * 64) + 63.5f;
          while (ty0 <= ty1 && !rect_may_touch(c0, c1, c2, fx0, fx1, (float)(ty0 * 64) + 0.5f, (float)(ty0 * 64) + 63.5f)) ty0++;
          while (ty1 >= ty0 && !rect_may_touch(c0, c1, c2, fx0, fx1, (float)(ty1 * 64) + 0.5f, (float)(ty1 * 64) + 63.5f)) ty1--;
          while (tx0 <= tx1 && !rect_may_touch(c0, c1, c2, (float)(tx0 * 64) + 0.5f, (float)(tx0 * 64) + 63.5f, fy0, fy1)) tx0++;
          while (tx1 >= tx0 && !rect_may_touch(c0, c1, c2, (float)(tx1 * 64) + 0.5f, (float)(tx1 * 64) + 63.5f, fy0, fy1)) tx1--;
        }
        nt = (tx1 >= tx0 && ty1 >= ty0) ? (uint32_t)((tx1 - tx0 + 1) * (ty1 - ty0 + 1)) : 0u;
        trange[tid] = (uint32_t)tx0 | ((uint32_t)ty0 << 8) | ((uint32_t)(tx1 - tx0 + 1) << 16);  // tiles per side <= 128
      }
      scan_tmp[tid] = nt;
      __syncthreads();
      for (int d = 1; d < BIN_THREADS; d <<= 1) {
        const uint32_t v = tid >= d ? scan_tmp[tid - d] : 0u;
        __syncthreads();
        scan_tmp[tid] += v;
        __syncthreads();
      }
      pref[tid] = scan_tmp[tid] - nt;  // exclusive; entries past cn repeat the total
      const uint32_t W = scan_tmp[BIN_THREADS - 1];
      if (tid == 0) pref[BIN_CHUNK] = W;
      __syncthreads();
      for (uint32_t w = (uint32_t)tid; w < W; w += BIN_THREADS) {
        // largest i with pref[i] <= w (pref non-decreasing, pref[0] = 0, pref[BIN_CHUNK] = W > w); triangles
        // past cn have pref == W and are never selected
        uint32_t lo = 0, hi = BIN_CHUNK;
#pragma unroll
        for (int step = 0; step < BIN_LOG2; step++) {
          const uint32_t mid = (lo + hi) >> 1;
          const bool le = pref[mid] <= w;
          lo = le ? mid : lo;
          hi = le ? hi : mid;
        }
        const uint32_t t = w - pref[lo];
        const uint2 bb = bbox[lo];
        const int x0 = (int)(bb.x & 0xFFFFu), y0 = (int)(bb.x >> 16), x1 = (int)(bb.y & 0xFFFFu), y1 = (int)(bb.y >> 16);
        const uint32_t tr = trange[lo];
        const int tx0 = (int)(tr & 0xFFu), ty0 = (int)((tr >> 8) & 0xFFu), ntx = (int)(tr >> 16);
        const int ty = (int)(((float)t + 0.5f) / (float)ntx), tx = (int)t - ty * ntx;  // t / ntx (t < 8192: exact)
        const uint4 c0 = coef[lo][0], c1 = coef[lo][1], c2 = coef[lo][2];
        uint32_t qm = 0;
        if (tile_may_touch(c0, c1, c2, (tx0 + tx) * 64, (ty0 + ty) * 64))
          qm = tile_quadrant_mask(c0, c1, c2, x0, y0, x1, y1, (tx0 + tx) * 64, (ty0 + ty) * 64);
        if (qm) {
          const uint32_t tile = (uint32_t)((ty0 + ty) * tiles_x + tx0 + tx);
          if (pass == 0) {
            atomicAdd(&tile_cnt[tile], 1u);
          } else {
            const uint32_t pos = atomicAdd(&tile_off[tile], 1u);
            if (pos < entry_cap) pent[pos] = (cbase + lo) | (qm << 28);
          }
        }
      }
    }
    __syncthreads();
    if (pass == 1) break;
    // exclusive scan of tile_cnt -> tile_off (thread t owns T/512 consecutive tiles); headers out
    const uint32_t per = (T + BIN_THREADS - 1u) / BIN_THREADS, lo = min((uint32_t)tid * per, T), hi = min(lo + per, T);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += tile_cnt[i];
    scan_tmp[tid] = sum;
    __syncthreads();
    for (int d = 1; d < BIN_THREADS; d <<= 1) {
      const uint32_t v = tid >= d ? scan_tmp[tid - d] : 0u;
      __syncthreads();
      scan_tmp[tid] += v;
      __syncthreads();
    }
    const uint32_t total = scan_tmp[BIN_THREADS - 1];
    uint32_t run = scan_tmp[tid] - sum;
    for (uint32_t i = lo; i < hi; i++) {
      tile_off[i] = run;
      hdr[i] = make_uint2(run, tile_cnt[i]);
      run += tile_cnt[i];
    }
    if (tid == 0) overflow[pose] = total > entry_cap ? 1u : 0u;
    if (total > entry_cap) return;  // uniform
  }
}

// F1..F3: perspective-correct tile coordinates -> atlas texel coordinates (shared by the alpha test
// R6 and the fragment stage).  `row_u`/`row_v`/`row_w` are fmaf(B, py, C) of the three planes.
struct TexelAt {
  int ix, iy;
  float dist;
};
__device__ __forceinline__ TexelAt texel_coords(const ShadeRec &s, float px, float row_w, float row_u,
                                                float row_v) {
  TexelAt t;
  const float rw = fmaf(s.wp[0], px, row_w);
  const float w = 1.0f / rw;
  const float tu = fmaf(s.up[0], px, row_u) * w;
  const float tv = fmaf(s.vp[0], px, row_v) * w;
  t.dist = w;
  // mod(x, y) = x - y * floor(x / y).  For power-of-two y, x / y == x * (1 / y) exactly, and 1 / y is
  // one integer subtraction on the exponent field: same bits as the division, a tenth of the cost.
  float qx, qy;
  if (s.flags & SHADE_POW2_X)
    qx = tu * __uint_as_float(0x7F000000u - __float_as_uint(s.size_x));
  else
    qx = tu / s.size_x;
  if (s.flags & SHADE_POW2_Y)
    qy = tv * __uint_as_float(0x7F000000u - __float_as_uint(s.size_y));
  else
    qy = tv / s.size_y;
  const float uvx = (tu - s.size_x * floorf(qx)) + s.atlas_u;
  const float uvy = (tv - s.size_y * floorf(qy)) + s.atlas_v;
  t.ix = (int)floorf(uvx);
  t.iy = (int)floorf(uvy);
  return t;
}

// F3: REPEAT + NEAREST on a power-of-two atlas; flats and walls live in one u16 store (see DeviceLevelView)
__device__ __forceinline__ uint32_t texel_offset(uint32_t flags, uint32_t tex, int ix, int iy) {
  const uint32_t wm = tex & 0xFFFFu, hm = tex >> 16, lw = (flags >> 8) & 15u, base = (flags >> 16) << 10;
  return base + ((((uint32_t)iy & hm) << lw) | ((uint32_t)ix & wm));
}
__device__ __forceinline__ uint32_t load_texel(const DeviceLevelView &lv, const ShadeRec &s, int ix, int iy) {
  return lv.texels[texel_offset(s.flags, s.tex, ix, iy)];
}

// =================================================================================================
// Rasteriser, per-entry part.  Rejection is hierarchical and exact: fmaf is monotone in each argument, so the
// extreme of a *computed* edge function, depth plane or 1/w plane over a pixel rectangle sits at a corner --
// per lane: nearest-corner depth against the lane's farthest pixel (early-z) first, then three edge corners,
// the depth range and the 1/w plane.  One __any() skips the 16-pixel body when no lane needs it.
// Winner = lexicographic min of (d24, primitive id): independent of processing order.
// =================================================================================================
// x > 0 for a non-NaN binary32, as an integer test on the bits: a scalar compare when x is wave-uniform
__device__ __forceinline__ bool pos(float x) { return (int)__float_as_uint(x) > 0; }

// One queue entry against one lane's 4x4 block: exact rejection (early-z first), then the pixel bodies (R1..R6).
// The coefficients arrive wave-uniform (v_readlane broadcasts), i.e. as SGPR operands.
template <bool STATS, int DBG, class ShadeFetch>
__device__ __forceinline__ void raster_entry(const DeviceLevelView &lv, const TriRec *__restrict__ prec, float e0a, float e0b,
                                             float e0c, float e1a, float e1b, float e1c, float e2a, float e2b, float e2c,
                                             float za, float zb, float zc, float wa, float wb, float wc, int x0, int y0,
                                             int x1, int y1, uint32_t flags, uint32_t ridx, int bx, int by, float pxlo,
                                             float pxhi, float pylo, float pyhi, uint32_t (&best_d)[16],
                                             uint32_t (&best_r)[16], uint32_t &lane_far, ShadeFetch fetch_shade,
                                             unsigned long long (&st)[16]) {
    // lane-level exact rejection over my 4x4 block.  Early-z first (most rejected triangles are simply hidden):
    // nearest depth of the plane over the block against the farthest depth I still hold
    const float zn = fmaf(za, pos(za) ? pxlo : pxhi, fmaf(zb, pos(zb) ? pylo : pyhi, zc));
    const uint32_t dn = __float2uint_rz(fmaf(fminf(fmaxf(zn, 0.0f), 1.0f), 16777215.0f, 0.5f));
    const bool zpass = (zn <= 1.0f) & (dn <= lane_far);
    if (!__any(zpass)) {
      if (STATS) st[15]++;
      return;
    }
    // largest edge values and farthest depth
    const float m0 = fmaf(e0a, pos(e0a) ? pxhi : pxlo, fmaf(e0b, pos(e0b) ? pyhi : pylo, e0c));
    const float m1 = fmaf(e1a, pos(e1a) ? pxhi : pxlo, fmaf(e1b, pos(e1b) ? pyhi : pylo, e1c));
    const float m2 = fmaf(e2a, pos(e2a) ? pxhi : pxlo, fmaf(e2b, pos(e2b) ? pyhi : pylo, e2c));
    const float zf = fmaf(za, pos(za) ? pxhi : pxlo, fmaf(zb, pos(zb) ? pyhi : pylo, zc));
    const bool need0 = zpass && bx <= x1 && bx + 3 >= x0 && by <= y1 && by + 3 >= y0 && m0 >= 0.0f && m1 >= 0.0f &&
                       m2 >= 0.0f && zf >= 0.0f;
    if (STATS && !__any(need0)) st[14]++;
    if (!__any(need0)) return;
    // R3 needs rw > 0: a block whose largest 1/w is not positive holds no coverable pixel (same corner argument)
    const float rwf = fmaf(wa, pos(wa) ? pxhi : pxlo, fmaf(wb, pos(wb) ? pyhi : pylo, wc));
    const bool need = need0 & (rwf > 0.0f);
    if (!__any(need)) return;
    if (DBG == 2) {
      if (need) best_r[0] = ridx;
      return;
    }
    if (STATS) st[2]++, st[3] += (unsigned long long)__popcll(__ballot(need));
    // Fast path (exact): block fully inside the bbox, whole block inside the depth range and in front
    // of the eye, texture rectangle fully opaque.  Edge ties and depth ties are only *detected* here and
    // replayed through the general path below, so the result is the same as running it everywhere.
    // A texture whose only transparent texels lie in the one-texel ring around its rectangle
    // (RASTER_MASKED_BORDER) is treated as opaque here; the rare pixel whose float mod lands on the ring
    // is caught by the fragment kernel (it sees a transparent texel) and re-resolved by fixup_kernel.
    const float rwn = fmaf(wa, pos(wa) ? pxlo : pxhi, fmaf(wb, pos(wb) ? pylo : pyhi, wc));
    const bool fast = need & (zn >= 0.0f) & (zf <= 1.0f) & (rwn > 0.0f) & ((flags & RASTER_MASKED_INTERIOR) == 0u);
    // pixels of my block outside the triangle's bbox (S6) never win: bit k of `outside` (k = 4 * row + column).
    // need guarantees the block overlaps the bbox, so the column and row ranges below are non-empty.
    uint32_t outside = 0u;
    if (__any(fast & !((bx >= x0) & (bx + 3 <= x1) & (by >= y0) & (by + 3 <= y1)))) {
      const int clo = max(x0 - bx, 0), chi = min(x1 - bx, 3), rlo = max(y0 - by, 0), rhi = min(y1 - by, 3);
      const uint32_t cm = ((2u << chi) - 1u) & ~((1u << clo) - 1u);                // columns inside, 4 bits
      const uint32_t rows = ((16u << (4 * rhi)) - 1u) & ~((1u << (4 * rlo)) - 1u);  // all pixels of the rows inside
      outside = ~((cm * 0x1111u) & rows) & 0xFFFFu;
    }
    bool redo = false, updated = false;
    if (STATS && __any(fast)) st[4]++, st[5] += (unsigned long long)__popcll(__ballot(fast));
    if (fast) {
#pragma unroll
      for (int ry = 0; ry < 4; ry++) {
        const float py = pylo + (float)ry;
        const float t0 = fmaf(e0b, py, e0c), t1 = fmaf(e1b, py, e1c), t2 = fmaf(e2b, py, e2c);
        const float tz = fmaf(zb, py, zc);
#pragma unroll
        for (int rx = 0; rx < 4; rx++) {
          const int k = ry * 4 + rx;
          const float px = pxlo + (float)rx;
          const float em = fminf(fminf(fmaf(e0a, px, t0), fmaf(e1a, px, t1)), fmaf(e2a, px, t2));
          const uint32_t d24 = __float2uint_rz(fmaf(fmaf(za, px, tz), 16777215.0f, 0.5f));
          const uint32_t d24m = d24 | (uint32_t)__builtin_amdgcn_sbfe((int)outside, k, 1);  // all ones when outside
          const bool win = (em > 0.0f) & (d24m < best_d[k]);
          redo |= (em == 0.0f) | ((em > 0.0f) & (d24 == best_d[k]));
          best_d[k] = win ? d24 : best_d[k];
          best_r[k] = win ? ridx : best_r[k];
          updated |= win;
        }
      }
    }
    if (__any(need & (!fast | redo))) {
      if (STATS) {
        st[6]++, st[7] += (unsigned long long)__popcll(__ballot(need & (!fast | redo)));
        // why: [10] depth range, [11] 1/w <= 0 in the block, [12] masked texture, [13] tie replay
        st[10] += (unsigned long long)__popcll(__ballot(need & !((zn >= 0.0f) & (zf <= 1.0f))));
        st[11] += (unsigned long long)__popcll(__ballot(need & !(rwn > 0.0f)));
        st[12] += (unsigned long long)__popcll(__ballot(need & ((flags & RASTER_MASKED_INTERIOR) != 0u)));
        st[13] += (unsigned long long)__popcll(__ballot(need & fast & redo));
      }
      if (need & (!fast | redo)) {
        const uint32_t prim = flags & 0xFFFFFFu;
#pragma unroll
        for (int ry = 0; ry < 4; ry++) {
          const int iy = by + ry;
          const float py = (float)iy + 0.5f;
          const float t0 = fmaf(e0b, py, e0c), t1 = fmaf(e1b, py, e1c), t2 = fmaf(e2b, py, e2c);
          const float tz = fmaf(zb, py, zc), tw = fmaf(wb, py, wc);
          const bool rowin = iy >= y0 && iy <= y1;
#pragma unroll
          for (int rx = 0; rx < 4; rx++) {
            const int ix = bx + rx;
            const float px = (float)ix + 0.5f;
            const float e0 = fmaf(e0a, px, t0), e1 = fmaf(e1a, px, t1), e2 = fmaf(e2a, px, t2);
            const bool in0 = (e0 > 0.0f) | ((e0 == 0.0f) & ((flags & (1u << 24)) != 0u));
            const bool in1 = (e1 > 0.0f) | ((e1 == 0.0f) & ((flags & (1u << 25)) != 0u));
            const bool in2 = (e2 > 0.0f) | ((e2 == 0.0f) & ((flags & (1u << 26)) != 0u));
            const float zw = fmaf(za, px, tz);
            const float rw = fmaf(wa, px, tw);
            const uint32_t d24 = __float2uint_rz(fmaf(fminf(fmaxf(zw, 0.0f), 1.0f), 16777215.0f, 0.5f));
            const int k = ry * 4 + rx;
            const uint32_t bd = best_d[k], br = best_r[k];
            bool pass = rowin & (ix >= x0) & (ix <= x1) & in0 & in1 & in2 & (zw >= 0.0f) & (zw <= 1.0f) &
                        (rw > 0.0f) & (d24 <= bd);
            if (pass && d24 == bd)  // depth tie (rare): the earlier primitive keeps the pixel
              pass = br == NONE || prim < (prec[br].r.flags & 0xFFFFFFu);
            if (pass && (flags & RASTER_MASKED_ANY) != 0u) {  // R6: alpha test before the depth write
              const ShadeRec sh = fetch_shade();
              const TexelAt t =
                  texel_coords(sh, px, tw, fmaf(sh.up[1], py, sh.up[2]), fmaf(sh.vp[1], py, sh.vp[2]));
              // texture rectangle fully opaque: only a coordinate that the float mod pushed just outside
              // the rectangle can hit a transparent neighbour texel -- fetch only then
              const bool must_fetch = (flags & RASTER_MASKED_INTERIOR) != 0u || t.ix < (int)sh.atlas_u ||
                                      t.ix >= (int)(sh.atlas_u + sh.size_x) || t.iy < (int)sh.atlas_v ||
                                      t.iy >= (int)(sh.atlas_v + sh.size_y);
              if (must_fetch) pass = (load_texel(lv, sh, t.ix, t.iy) & 0x8000u) == 0u;
            }
            if (pass) {
              best_d[k] = d24;
              best_r[k] = ridx;
              updated = true;
            }
          }
        }
      }
    }
    if (updated) {
      uint32_t m = best_d[0];
#pragma unroll
      for (int k = 1; k < 16; k++) m = max(m, best_d[k]);
      lane_far = m;
    }
}

// =================================================================================================
// Kernel 2: tiled rasteriser, wave-autonomous.  One 256-thread workgroup per (pose, 64x64 tile); each wavefront
// owns a 32x32 quadrant and runs on its own (no LDS staging of records, no workgroup barriers); each lane owns a
// 4x4 pixel block whose depth / winner live in registers.  blockIdx -> (pose, tile) keeps all tiles of a pose on
// one XCD (b % 8): its records stay in that XCD's L2.
//   * candidates  64 tile-list entries at a time, one per lane; the lanes whose entry touches this wave's
//                 quadrant are ranked by record index (= depth rank) with readlane broadcasts and compacted
//                 through a 256-byte per-wave LDS scratch;
//   * records     lane s gathers the 80-byte raster record of the s-th entry straight into registers and
//                 computes, for all entries at once, the nearest depth of the triangle over the quadrant;
//   * walk        entry s is broadcast with v_readlane: its coefficients become wave-uniform SGPR operands.
//                 One compare against the lanes' farthest depths skips a hidden triangle before anything
//                 else is touched (most rejections are of this kind).
// =================================================================================================
template <bool STATS, int DBG = 0>  // DBG: timing experiments only (1 = no queue walk, 2 = reject tests but no pixel bodies)
__global__ __launch_bounds__(256, 4) void raster_wave_kernel(DeviceLevelView lv, const TriRec *__restrict__ recs,
                                                             const uint4 *__restrict__ sorted,
                                                             const uint32_t *__restrict__ counts, uint32_t cap,
                                                             uint32_t n_poses, int width, int height, int tiles_x,
                                                             int tiles_y, const uint2 *__restrict__ tile_hdr,
                                                             const uint32_t *__restrict__ entries, uint32_t entry_cap,
                                                             const uint32_t *__restrict__ overflow,
                                                             uint32_t *__restrict__ vis, uint32_t vis16,
                                                             uint32_t *__restrict__ prim_out,
                                                             unsigned long long *__restrict__ stats) {
  unsigned long long st[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  __shared__ uint32_t wq[4][64];
  const uint32_t b = blockIdx.x;
  const uint32_t T = (uint32_t)(tiles_x * tiles_y);
  const uint32_t g = b >> 3;
  const uint32_t pose = (g / T) * 8u + (b & 7u);
  const uint32_t tile = g % T;
  if (pose >= n_poses) return;
  const int tx0 = (int)(tile % (uint32_t)tiles_x) * TILE_W, ty0 = (int)(tile / (uint32_t)tiles_x) * TILE_H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int qx0 = tx0 + (wave & 1) * 32, qy0 = ty0 + (wave >> 1) * 32;  // this wave's quadrant
  const int bx = qx0 + (lane & 7) * 4, by = qy0 + (lane >> 3) * 4;      // this lane's 4x4 block
  const float pxlo = (float)bx + 0.5f, pxhi = (float)bx + 3.5f, pylo = (float)by + 0.5f, pyhi = (float)by + 3.5f;
  const float qxl = (float)qx0 + 0.5f, qxh = (float)qx0 + 31.5f, qyl = (float)qy0 + 0.5f, qyh = (float)qy0 + 31.5f;
  uint32_t best_d[16], best_r[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    best_d[k] = NONE;
    best_r[k] = NONE;
  }
  uint32_t lane_far = NONE;  // max of best_d: the farthest depth this lane still holds
  const TriRec *prec = recs + (size_t)pose * cap;
  const uint4 *psorted = sorted + (size_t)pose * cap;
  const bool binned = overflow[pose] == 0u;  // the pose's per-tile lists are complete
  const uint2 hdr = binned ? tile_hdr[(size_t)pose * T + tile] : make_uint2(0u, counts[pose]);
  const uint32_t *pent = entries + (size_t)pose * entry_cap + hdr.x;
  const uint32_t count = DBG == 1 ? 0u : hdr.y;
  uint32_t *myq = wq[wave];
  for (uint32_t base = 0; base < count; base += 64u) {
    // ---- candidates: one per lane ------------------------------------------------------------------
    const uint32_t i = base + (uint32_t)lane;
    uint32_t cand = 0;
    bool rel = false;
    if (i < count) {
      if (binned) {
        const uint32_t e = pent[i];
        cand = e & 0x0FFFFFFFu;
        rel = ((e >> (28 + wave)) & 1u) != 0u;  // exact quadrant test done by the binning kernel
      } else {
        // pose without complete bins: every visible triangle is a candidate; bbox, then the exact quadrant test
        const uint4 bb = psorted[i];  // (bb0, bb1, record index, depth bucket), near to far
        cand = bb.z;
        const int x0 = (int)(bb.x & 0xFFFFu), y0 = (int)(bb.x >> 16), x1 = (int)(bb.y & 0xFFFFu), y1 = (int)(bb.y >> 16);
        if (x0 <= qx0 + 31 && x1 >= qx0 && y0 <= qy0 + 31 && y1 >= qy0) {
          const uint4 *rp = reinterpret_cast<const uint4 *>(&prec[cand]);
          rel = rect_may_touch(rp[0], rp[1], rp[2], qxl, qxh, qyl, qyh);
        }
      }
    }
    if (STATS && !binned) st[8] += (unsigned long long)__popcll(__ballot(i < count)), st[9] += (unsigned long long)__popcll(__ballot(rel));
    const unsigned long long rm = __ballot(rel);
    const uint32_t n = (uint32_t)__popcll(rm);
    if (n == 0u) continue;
    // rank of my entry among the relevant ones (record index = depth rank; the lists are near-sorted already)
    uint32_t rank = 0;
    for (unsigned long long m = rm; m; m &= m - 1ull) {
      const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)cand, (int)__builtin_ctzll(m));
      rank += kj < cand ? 1u : 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // earlier readers of myq are done
    if (rel) myq[rank] = cand;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- records: lane s holds entry s ---------------------------------------------------------------
    const bool have = (uint32_t)lane < n;
    const uint32_t myrec = have ? myq[lane] : 0u;
    uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0, c2 = c0, c3 = c0, c4 = c0;
    if (have) {
      const uint4 *rp = reinterpret_cast<const uint4 *>(&prec[myrec]);
      c0 = rp[0], c1 = rp[1], c2 = rp[2], c3 = rp[3], c4 = rp[4];
    }
    // nearest depth of my entry's plane over the quadrant (exact corner argument), as d24; none if beyond far
    uint32_t dnq = NONE;
    {
      const float za = __uint_as_float(c2.y), zb = __uint_as_float(c2.z), zc = __uint_as_float(c2.w);
      const float zn = fmaf(za, za > 0.0f ? qxl : qxh, fmaf(zb, zb > 0.0f ? qyl : qyh, zc));
      if (have && zn <= 1.0f) dnq = __float2uint_rz(fmaf(fminf(fmaxf(zn, 0.0f), 1.0f), 16777215.0f, 0.5f));
    }
    // ---- walk ------------------------------------------------------------------------------------------
    for (uint32_t s = 0; s < n; s++) {
      if (STATS) st[0]++;
      const uint32_t dq = (uint32_t)__builtin_amdgcn_readlane((int)dnq, (int)s);
      // hidden in the whole quadrant: every lane's nearest depth is >= dq (its block lies inside the quadrant)
      if (!__any(dq <= lane_far)) {
        if (STATS) st[15]++;
        continue;
      }
      if (STATS) st[1]++;
      auto bc = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)s); };
      auto bf = [&](uint32_t v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)v, (int)s)); };
      const uint32_t bb0 = bc(c3.w), bb1 = bc(c4.x), flags = bc(c4.y), ridx = bc(myrec);
      const int x0 = (int)(bb0 & 0xFFFFu), y0 = (int)(bb0 >> 16), x1 = (int)(bb1 & 0xFFFFu), y1 = (int)(bb1 >> 16);
      raster_entry<STATS, DBG>(lv, prec, bf(c0.x), bf(c0.y), bf(c0.z), bf(c0.w), bf(c1.x), bf(c1.y), bf(c1.z), bf(c1.w), bf(c2.x),
                               bf(c2.y), bf(c2.z), bf(c2.w), bf(c3.x), bf(c3.y), bf(c3.z), x0, y0, x1, y1, flags, ridx, bx, by,
                               pxlo, pxhi, pylo, pyhi, best_d, best_r, lane_far,
                               [&]() -> ShadeRec { return prec[ridx].s; }, st);
    }
  }
  if (STATS && lane == 0)
    for (int k = 0; k < 16; k++) atomicAdd(&stats[k], st[k]);
#pragma unroll
  for (int ry = 0; ry < 4; ry++) {
    const int iy = by + ry;
    if (iy < height && bx < width) {
      const size_t o = ((size_t)pose * (size_t)height + (size_t)iy) * (size_t)width + (size_t)bx;
      if (vis16)  // record indices fit 16 bits (0xFFFF = none): half the visibility traffic
        *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(vis) + o) =
            make_uint2((best_r[ry * 4] & 0xFFFFu) | (best_r[ry * 4 + 1] << 16),
                       (best_r[ry * 4 + 2] & 0xFFFFu) | (best_r[ry * 4 + 3] << 16));
      else
        *reinterpret_cast<uint4 *>(vis + o) =
            make_uint4(best_r[ry * 4], best_r[ry * 4 + 1], best_r[ry * 4 + 2], best_r[ry * 4 + 3]);
      if (prim_out) {
        uint32_t p[4];
#pragma unroll
        for (int rx = 0; rx < 4; rx++)
          p[rx] = best_r[ry * 4 + rx] == NONE ? NONE : (prec[best_r[ry * 4 + rx]].r.flags & 0xFFFFFFu);
        *reinterpret_cast<uint4 *>(prim_out + o) = make_uint4(p[0], p[1], p[2], p[3]);
      }
    }
  }
}

// =================================================================================================
// Kernel 3: fragment kernel (F1..F6): visibility record -> atlas texel -> COLORMAP row -> 8-bit
// palette index.  One lane per run of 8 (or 4) horizontally adjacent pixels: one 16-byte visibility
// load, one 8-byte packed store; COLORMAP (8 KiB) is staged in LDS once per workgroup, which walks
// FRAG_CHUNK consecutive slabs of one pose (all blocks of a pose run on one XCD).
//
// Packed path (96 % of the runs of an E1M1 sweep): the pixels of the run see the same flat/wall triangle
// whose tile sizes are powers of two or integers.  Its 64-byte shade record is loaded once and the pixels
// are shaded branch-free two at a time (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 evaluate the same IEEE
// operations as their scalar forms, half by half).  Exactness devices, all verified or proven
// (fastmath.hpp; rdoom_selftest_fastmath sweeps them on the device, tests/test_gpu_fastmath.py):
//   * 1/rw        = rcp, fma, fma       -- equals the correctly rounded quotient for EVERY binary32 x with
//   * 0.9/(d+0.9) = rcp, mul, fma, fma     2^-100 <= |x| <= 2^100 (exhaustive sweep on gfx950)
//   * mod by a power of two: x / 2^k == x * 2^-k, and y * floor(q) is exact, so fma(-y, f, x) == x - y * f
//   * mod by an integer tile size: floor(t * RN(1/size)) is certified by a remainder test (see F2)
//   * COLORMAP row: every operation of F1/F4/F5 is monotone and rw is monotone along the run, so when the
//     rows of the two end pixels agree every pixel between them has that row too
// A run of sky is shaded from per-batch ndc tables.  A run that fails any precondition (mixed triangles,
// decor, rw outside the verified range, an uncertified mod, a transparent texel) is appended, quad by quad,
// to a per-wave LDS list and shaded afterwards by the general per-pixel body, lane per pixel -- same
// results, one code path for everything unusual.
// =================================================================================================
constexpr int FRAG_CHUNK = 16;
constexpr int FRAG_WLIST = 160;  // per-wave list of unfinished quads: at most 15 carried over + 64 x 2 new

__device__ __forceinline__ uint32_t shade_sky(const DeviceLevelView &lv, const uint8_t *cmap, float px, float py,
                                              int width, int height, float vr0, float vr1) {
  const float ndc_x = px / (0.5f * (float)width) - 1.0f;
  const float ndc_y = py / (0.5f * (float)height) - 1.0f;
  float uvx = ndc_x;
  float uvy = -ndc_y;
  uvx = uvx - 4.0f * vr0 / 3.14159265358f;
  uvy = (uvy + 1.0f) + vr1;
  const float band = lv.sky_band;
  if (uvy < 0.0f) {
    uvy = fabsf(glsl_mod(-uvy + band, band * 2.0f) - band);
  } else if (uvy >= 2.0f) {
    uvy = fabsf(glsl_mod((uvy - 2.0f) + band, band * 2.0f) - band);
  } else if (uvy >= 1.0f) {
    uvy = 1.0f - uvy;
  }
  const float fx = uvx - floorf(uvx), fy = uvy - floorf(uvy);
  int ix = (int)floorf(fx * (float)lv.sky_w), iy = (int)floorf(fy * (float)lv.sky_h);
  if (ix >= (int)lv.sky_w) ix = (int)lv.sky_w - 1;
  if (iy >= (int)lv.sky_h) iy = (int)lv.sky_h - 1;
  const uint32_t texel = lv.sky_tex[(size_t)iy * lv.sky_w + (size_t)ix];
  return cmap[texel & 0xFFu];
}

// returns the palette index, or 0x100 | index when the winning wall fragment's texel is transparent (the
// rasteriser treated a border-masked texture as opaque and the coordinate leaked onto the ring)
__device__ __forceinline__ uint32_t shade_pixel(const DeviceLevelView &lv, const uint8_t *cmap, const ShadeRec &s,
                                                float px, float py, float row_w, float row_u, float row_v,
                                                int width, int height, const PoseConst &pc) {
  const uint32_t kind = s.flags & 3u;
  if (kind == RDOOM_KIND_SKY) return shade_sky(lv, cmap, px, py, width, height, s.atlas_u, s.atlas_v);
  const TexelAt t = texel_coords(s, px, row_w, row_u, row_v);
  const uint32_t texel = load_texel(lv, s, t.ix, t.iy);
  if (kind != RDOOM_KIND_FLAT && (texel & 0x8000u)) return 0x100u;
  float light;
  if (kind == RDOOM_KIND_DECOR) {  // sprite.frag:22-25: DIST_SCALE = 1, light = min(v_light, 2 v_light - dist_term)
    const float dist_term = fminf(1.0f, 1.0f - 1.0f / (t.dist + 1.0f));
    light = fminf(s.light, s.light * 2.0f - dist_term);
  } else {
    const float dist_term = fminf(1.0f, 1.0f - 0.9f / (t.dist + 0.9f));
    light = s.light * 2.0f - dist_term;
  }
  const float tt = (1.0f - light) * 32.0f;
  const int rowc = tt < 0.0f ? 0 : (tt >= 32.0f ? 31 : (int)floorf(tt));
  return cmap[rowc * 256 + (int)(texel & 0xFFu)];
}

// idx / d for idx < 2^24 by multiply-high (m, sh) computed and verified on the host
__device__ __forceinline__ uint32_t fast_div(uint32_t idx, uint32_t m, uint32_t sh) { return __umulhi(idx, m) >> sh; }

template <int NQ, int DBG, bool VIS16>  // NQ: adjacent quads per lane (1 or 2; the frame width is a multiple of 4 NQ);
                                        // DBG: timing experiments only; VIS16: 16-bit visibility words (0xFFFF = none)
__global__ __launch_bounds__(256) void fragment_kernel(DeviceLevelView lv, const TriRec *__restrict__ recs,
                                                       uint32_t cap, const PoseConst *__restrict__ poses,
                                                       const uint32_t *__restrict__ vis, uint32_t n_poses,
                                                       uint32_t chunks_per_pose, uint32_t chunk_iters,
                                                       uint32_t quads_per_pose,
                                                       uint32_t quads_per_row, uint32_t div_m, uint32_t div_sh,
                                                       int width, int height, const float *__restrict__ ndc_tab,
                                                       uint8_t *__restrict__ fb, uint32_t *__restrict__ fix_count,
                                                       uint2 *__restrict__ fix_list, uint32_t fix_cap,
                                                       uint32_t debug_leak_mod) {
  constexpr int NP = 2 * NQ, NPX = 4 * NQ;  // float2 pairs and pixels per lane
  __shared__ uint8_t cmap[32 * 256];
  __shared__ uint32_t wlist[4][FRAG_WLIST];
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(lv.colormap);
    uint4 *dst = reinterpret_cast<uint4 *>(cmap);
    dst[threadIdx.x] = src[threadIdx.x];
    dst[threadIdx.x + 256] = src[threadIdx.x + 256];
  }
  __syncthreads();
  // blockIdx -> (pose, chunk): all chunks of a pose on one XCD (b % 8), like the rasteriser
  const uint32_t g = blockIdx.x >> 3;
  const uint32_t pose = (g / chunks_per_pose) * 8u + (blockIdx.x & 7u);
  const uint32_t chunk = g % chunks_per_pose;
  if (pose >= n_poses) return;
  const TriRec *prec = recs + (size_t)pose * cap;
  constexpr uint32_t NONE_ID = VIS16 ? 0xFFFFu : NONE;
  const uint32_t *pvis32 = vis + (size_t)pose * quads_per_pose * 4u;
  const uint16_t *pvis16 = reinterpret_cast<const uint16_t *>(vis) + (size_t)pose * quads_per_pose * 4u;
  uint32_t *pfb = reinterpret_cast<uint32_t *>(fb) + (size_t)pose * quads_per_pose;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t *mylist = wlist[threadIdx.x >> 6];
  uint32_t wn = 0;  // wave-uniform: quads waiting in mylist
  const PoseConst &pc = poses[pose];
  // general body: 16 listed quads at a time, lane per pixel.  The list is private to the wave (LDS operations of
  // one wave execute in order), so no workgroup barrier is involved and waves never wait for each other.
  auto shade_listed = [&](uint32_t first, uint32_t count) {
    const uint32_t j = lane >> 2, k = lane & 3u;
    if (j < count) {
      const uint32_t qi = mylist[first + j];
      const uint32_t row = fast_div(qi, div_m, div_sh), qx = qi - row * quads_per_row;
      const uint32_t id = VIS16 ? (uint32_t)pvis16[(size_t)qi * 4u + k] : pvis32[(size_t)qi * 4u + k];
      uint32_t c = 0;
      const uint32_t pix = (row * quads_per_row + qx) * 4u + k;
      if (id != NONE_ID) {
        const ShadeRec cur = prec[id].s;
        const float py = (float)row + 0.5f, px = (float)(qx * 4u + k) + 0.5f;
        c = shade_pixel(lv, cmap, cur, px, py, fmaf(cur.wp[1], py, cur.wp[2]), fmaf(cur.up[1], py, cur.up[2]),
                        fmaf(cur.vp[1], py, cur.vp[2]), width, height, pc);
        // debug_leak_mod != 0 (tests only): pretend every n-th pixel leaked, so fixup_kernel's general rule
        // is exercised on ordinary pixels too -- the output must not change
        const bool forced = debug_leak_mod != 0u && pix % debug_leak_mod == 0u;
        if ((c & 0x100u) || forced) {  // rare: alpha leak, queue the pixel for exact re-resolution
          const uint32_t slot = atomicAdd(fix_count, 1u);
          if (slot < fix_cap) fix_list[slot] = make_uint2(pose, pix);
        }
      }
      uint32_t v = (c & 0xFFu) << (8u * k);
      v |= __shfl_xor(v, 1);
      v |= __shfl_xor(v, 2);
      if (k == 0) pfb[qi] = v;
    }
  };
  const uint32_t units_per_pose = quads_per_pose / (uint32_t)NQ;  // a unit = the NQ adjacent quads of one lane
  for (uint32_t it = 0; it < chunk_iters; it++) {
    const uint32_t ui = (chunk * chunk_iters + it) * 256u + threadIdx.x;
    if (ui - lane >= units_per_pose) break;  // wave-uniform: the whole wave is past the end of the frame
    const bool valid = ui < units_per_pose;
    const uint32_t q0 = ui * (uint32_t)NQ;
    // visibility words of my NPX pixels
    uint32_t id[NPX];
#pragma unroll
    for (int k = 0; k < NPX; k++) id[k] = NONE_ID;
    if (valid) {
      if (VIS16) {
        if (NQ == 2) {
          const uint4 v = *reinterpret_cast<const uint4 *>(pvis16 + (size_t)q0 * 4u);
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < NPX; k++) id[k] = (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xFFFFu);
        } else {
          const uint2 v = *reinterpret_cast<const uint2 *>(pvis16 + (size_t)q0 * 4u);
          id[0] = v.x & 0xFFFFu, id[1] = v.x >> 16, id[2] = v.y & 0xFFFFu, id[3] = v.y >> 16;
        }
      } else {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const uint4 v = *reinterpret_cast<const uint4 *>(pvis32 + ((size_t)q0 + (size_t)q) * 4u);
          id[4 * q] = v.x, id[4 * q + 1] = v.y, id[4 * q + 2] = v.z, id[4 * q + 3] = v.w;
        }
      }
    }
    bool uniform = true;
#pragma unroll
    for (int k = 1; k < NPX; k++) uniform &= id[k] == id[0];
    bool done = false;
    uint32_t out[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) out[q] = 0;
    if (uniform & (id[0] == NONE_ID)) done = true;  // background (or past the end: nothing is stored)
    if (uniform & (id[0] != NONE_ID) & (debug_leak_mod == 0u)) {
      const uint4 *rp = reinterpret_cast<const uint4 *>(&prec[id[0]].s);
      const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
      const uint32_t flags = r3.z, tex = r3.w;
      // (all four loads are issued before the flag is examined: one memory latency, not two)
      asm volatile("" ::"v"(r0.x), "v"(r1.x), "v"(r2.x));
      if (flags & SHADE_FAST) {
        const float wa = __uint_as_float(r0.x), wb = __uint_as_float(r0.y), wc = __uint_as_float(r0.z),
                    ua = __uint_as_float(r0.w), ub = __uint_as_float(r1.x), uc = __uint_as_float(r1.y),
                    va = __uint_as_float(r1.z), vb = __uint_as_float(r1.w), vc = __uint_as_float(r2.x),
                    atlas_u = __uint_as_float(r2.y), atlas_v = __uint_as_float(r2.z), size_x = __uint_as_float(r2.w),
                    size_y = __uint_as_float(r3.x), light = __uint_as_float(r3.y);
        const uint32_t row = fast_div(q0, div_m, div_sh), qx = q0 - row * quads_per_row;
        const float py = (float)row + 0.5f;
        const float px0 = (float)(qx * 4u) + 0.5f;
        const float row_w = fmaf(wb, py, wc), row_u = fmaf(ub, py, uc), row_v = fmaf(vb, py, vc);
        // F2 preparation: q0 = t * RN(1/size) equals the quotient exactly for a power-of-two size; for an integer
        // size it is within |t/size| * 2^-23 of it, and the remainder test below certifies
        // floor(q0) == floor(RN(t / size)) (else the run goes to the general body).
        const f32x2 inv_s = exact_rcp2(f32x2{size_x, size_y});
        // F3 parameters: one u16 texel store, REPEAT = masks
        const uint32_t wm = tex & 0xFFFFu, hm = tex >> 16, lw = (flags >> 8) & 15u, base = (flags >> 16) << 10;
        const uint32_t base2 = base * 2u;  // byte offsets < 2^27: one 32-bit VGPR offset from the uniform base pointer
        const char *tb = reinterpret_cast<const char *>(lv.texels);
        // Certificate for integer (non-power-of-two) tile sizes, evaluated only in waves that hold such a record
        // (fastmath.hpp, mod_cert): with guard >= 2^-20 * max(|x|, y), guard <= r <= y - guard and |x| < 2^23 imply that
        // no integer lies between x * RN(1/y) and RN(x / y) and that y * floor is exact.  One guard per run and axis:
        // |x_k| = |n_k * w_k| <= max(|n_first|, |n_last|) * max(w_first, w_last) because the numerator plane n and, for
        // rw > 0, w = 1/rw are monotone along the run (and rounding is monotone).  The run's guard is at least every
        // pixel's own guard, so passing here implies mod_cert() for each pixel -- the form the on-device self-test sweeps.
        // Power-of-two axes always pass.
        const bool any_np2 = __any((flags & SHADE_NP2) != 0u);
        const bool p2x = (flags & SHADE_POW2_X) != 0u, p2y = (flags & SHADE_POW2_Y) != 0u;
        bool mod_ok = true;
        float lox = 0.0f, hix = 0.0f, loy = 0.0f, hiy = 0.0f;
        if (any_np2) {
          const float pxl = px0 + (float)(NPX - 1);
          const f32x2 w_ends = exact_rcp2(f32x2{fmaf(wa, px0, row_w), fmaf(wa, pxl, row_w)});
          const float w_hi = fmaxf(w_ends.x, w_ends.y);
          const float bu = fmaxf(fabsf(fmaf(ua, px0, row_u)), fabsf(fmaf(ua, pxl, row_u))) * w_hi;
          const float bv = fmaxf(fabsf(fmaf(va, px0, row_v)), fabsf(fmaf(va, pxl, row_v))) * w_hi;
          lox = fmaxf(bu, size_x) * 0x1p-20f, hix = size_x - lox;
          loy = fmaxf(bv, size_y) * 0x1p-20f, hiy = size_y - loy;
          mod_ok = (p2x | (bu < 0x1p23f)) & (p2y | (bv < 0x1p23f));
        }
        f32x2 ww[NP];
        uint32_t texel[NPX], any_texel = 0;
        float rw_first = 0.0f, rw_last = 0.0f;
#pragma unroll
        for (int p = 0; p < NP; p++) {  // one pair of pixels at a time, straight through to its two texel loads
          const f32x2 px = {px0 + (float)(2 * p), px0 + (float)(2 * p + 1)};
          const f32x2 rw = pk_fma(splat(wa), px, splat(row_w));  // F1
          if (p == 0) rw_first = rw.x;
          if (p == NP - 1) rw_last = rw.y;
          ww[p] = exact_rcp2(rw);
          const f32x2 tu = pk_fma(splat(ua), px, splat(row_u)) * ww[p];
          const f32x2 tv = pk_fma(splat(va), px, splat(row_v)) * ww[p];
          f32x2 fq = tu * splat(inv_s.x);  // F2: mod(t, size) = t - size * floor(t / size)
          fq = f32x2{floorf(fq.x), floorf(fq.y)};
          const f32x2 rx = pk_fma(splat(-size_x), fq, tu);
          f32x2 fh = tv * splat(inv_s.y);
          fh = f32x2{floorf(fh.x), floorf(fh.y)};
          const f32x2 ry = pk_fma(splat(-size_y), fh, tv);
          if (any_np2)
            mod_ok = mod_ok & (p2x | ((rx.x >= lox) & (rx.x <= hix) & (rx.y >= lox) & (rx.y <= hix))) &
                     (p2y | ((ry.x >= loy) & (ry.x <= hiy) & (ry.y >= loy) & (ry.y <= hiy)));
          const f32x2 ux = rx + splat(atlas_u), uy = ry + splat(atlas_v);  // F3
          const uint32_t o0 = (((uint32_t)cvt_floor_i32(uy.x) & hm) << lw) | ((uint32_t)cvt_floor_i32(ux.x) & wm);
          const uint32_t o1 = (((uint32_t)cvt_floor_i32(uy.y) & hm) << lw) | ((uint32_t)cvt_floor_i32(ux.y) & wm);
          texel[2 * p] = (DBG & 2) ? (o0 & 255u) : *reinterpret_cast<const uint16_t *>(tb + (o0 * 2u + base2));
          texel[2 * p + 1] = (DBG & 2) ? (o1 & 255u) : *reinterpret_cast<const uint16_t *>(tb + (o1 * 2u + base2));
        }
        // rw is monotone along the run: both ends inside the verified range of the exact reciprocal forms
        const bool in_range = (fminf(rw_first, rw_last) >= 0x1p-100f) & (fmaxf(rw_first, rw_last) <= 0x1p100f);
#pragma unroll
        for (int k = 0; k < NPX; k++) any_texel |= texel[k];
        // F4, F5 at the two end pixels; the pixels between them only when the ends disagree
        auto rows_of = [&](f32x2 dist) {
          const f32x2 dterm = splat(1.0f) - exact_div09_2(dist + splat(0.9f));
          const f32x2 lgt = splat(light * 2.0f) - f32x2{fminf(1.0f, dterm.x), fminf(1.0f, dterm.y)};
          const f32x2 tt = (splat(1.0f) - lgt) * splat(32.0f);
          return f32x2{fminf(fmaxf(floorf(tt.x), 0.0f), 31.0f), fminf(fmaxf(floorf(tt.y), 0.0f), 31.0f)};
        };
        const f32x2 rf_ends = rows_of(f32x2{ww[0].x, ww[NP - 1].y});
        f32x2 rf[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) rf[p] = splat(rf_ends.x);
        if (rf_ends.x != rf_ends.y) {
#pragma unroll
          for (int p = 0; p < NP; p++) rf[p] = rows_of(ww[p]);
        }
        const bool opaque = (any_texel & 0x8000u) == 0u;
        if (in_range & mod_ok & opaque) {
#pragma unroll
          for (int p = 0; p < NP; p++) {
            const uint32_t c0 = cmap[((uint32_t)(int)rf[p].x << 8) | (texel[2 * p] & 0xFFu)],
                           c1 = cmap[((uint32_t)(int)rf[p].y << 8) | (texel[2 * p + 1] & 0xFFu)];
            out[p >> 1] |= (c0 | (c1 << 8)) << (16 * (p & 1));
          }
          done = true;
        }
      } else if ((flags & 3u) == RDOOM_KIND_SKY) {
        // a run of sky (sky.frag:12-26): the colour depends on the pixel and the pose only.  ndc_tab holds
        // p / (size / 2) - 1 for every column and row of the frame (computed once per batch with the same two
        // operations), the record carries v_r.y and 4 v_r.x / 3.14159265358; the row part is evaluated once per run.
        const uint32_t row = fast_div(q0, div_m, div_sh), qx = q0 - row * quads_per_row;
        const float ushift = __uint_as_float(r2.w), vr1 = __uint_as_float(r2.z), band = lv.sky_band;
        float uvy = (-ndc_tab[(uint32_t)width + row] + 1.0f) + vr1;
        if (uvy < 0.0f) {
          uvy = fabsf(glsl_mod(-uvy + band, band * 2.0f) - band);
        } else if (uvy >= 2.0f) {
          uvy = fabsf(glsl_mod((uvy - 2.0f) + band, band * 2.0f) - band);
        } else if (uvy >= 1.0f) {
          uvy = 1.0f - uvy;
        }
        const float fy = uvy - floorf(uvy);
        int iy = (int)floorf(fy * (float)lv.sky_h);
        if (iy >= (int)lv.sky_h) iy = (int)lv.sky_h - 1;
        const uint16_t *srow = lv.sky_tex + (size_t)iy * lv.sky_w;
        uint32_t c[NPX];
#pragma unroll
        for (int k = 0; k < NPX; k++) {
          const float uvx = ndc_tab[qx * 4u + (uint32_t)k] - ushift;
          const float fx = uvx - floorf(uvx);
          int ix = (int)floorf(fx * (float)lv.sky_w);
          if (ix >= (int)lv.sky_w) ix = (int)lv.sky_w - 1;
          c[k] = cmap[srow[ix] & 0xFFu];
        }
#pragma unroll
        for (int k = 0; k < NPX; k++) out[k >> 2] |= c[k] << (8 * (k & 3));
        done = true;
      }
    }
    if (done & valid) {
      if (NQ == 2)
        *reinterpret_cast<uint2 *>(pfb + q0) = make_uint2(out[0], out[NQ - 1]);
      else
        pfb[q0] = out[0];
    }
    const unsigned long long sm = __ballot(!done);
    if (sm) {  // ordered append of this wave's unfinished quads, then shade full groups of 16
      if (!done) {
        const uint32_t at = wn + (uint32_t)NQ * (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
#pragma unroll
        for (int q = 0; q < NQ; q++) mylist[at + (uint32_t)q] = q0 + (uint32_t)q;
      }
      wn += (uint32_t)NQ * (uint32_t)__popcll(sm);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      while (wn >= 16u) {
        wn -= 16u;
        shade_listed(wn, 16u);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (wn) shade_listed(0u, wn);
}

// =================================================================================================
// Kernel 4: fixup.  Re-resolves the (rare) pixels queued by the fragment kernel with the general rule
// R1..R6 applied to every candidate of the pixel's tile: lanes = candidates, lexicographic wave-min of
// (d24, primitive), then the winner is shaded.  One wave per queued pixel; the list is usually empty.
// =================================================================================================
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_xor(v, d);
    v = o < v ? o : v;
  }
  return v;
}

__global__ __launch_bounds__(256) void fixup_kernel(DeviceLevelView lv, const TriRec *__restrict__ recs,
                                                    const uint4 *__restrict__ sorted,
                                                    const uint32_t *__restrict__ counts, uint32_t cap,
                                                    const PoseConst *__restrict__ poses, int width, int height,
                                                    int tiles_x, int tiles_y, const uint2 *__restrict__ tile_hdr,
                                                    const uint32_t *__restrict__ entries, uint32_t entry_cap,
                                                    const uint32_t *__restrict__ overflow,
                                                    const uint32_t *__restrict__ fix_count,
                                                    const uint2 *__restrict__ fix_list, uint32_t fix_cap,
                                                    uint32_t *__restrict__ vis, uint32_t vis16,
                                                    uint32_t *__restrict__ prim_out, uint8_t *__restrict__ fb,
                                                    uint32_t *__restrict__ error_flag) {
  const uint32_t total = *fix_count;
  if (total > fix_cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *error_flag = 1u;
    return;
  }
  const uint32_t lane = threadIdx.x & 63u, wave_id = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
  for (uint32_t item = wave_id; item < total; item += n_waves) {
    const uint2 it = fix_list[item];
    const uint32_t pose = it.x, pix = it.y;
    const int iy = (int)(pix / (uint32_t)width), ix = (int)(pix - (uint32_t)iy * (uint32_t)width);
    const float px = (float)ix + 0.5f, py = (float)iy + 0.5f;
    const TriRec *prec = recs + (size_t)pose * cap;
    const bool binned = overflow[pose] == 0u;
    const uint32_t T = (uint32_t)(tiles_x * tiles_y), tile = (uint32_t)((iy >> 6) * tiles_x + (ix >> 6));
    const uint2 hdr = binned ? tile_hdr[(size_t)pose * T + tile] : make_uint2(0u, counts[pose]);
    unsigned long long best = ~0ull;
    uint32_t best_rec = NONE;
    for (uint32_t base = 0; base < hdr.y; base += 64u) {
      const uint32_t e = base + lane;
      unsigned long long key = ~0ull;
      uint32_t rec = NONE;
      if (e < hdr.y) {
        rec = binned ? (entries[(size_t)pose * entry_cap + hdr.x + e] & 0x0FFFFFFFu) : sorted[(size_t)pose * cap + e].z;
        const RasterRec r = prec[rec].r;
        const int x0 = (int)(r.bb0 & 0xFFFFu), y0 = (int)(r.bb0 >> 16), x1 = (int)(r.bb1 & 0xFFFFu),
                  y1 = (int)(r.bb1 >> 16);
        const float e0 = fmaf(r.e[0], px, fmaf(r.e[1], py, r.e[2])), e1 = fmaf(r.e[3], px, fmaf(r.e[4], py, r.e[5])),
                    e2 = fmaf(r.e[6], px, fmaf(r.e[7], py, r.e[8]));
        const bool in0 = (e0 > 0.0f) | ((e0 == 0.0f) & ((r.flags & (1u << 24)) != 0u));
        const bool in1 = (e1 > 0.0f) | ((e1 == 0.0f) & ((r.flags & (1u << 25)) != 0u));
        const bool in2 = (e2 > 0.0f) | ((e2 == 0.0f) & ((r.flags & (1u << 26)) != 0u));
        const float zw = fmaf(r.zp[0], px, fmaf(r.zp[1], py, r.zp[2]));
        const float rw = fmaf(r.wp[0], px, fmaf(r.wp[1], py, r.wp[2]));
        bool pass = (ix >= x0) & (ix <= x1) & (iy >= y0) & (iy <= y1) & in0 & in1 & in2 & (zw >= 0.0f) & (zw <= 1.0f) &
                    (rw > 0.0f);
        if (pass && (r.flags & RASTER_MASKED_ANY) != 0u) {
          const ShadeRec sh = prec[rec].s;
          const TexelAt t = texel_coords(sh, px, fmaf(sh.wp[1], py, sh.wp[2]), fmaf(sh.up[1], py, sh.up[2]),
                                         fmaf(sh.vp[1], py, sh.vp[2]));
          pass = (load_texel(lv, sh, t.ix, t.iy) & 0x8000u) == 0u;
        }
        if (pass) {
          const uint32_t d24 = __float2uint_rz(fmaf(fminf(fmaxf(zw, 0.0f), 1.0f), 16777215.0f, 0.5f));
          key = ((unsigned long long)d24 << 32) | (unsigned long long)(r.flags & 0xFFFFFFu);
        }
      }
      const unsigned long long m = wave_min_u64(key);
      if (m < best) {
        best = m;
        const unsigned long long who = __ballot(key == m);
        best_rec = __shfl(rec, __ffsll((long long)who) - 1);
      }
    }
    if (lane == 0) {
      const size_t o = ((size_t)pose * (size_t)height + (size_t)iy) * (size_t)width + (size_t)ix;
      uint32_t colour = 0;
      if (best_rec != NONE) {
        const ShadeRec sh = prec[best_rec].s;
        colour = shade_pixel(lv, lv.colormap, sh, px, py, fmaf(sh.wp[1], py, sh.wp[2]), fmaf(sh.up[1], py, sh.up[2]),
                             fmaf(sh.vp[1], py, sh.vp[2]), width, height, poses[pose]) & 0xFFu;
      }
      if (vis16)
        reinterpret_cast<uint16_t *>(vis)[o] = (uint16_t)best_rec;  // NONE -> 0xFFFF
      else
        vis[o] = best_rec;
      if (prim_out) prim_out[o] = best_rec == NONE ? NONE : (uint32_t)(best & 0xFFFFFFull);
      fb[o] = (uint8_t)colour;
    }
  }
}

// =================================================================================================
// host side
// =================================================================================================
#define HIP_TRY(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess)                                                                                 \
      return rdoom::fail(_e == hipErrorOutOfMemory ? RDOOM_OOM : RDOOM_HIP_ERROR, "%s failed: %s", #expr, \
                         hipGetErrorString(_e));                                                          \
  } while (0)

bool is_pow2(uint32_t x) { return x != 0 && (x & (x - 1)) == 0; }

}  // namespace

struct rdoom_level {
  int device = 0;
  DeviceLevelView view{};
  void *d_tris = nullptr, *d_flat = nullptr, *d_wall = nullptr, *d_sky = nullptr, *d_cmap = nullptr;
  uint32_t ntri = 0, n_objects = 1;
};

struct rdoom_batch {
  const rdoom_level *level = nullptr;
  uint32_t width = 0, height = 0, max_poses = 0, cap = 0, last_n = 0;
  PoseConst *d_poses = nullptr;
  TriRec *d_recs = nullptr;   // max_poses x cap records in near-to-far order (setup -> bin, raster, fragment)
  TriRec *d_tmp_recs = nullptr;  // same size: setup's compaction-order staging
  uint4 *d_sorted = nullptr;  // per pose: (bbox, record index, depth bucket) near-to-far (coarse test input)
  uint2 *d_tile_hdr = nullptr;     // per (pose, tile): (first entry, entry count)
  uint32_t *d_entries = nullptr;   // per pose: entry_cap tile-list entries (record index | quadrant mask << 28)
  uint32_t *d_overflow = nullptr;  // per pose: 1 = bins incomplete, rasteriser scans the sorted list
  uint32_t entry_cap = 0, n_tiles = 0;
  uint32_t *d_fix_count = nullptr;  // [0] = queued pixels, [1] = error flag (fixup list overflow)
  uint2 *d_fix_list = nullptr;
  uint32_t fix_cap = 1u << 20;
  uint32_t *d_counts = nullptr, *d_vis = nullptr, *d_prim = nullptr;
  uint8_t *d_fb = nullptr;
  PoseConst *h_poses = nullptr;  // pinned staging for the per-pose constants
  ObjectConst *d_objects = nullptr, *h_objects = nullptr;  // max_poses x n_objects, allocated on first use
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_copy = nullptr;  // H2D of h_poses finished: staging may be rewritten
  bool want_prim = false;
  bool vis16 = false;  // record indices fit 16 bits: visibility words are u16
  float *d_ndc = nullptr;  // (ix + 0.5) / (width / 2) - 1 for every column, then (iy + 0.5) / (height / 2) - 1 for every row
};

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
  for (void *p : {level->d_tris, level->d_flat, level->d_wall, level->d_sky, level->d_cmap})
    if (p) (void)hipFree(p);
  delete level;
}

rdoom_status rdoom_level_create(const rdoom_level_desc *d, rdoom_level **out_level) {
  if (!d || !out_level) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out_level = nullptr;
  if (!d->colormap) return rdoom::fail(RDOOM_BAD_ARG, "colormap is null");
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
  std::vector<LevelTri> tris;
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
  rdoom_level *lv = new rdoom_level;
  (void)hipGetDevice(&lv->device);
  lv->ntri = (uint32_t)tris.size();
  lv->n_objects = n_objects;
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
  // unified u16 texel store: wall atlas, then (at a multiple of 1024 elements) the flat atlas promoted to u16
  const size_t wall_n = d->wall_atlas ? (size_t)d->wall_w * d->wall_h : 0;
  const size_t flat_n = d->flat_atlas ? (size_t)d->flat_w * d->flat_h : 0;
  const size_t decor_n = d->decor_atlas ? (size_t)d->decor_w * d->decor_h : 0;
  const size_t flat_base = (wall_n + 1023) / 1024 * 1024;
  const size_t decor_base = (flat_base + flat_n + 1023) / 1024 * 1024;
  if (decor_base + decor_n >= ((size_t)1 << 26)) {
    rdoom_level_destroy(lv);
    return rdoom::fail(RDOOM_BAD_LEVEL, "atlases too large (%zu texels)", decor_base + decor_n);
  }
  std::vector<uint16_t> texels(decor_base + decor_n + 1, 0);  // never empty: masked-off lanes read element 0
  if (wall_n) std::memcpy(texels.data(), d->wall_atlas, wall_n * 2);
  for (size_t i = 0; i < flat_n; i++) texels[flat_base + i] = d->flat_atlas[i];
  if (decor_n) std::memcpy(texels.data() + decor_base, d->decor_atlas, decor_n * 2);
  if (e == hipSuccess) e = upload(&lv->d_wall, texels.data(), texels.size() * 2);
  if (e == hipSuccess) e = upload(&lv->d_sky, d->sky_texture, (size_t)d->sky_w * d->sky_h * 2);
  if (e == hipSuccess) e = upload(&lv->d_cmap, d->colormap, 32 * 256);
  if (e != hipSuccess) {
    rdoom_level_destroy(lv);
    return rdoom::fail(e == hipErrorOutOfMemory ? RDOOM_OOM : RDOOM_HIP_ERROR, "level upload failed: %s",
                       hipGetErrorString(e));
  }
  lv->view.tris = (const LevelTri *)lv->d_tris;
  lv->view.ntri = lv->ntri;
  lv->view.texels = (const uint16_t *)lv->d_wall;
  lv->view.flat_base = (uint32_t)flat_base;
  lv->view.decor_base = (uint32_t)decor_base;
  lv->view.decor_w = d->decor_atlas ? d->decor_w : 0;
  lv->view.decor_h = d->decor_atlas ? d->decor_h : 0;
  lv->view.flat_w = d->flat_w;
  lv->view.flat_h = d->flat_h;
  lv->view.wall_w = d->wall_w;
  lv->view.wall_h = d->wall_h;
  lv->view.sky_tex = (const uint16_t *)lv->d_sky;
  lv->view.sky_w = lv->d_sky ? d->sky_w : 0;
  lv->view.sky_h = lv->d_sky ? d->sky_h : 0;
  lv->view.sky_band = d->sky_tiled_band_size;
  lv->view.colormap = (const uint8_t *)lv->d_cmap;
  *out_level = lv;
  return RDOOM_OK;
}

void rdoom_batch_destroy(rdoom_batch *b) {
  if (!b) return;
  for (void *p : {(void *)b->d_poses, (void *)b->d_recs, (void *)b->d_tmp_recs, (void *)b->d_sorted, (void *)b->d_tile_hdr, (void *)b->d_entries,
                  (void *)b->d_overflow, (void *)b->d_fix_count, (void *)b->d_fix_list, (void *)b->d_counts, (void *)b->d_vis,
                  (void *)b->d_prim, (void *)b->d_fb})
    if (p) (void)hipFree(p);
  for (auto &e : b->ev)
    if (e) (void)hipEventDestroy(e);
  if (b->ev_copy) (void)hipEventDestroy(b->ev_copy);
  if (b->h_poses) (void)hipHostFree(b->h_poses);
  if (b->h_objects) (void)hipHostFree(b->h_objects);
  if (b->d_ndc) (void)hipFree(b->d_ndc);
  if (b->d_objects) (void)hipFree(b->d_objects);
  delete b;
}

rdoom_status rdoom_batch_create(const rdoom_level *level, uint32_t width, uint32_t height, uint32_t max_poses,
                                rdoom_batch **out_batch) {
  if (!level || !out_batch) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out_batch = nullptr;
  if (width == 0 || height == 0 || max_poses == 0 || width % 4 != 0 || width > 16384 || height > 16384)
    return rdoom::fail(RDOOM_BAD_ARG, "bad frame size %ux%u (width must be a multiple of 4) or max_poses %u", width,
                       height, max_poses);
  rdoom_batch *b = new rdoom_batch;
  b->level = level;
  b->width = width;
  b->height = height;
  b->max_poses = max_poses;
  b->cap = level->ntri ? level->ntri : 1;
  b->vis16 = b->cap < 0xFFFFu && getenv("RDOOM_VIS32") == nullptr;  // RDOOM_VIS32: tests force the 32-bit words
  const size_t npx = (size_t)width * height * max_poses;
  hipError_t e = hipMalloc((void **)&b->d_poses, sizeof(PoseConst) * max_poses);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_recs, sizeof(TriRec) * (size_t)b->cap * max_poses);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_tmp_recs, sizeof(TriRec) * (size_t)b->cap * max_poses);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_sorted, sizeof(uint4) * (size_t)b->cap * max_poses);
  b->n_tiles = ((width + TILE_W - 1) / TILE_W) * ((height + TILE_H - 1) / TILE_H);
  b->entry_cap = std::max<uint32_t>(65536u, 32u * b->n_tiles);  // tile-list entries per pose; beyond it the pose is scanned
  if (const char *dbg = getenv("RDOOM_ENTRY_CAP")) b->entry_cap = (uint32_t)std::max(1, atoi(dbg));  // tests: force that fallback
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_tile_hdr, sizeof(uint2) * (size_t)b->n_tiles * max_poses);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_entries, sizeof(uint32_t) * (size_t)b->entry_cap * max_poses);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_overflow, sizeof(uint32_t) * max_poses);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_fix_count, 2 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_fix_list, sizeof(uint2) * (size_t)b->fix_cap);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_counts, sizeof(uint32_t) * max_poses);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_vis, sizeof(uint32_t) * npx);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_fb, npx);
  if (e == hipSuccess) {  // sky.frag:13's ndc per column / row, same two operations as the per-pixel form
    std::vector<float> ndc(width + height);
    for (uint32_t i = 0; i < width; i++) ndc[i] = ((float)i + 0.5f) / (0.5f * (float)width) - 1.0f;
    for (uint32_t i = 0; i < height; i++) ndc[width + i] = ((float)i + 0.5f) / (0.5f * (float)height) - 1.0f;
    e = hipMalloc((void **)&b->d_ndc, sizeof(float) * ndc.size());
    if (e == hipSuccess) e = hipMemcpy(b->d_ndc, ndc.data(), sizeof(float) * ndc.size(), hipMemcpyHostToDevice);
  }
  for (auto &ev : b->ev)
    if (e == hipSuccess) e = hipEventCreate(&ev);
  if (e == hipSuccess) e = hipEventCreate(&b->ev_copy);
  if (e == hipSuccess) e = hipHostMalloc((void **)&b->h_poses, sizeof(PoseConst) * max_poses, hipHostMallocDefault);
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

static rdoom_status render_impl(rdoom_batch *b, const rdoom_pose *poses, const uint8_t *lights, uint32_t lights_stride,
                                uint32_t n, uint32_t kinds_mask, hipStream_t st, rdoom_timings *tm,
                                const float *object_modelviews = nullptr, uint32_t n_objects = 0) {
  if (!b || !poses || !lights) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  if (n == 0 || n > b->max_poses) return rdoom::fail(RDOOM_BAD_ARG, "n_poses %u outside 1..%u", n, b->max_poses);
  const rdoom_level *lv = b->level;
  if (object_modelviews && n_objects < lv->n_objects)
    return rdoom::fail(RDOOM_BAD_ARG, "n_objects %u but the level draws objects 0..%u", n_objects, lv->n_objects - 1);
  HIP_TRY(hipEventSynchronize(b->ev_copy));  // previous render's H2D must be done before restaging
  if (object_modelviews) {
    const size_t count = (size_t)b->max_poses * lv->n_objects;
    if (!b->d_objects) HIP_TRY(hipMalloc((void **)&b->d_objects, sizeof(ObjectConst) * count));
    if (!b->h_objects) HIP_TRY(hipHostMalloc((void **)&b->h_objects, sizeof(ObjectConst) * count, hipHostMallocDefault));
    for (uint32_t p = 0; p < n; p++)
      for (uint32_t o = 0; o < lv->n_objects; o++) {
        ObjectConst &oc = b->h_objects[(size_t)p * lv->n_objects + o];
        const float *M = object_modelviews + ((size_t)p * n_objects + o) * 16;
        mat_mul_v1(poses[p].projection, M, oc.pm);
        std::memcpy(oc.mv, M, sizeof oc.mv);
        oc.vr0 = atan2f(oc.pm[8], oc.pm[10]);  // sky.vert:10-12
        oc.vr1 = oc.pm[9] / oc.pm[11];
        oc.pad0 = oc.pad1 = 0;
      }
  }
  for (uint32_t p = 0; p < n; p++) {  // V1: PM = P * M, plain multiply/add, left to right
    PoseConst &pc = b->h_poses[p];
    const float *P = poses[p].projection, *M = poses[p].modelview;
    mat_mul_v1(P, M, pc.pm);
    std::memcpy(pc.mv, M, sizeof pc.mv);
    std::memcpy(pc.proj, P, sizeof pc.proj);
    pc.time = poses[p].time;
    pc.vr0 = atan2f(pc.pm[8], pc.pm[10]);  // sky.vert:10-12
    pc.vr1 = pc.pm[9] / pc.pm[11];
    pc.pad = 0;
    std::memcpy(pc.lights, lights + (size_t)p * lights_stride, 256);
  }
  b->last_n = n;
  if (tm) HIP_TRY(hipEventRecord(b->ev[0], st));
  HIP_TRY(hipMemcpyAsync(b->d_poses, b->h_poses, sizeof(PoseConst) * n, hipMemcpyHostToDevice, st));
  if (object_modelviews)
    HIP_TRY(hipMemcpyAsync(b->d_objects, b->h_objects, sizeof(ObjectConst) * (size_t)n * lv->n_objects,
                           hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(b->ev_copy, st));
  const int W = (int)b->width, H = (int)b->height;
  if (lv->ntri) {
    hipLaunchKernelGGL(setup_kernel, dim3(n), dim3(256), 0, st, lv->view, b->d_poses,
                       object_modelviews ? (const ObjectConst *)b->d_objects : (const ObjectConst *)nullptr,
                       lv->n_objects, W, H, kinds_mask, b->d_recs,
                       b->d_tmp_recs, b->d_sorted, b->d_counts, b->cap);
  }
  const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
  static const bool no_bins = getenv("RDOOM_NO_BINS") != nullptr;  // debug: exercise the fallback scan
  if (lv->ntri && !no_bins) {
    const uint32_t bin_tiles = std::min<uint32_t>((uint32_t)(tiles_x * tiles_y), MAX_TILES);
    static const int bin_threads = getenv("RDOOM_BIN_THREADS") ? atoi(getenv("RDOOM_BIN_THREADS")) : 256;  // tuning switch
    auto bk = bin_threads == 512 ? bin_kernel<512, 9> : (bin_threads == 128 ? bin_kernel<128, 7> : bin_kernel<256, 8>);
    const int bt = bin_threads == 512 ? 512 : (bin_threads == 128 ? 128 : 256);
    hipLaunchKernelGGL(bk, dim3(n), dim3(bt), 2 * sizeof(uint32_t) * bin_tiles, st, b->d_recs,
                       b->d_sorted, b->d_counts, b->cap, tiles_x, tiles_y, b->d_tile_hdr, b->d_entries, b->entry_cap,
                       b->d_overflow);
  } else {
    HIP_TRY(hipMemsetAsync(b->d_overflow, 0xFF, sizeof(uint32_t) * n, st));
    if (!lv->ntri) HIP_TRY(hipMemsetAsync(b->d_counts, 0, sizeof(uint32_t) * n, st));
  }
  if (tm) HIP_TRY(hipEventRecord(b->ev[1], st));
  const uint64_t nblocks = (uint64_t)((n + 7) / 8) * 8ull * (uint64_t)(tiles_x * tiles_y);
  if (nblocks > 0x7FFFFFFFull) return rdoom::fail(RDOOM_BAD_ARG, "batch too large for one launch");
  static const bool want_stats = getenv("RDOOM_STATS") != nullptr;
  if (want_stats) {
    unsigned long long *d_stats = nullptr, h[16];
    HIP_TRY(hipMalloc((void **)&d_stats, sizeof h));
    HIP_TRY(hipMemsetAsync(d_stats, 0, sizeof h, st));
    hipLaunchKernelGGL(raster_wave_kernel<true>, dim3((uint32_t)nblocks), dim3(256), 0, st, lv->view, b->d_recs,
                       b->d_sorted, b->d_counts, b->cap, n, W, H, tiles_x, tiles_y, b->d_tile_hdr, b->d_entries,
                       b->entry_cap, b->d_overflow, b->d_vis, b->vis16 ? 1u : 0u, b->want_prim ? b->d_prim : nullptr, d_stats);
    HIP_TRY(hipMemcpy(h, d_stats, sizeof h, hipMemcpyDeviceToHost));
    (void)hipFree(d_stats);
    const double waves = (double)nblocks * 4.0;
    fprintf(stderr,
            "[rdoom stats] per wave: queue %.1f  quadrant-bbox %.1f  need-any %.1f (lanes %.1f)  fast %.1f (lanes %.1f)"
            "  general %.2f (lanes %.1f: zrange %.1f, rw<=0 %.1f, masked %.1f, tie %.1f) | rejected: early-z %.2f, then geometry %.2f | coarse tests/block %.0f hits %.1f\n",
            h[0] / waves, h[1] / waves, h[2] / waves, h[2] ? (double)h[3] / h[2] : 0.0, h[4] / waves,
            h[4] ? (double)h[5] / h[4] : 0.0, h[6] / waves, h[6] ? (double)h[7] / h[6] : 0.0,
            h[6] ? (double)h[10] / h[6] : 0.0, h[6] ? (double)h[11] / h[6] : 0.0, h[6] ? (double)h[12] / h[6] : 0.0,
            h[6] ? (double)h[13] / h[6] : 0.0, h[15] / waves, h[14] / waves, (double)h[8] / (double)nblocks,
            (double)h[9] / (double)nblocks);
  } else {
    static const int raster_dbg = getenv("RDOOM_RASTER_DBG") ? atoi(getenv("RDOOM_RASTER_DBG")) : 0;  // timing experiments
    auto rk = raster_dbg == 1 ? raster_wave_kernel<false, 1> : (raster_dbg == 2 ? raster_wave_kernel<false, 2> : raster_wave_kernel<false, 0>);
    hipLaunchKernelGGL(rk, dim3((uint32_t)nblocks), dim3(256), 0, st, lv->view, b->d_recs,
                       b->d_sorted, b->d_counts, b->cap, n, W, H, tiles_x, tiles_y, b->d_tile_hdr, b->d_entries,
                       b->entry_cap, b->d_overflow, b->d_vis, b->vis16 ? 1u : 0u, b->want_prim ? b->d_prim : nullptr,
                       (unsigned long long *)nullptr);
  }
  if (tm) HIP_TRY(hipEventRecord(b->ev[2], st));
  const uint32_t qpr = (uint32_t)W / 4u, qpp = qpr * (uint32_t)H;
  if (qpp >= (1u << 24)) return rdoom::fail(RDOOM_BAD_ARG, "frame too large");
  // multiply-high divisor for idx / qpr, idx < 2^24 (checked exhaustively at the only places it can fail)
  if (qpr < 2) return rdoom::fail(RDOOM_BAD_ARG, "width must be at least 8");
  uint32_t div_sh = 0;
  while ((2u << div_sh) <= qpr) div_sh++;      // floor(log2(qpr))
  if ((qpr & (qpr - 1)) == 0) div_sh -= 1;     // power of two: m = 2^31
  const uint32_t div_m = (uint32_t)((((uint64_t)1 << (32 + div_sh)) + qpr - 1) / qpr);
  for (uint32_t k = 1; k * qpr <= qpp; k++) {
    const uint32_t lo = k * qpr - 1, hi = k * qpr;
    if ((uint32_t)(((uint64_t)lo * div_m) >> 32) >> div_sh != k - 1 ||
        (hi < qpp && (uint32_t)(((uint64_t)hi * div_m) >> 32) >> div_sh != k))
      return rdoom::fail(RDOOM_BAD_ARG, "internal: fast_div constants invalid for width %d", W);
  }
  static const uint32_t debug_leak_mod = getenv("RDOOM_DEBUG_LEAK_MOD") ? (uint32_t)atoi(getenv("RDOOM_DEBUG_LEAK_MOD")) : 0u;
  static const int frag_nq_env = getenv("RDOOM_FRAG_NQ") ? atoi(getenv("RDOOM_FRAG_NQ")) : 2;  // tuning switch / tests
  static const int frag_dbg = getenv("RDOOM_FRAG_DBG") ? atoi(getenv("RDOOM_FRAG_DBG")) : 0;  // timing experiments (wrong images)
  const int nq = (frag_nq_env == 2 && W % 8 == 0) ? 2 : 1;  // quads per lane: two when rows divide into 8-pixel runs
  const uint32_t units = qpp / (uint32_t)nq;
  static const uint32_t frag_chunk = getenv("RDOOM_FRAG_CHUNK") ? (uint32_t)std::max(1, atoi(getenv("RDOOM_FRAG_CHUNK"))) : (uint32_t)FRAG_CHUNK;  // tuning switch
  const uint32_t fblocks = (units + frag_chunk * 256 - 1) / (frag_chunk * 256);
  HIP_TRY(hipMemsetAsync(b->d_fix_count, 0, 2 * sizeof(uint32_t), st));
  const uint64_t fgrid = (uint64_t)((n + 7) / 8) * 8ull * fblocks;
  if (fgrid > 0x7FFFFFFFull) return rdoom::fail(RDOOM_BAD_ARG, "batch too large for one launch");
  auto frag = nq == 2 ? (b->vis16 ? fragment_kernel<2, 0, true> : fragment_kernel<2, 0, false>)
                      : (b->vis16 ? fragment_kernel<1, 0, true> : fragment_kernel<1, 0, false>);
  if (frag_dbg == 2) frag = b->vis16 ? fragment_kernel<1, 2, true> : fragment_kernel<1, 2, false>;

  hipLaunchKernelGGL(frag, dim3((uint32_t)fgrid), dim3(256), 0, st, lv->view, b->d_recs, b->cap, b->d_poses,
                     b->d_vis, n, fblocks, frag_chunk, qpp, qpr, div_m, div_sh, W, H, b->d_ndc, b->d_fb, b->d_fix_count,
                     b->d_fix_list, b->fix_cap, debug_leak_mod);
  hipLaunchKernelGGL(fixup_kernel, dim3(64), dim3(256), 0, st, lv->view, b->d_recs, b->d_sorted, b->d_counts, b->cap,
                     b->d_poses, W, H, tiles_x, tiles_y, b->d_tile_hdr, b->d_entries, b->entry_cap, b->d_overflow,
                     b->d_fix_count, b->d_fix_list, b->fix_cap, b->d_vis, b->vis16 ? 1u : 0u,
                     b->want_prim ? b->d_prim : nullptr, b->d_fb,
                     b->d_fix_count + 1);
  HIP_TRY(hipGetLastError());
  if (tm) {
    HIP_TRY(hipEventRecord(b->ev[3], st));
    HIP_TRY(hipEventSynchronize(b->ev[3]));
    HIP_TRY(hipEventElapsedTime(&tm->setup_ms, b->ev[0], b->ev[1]));
    HIP_TRY(hipEventElapsedTime(&tm->raster_ms, b->ev[1], b->ev[2]));
    HIP_TRY(hipEventElapsedTime(&tm->fragment_ms, b->ev[2], b->ev[3]));
    HIP_TRY(hipEventElapsedTime(&tm->total_ms, b->ev[0], b->ev[3]));
    tm->pixels = (uint64_t)n * W * H;
    std::vector<uint32_t> counts(n);
    HIP_TRY(hipMemcpy(counts.data(), b->d_counts, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    tm->visible_triangles = 0;
    for (uint32_t c : counts) tm->visible_triangles += c;
    uint32_t fix[2] = {0, 0};
    HIP_TRY(hipMemcpy(fix, b->d_fix_count, sizeof fix, hipMemcpyDeviceToHost));
    tm->fixup_pixels = fix[0];
    if (fix[1]) return rdoom::fail(RDOOM_BAD_LEVEL, "alpha-leak fixup list overflow (%u pixels)", fix[0]);
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

rdoom_status rdoom_batch_render_objects(rdoom_batch *batch, const rdoom_pose *poses, const uint8_t *lights,
                                        uint32_t lights_stride, uint32_t n_poses, uint32_t kinds_mask, void *stream,
                                        const float *object_modelviews, uint32_t n_objects) {
  if (!object_modelviews) return rdoom::fail(RDOOM_BAD_ARG, "object_modelviews is null");
  return render_impl(batch, poses, lights, lights_stride, n_poses, kinds_mask, (hipStream_t)stream, nullptr,
                     object_modelviews, n_objects);
}

rdoom_status rdoom_level_num_objects(const rdoom_level *level, uint32_t *out) {
  if (!level || !out) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out = level->n_objects;
  return RDOOM_OK;
}

rdoom_status rdoom_batch_framebuffer_device(const rdoom_batch *batch, uint8_t **out_device_ptr) {
  if (!batch || !out_device_ptr) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out_device_ptr = batch->d_fb;
  return RDOOM_OK;
}

rdoom_status rdoom_batch_read_framebuffer(rdoom_batch *b, uint32_t first, uint32_t count, uint8_t *host_out) {
  if (!b || !host_out) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  if ((uint64_t)first + count > b->last_n) return rdoom::fail(RDOOM_BAD_ARG, "frame range outside the last render");
  const size_t frame = (size_t)b->width * b->height;
  HIP_TRY(hipDeviceSynchronize());
  uint32_t fix[2] = {0, 0};
  HIP_TRY(hipMemcpy(fix, b->d_fix_count, sizeof fix, hipMemcpyDeviceToHost));
  if (fix[1]) return rdoom::fail(RDOOM_BAD_LEVEL, "alpha-leak fixup list overflow (%u pixels)", fix[0]);
  HIP_TRY(hipMemcpy(host_out, b->d_fb + frame * first, frame * count, hipMemcpyDeviceToHost));
  return RDOOM_OK;
}

rdoom_status rdoom_batch_enable_primitive_ids(rdoom_batch *b) {
  if (!b) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  if (!b->d_prim) {
    const size_t npx = (size_t)b->width * b->height * b->max_poses;
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
  const size_t frame = (size_t)b->width * b->height;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(host_out, b->d_prim + frame * first, frame * count * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return RDOOM_OK;
}

}  // extern "C"
