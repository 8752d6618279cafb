// Device-side records and helpers shared by the renderer's translation units (setup, binning, rasteriser,
// fragment / fixup kernels, host side).  The arithmetic every kernel must reproduce is specified in DESIGN.md
// "Raster arithmetic" (steps V1.., S1.., R1.., F1..).  Built with -ffp-contract=off: a*b+c is never fused unless
// written as fmaf.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../common.hpp"
#include "fastmath.hpp"

#pragma clang fp contract(off)

namespace rdoom_dev {
using namespace rdoom_fm;

constexpr uint32_t NONE = 0xFFFFFFFFu;
// A tile-list entry is record index | quadrant bits << 28; a quadrant-table entry is record index | QTAB_HANDLED (or NONE).
// One mask for every reader: a level has fewer than 2^24 triangles (rdoom_level_create checks).
constexpr uint32_t ENTRY_REC_MASK = 0x00FFFFFFu;
// Quadrant table, bit 24: fragment_quadrant_kernel has already shaded every pixel of this quadrant (it runs before
// fragment_kernel, which then skips the blocks that lie in handled quadrants).  NONE has the bit set too: test NONE first.
constexpr uint32_t QTAB_HANDLED = 1u << 24;
__host__ __device__ inline bool qtab_handled(uint32_t e) { return e != NONE && (e & QTAB_HANDLED) != 0u; }
__host__ __device__ inline uint32_t qtab_record(uint32_t e) { return e == NONE ? NONE : (e & ENTRY_REC_MASK); }
constexpr int TILE_W = 64, TILE_H = 64;  // one wavefront per tile: four 32x32 quadrants in turn, 4x4 pixels per lane

// ---- level-constant triangle record (built once per level on the host) -----------------------
struct alignas(16) LevelTri {  // 96 bytes
  float pos[9];
  float uv[6];
  float scroll[3];              // a_scroll_rate per vertex; for decor triangles: a_local_x per vertex
  float atlas_u, atlas_v, size_x, size_y, row_height;
  uint32_t packed;  // num_frames | light << 8 | kind << 16 | masked border << 18 | masked interior << 19 | object id << 20
};
static_assert(sizeof(LevelTri) == 96, "LevelTri layout");

struct alignas(16) PoseConst {  // 480 bytes
  float pm[16];                 // projection * modelview (V1)
  float time, vr0, vr1, zk;     // zk: depth conditioning constant P[2][2] / P[2][3] (S5)
  uint8_t lights[256];
  float mv[16], proj[16];       // the two uniforms themselves: sprite.vert transforms in two steps (D1..D3)
  uint32_t level, pad0, pad1, pad2;  // which level of the resident set this pose looks at (DeviceLevelView::slices; 0 for a single level)
};
static_assert(sizeof(PoseConst) == 480, "PoseConst layout");

// Per (pose, object) uniforms when objects move (doors, lifts): the reference sets u_modelview = view o model
// transform for the draws of each object (engine/src/renderer.rs:120-132, game/src/level.rs:203-255).
struct alignas(16) ObjectConst {  // 144 bytes
  float pm[16];                   // projection * object modelview (V1)
  float mv[16];                   // object modelview
  float vr0, vr1, pad0, pad1;     // sky.vert:10-12 from this object's transform
};
static_assert(sizeof(ObjectConst) == 144, "ObjectConst layout");

// ---- per (pose, visible triangle) records -----------------------------------------------------
// (round 5: 64 bytes -- the 1/w plane is the shade part's copy, ShadeRec::wp, which follows immediately: the rasteriser reads
// the record's first five 16-byte words, c0..c2 = edges + depth plane, c3 = bbox + flags, c4 = (1/w plane, up[0]))
struct alignas(16) RasterRec {  // 64 bytes
  float e[9];                   // edge functions A,B,C x3
  float zp[3];                  // window-depth plane
  uint32_t bb0, bb1;            // x0 | y0 << 16, x1 | y1 << 16 (inclusive)
  uint32_t flags;               // prim id (24 bits) | tl << 24 | kind << 27 | RASTER_MASKED_*
  uint32_t pad;
};
static_assert(sizeof(RasterRec) == 64 && offsetof(RasterRec, zp) == 36 && offsetof(RasterRec, bb0) == 48, "RasterRec layout");

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

// 128 bytes, 128-byte aligned: ONE cache line per record.  With the 144-byte record of rounds 1-4 every line of the record array
// was shared by two records that different lanes, waves and workgroups of the set-up kernel store at different times (records go
// to their depth rank): each line went to memory twice, partially filled -- and the set-up and binning kernels of the large-level
// and small-frame workloads are bound by exactly this traffic (0.95 GB of records per step at 320 x 200 x 8 192 poses).
struct alignas(128) TriRec {
  RasterRec r;
  ShadeRec s;  // s.wp is the 1/w plane of both parts
};
static_assert(sizeof(TriRec) == 128 && offsetof(TriRec, s) == 64, "TriRec layout");
// the rasteriser's five words of a record in the order its code names them (rounds 1-4's layout): c3 = (1/w plane, bb0),
// c4 = (bb1, flags)
__device__ __forceinline__ void raster_words(const TriRec *rec, uint4 &c0, uint4 &c1, uint4 &c2, uint4 &c3, uint2 &c4) {
  const uint4 *rp = reinterpret_cast<const uint4 *>(rec);
  c0 = rp[0], c1 = rp[1], c2 = rp[2];
  const uint4 b = rp[3], w = rp[4];  // (bb0, bb1, flags, pad), (wp[0], wp[1], wp[2], up[0])
  c3 = make_uint4(w.x, w.y, w.z, b.x);
  c4 = make_uint2(b.y, b.z);
}

// Up to 32 consecutive triangles of one object with the bounding box of their vertices: the set-up kernel discards a
// whole cluster when the box proves that every triangle in it fails S1 or S6 (corner arguments on the same fmaf chains).
constexpr uint32_t CLUSTER_TRIS = 32;
struct alignas(16) Cluster {  // 32 bytes
  float lo[3], hi[3];
  uint32_t first;         // first triangle
  uint32_t count_object;  // count | object id << 8 | never-cull << 31 (decor: sprite.vert moves the vertices per pose)
};
static_assert(sizeof(Cluster) == 32, "Cluster layout");

// One level of the resident set (rdoom_levelset_create; rdoom_level_create = a set of one): its share of the set's triangle and
// cluster arrays, where its atlases sit in the set's ONE u16 texel store, its sky.  The reference keeps one level loaded at a time
// (game/src/level.rs:330-496 builds it, engine/src/renderer.rs:98-157 draws it); a pose batch over several levels -- BASELINE
// config 4's share of one GPU -- would otherwise be one launch set per level.  A pose names its level (PoseConst::level); only the
// cull / set-up kernels and the fragment stage's sky path look the slice up (scalar loads: a workgroup works on one pose), every
// other reader finds what it needs in the pose's records (texel base, atlas masks).
struct alignas(16) LevelSlice {  // 80 bytes
  uint32_t first_cluster, n_clusters, first_tri, ntri;
  // texel offsets in the set's store (multiples of 1024): the wall atlas (lo = palette index, bit 15 = transparent), the flat
  // atlas promoted to u16 (hi byte 0: never transparent), the decor (sprite) atlas
  uint32_t wall_base, flat_base, decor_base, sky_base;  // sky_base: offset of this level's sky texture in DeviceLevelView::sky_texels
  uint32_t wall_w, wall_h, flat_w, flat_h;
  uint32_t decor_w, decor_h, sky_w, sky_h;
  float sky_band;
  uint32_t pad0, pad1, pad2;
};
static_assert(sizeof(LevelSlice) == 80, "LevelSlice layout");

struct DeviceLevelView {
  const LevelTri *tris;      // all levels' triangles, level after level; primitive id = index - slice.first_tri
  const Cluster *clusters;   // Cluster::first indexes `tris`
  const uint16_t *texels;    // one u16 texel store for every atlas of every level of the set
  const uint16_t *sky_texels;
  const uint8_t *colormap;   // COLORMAP of the IWAD (the levels of a set share it: rdoom_levelset_create checks)
  const LevelSlice *slices;
  uint32_t n_slices;
  uint32_t max_clusters;     // the largest slice's cluster / triangle count: grid sizes and per-pose record strides
  uint32_t max_ntri;
};

__device__ __forceinline__ float plane3(const float *p, float px, float py) {
  return fmaf(p[0], px, fmaf(p[1], py, p[2]));
}
__device__ __forceinline__ float glsl_mod(float x, float y) { return x - y * floorf(x / y); }


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


// F1..F3: perspective-correct tile coordinates -> atlas texel coordinates (shared by the alpha test
// R6 and the fragment stage).  `row_u`/`row_v`/`row_w` are fmaf(B, py, C) of the three planes.
struct TexelAt {
  int ix, iy;
  float dist;
};
// RCP_EXACT: the caller guarantees 2^-100 <= rw <= 2^100, where fastmath.hpp's exact_rcp equals the division bit for bit.
template <bool RCP_EXACT = false>
__device__ __forceinline__ TexelAt texel_coords(const ShadeRec &s, float px, float row_w, float row_u,
                                                float row_v) {
  TexelAt t;
  const float rw = fmaf(s.wp[0], px, row_w);
  const float w = RCP_EXACT ? exact_rcp(rw) : 1.0f / rw;
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
__device__ __forceinline__ uint32_t load_texel(const uint16_t *__restrict__ texels, const ShadeRec &s, int ix, int iy) {
  return texels[texel_offset(s.flags, s.tex, ix, iy)];
}

// x > 0 for a non-NaN binary32, as an integer test on the bits: a scalar compare when x is wave-uniform
__device__ __forceinline__ bool pos(float x) { return (int)__float_as_uint(x) > 0; }

// idx / d for idx < 2^24 by multiply-high (m, sh) computed and verified on the host
__device__ __forceinline__ uint32_t fast_div(uint32_t idx, uint32_t m, uint32_t sh) { return __umulhi(idx, m) >> sh; }

}  // namespace rdoom_dev
