// Kernel 2: tiled rasteriser (wave-autonomous) -> visibility words.
//
// Part of the pose-batch renderer for gfx950 (MI355X) that replaces the reference's GL draw path:
// assets/shaders/static.{vert,frag}, sky.{vert,frag}, sprite.{vert,frag} and the fixed-function state of
// engine/src/renderer.rs:49-57 + engine/src/window.rs:12,40-44.  The arithmetic is specified in DESIGN.md
// "Raster arithmetic"; operation order follows that text, not the oracle's source.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "kernels.hpp"

#pragma clang fp contract(off)

namespace rdoom_dev {
namespace {

// Section timers (tools/variant.sh NAME raster -DRDOOM_RASTER_TIMERS; never in the shipped library): where does a wave's
// time go?  Every mark waits for the wave's outstanding memory operations, reads the shader clock (s_memtime) and
// charges the cycles since the previous mark to a section; lane 0 adds the wave's sums to g_raster_t at the end.  The
// waits serialise what would overlap and the reads cost cycles themselves: the SHARES are the result, not the total.
#ifdef RDOOM_CENSUS_TWO
__device__ unsigned long long g_raster_census[64];
#endif
#ifdef RDOOM_RASTER_TIMERS
__device__ unsigned long long g_raster_t[16];
__device__ __forceinline__ unsigned long long rt_now() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  return __builtin_readcyclecounter();
}
#define RT_DECL uint32_t rt_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long rt_last = rt_now();
#define RT_MARK(i) do { const unsigned long long n_ = rt_now(); rt_acc[i] += (uint32_t)(n_ - rt_last); rt_last = n_; } while (0)
#define RT_FLUSH() do { if (lane == 0) { for (int k_ = 0; k_ < 10; k_++) atomicAdd(&g_raster_t[k_], (unsigned long long)rt_acc[k_]); atomicAdd(&g_raster_t[15], 1ull); } } while (0)
#else
#define RT_DECL
#define RT_MARK(i) do { } while (0)
#define RT_FLUSH() do { } while (0)
#endif

// =================================================================================================
// Rasteriser, per-entry part.  Rejection is hierarchical and exact: fmaf is monotone in each argument, so the
// extreme of a *computed* edge function, depth plane or 1/w plane over a pixel rectangle sits at a corner --
// per lane: nearest-corner depth against the lane's farthest pixel (early-z) first, then three edge corners,
// the depth range and the 1/w plane.  One __any() skips the 16-pixel body when no lane needs it.
// Winner = lexicographic min of (d24, primitive id): independent of processing order.
// =================================================================================================
// One queue entry against one lane's 4x4 block: exact rejection (early-z first), then the pixel bodies (R1..R6).
// The coefficients arrive wave-uniform (v_readlane broadcasts), i.e. as SGPR operands.
template <bool STATS, class ShadeFetch>
__device__ __forceinline__ void raster_entry(const DeviceLevelView &lv, const TriRec *__restrict__ prec, float e0a, float e0b,
                                             float e0c, float e1a, float e1b, float e1c, float e2a, float e2b, float e2c,
                                             float za, float zb, float zc, float wa, float wb, float wc, int x0, int y0,
                                             int x1, int y1, uint32_t flags, uint32_t ridx, int bx, int by, float pxlo,
                                             float pxhi, float pylo, float pyhi, uint32_t (&best_d)[16],
                                             uint32_t (&best_r)[16], uint32_t &lane_far, ShadeFetch fetch_shade,
                                             unsigned long long (&st)[20]) {
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
    if (STATS) st[2]++, st[3] += (unsigned long long)__popcll(__ballot(need)), st[16] += need ? 1ull : 0ull;  // [16]: THIS lane's bodies in the current pass
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
    // Ties are folded into ONE unsigned minimum so that no per-pixel compare mask has to stay alive (sixteen of them
    // do not fit the scalar registers): tiez = 0 iff some pixel has an edge function exactly zero, or lies inside with
    // a depth equal to the one it holds.  d24m - best is computed once; its borrow is the depth test.
    uint32_t tiez = NONE;
    bool updated = false;
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
          uint32_t diff;
          const bool nearer = __builtin_usub_overflow(d24m, best_d[k], &diff);  // borrow: d24m < best_d[k]
          const int ei = (int)__float_as_uint(em);
          const bool inside = ei > 0;  // em > 0
          // inside: zero iff the depths are equal; not inside: zero iff em is +0 or -0 (an edge tie)
          tiez = min(tiez, inside ? diff : ((uint32_t)ei << 1));
          const bool win = inside & nearer;
          best_d[k] = win ? d24 : best_d[k];
          best_r[k] = win ? ridx : best_r[k];
          updated |= win;
        }
      }
    }
    const bool redo = tiez == 0u;
    if (__any(need & (!fast | redo))) {
      if (STATS) {
        st[6]++, st[7] += (unsigned long long)__popcll(__ballot(need & (!fast | redo)));
        // why: [12] masked texture, [13] tie replay
        st[12] += (unsigned long long)__popcll(__ballot(need & ((flags & RASTER_MASKED_INTERIOR) != 0u)));
        st[13] += (unsigned long long)__popcll(__ballot(need & fast & redo));
      }
      if (need & (!fast | redo)) {
        // The general rule R1..R6, a row of four pixels at a time and branch-free but for two rare cases (a depth tie
        // that the primitive order must settle; texels to look at): compare results go straight into bit masks.
        const uint32_t prim = flags & 0xFFFFFFu;
        const bool tl0 = (flags & (1u << 24)) != 0u, tl1 = (flags & (1u << 25)) != 0u, tl2 = (flags & (1u << 26)) != 0u;
        const bool masked = (flags & RASTER_MASKED_ANY) != 0u, interior = (flags & RASTER_MASKED_INTERIOR) != 0u;  // uniform
        ShadeRec sh;
        if (masked) sh = fetch_shade();
#pragma unroll
        for (int ry = 0; ry < 4; ry++) {
          const int iy = by + ry;
          const float py = pylo + (float)ry;  // == (float)iy + 0.5f exactly
          const float t0 = fmaf(e0b, py, e0c), t1 = fmaf(e1b, py, e1c), t2 = fmaf(e2b, py, e2c);
          const float tz = fmaf(zb, py, zc), tw = fmaf(wb, py, wc);
          const bool rowin = iy >= y0 && iy <= y1;
          uint32_t passm = 0u, tiem = 0u, d24s[4];
#pragma unroll
          for (int rx = 0; rx < 4; rx++) {
            const int ix = bx + rx;
            const float px = pxlo + (float)rx;  // == (float)ix + 0.5f exactly
            const float e0 = fmaf(e0a, px, t0), e1 = fmaf(e1a, px, t1), e2 = fmaf(e2a, px, t2);
            const bool in0 = (e0 > 0.0f) | ((e0 == 0.0f) & tl0);
            const bool in1 = (e1 > 0.0f) | ((e1 == 0.0f) & tl1);
            const bool in2 = (e2 > 0.0f) | ((e2 == 0.0f) & tl2);
            const float zw = fmaf(za, px, tz);
            const float rw = fmaf(wa, px, tw);
            const uint32_t d24 = __float2uint_rz(fmaf(fminf(fmaxf(zw, 0.0f), 1.0f), 16777215.0f, 0.5f));
            const uint32_t bd = ry == 0 ? best_d[rx] : (ry == 1 ? best_d[4 + rx] : (ry == 2 ? best_d[8 + rx] : best_d[12 + rx]));
            const bool geom = rowin & (ix >= x0) & (ix <= x1) & in0 & in1 & in2 & (zw >= 0.0f) & (zw <= 1.0f) & (rw > 0.0f);
            d24s[rx] = d24;
            passm |= (geom & (d24 < bd)) ? (1u << rx) : 0u;
            tiem |= (geom & (d24 == bd)) ? (1u << rx) : 0u;
          }
          if (tiem != 0u) {  // depth tie (rare): the earlier primitive keeps the pixel
#pragma unroll
            for (int rx = 0; rx < 4; rx++)
              if ((tiem >> rx) & 1u) {
                const uint32_t br = ry == 0 ? best_r[rx] : (ry == 1 ? best_r[4 + rx] : (ry == 2 ? best_r[8 + rx] : best_r[12 + rx]));
                if (br == NONE || prim < (prec[br].r.flags & 0xFFFFFFu)) passm |= 1u << rx;
              }
          }
          if (masked && passm != 0u) {  // R6: alpha test before the depth write; the row's texel fetches in flight together
            const float tu = fmaf(sh.up[1], py, sh.up[2]), tv = fmaf(sh.vp[1], py, sh.vp[2]);
            uint32_t toff[4], fetchm = 0u;
#pragma unroll
            for (int rx = 0; rx < 4; rx++) {
              // pixels that did not pass compute garbage coordinates: the offset stays inside the texel store, the result is ignored
              const TexelAt t = texel_coords(sh, pxlo + (float)rx, tw, tu, tv);
              // texture rectangle fully opaque: only a coordinate that the float mod pushed just outside the rectangle
              // can hit a transparent neighbour texel -- look only then
              const bool must_fetch = interior | (t.ix < (int)sh.atlas_u) | (t.ix >= (int)(sh.atlas_u + sh.size_x)) |
                                      (t.iy < (int)sh.atlas_v) | (t.iy >= (int)(sh.atlas_v + sh.size_y));
              toff[rx] = texel_offset(sh.flags, sh.tex, t.ix, t.iy);
              fetchm |= must_fetch ? (1u << rx) : 0u;
            }
            fetchm &= passm;
            if (fetchm != 0u) {
              uint32_t tx[4];
#pragma unroll
              for (int rx = 0; rx < 4; rx++) tx[rx] = lv.texels[toff[rx]];
#pragma unroll
              for (int rx = 0; rx < 4; rx++)
                if (tx[rx] & 0x8000u) passm &= ~(fetchm & (1u << rx));
            }
          }
          updated |= passm != 0u;
#pragma unroll
          for (int rx = 0; rx < 4; rx++) {
            const bool pass = ((passm >> rx) & 1u) != 0u;
#pragma unroll
            for (int r2 = 0; r2 < 4; r2++)
              if (r2 == ry) {
                best_d[r2 * 4 + rx] = pass ? d24s[rx] : best_d[r2 * 4 + rx];
                best_r[r2 * 4 + rx] = pass ? ridx : best_r[r2 * 4 + rx];
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
// Kernel 2: tiled rasteriser, wave-autonomous.  One wavefront per (pose, 64x64 tile), one tile per 64-thread workgroup
// (a workgroup's registers and LDS come back when its last wave ends: with four tiles per workgroup three finished
// waves waited for the slowest), no workgroup barriers.  The wave gathers the tile's list once and then rasterises the tile's four
// 32x32 quadrants one after the other: in a quadrant each lane owns a 4x4 pixel block whose depth / winner live
// in registers.  blockIdx -> (pose, tiles) keeps all tiles of a pose on one XCD (b % 8): its records stay in that
// XCD's L2.
//   * candidates  64 tile-list entries at a time, one per lane, ranked by record index (= depth rank) with
//                 readlane broadcasts and compacted through a 256-byte per-wave LDS scratch;
//   * records     lane s gathers the 80-byte raster record of the s-th entry, parks 15 of its words in LDS and
//                 computes, for all four quadrants at once, the nearest depth of the triangle over the quadrant
//                 and whether the triangle covers the quadrant entirely (exact corner arguments);
//   * walk        per quadrant, entry s is broadcast with v_readlane / uniform LDS reads: its coefficients become
//                 wave-uniform SGPR operands.  One compare against the lanes' farthest depths skips a hidden
//                 triangle before anything else is touched (most rejections are of this kind); a covering
//                 triangle runs a depth-only pixel body.
// A tile with more than 64 entries re-gathers each 64-entry batch once per quadrant.
// =================================================================================================
// max of v over the wavefront (DPP within rows of 16 lanes, then the four row results): uniform
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));   // quad_perm [2, 3, 0, 1]
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));  // row_half_mirror
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));  // row_mirror
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
                 c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {  // (the same reduction with min)
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0xB1, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x4E, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x141, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x140, 0xF, 0xF, false));
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
                 c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, d));
}

// =================================================================================================
// Kernel 2a: settle.  Seven quadrants in ten of a 1080p sweep show ONE triangle: the nearest entry of the tile's list covers the
// quadrant entirely and every other entry lies strictly behind its farthest depth there -- the rasteriser's one-entry shortcut.
// The rasteriser finds that out with a WAVE per tile: header, entries and 80-byte records by three dependent memory round
// trips, the per-quadrant corner arithmetic of every entry with 7 of 64 lanes busy, wave-wide minima and ballots -- and then
// stores four bytes.  Here a LANE per (tile, quadrant) walks the tile's list itself (lists of at most `max_list` entries,
// stored whole): of each entry that touches its quadrant (the binning kernel's bits) it needs only the depth plane -- the
// nearest depth over the quadrant, a corner value --, keeps the nearest and the second nearest, and evaluates the cover test
// and the farthest depth once, for the nearest entry.  Same corner arguments, same strictness, hence the same decisions as the
// shortcut (a depth tie between the two nearest leaves the quadrant to the rasteriser, as there).  A settled quadrant gets its
// table entry HERE; every other quadrant inside the frame gets QTAB_TODO, and the rasteriser's wave for the tile starts by
// reading the tile's four entries: nothing to do -> it ends after one scalar load; else it passes only the TODO quadrants.
// (Instantiations that store visibility words or primitive ids for described quadrants -- tests -- ignore this and do everything.)
// Poses whose bins overflowed are left alone: the rasteriser scans their sorted lists as before.
// =================================================================================================
constexpr uint32_t QTAB_TODO = 0xFFFFFFFEu;  // (not NONE, not a record index: the rasteriser replaces every one before anybody else reads the table)
__global__ __launch_bounds__(256) void settle_kernel(const TriRec *__restrict__ recs, uint32_t cap, uint32_t n_poses, int width, int height,
                                                     int tiles_x, int tiles_y, const uint2 *__restrict__ tile_hdr,
                                                     const uint32_t *__restrict__ entries, uint32_t entry_cap,
                                                     const uint32_t *__restrict__ overflow, uint32_t *__restrict__ qtab, uint32_t max_list) {
  const uint32_t T = (uint32_t)(tiles_x * tiles_y);
  const uint32_t pose = blockIdx.z * 8u + (blockIdx.x & 7u);  // a pose's tiles on one XCD, like the rasteriser's
  const uint32_t idx = (blockIdx.x >> 3) * 256u + threadIdx.x, tile = idx >> 2, q = idx & 3u;
  if (pose >= n_poses || tile >= T) return;
  if (overflow[pose] != 0u) return;
  const int tx0 = (int)(tile % (uint32_t)tiles_x) * TILE_W, ty0 = (int)(tile / (uint32_t)tiles_x) * TILE_H;
  const int rx0 = tx0 + (int)(q & 1u) * 32, ry0 = ty0 + (int)(q >> 1) * 32;
  if (rx0 >= width || ry0 >= height) return;  // outside the frame: NONE from the start, never written
  const TriRec *prec = recs + (size_t)pose * cap;
  const uint2 hdr = tile_hdr[(size_t)pose * T + tile];
  uint32_t out = QTAB_TODO;
  if ((hdr.y & TILE_SPLIT) == 0u && hdr.y != 0u && hdr.y <= max_list) {
    const uint32_t *pent = entries + (size_t)pose * entry_cap + hdr.x;
    const float xl = (float)rx0 + 0.5f, xh = (float)rx0 + 31.5f, yl = (float)ry0 + 0.5f, yh = (float)ry0 + 31.5f;
    uint32_t m1 = NONE, m2 = NONE, best = NONE;
    // four entries at a time: their words first, then the depth planes of those that touch my quadrant -- the loads of a group
    // are in flight together (a lane that walked its list one entry at a time spent its life in two dependent round trips per entry)
    for (uint32_t i = 0; i < hdr.y; i += 4u) {
      uint32_t e[4];
      uint4 c2[4];
#pragma unroll
      for (uint32_t k = 0; k < 4u; k++) e[k] = i + k < hdr.y ? pent[i + k] : 0u;  // (0: touches nothing)
#pragma unroll
      for (uint32_t k = 0; k < 4u; k++)
        if ((e[k] >> (28u + q)) & 1u) c2[k] = reinterpret_cast<const uint4 *>(&prec[e[k] & ENTRY_REC_MASK])[2];
#pragma unroll
      for (uint32_t k = 0; k < 4u; k++) {
        if (((e[k] >> (28u + q)) & 1u) == 0u) continue;  // does not touch my quadrant (exact tests of the binning kernel)
        const float za = __uint_as_float(c2[k].y), zb = __uint_as_float(c2[k].z), zc = __uint_as_float(c2[k].w);
        const float zn = fmaf(za, pos(za) ? xl : xh, fmaf(zb, pos(zb) ? yl : yh, zc));
        const uint32_t dn = zn <= 1.0f ? __float2uint_rz(fmaf(fminf(fmaxf(zn, 0.0f), 1.0f), 16777215.0f, 0.5f)) : NONE;
        if (dn < m1) {
          m2 = m1, m1 = dn, best = e[k] & ENTRY_REC_MASK;
        } else {
          m2 = min(m2, dn);  // (dn == m1: a tie of the two nearest -- m2 = m1 cannot be strictly behind anything)
        }
      }
    }
    if (best != NONE) {
      uint4 c0, c1, c2, c3;
      uint2 c4;
      raster_words(&prec[best], c0, c1, c2, c3, c4);
      const float e0a = __uint_as_float(c0.x), e0b = __uint_as_float(c0.y), e0c = __uint_as_float(c0.z),
                  e1a = __uint_as_float(c0.w), e1b = __uint_as_float(c1.x), e1c = __uint_as_float(c1.y),
                  e2a = __uint_as_float(c1.z), e2b = __uint_as_float(c1.w), e2c = __uint_as_float(c2.x);
      const float za = __uint_as_float(c2.y), zb = __uint_as_float(c2.z), zc = __uint_as_float(c2.w);
      const float wa = __uint_as_float(c3.x), wb = __uint_as_float(c3.y), wc = __uint_as_float(c3.z);
      const int x0 = (int)(c3.w & 0xFFFFu), y0 = (int)(c3.w >> 16), x1 = (int)(c4.x & 0xFFFFu), y1 = (int)(c4.x >> 16);
      // the cover test of the rasteriser's gather, for this one entry and quadrant: smallest corner values of the edge
      // functions and of 1/w, the depth range over the quadrant, the bbox (clipped to the frame), an opaque texture rectangle
      const float zn = fmaf(za, pos(za) ? xl : xh, fmaf(zb, pos(zb) ? yl : yh, zc));
      const float zf = fmaf(za, pos(za) ? xh : xl, fmaf(zb, pos(zb) ? yh : yl, zc));
      const float n0 = fmaf(e0a, pos(e0a) ? xl : xh, fmaf(e0b, pos(e0b) ? yl : yh, e0c));
      const float n1 = fmaf(e1a, pos(e1a) ? xl : xh, fmaf(e1b, pos(e1b) ? yl : yh, e1c));
      const float n2 = fmaf(e2a, pos(e2a) ? xl : xh, fmaf(e2b, pos(e2b) ? yl : yh, e2c));
      const float rwn = fmaf(wa, pos(wa) ? xl : xh, fmaf(wb, pos(wb) ? yl : yh, wc));
#ifndef RDOOM_NO_EDGE_COVER
      const int qx1 = min(rx0 + 31, width - 1), qy1 = min(ry0 + 31, height - 1);
#else
      const int qx1 = rx0 + 31, qy1 = ry0 + 31;
#endif
      const bool cover = (zn >= 0.0f) & (zf <= 1.0f) & (rwn > 0.0f) & (x0 <= rx0) & (x1 >= qx1) & (y0 <= ry0) & (y1 >= qy1) &
                         ((c4.y & RASTER_MASKED_INTERIOR) == 0u) & (n0 > 0.0f) & (n1 > 0.0f) & (n2 > 0.0f);
      if (cover) {
        const uint32_t df0 = __float2uint_rz(fmaf(zf, 16777215.0f, 0.5f));  // zf in [0, 1]: the entry covers
        if (m2 > df0) out = best;  // everything else strictly behind its farthest depth: it wins all 1024 pixels
      }
    }
  }
  qtab[((size_t)pose * T + tile) * 4u + q] = out;
}

#ifndef RDOOM_RANK_LIMIT
#define RDOOM_RANK_LIMIT 0  // a batch of a binned list with more entries than this is walked in list order, unranked (0: always --
                            // ranking every batch by record index measured 1 % slower at 1080p and 8 % slower on the large level)
#endif
#ifndef RDOOM_SETTLE_MAX
#define RDOOM_SETTLE_MAX 32  // settle_kernel examines whole lists of at most this many entries (a lane walks the list: longer ones are the rasteriser's)
#endif
#ifndef RDOOM_RASTER_OCC
#define RDOOM_RASTER_OCC 4  // waves per SIMD the register allocation aims at (128 VGPRs)
#endif
#ifndef RDOOM_RASTER_WAVES
#define RDOOM_RASTER_WAVES 1
#endif
constexpr uint32_t RASTER_WAVES = RDOOM_RASTER_WAVES;  // tiles (= waves) per workgroup

// VIS16: 16-bit visibility words (record indices below 65 535; 0xFFFF = none) -- a compile-time choice: as a run-time
// flag the compiler kept it as a per-lane boolean and spilled that register to scratch
// PRIM: the winning primitive ids are written as well (tests; rdoom_batch_enable_primitive_ids)
// SKIPVIS: a quadrant the table describes ("all 1024 pixels show record r") gets NO visibility words: the fragment kernel
// takes the record from the table wherever an entry exists and reads visibility words only where it says NONE
// (launch_fragment's plan decides; 60 % of the quadrant passes of the 1080p sweep then store four bytes instead of 2 KB)
// SPLIT: the binning kernel may have stored long lists per quadrant (bin.hip); without it the instantiation is the kernel as
// it was before such lists existed (the caller chooses per render: renderer.hip)
template <bool STATS, bool VIS16, bool PRIM, bool SKIPVIS, bool SPLIT>
__global__ __launch_bounds__(64 * RDOOM_RASTER_WAVES, RDOOM_RASTER_OCC) void raster_wave_kernel(DeviceLevelView lv, const TriRec *__restrict__ recs,
                                                             const uint32_t *__restrict__ counts, uint32_t cap,
                                                             uint32_t n_poses, int width, int height, int tiles_x,
                                                             int tiles_y, const uint2 *__restrict__ tile_hdr,
                                                             const uint32_t *__restrict__ entries, uint32_t entry_cap,
                                                             const uint32_t *__restrict__ overflow,
                                                             uint32_t *__restrict__ vis,
                                                             uint32_t *__restrict__ prim_out, uint32_t no_cover,
                                                             uint32_t *__restrict__ qtab, uint32_t settled,
                                                             unsigned long long *__restrict__ stats) {
  unsigned long long st[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  __shared__ uint32_t wq[RASTER_WAVES][64];
  __shared__ uint4 wrec[RASTER_WAVES][64][4];  // per wave: 15 words of each of the 64 gathered raster records
  // per entry: a sign-blind hash of each edge's three coefficients (a shared edge has exactly negated coefficients in the two
  // triangles, S3) and, per quadrant, "covers the quadrant but for edge k" (bit 4 k + q) -- the two-entry shortcut below
  __shared__ uint4 whash[RASTER_WAVES][64];
  // grid = (8 tiles_x, tiles_y, pose groups of 8): workgroups are dispatched x-fastest and dealt to the eight XCDs in turn,
  // so blockIdx.x & 7 is the XCD and all tiles of a pose land on one of them; no division is needed to find (pose, tile)
  static_assert(RASTER_WAVES == 1, "the 3-D grid maps one tile to one workgroup");
  const uint32_t T = (uint32_t)(tiles_x * tiles_y);
#ifndef RDOOM_RASTER_ROWS_INNER
  // Tile ROWS are the slowest grid dimension, the frame's middle rows first (workgroups are dispatched x-fastest, then y, then z):
  // the horizon rows hold the long lists, and a wave that walks 60 entries through four quadrants lives ten times longer than the
  // average one -- dispatched with the last pose group it WAS the kernel's tail (a fixed ~0.1 ms per launch whatever the batch
  // size, which the 128-pose renders of a strong-scaling share paid nine times per step).  Now every pose's heavy rows start
  // first and the launch ends with the cheap rows (floor, ceiling, sky: mostly settled, their waves end after one load).
  const uint32_t pose = blockIdx.y * 8u + (blockIdx.x & 7u);
  const uint32_t zk = blockIdx.z, zc = (uint32_t)tiles_y >> 1;
  const uint32_t tile_y = (zk & 1u) ? zc - ((zk + 1u) >> 1) : zc + (zk >> 1);  // c, c - 1, c + 1, c - 2, ...
#else
  const uint32_t pose = blockIdx.z * 8u + (blockIdx.x & 7u);
  const uint32_t tile_y = blockIdx.y;
#endif
  const int tid = threadIdx.x, wave = 0, lane = tid & 63;
  const uint32_t tile_x = blockIdx.x >> 3, tile = tile_y * (uint32_t)tiles_x + tile_x;
  if (pose >= n_poses) return;
  RT_DECL
  const int tx0 = (int)tile_x * TILE_W, ty0 = (int)tile_y * TILE_H;
  const int lx = (lane & 7) * 4, ly = (lane >> 3) * 4;  // this lane's 4x4 block inside a quadrant
  const TriRec *prec = recs + (size_t)pose * cap;
  // (three independent scalar loads in flight at once, not a chain: the header is read whether or not it will be used)
  const uint32_t over = overflow[pose], all_visible = counts[pose];
  const uint2 hdr_binned = tile_hdr[(size_t)pose * T + tile];
  const bool binned = over == 0u;  // the pose's per-tile lists are complete
  // settle_kernel ran over this pose's lists (settled != 0, bins complete): the tile's four table entries say which quadrants are
  // still to do.  Instantiations that owe visibility words or primitive ids to EVERY quadrant do them all, as before.
  uint32_t todo = 0xFu;
  if (SKIPVIS && !PRIM && settled != 0u && binned) {
    const uint4 qt = *reinterpret_cast<const uint4 *>(qtab + ((size_t)pose * T + tile) * 4u);  // (uniform address: a scalar load)
    todo = (qt.x == QTAB_TODO ? 1u : 0u) | (qt.y == QTAB_TODO ? 2u : 0u) | (qt.z == QTAB_TODO ? 4u : 0u) | (qt.w == QTAB_TODO ? 8u : 0u);
    if (todo == 0u) return;
  }
  // A tile with a long list (more than 64 entries: far geometry, small triangles that touch one quadrant each) comes with a list
  // PER QUADRANT (bin.hip, "split lists"): each quadrant's pass starts with the gather of ITS list -- most of those fit one
  // batch again (one gather, the shortcuts apply), and a quadrant's pass no longer gathers the records of the other three.
  // pent / count / single then describe the current quadrant's list.
  const bool split = SPLIT && binned && (hdr_binned.y & TILE_SPLIT) != 0u;
  const uint2 hdr = binned ? hdr_binned : make_uint2(0u, all_visible);
  const uint32_t *pent = entries + (size_t)pose * entry_cap + hdr.x;  // (a split tile: set per quadrant)
  uint32_t count = split ? 0u : hdr.y;
  bool single = count <= 64u;  // the usual case: one gather serves all four quadrants
  uint32_t *myq = wq[wave];
  // what lane s keeps of the s-th entry of the current batch: record index, quadrant bits (touches: 0..3, covers:
  // 4..7), the depth plane, the nearest depth over each quadrant
  // (record index and quadrant bits share one register: a level has fewer than 2^24 triangles, rdoom_level_create checks)
  uint32_t n = 0, myrq = 0, zpa = 0, zpb = 0, zpc = 0, dnq0 = NONE, dnq1 = NONE, dnq2 = NONE, dnq3 = NONE;
  // Gathers one batch of 64 list entries into the lanes (see the header: candidates, ranking, records, the per-quadrant
  // nearest depths and cover flags).  The usual tile has one batch, gathered once for its four quadrants.
  auto gather = [&](uint32_t base, const int qonly) {
    // ---- candidates: one per lane ----------------------------------------------------------------
    const uint32_t i = base + (uint32_t)lane;
    uint32_t cand = 0, qb = 0;
    if (i < count) {
      if (binned) {
        const uint32_t e = pent[i];
        cand = e & ENTRY_REC_MASK;
        qb = e >> 28;  // exact quadrant tests done by the binning kernel
      } else {
        // pose without complete bins: every visible triangle is a candidate; bbox, then the exact quadrant tests
        const uint4 bb = reinterpret_cast<const uint4 *>(&prec[i])[3];  // (bb0, bb1, flags, pad): records lie near to far
        cand = i;
        const int x0 = (int)(bb.x & 0xFFFFu), y0 = (int)(bb.x >> 16), x1 = (int)(bb.y & 0xFFFFu), y1 = (int)(bb.y >> 16);
        if (x0 <= tx0 + 63 && x1 >= tx0 && y0 <= ty0 + 63 && y1 >= ty0) {
          const uint4 *rp = reinterpret_cast<const uint4 *>(&prec[cand]);
          qb = tile_quadrant_mask(rp[0], rp[1], rp[2], x0, y0, x1, y1, tx0, ty0);
        }
      }
    }

    const unsigned long long rm = __ballot(qb != 0u);
    n = (uint32_t)__popcll(rm);
    if (n != 0u) {
      // rank of my entry among the relevant ones (record index = depth rank; the lists are near-sorted already).  A batch of
      // more than RDOOM_RANK_LIMIT entries of a binned list is taken in list order instead (every entry of such a list is
      // relevant, lane s already holds the s-th; the binning kernel's lists are near to far up to its window of 256 records).
      uint32_t e = cand | (qb << 28);
      if (!binned || n <= RDOOM_RANK_LIMIT) {
        uint32_t rank = 0;
        for (unsigned long long m = rm; m; m &= m - 1ull) {
          const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)cand, (int)__builtin_ctzll(m));
          rank += kj < cand ? 1u : 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // earlier readers of myq / wrec are done
        if (qb != 0u) myq[rank] = e;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if ((uint32_t)lane < n) e = myq[lane];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // earlier readers of wrec are done
      // ---- records: lane s gathers entry s -------------------------------------------------------
      const bool have = (uint32_t)lane < n;
      myrq = 0u, zpa = zpb = zpc = 0u, dnq0 = dnq1 = dnq2 = dnq3 = NONE;
      if (have) {
        const uint32_t myrec = e & ENTRY_REC_MASK;
        uint32_t myqb = e >> 28;
        uint4 c0, c1, c2, c3;
        uint2 c4;
        raster_words(&prec[myrec], c0, c1, c2, c3, c4);
        uint4 *mine = wrec[wave][lane];
        mine[0] = c0, mine[1] = c1, mine[2] = c3, mine[3] = make_uint4(c2.x, c4.x, c4.y, 0u);
        zpa = c2.y, zpb = c2.z, zpc = c2.w;
        // per quadrant: nearest depth of my entry's plane over it as d24 (none if beyond far), and whether my
        // entry covers it entirely -- the smallest computed value of each edge function and of 1/w and the
        // extremes of the depth over the quadrant sit at corners (fmaf is monotone in each argument)
        const float e0a = __uint_as_float(c0.x), e0b = __uint_as_float(c0.y), e0c = __uint_as_float(c0.z),
                    e1a = __uint_as_float(c0.w), e1b = __uint_as_float(c1.x), e1c = __uint_as_float(c1.y),
                    e2a = __uint_as_float(c1.z), e2b = __uint_as_float(c1.w), e2c = __uint_as_float(c2.x);
        const float za = __uint_as_float(c2.y), zb = __uint_as_float(c2.z), zc = __uint_as_float(c2.w);
        const float wa = __uint_as_float(c3.x), wb = __uint_as_float(c3.y), wc = __uint_as_float(c3.z);
        const int x0 = (int)(c3.w & 0xFFFFu), y0 = (int)(c3.w >> 16), x1 = (int)(c4.x & 0xFFFFu), y1 = (int)(c4.x >> 16);
        uint32_t dn[4], covx = 0u;
#pragma unroll
        for (int qi = 0; qi < 4; qi++) {
          if (qi != qonly && qonly >= 0) {  // (uniform) a split tile's gather serves one quadrant: the others' values are not used
            dn[qi] = NONE;
            continue;
          }
          const int rx0 = tx0 + (qi & 1) * 32, ry0 = ty0 + (qi >> 1) * 32;
          const float xl = (float)rx0 + 0.5f, xh = (float)rx0 + 31.5f, yl = (float)ry0 + 0.5f, yh = (float)ry0 + 31.5f;
          const float zn = fmaf(za, pos(za) ? xl : xh, fmaf(zb, pos(zb) ? yl : yh, zc));
          const float zf = fmaf(za, pos(za) ? xh : xl, fmaf(zb, pos(zb) ? yh : yl, zc));
          dn[qi] = zn <= 1.0f ? __float2uint_rz(fmaf(fminf(fmaxf(zn, 0.0f), 1.0f), 16777215.0f, 0.5f)) : NONE;
          const float n0 = fmaf(e0a, pos(e0a) ? xl : xh, fmaf(e0b, pos(e0b) ? yl : yh, e0c));
          const float n1 = fmaf(e1a, pos(e1a) ? xl : xh, fmaf(e1b, pos(e1b) ? yl : yh, e1c));
          const float n2 = fmaf(e2a, pos(e2a) ? xl : xh, fmaf(e2b, pos(e2b) ? yl : yh, e2c));
          const float rwn = fmaf(wa, pos(wa) ? xl : xh, fmaf(wb, pos(wb) ? yl : yh, wc));
          // (the bbox is clipped to the frame: a quadrant that crosses the frame's right or bottom edge is covered when the
          // bbox reaches that edge -- the pixels beyond it do not exist; the corner values above are those of the whole
          // quadrant, which only asks for more)
#ifndef RDOOM_NO_EDGE_COVER
          const int qx1 = min(rx0 + 31, width - 1), qy1 = min(ry0 + 31, height - 1);
#else
          const int qx1 = rx0 + 31, qy1 = ry0 + 31;
#endif
          const bool common = (zn >= 0.0f) & (zf <= 1.0f) & (rwn > 0.0f) & (x0 <= rx0) & (x1 >= qx1) & (y0 <= ry0) & (y1 >= qy1) &
                              ((c4.y & RASTER_MASKED_INTERIOR) == 0u);
          const bool p0 = n0 > 0.0f, p1 = n1 > 0.0f, p2 = n2 > 0.0f;
          const bool cov = common & p0 & p1 & p2;
          myqb |= (cov && !(no_cover & 1u)) ? (16u << qi) : 0u;  // no_cover bit 0: test hook "no_cover"
          covx |= ((common & p1 & p2) ? (1u << qi) : 0u) | ((common & p0 & p2) ? (16u << qi) : 0u) | ((common & p0 & p1) ? (256u << qi) : 0u);
        }
        {
          auto mag = [](uint32_t v) { return v & 0x7FFFFFFFu; };
          auto edge_hash = [&](uint32_t a, uint32_t b, uint32_t c) {
            return mag(a) ^ __builtin_amdgcn_alignbit(mag(b), mag(b), 21) ^ __builtin_amdgcn_alignbit(mag(c), mag(c), 11);
          };
          whash[wave][lane] = make_uint4(edge_hash(c0.x, c0.y, c0.z), edge_hash(c0.w, c1.x, c1.y), edge_hash(c1.z, c1.w, c2.x), no_cover ? 0u : covx);  // (bit 1 alone: test hook "no_pair")
        }
        dnq0 = dn[0], dnq1 = dn[1], dnq2 = dn[2], dnq3 = dn[3];
        myrq = myrec | (myqb << 24);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  };
#ifdef RDOOM_RASTER_TIMERS
  asm volatile("" ::"s"(count));
#endif
  RT_MARK(0);  // header, overflow flag, count
  if (single && count != 0u) gather(0u, -1);
  RT_MARK(1);  // list gather: entries, ranking, records, per-quadrant nearest depths and cover flags
#ifdef RDOOM_TIMING_EXPERIMENTS  // the list gather and record set-up run twice: the difference in kernel time is their cost
  asm volatile("" ::: "memory");
  if (single && count != 0u) gather(0u, -1);
#endif
#ifndef RDOOM_NO_TILE_SHORTCUT
  // The same shortcut one level up: the tile's nearest entry (by its nearest depth over the quadrants it touches, not by its
  // place in the ranked list) covers all four quadrants and every other entry lies, in every quadrant it touches, strictly
  // behind that entry's farthest depth over the whole tile.
  // (Whole tiles only: its stores carry no frame checks.  The quadrants of a tile that crosses the frame's edge take the
  // quadrant-level shortcut below.)
  if (!split && single && n != 0u && todo == 0xFu && tx0 + TILE_W <= width && ty0 + TILE_H <= height) {
    // my entry's nearest depth over the quadrants it touches (lanes without an entry hold NONE everywhere)
    const uint32_t tq = myrq >> 24;
    const uint32_t near_all = min(min((tq & 1u) ? dnq0 : NONE, (tq & 2u) ? dnq1 : NONE), min((tq & 4u) ? dnq2 : NONE, (tq & 8u) ? dnq3 : NONE));
#ifndef RDOOM_RANKED_SHORTCUT
    const uint32_t sc = (uint32_t)__builtin_ctzll(__ballot(near_all == wave_min_u32(near_all)));  // the nearest entry of the tile
#else
    const uint32_t sc = 0u;  // the first of the ranked list
#endif
    const uint32_t rq0 = (uint32_t)__builtin_amdgcn_readlane((int)myrq, (int)sc);
    if ((rq0 >> 28) == 0xFu) {
      const float za0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)zpa, (int)sc)),
                  zb0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)zpb, (int)sc)),
                  zc0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)zpc, (int)sc));
      const float xl = (float)tx0 + 0.5f, xh = (float)tx0 + 63.5f, yl = (float)ty0 + 0.5f, yh = (float)ty0 + 63.5f;
      const float zf0 = fmaf(za0, pos(za0) ? xh : xl, fmaf(zb0, pos(zb0) ? yh : yl, zc0));  // in [0, 1]: the entry covers
      const uint32_t df0 = __float2uint_rz(fmaf(zf0, 16777215.0f, 0.5f));
      if ((__ballot(near_all <= df0) & ~(1ull << sc)) == 0ull) {
        const uint32_t r0 = rq0 & 0xFFFFFFu;
        const uint32_t p0 = PRIM ? (prec[r0].r.flags & 0xFFFFFFu) : 0u;
        if (STATS) st[0] += (unsigned long long)n, st[9] += 4ull;
        if (qtab && lane < 4) qtab[((size_t)pose * T + tile) * 4u + (uint32_t)lane] = r0;
        if (!SKIPVIS || PRIM) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int bx = tx0 + (q & 1) * 32 + lx, by = ty0 + (q >> 1) * 32 + ly;
            const size_t o0 = ((size_t)pose * (size_t)height + (size_t)by) * (size_t)width + (size_t)bx;
#pragma unroll
            for (int ry = 0; ry < 4; ry++) {
              const size_t o = o0 + (size_t)(ry * width);
              if (!SKIPVIS) {
                if (VIS16)
                  *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(vis) + o) = make_uint2(r0 | (r0 << 16), r0 | (r0 << 16));
                else
                  *reinterpret_cast<uint4 *>(vis + o) = make_uint4(r0, r0, r0, r0);
              }
              if (PRIM) *reinterpret_cast<uint4 *>(prim_out + o) = make_uint4(p0, p0, p0, p0);
            }
          }
        }
        if (STATS && lane == 0)
          for (int k = 0; k < 20; k++) atomicAdd(&stats[k], st[k]);
        RT_MARK(2);  // tile-level shortcut, taken
        RT_FLUSH();
        return;
      }
    }
  }
#endif
  RT_MARK(3);  // tile-level shortcut, not taken
#pragma unroll 1
  for (int q = 0; q < 4; q++) {
    const int qx0 = tx0 + (q & 1) * 32, qy0 = ty0 + (q >> 1) * 32;  // this quadrant
    if (qx0 >= width || qy0 >= height) continue;                    // entirely outside the frame (partial tiles)
    if (((todo >> q) & 1u) == 0u) continue;                         // settled by settle_kernel: its table entry stands
    const int bx = qx0 + lx, by = qy0 + ly;                         // this lane's 4x4 block
    const float pxlo = (float)bx + 0.5f, pxhi = (float)bx + 3.5f, pylo = (float)by + 0.5f, pyhi = (float)by + 3.5f;
    if (split) {  // this quadrant's own list: (first entry, count) from the tile's sub-header
      const uint32_t *sh = entries + (size_t)pose * entry_cap + hdr.x + 2u * (uint32_t)q;
      pent = entries + (size_t)pose * entry_cap + sh[0];
      count = sh[1];
      single = count <= 64u;
      n = 0u;
      if (single && count != 0u) gather(0u, q);
    }
    if (single && n != 0u) {
      // Shortcut for the commonest quadrant of all: its nearest entry covers it entirely and every other entry of the
      // (complete, single-batch) list lies strictly behind that entry's FARTHEST depth over the quadrant -- the entry
      // wins all 1024 pixels without a compare (extremes of the computed depth sit at corners; strictness rules out
      // ties).  Nothing is initialised for such a quadrant; the visibility words are one broadcast value.
      const uint32_t dnq_s = q == 0 ? dnq0 : (q == 1 ? dnq1 : (q == 2 ? dnq2 : dnq3));
      const unsigned long long touch_s = __ballot(((myrq >> (24 + q)) & 1u) != 0u);
      const unsigned long long cover_s = __ballot(((myrq >> (28 + q)) & 1u) != 0u);
      if (touch_s != 0ull) {
#ifndef RDOOM_RANKED_SHORTCUT
        // the entry with the nearest depth over THIS quadrant, wherever it stands in the tile's ranking (the ranking is by
        // the triangles' depth as a whole: a floor triangle that starts at the camera's feet precedes the wall it ends at)
        const uint32_t near_q = wave_min_u32(((myrq >> (24 + q)) & 1u) ? dnq_s : NONE);
        const uint32_t s0 = (uint32_t)__builtin_ctzll(__ballot(dnq_s == near_q) & touch_s);
#else
        const uint32_t s0 = (uint32_t)__builtin_ctzll(touch_s);
#endif
        if ((cover_s >> s0) & 1ull) {
          const float za0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)zpa, (int)s0)),
                      zb0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)zpb, (int)s0)),
                      zc0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)zpc, (int)s0));
          const float xl = (float)qx0 + 0.5f, xh = (float)qx0 + 31.5f, yl = (float)qy0 + 0.5f, yh = (float)qy0 + 31.5f;
          const float zf0 = fmaf(za0, pos(za0) ? xh : xl, fmaf(zb0, pos(zb0) ? yh : yl, zc0));  // in [0, 1]: the entry covers
          const uint32_t df0 = __float2uint_rz(fmaf(zf0, 16777215.0f, 0.5f));
          if ((__ballot(dnq_s <= df0) & touch_s & ~(1ull << s0)) == 0ull) {
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)myrq, (int)s0) & 0xFFFFFFu;
            if (STATS) st[0] += (unsigned long long)__popcll(touch_s), st[9]++;
            // the quadrant table: "all 1024 pixels show record r0" -- the fragment kernel's waves then take the record
            // by scalar loads without reading (or comparing) the visibility words of this quadrant
            if (qtab && lane == 0) qtab[((size_t)pose * T + tile) * 4u + (uint32_t)q] = r0;
            if ((!SKIPVIS || PRIM) && bx < width) {
              const size_t o0 = ((size_t)pose * (size_t)height + (size_t)by) * (size_t)width + (size_t)bx;
              const uint32_t p0 = PRIM ? (prec[r0].r.flags & 0xFFFFFFu) : 0u;
#pragma unroll
              for (int ry = 0; ry < 4; ry++) {
                if (by + ry < height) {
                  const size_t o = o0 + (size_t)(ry * width);
                  if (!SKIPVIS) {
                    if (VIS16)
                      *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(vis) + o) = make_uint2(r0 | (r0 << 16), r0 | (r0 << 16));
                    else
                      *reinterpret_cast<uint4 *>(vis + o) = make_uint4(r0, r0, r0, r0);
                  }
                  if (PRIM) *reinterpret_cast<uint4 *>(prim_out + o) = make_uint4(p0, p0, p0, p0);
                }
              }
            }
            RT_MARK(4);  // quadrant shortcut, taken
            continue;
          }
        }
#ifndef RDOOM_NO_PAIR_SHORTCUT
        // The two-entry shortcut.  A third of the quadrants that show more than one triangle show exactly TWO that share an edge
        // -- the diagonal of a wall quad, a spoke of a floor fan (census with the oracle's winners: 9.6 % of all quadrants at
        // 1080p).  Set-up gives a shared edge exactly negated coefficients in the two triangles (S3), so e_B(pixel) = -e_A(pixel)
        // bit for bit and the tie flags of the two sides are complementary: every pixel lies inside exactly one of them as far
        // as that edge is concerned.  When both triangles cover the quadrant but for that edge (their other edges, the depth
        // range, 1/w and the bbox hold at the quadrant's corners -- the same corner arguments as the one-entry shortcut) and
        // every other entry lies strictly behind the farther of their farthest depths, the winner of a pixel is decided by the
        // sign of ONE edge function: no depth is evaluated, nothing is initialised.  A = the entry that is nearest over the
        // quadrant, B = a touching entry one of whose edge hashes equals one of A's (verified exactly below).
        const uint4 hA4 = whash[wave][s0];
        const uint32_t cxA = ((uint32_t)__builtin_amdgcn_readfirstlane((int)hA4.w) >> q) & 0x111u;  // A covers but for edge 0 / 1 / 2: bits 0 / 4 / 8
        if (cxA != 0u && __popcll(touch_s) >= 2) {  // (most quadrants that show several triangles leave here: no edge of A to share)
          const uint4 hme = whash[wave][lane];
          const uint32_t hA0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)hA4.x), hA1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)hA4.y),
                         hA2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)hA4.z);
          const uint32_t cxm = (hme.w >> q) & 0x111u;
          // bit 3 i + j: A's edge i and my edge j have equal hashes, and each of us covers the quadrant but for that edge
          const bool m0 = (cxm & 1u) != 0u, m1 = (cxm & 16u) != 0u, m2 = (cxm & 256u) != 0u;
          const bool A0 = (cxA & 1u) != 0u, A1 = (cxA & 16u) != 0u, A2 = (cxA & 256u) != 0u;
          const uint32_t mm = (((hme.x == hA0) & m0 & A0) ? 1u : 0u) | (((hme.y == hA0) & m1 & A0) ? 2u : 0u) | (((hme.z == hA0) & m2 & A0) ? 4u : 0u) |
                              (((hme.x == hA1) & m0 & A1) ? 8u : 0u) | (((hme.y == hA1) & m1 & A1) ? 16u : 0u) | (((hme.z == hA1) & m2 & A1) ? 32u : 0u) |
                              (((hme.x == hA2) & m0 & A2) ? 64u : 0u) | (((hme.y == hA2) & m1 & A2) ? 128u : 0u) | (((hme.z == hA2) & m2 & A2) ? 256u : 0u);
          unsigned long long cm = __ballot(mm != 0u) & touch_s & ~(1ull << s0);
          bool paired = false;
          while (cm != 0ull && !paired) {
            const uint32_t sB = (uint32_t)__builtin_ctzll(cm);
            cm &= cm - 1ull;
            const uint32_t mB = (uint32_t)__builtin_amdgcn_readlane((int)mm, (int)sB);
            const uint32_t ij = (uint32_t)__builtin_ctz(mB), ei = ij / 3u, ej = ij - ei * 3u;
            // the two edges, exactly: words 3 k .. 3 k + 2 of the nine edge words (parked as words 0..7 and 12 of the record's slot)
            const uint32_t *wa = reinterpret_cast<const uint32_t *>(wrec[wave][s0]), *wb = reinterpret_cast<const uint32_t *>(wrec[wave][sB]);
            auto word = [](const uint32_t *w, uint32_t k) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)w[k < 8u ? k : 12u]); };  // edge words
            auto slot = [](const uint32_t *w, uint32_t k) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)w[k]); };                  // any parked word
            const uint32_t a0 = word(wa, 3u * ei), a1 = word(wa, 3u * ei + 1u), a2 = word(wa, 3u * ei + 2u);
            const uint32_t b0 = word(wb, 3u * ej), b1 = word(wb, 3u * ej + 1u), b2 = word(wb, 3u * ej + 2u);
            auto negated = [](uint32_t x, uint32_t y) { return ((x ^ y) == 0x80000000u) | (((x | y) << 1) == 0u); };  // y = -x, or both are zeros
            const uint32_t fA = slot(wa, 14u), fB = slot(wb, 14u);  // flags: primitive id, tie bits 24..26 (slot word 14 = c4.y)
            const bool tlA = ((fA >> (24u + ei)) & 1u) != 0u, tlB = ((fB >> (24u + ej)) & 1u) != 0u;
            bool ok = negated(a0, b0) & negated(a1, b1) & negated(a2, b2) & (tlA != tlB);  // (the cover-but-for-this-edge bits are in mm)
            if (!ok) continue;
            const float ea = __uint_as_float(a0), eb = __uint_as_float(a1), ec = __uint_as_float(a2);
            const float xl = (float)qx0 + 0.5f, xh = (float)qx0 + 31.5f, yl = (float)qy0 + 0.5f, yh = (float)qy0 + 31.5f;
            // the shared edge function is finite at the four corners, hence (fmaf is monotone in each argument) at every pixel
            const float tl_ = fmaf(eb, yl, ec), th_ = fmaf(eb, yh, ec);
            const bool finite = ((__float_as_uint(fmaf(ea, xl, tl_)) & 0x7F800000u) != 0x7F800000u) & ((__float_as_uint(fmaf(ea, xh, tl_)) & 0x7F800000u) != 0x7F800000u) &
                                ((__float_as_uint(fmaf(ea, xl, th_)) & 0x7F800000u) != 0x7F800000u) & ((__float_as_uint(fmaf(ea, xh, th_)) & 0x7F800000u) != 0x7F800000u);
            // farthest depth of the two over the quadrant; everything else strictly behind it
            auto far_of = [&](uint32_t sl) {
              const float za = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)zpa, (int)sl)), zb = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)zpb, (int)sl)),
                          zc = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)zpc, (int)sl));
              const float zf = fmaf(za, pos(za) ? xh : xl, fmaf(zb, pos(zb) ? yh : yl, zc));  // in [0, 1]: the entry's depth range holds over the quadrant
              return __float2uint_rz(fmaf(zf, 16777215.0f, 0.5f));
            };
            const uint32_t dfar = max(far_of(s0), far_of(sB));
            ok = finite & ((__ballot(dnq_s <= dfar) & touch_s & ~(1ull << s0) & ~(1ull << sB)) == 0ull);
            if (!ok) continue;
            paired = true;
            const uint32_t rA = (uint32_t)__builtin_amdgcn_readlane((int)myrq, (int)s0) & 0xFFFFFFu, rB = (uint32_t)__builtin_amdgcn_readlane((int)myrq, (int)sB) & 0xFFFFFFu;
            if (STATS) st[0] += (unsigned long long)__popcll(touch_s), st[8]++;
            if (qtab && lane == 0) qtab[((size_t)pose * T + tile) * 4u + (uint32_t)q] = NONE;  // two records: not described
            if (bx < width) {
              const size_t o0 = ((size_t)pose * (size_t)height + (size_t)by) * (size_t)width + (size_t)bx;
              const uint32_t pA = PRIM ? (fA & 0xFFFFFFu) : 0u, pB = PRIM ? (fB & 0xFFFFFFu) : 0u;
#pragma unroll
              for (int ry = 0; ry < 4; ry++) {
                const float trow = fmaf(eb, pylo + (float)ry, ec);
                uint32_t w[4];
                bool in[4];
#pragma unroll
                for (int rx = 0; rx < 4; rx++) {
                  const float e = fmaf(ea, pxlo + (float)rx, trow);  // R1 for A's side of the shared edge
                  in[rx] = (e > 0.0f) | ((e == 0.0f) & tlA);
                  w[rx] = in[rx] ? rA : rB;
                }
                if (by + ry < height) {
                  const size_t o = o0 + (size_t)(ry * width);
                  if (VIS16)
                    *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(vis) + o) =
                        make_uint2(__builtin_amdgcn_perm(w[1], w[0], 0x05040100u), __builtin_amdgcn_perm(w[3], w[2], 0x05040100u));
                  else
                    *reinterpret_cast<uint4 *>(vis + o) = make_uint4(w[0], w[1], w[2], w[3]);
                  if (PRIM) *reinterpret_cast<uint4 *>(prim_out + o) = make_uint4(in[0] ? pA : pB, in[1] ? pA : pB, in[2] ? pA : pB, in[3] ? pA : pB);
                }
              }
            }
          }
          if (paired) {
            RT_MARK(4);
            continue;
          }
#ifdef RDOOM_CENSUS_TWO  // census build only (tools/variant.sh rcensus raster -DRDOOM_CENSUS_TWO): what do the remaining full passes look like?
          {
            const uint32_t nt = min((uint32_t)__popcll(touch_s), 7u);
            const uint32_t ncx = min((uint32_t)__popcll(__ballot(((hme.w >> q) & 0x111u) != 0u) & touch_s), 3u);   // entries that cover but for one edge
            if (lane == 0) atomicAdd(&g_raster_census[nt * 4u + ncx], 1ull);
          }
#endif
        }
#endif
      }
    }
    RT_MARK(5);  // quadrant shortcut, not taken
    uint32_t best_d[16], best_r[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      best_d[k] = NONE;
      best_r[k] = NONE;
    }
    // A frame height that is not a multiple of 4: the bottom block row of the frame has pixel rows below it.  They start at depth 0
    // -- nothing is nearer, so they never win (a depth equal to 0 replays through the general rule, whose bounding-box test
    // excludes them) -- instead of "none", which kept such a lane's farthest depth, and with it the wave's, at "none" for the
    // whole pass: no early-z on the bottom quadrant row.  (uniform branch: only quadrants that cross the frame's bottom edge)
    const bool rows_below = qy0 + 32 > height;
    if (rows_below) {
#pragma unroll
      for (int ry = 1; ry < 4; ry++)
        if (by + ry >= height && by < height) best_d[4 * ry] = best_d[4 * ry + 1] = best_d[4 * ry + 2] = best_d[4 * ry + 3] = 0u;
    }
    // max of best_d: the farthest depth this lane still holds.  A lane whose block lies outside the frame (partial tiles:
    // the bottom row at 1080 = 16 * 64 + 56) stores nothing and must not keep the wave-wide farthest depth at "none"
    uint32_t lane_far = ((bx >= width) | (by >= height)) ? 0u : NONE;
    bool had_cover = false;  // (uniform) some entry covers the whole quadrant
#pragma unroll 1
    for (uint32_t base = 0; base < count; base += 64u) {
      if (!single) gather(base, split ? q : -1);
      if (n == 0u) continue;
      // ---- walk: the entries that touch this quadrant, near to far -------------------------------------------
      const uint32_t dnq = q == 0 ? dnq0 : (q == 1 ? dnq1 : (q == 2 ? dnq2 : dnq3));
      const unsigned long long qcm = __ballot(((myrq >> (28 + q)) & 1u) != 0u);
      had_cover |= qcm != 0ull;
      // An entry is hidden in the whole quadrant when its nearest depth over the quadrant (lane s holds entry s's) is
      // beyond the farthest depth ANY lane still holds: all entries are tested at once against that wave-wide maximum,
      // again whenever a body has brought some lane's depths nearer.
      const unsigned long long touch = __ballot(((myrq >> (24 + q)) & 1u) != 0u);
      uint32_t wave_far = wave_max_u32(lane_far);
      unsigned long long wm = touch & __ballot(dnq <= wave_far);
      if (STATS) st[0] += (unsigned long long)__popcll(touch), st[15] += (unsigned long long)__popcll(touch & ~wm);
#ifndef RDOOM_RANKED_WALK
      // The covering entry that is nearest over this quadrant goes first, wherever it stands in the ranking (the winner does
      // not depend on the order): its depth-only body then puts a bound on every lane, and entries ranked before it that lie
      // behind it are dropped by the compare below instead of being rasterised.
      uint32_t s_first = 64u;
      if (qcm & wm) s_first = (uint32_t)__builtin_ctzll(__ballot(dnq == wave_min_u32(((qcm >> lane) & 1ull) ? dnq : NONE)) & qcm & wm);
#endif
      RT_MARK(6);  // quadrant pass set-up: initialisation, wave-wide farthest depth, first covering entry
      while (wm) {
#ifndef RDOOM_RANKED_WALK
        const uint32_t s = s_first < 64u ? s_first : (uint32_t)__builtin_ctzll(wm);
        s_first = 64u;
        wm &= ~(1ull << s);
#else
        const uint32_t s = (uint32_t)__builtin_ctzll(wm);
        wm &= wm - 1ull;
#endif
        const uint32_t far_before = lane_far;
        auto refresh = [&]() {  // after a body: drop what is hidden now
          if (__any(lane_far != far_before)) {
            wave_far = wave_max_u32(lane_far);
            const unsigned long long alive = __ballot(dnq <= wave_far);
            if (STATS) st[15] += (unsigned long long)__popcll(wm & ~alive);
            wm &= alive;
          }
        };
        if (STATS) st[1]++;
        auto bc = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)s); };
        auto bf = [&](uint32_t v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)v, (int)s)); };
        const float za = bf(zpa), zb = bf(zpb), zc = bf(zpc);
        const uint32_t ridx = bc(myrq) & 0xFFFFFFu;
        if ((qcm >> s) & 1ull) {
          // the triangle covers the whole quadrant, inside its bbox, the depth range and in front of the eye, texture
          // rectangle opaque: depth compares only.  (Lanes whose block is hidden lose every compare.)  A depth tie
          // is replayed through the regular path below, which re-resolves the block exactly.
          if (STATS) st[10]++;
          uint32_t tiez = NONE;  // min over the block of (d24 - best): zero iff some depth ties
          bool updated = false;
#pragma unroll
          for (int ry = 0; ry < 4; ry++) {
            const float tz = fmaf(zb, pylo + (float)ry, zc);
#pragma unroll
            for (int rx = 0; rx < 4; rx++) {
              const int k = ry * 4 + rx;
              const uint32_t d24 = __float2uint_rz(fmaf(fmaf(za, pxlo + (float)rx, tz), 16777215.0f, 0.5f));
              uint32_t diff;
              const bool win = __builtin_usub_overflow(d24, best_d[k], &diff);
              tiez = min(tiez, diff);
              best_d[k] = win ? d24 : best_d[k];
              best_r[k] = win ? ridx : best_r[k];
              updated |= win;
            }
          }
          if (updated) {
            uint32_t m = best_d[0];
#pragma unroll
            for (int k = 1; k < 16; k++) m = max(m, best_d[k]);
            lane_far = m;
          }
          if (!__any(tiez == 0u)) {
            refresh();
            RT_MARK(7);  // covering entry: depth-only body
            continue;
          }
        }
        // the rest of the record: uniform LDS reads, made SGPR operands
        const uint4 *wr = wrec[wave][s];
        const uint4 r0 = wr[0], r1 = wr[1], r2 = wr[2], r3 = wr[3];
        auto uf = [](uint32_t v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)v)); };
        auto uu = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        const uint32_t bb0 = uu(r2.w), bb1 = uu(r3.y), flags = uu(r3.z);
        const int x0 = (int)(bb0 & 0xFFFFu), y0 = (int)(bb0 >> 16), x1 = (int)(bb1 & 0xFFFFu), y1 = (int)(bb1 >> 16);
        raster_entry<STATS>(lv, prec, uf(r0.x), uf(r0.y), uf(r0.z), uf(r0.w), uf(r1.x), uf(r1.y), uf(r1.z), uf(r1.w),
                                 uf(r3.x), za, zb, zc, uf(r2.x), uf(r2.y), uf(r2.z), x0, y0, x1, y1, flags, ridx, bx, by, pxlo,
                                 pxhi, pylo, pyhi, best_d, best_r, lane_far, [&]() -> ShadeRec { return prec[ridx].s; }, st);
        refresh();
        RT_MARK(8);  // other entry: record broadcast, rejection tests, pixel bodies
      }
    }
    if (STATS) {  // the census a per-lane choice of entries would be judged by: the pass's bodies (st[2]) against the most any ONE lane needed
      const uint32_t mx = wave_max_u32((uint32_t)st[16]);
      st[17] += (unsigned long long)mx, st[18] += mx != 0u ? 1ull : 0ull, st[16] = 0ull;
    }
    // ---- this quadrant's visibility words ------------------------------------------------------------------------
    // One record after all?  (The shortcut above needs the winner to lie strictly in front of everything else over the whole
    // quadrant; a quadrant can still end up with one winner.)  Lane 0's first pixel lies inside the frame; lanes whose block
    // lies outside it do not count -- the table speaks about the pixels of the frame.
    uint32_t described = NONE;
#ifndef RDOOM_NO_LATE_TABLE
    if (qtab && had_cover) {  // (one winner needs a triangle that covers the quadrant: without one the check is skipped)
      const uint32_t rf = (uint32_t)__builtin_amdgcn_readfirstlane((int)best_r[0]);
      uint32_t lo = best_r[0], hi = best_r[0];  // (three-operand min / max: 16 instructions for the 16 winners)
      if (!rows_below) {
#pragma unroll
        for (int k = 1; k < 15; k += 2) {
          lo = min(lo, min(best_r[k], best_r[k + 1]));
          hi = max(hi, max(best_r[k], best_r[k + 1]));
        }
        lo = min(lo, best_r[15]), hi = max(hi, best_r[15]);
      } else {  // the pixel rows below the frame (winner "none" for ever) do not count: the table speaks about the pixels of the frame
#pragma unroll
        for (int k = 1; k < 16; k++) {
          const uint32_t v = (by + (k >> 2) >= height) ? rf : best_r[k];
          lo = min(lo, v), hi = max(hi, v);
        }
      }
      const bool outside_lane = (bx >= width) | (by >= height);
      if (rf != NONE && __all(outside_lane | ((lo == rf) & (hi == rf)))) described = rf;
    }
#endif
    if (STATS && described != NONE) st[11]++;
    if (qtab && lane == 0) qtab[((size_t)pose * T + tile) * 4u + (uint32_t)q] = described;  // NONE: not known to be uniform
    const bool want_vis = !SKIPVIS || described == NONE;  // (uniform) a described quadrant needs no visibility words
    if ((want_vis || PRIM) && bx < width) {
      const size_t o0 = ((size_t)pose * (size_t)height + (size_t)by) * (size_t)width + (size_t)bx;
#pragma unroll
      for (int ry = 0; ry < 4; ry++) {
        if (by + ry < height) {
          const size_t o = o0 + (size_t)(ry * width);
          if (!want_vis) {
          } else if (VIS16)  // record indices fit 16 bits (0xFFFF = none): half the visibility traffic
            *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(vis) + o) =
                make_uint2(__builtin_amdgcn_perm(best_r[ry * 4 + 1], best_r[ry * 4], 0x05040100u),
                           __builtin_amdgcn_perm(best_r[ry * 4 + 3], best_r[ry * 4 + 2], 0x05040100u));
          else
            *reinterpret_cast<uint4 *>(vis + o) =
                make_uint4(best_r[ry * 4], best_r[ry * 4 + 1], best_r[ry * 4 + 2], best_r[ry * 4 + 3]);
          if (PRIM) {
            uint32_t p[4];
#pragma unroll
            for (int rx = 0; rx < 4; rx++)
              p[rx] = best_r[ry * 4 + rx] == NONE ? NONE : (prec[best_r[ry * 4 + rx]].r.flags & 0xFFFFFFu);
            *reinterpret_cast<uint4 *>(prim_out + o) = make_uint4(p[0], p[1], p[2], p[3]);
          }
        }
      }
    }
    RT_MARK(9);  // one-winner check, table entry, visibility words
  }
  RT_FLUSH();
  if (STATS && lane == 0)
    for (int k = 0; k < 20; k++) atomicAdd(&stats[k], st[k]);
}

}  // namespace

rdoom_status launch_raster(hipStream_t st, uint32_t n_poses, const DeviceLevelView &lv, const TriRec *recs,
                           const uint32_t *counts, uint32_t cap, int width, int height, int tiles_x,
                           int tiles_y, const uint2 *tile_hdr, const uint32_t *entries, uint32_t entry_cap,
                           const uint32_t *overflow, uint32_t *vis, bool vis16, uint32_t *prim_out, uint32_t *qtab,
                           bool skip_described_vis, bool split_lists, bool bins_launched) {
  const uint32_t n = n_poses;
  const uint32_t groups = (n + 7u) / 8u;  // pose groups of eight: one pose per XCD
  if (groups > 65535u || tiles_y > 65535 || (uint64_t)tiles_x * 8ull > 0x7FFFFFFFull)
    return rdoom::fail(RDOOM_BAD_ARG, "batch too large for one launch");
  const rdoom::DebugOptions &dbg = rdoom::debug_options();
  unsigned long long *d_stats = nullptr;
  if (dbg.raster_stats) {
    HIP_TRY(hipMalloc((void **)&d_stats, 20 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(d_stats, 0, 20 * sizeof(unsigned long long), st));
  }
  const bool skip = skip_described_vis && qtab != nullptr;  // (without a table every visibility word is needed)
  auto pick = [&](auto stats, auto skipvis, auto splt) {
    constexpr bool S = decltype(stats)::value, K = decltype(skipvis)::value, P = decltype(splt)::value;
    return vis16 ? (prim_out ? raster_wave_kernel<S, true, true, K, P> : raster_wave_kernel<S, true, false, K, P>)
                 : (prim_out ? raster_wave_kernel<S, false, true, K, P> : raster_wave_kernel<S, false, false, K, P>);
  };
  auto pick2 = [&](auto splt) {
    return dbg.raster_stats ? (skip ? pick(std::true_type{}, std::true_type{}, splt) : pick(std::true_type{}, std::false_type{}, splt))
                            : (skip ? pick(std::false_type{}, std::true_type{}, splt) : pick(std::false_type{}, std::false_type{}, splt));
  };
  auto rk = split_lists ? pick2(std::true_type{}) : pick2(std::false_type{});
  // settle_kernel first (see there): only where the rasteriser may skip what it settles -- the table is in use, no visibility
  // words or primitive ids are owed for described quadrants, the one-entry shortcut is not switched off by a hook
  const uint32_t max_list = dbg.settle_max > 0 ? (uint32_t)dbg.settle_max : RDOOM_SETTLE_MAX;
  const bool settle = skip && !prim_out && !dbg.no_settle && !dbg.no_cover && !dbg.raster_stats && bins_launched;
  if (settle) {
    const uint32_t T4 = (uint32_t)(tiles_x * tiles_y) * 4u;
    hipLaunchKernelGGL(settle_kernel, dim3(((T4 + 255u) / 256u) * 8u, 1, groups), dim3(256), 0, st, recs, cap, n, width, height, tiles_x, tiles_y, tile_hdr,
                       entries, entry_cap, overflow, qtab, max_list);
  }
#ifndef RDOOM_RASTER_ROWS_INNER
  const dim3 rgrid((uint32_t)tiles_x * 8u, groups, (uint32_t)tiles_y);
#else
  const dim3 rgrid((uint32_t)tiles_x * 8u, (uint32_t)tiles_y, groups);
#endif
  hipLaunchKernelGGL(rk, rgrid, dim3(64 * RASTER_WAVES), 0, st, lv, recs, counts, cap, n, width, height, tiles_x,
                     tiles_y, tile_hdr, entries, entry_cap, overflow, vis, prim_out, (dbg.no_cover ? 1u : 0u) | (dbg.no_pair ? 2u : 0u),
                     qtab, settle ? 1u : 0u, d_stats);
#ifdef RDOOM_CENSUS_TWO
  {
    unsigned long long h[64], zero[64] = {};
    (void)hipStreamSynchronize(st);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_raster_census), sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_raster_census), zero, sizeof zero);
    fprintf(stderr, "[raster census] passes that reach the two-entry test with an edge of the nearest entry to share, by touching entries (rows 0..7+) x entries that cover but for one edge (0..3+):\n");
    for (int t = 0; t < 8; t++) fprintf(stderr, "   %d: %llu %llu %llu %llu\n", t, h[4 * t], h[4 * t + 1], h[4 * t + 2], h[4 * t + 3]);
  }
#endif
#ifdef RDOOM_RASTER_TIMERS
  {
    unsigned long long h[16], zero[16] = {};
    (void)hipStreamSynchronize(st);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_raster_t), sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_raster_t), zero, sizeof zero);
    unsigned long long total = 0;
    for (int k = 0; k < 10; k++) total += h[k];
    static const char *names[10] = {"header", "gather", "tile shortcut taken", "tile shortcut not taken", "quadrant shortcut taken",
                                    "quadrant shortcut not taken", "pass set-up", "cover body", "other entry", "final check + stores"};
    fprintf(stderr, "[raster timers] %llu waves, %.0f cycles per wave:", h[15], h[15] ? (double)total / (double)h[15] : 0.0);
    for (int k = 0; k < 10; k++) fprintf(stderr, "  %s %.1f %%", names[k], total ? 100.0 * (double)h[k] / (double)total : 0.0);
    fprintf(stderr, "\n");
  }
#endif
  if (d_stats) {
    unsigned long long h[20];
    HIP_TRY(hipMemcpy(h, d_stats, sizeof h, hipMemcpyDeviceToHost));
    (void)hipFree(d_stats);
    {  // census of the tile lists' lengths (a tile with more than 64 entries gets a list per quadrant: bin.hip)
      const size_t nt = (size_t)n * (size_t)(tiles_x * tiles_y);
      std::vector<uint2> hdrs(nt);
      HIP_TRY(hipMemcpy(hdrs.data(), tile_hdr, sizeof(uint2) * nt, hipMemcpyDeviceToHost));
      static const uint32_t edge[8] = {0u, 8u, 16u, 32u, 64u, 128u, 256u, 0xFFFFFFFFu};
      unsigned long long tiles_in[8] = {}, entries_in[8] = {}, total = 0;
      for (const uint2 &hd : hdrs) {
        const uint32_t c = hd.y & ~TILE_SPLIT;
        int k = 0;
        while (c > edge[k]) k++;
        tiles_in[k]++, entries_in[k] += c, total += c;
      }
      std::vector<uint32_t> ov(n);
      HIP_TRY(hipMemcpy(ov.data(), overflow, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
      uint32_t n_over = 0;
      for (uint32_t o : ov) n_over += o != 0u;
      fprintf(stderr, "[rdoom stats] poses whose bins overflowed (rasterised from the sorted list; their headers below are stale): %u of %u\n", n_over, n);
      fprintf(stderr, "[rdoom stats] tile lists: %.1f entries per tile;", (double)total / (double)nt);
      static const char *names[8] = {"0", "1-8", "9-16", "17-32", "33-64", "65-128", "129-256", ">256"};
      for (int k = 0; k < 8; k++) fprintf(stderr, "  %s: %.1f %% of the tiles, %.1f %% of the entries;", names[k], 100.0 * tiles_in[k] / (double)nt, total ? 100.0 * entries_in[k] / (double)total : 0.0);
      fprintf(stderr, "\n");
    }
    const double waves = (double)groups * 8.0 * (double)(tiles_x * tiles_y) * 4.0;  // (pose, quadrant) passes
    fprintf(stderr,
            "[rdoom stats] per quadrant pass: queue %.1f  past quadrant early-z %.1f  quadrant-cover %.2f  need-any %.1f (lanes %.1f)  fast %.1f (lanes %.1f)"
            "  general %.3f (lanes %.1f: masked %.1f, tie %.1f) | rejected: early-z %.2f, then geometry %.2f | one-entry shortcut %.3f, one winner in the end %.3f, two-entry shortcut %.4f (%llu of %.0f passes)\n",
            h[0] / waves, h[1] / waves, h[10] / waves, h[2] / waves, h[2] ? (double)h[3] / h[2] : 0.0, h[4] / waves,
            h[4] ? (double)h[5] / h[4] : 0.0, h[6] / waves, h[6] ? (double)h[7] / h[6] : 0.0,
            h[6] ? (double)h[12] / h[6] : 0.0, h[6] ? (double)h[13] / h[6] : 0.0, h[15] / waves, h[14] / waves, h[9] / waves, h[11] / waves, h[8] / waves, h[8], waves);
    fprintf(stderr, "[rdoom stats] sixteen-pixel bodies: %llu in %llu passes that ran any (%.2f per such pass, %.1f lanes each); the most ONE lane needed, summed over those passes: %llu (%.2f per pass) -- "
            "what a body with a per-lane choice of entry would run; all needs / 64: %.2f per pass\n",
            h[2], h[18], h[18] ? (double)h[2] / h[18] : 0.0, h[2] ? (double)h[3] / h[2] : 0.0, h[17], h[18] ? (double)h[17] / h[18] : 0.0, h[18] ? (double)h[3] / 64.0 / h[18] : 0.0);
  }
  return RDOOM_OK;
}

}  // namespace rdoom_dev
