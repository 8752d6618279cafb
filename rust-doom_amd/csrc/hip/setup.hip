// Kernel 1: vertex stage + triangle setup + near-to-far ordering.
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

#include "kernels.hpp"

#pragma clang fp contract(off)

namespace rdoom_dev {
namespace {

// =================================================================================================
// Kernel 1: vertex stage + triangle setup (V2..V5, S1..S6) and front-to-back ordering.
// Three launches, records written once:
//   cull_kernel       a few workgroups per pose, each with a share of the level's clusters (runs of <= 32 triangles with a
//                     bounding box): coarse cull of whole clusters, then the cheap half of the set-up (transform, S1, S4,
//                     S6) for four triangles per thread; the visible triangles are listed and their log-depth buckets
//                     (exponent + top mantissa bits of the nearest vertex's w) counted;
//   sort_scan_kernel  exclusive scan of each pose's bucket histogram;
//   setup_kernel      one lane per VISIBLE triangle does the full set-up -- all lanes busy, where a single pass would run
//                     the whole set-up for every wave that holds one visible triangle -- and stores the record at its
//                     near-to-far position.
// Near geometry first lets the rasteriser's exact early-z test reject most occluded triangles.  The order only affects
// speed: the winner is the lexicographic min of (d24, primitive).
// =================================================================================================
constexpr uint32_t SORT_BUCKETS = 2048;
constexpr uint32_t CULL_CHUNK = 1024;  // triangle slots culled per outer iteration (four per thread), survivors set up densely
constexpr uint32_t CLUSTER_LIST = 1024;  // clusters of a workgroup's share whose survivors fit the LDS list (16 workgroups per pose: levels of up to
                                         // half a million triangles; larger ones: no coarse cull)

// Coarse cull of a cluster (exact implications, no new rule): true only if EVERY triangle whose vertices lie in the box
// fails S1 or S6.  A clip coordinate is one fmaf chain over (x, y, z), monotone in each of them, so its extremes over the
// box sit at corners chosen by the signs of the coefficients; rounding, multiplication by a positive constant and IEEE
// division are monotone, so the bounds survive S2 and S6's own operations:
//   S1  every vertex has w <= max w <= 0;
//   S6  needs min w >= 1e-5 (its bounding-box branch), then e.g. left of the frame: every vertex has
//       sx = ((X + W) * hw) / w <= ((Xmax + Wmax) * hw) / Wmax < -2, hence ceil(max sx) + 1 < 0.
__device__ __forceinline__ bool cluster_culled(const Cluster &c, const float *pm, int width, int height) {
  auto ext = [&](int r, bool want_max) {
    const float a = pm[r], b = pm[4 + r], cc = pm[8 + r];
    const float x = ((a > 0.0f) == want_max) ? c.hi[0] : c.lo[0];
    const float y = ((b > 0.0f) == want_max) ? c.hi[1] : c.lo[1];
    const float z = ((cc > 0.0f) == want_max) ? c.hi[2] : c.lo[2];
    return fmaf(cc, z, fmaf(b, y, fmaf(a, x, pm[12 + r])));
  };
  const float wmax = ext(3, true);
  if (wmax <= 0.0f) return true;  // S1
  const float wmin = ext(3, false);
  if (!(wmin >= 1e-5f)) return false;
  const float hw = 0.5f * (float)width, hh = 0.5f * (float)height;
  const float xhi = ((ext(0, true) + wmax) * hw), xlo = ((ext(0, false) + wmin) * hw);
  const float yhi = ((ext(1, true) + wmax) * hh), ylo = ((ext(1, false) + wmin) * hh);
  // a negative numerator divided by the LARGEST w is the quotient closest to zero; likewise a positive one
  const bool left = xhi < 0.0f && xhi / wmax < -2.0f;
  const bool right = xlo > 0.0f && xlo / wmax >= (float)width + 1.0f;
  const bool below = yhi < 0.0f && yhi / wmax < -2.0f;
  const bool above = ylo > 0.0f && ylo / wmax >= (float)height + 1.0f;
  return left | right | below | above;
}

__device__ __forceinline__ bool setup_triangle(const LevelSlice &lv, const PoseConst &pc,
                                               const ObjectConst *__restrict__ objs, uint32_t t, const LevelTri &tri,
                                               int width, int height, uint32_t kinds_mask, RasterRec &rr, ShadeRec &sr,
                                               float &wkey) {
  wkey = 0.0f;
  bool ok = true;
  {
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
        // S3..S5 in BINARY64 on the binary32 inputs (round 6): a product of two binary32 values is exact in
        // binary64, so an edge coefficient is ONE rounded difference -- still exactly the negative of the neighbouring triangle's
        // across a shared edge -- and the determinant and the plane numerators keep their leading digits where a triangle is thin on
        // the screen (the census against Mesa found u/w, v/w, 1/w up to 3 texels / 0.1 % off there with the binary32 set-up).  Every
        // operation is one IEEE binary64 operation (the unit is compiled with -ffp-contract=off); what the records store is
        // rounded to binary32 once.  The cull kernel's instantiation runs the same operations up to the determinant's sign.
        uint32_t tl = 0;
        double ed[9];
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const int j = (i + 1) % 3, k = (i + 2) % 3;
          ed[3 * i] = (double)yw[j] * (double)w[k] - (double)yw[k] * (double)w[j];
          ed[3 * i + 1] = (double)xw[k] * (double)w[j] - (double)xw[j] * (double)w[k];
          ed[3 * i + 2] = (double)xw[j] * (double)yw[k] - (double)xw[k] * (double)yw[j];
          const float A = (float)ed[3 * i], B = (float)ed[3 * i + 1], C = (float)ed[3 * i + 2];
          rr.e[3 * i] = A;
          rr.e[3 * i + 1] = B;
          rr.e[3 * i + 2] = C;
          if ((A > 0.0f) || (A == 0.0f && B > 0.0f)) tl |= 1u << i;
        }
        const double det = (double)w[0] * ed[2] + ((double)yw[0] * ed[1] + (double)xw[0] * ed[0]);
        ok = det > 0.0;
        if (ok) {
          // S5: the depth plane interpolates Z - zk * W (a small residual: the constant P[3][2] for a perspective
          // matrix) instead of Z, whose dominant part zk * W interpolates to the constant zk exactly; errors of the
          // edge functions then no longer leak |Z| ~ |W| into window depth
          const float rz[3] = {fmaf(-pc.zk, clip[0][3], clip[0][2]), fmaf(-pc.zk, clip[1][3], clip[1][2]),
                               fmaf(-pc.zk, clip[2][3], clip[2][2])};
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const double e0 = ed[c], e1 = ed[3 + c], e2 = ed[6 + c];
            const double nz = (double)rz[2] * e2 + ((double)rz[1] * e1 + (double)rz[0] * e0);
            const double n1 = (e0 + e1) + e2;
            const double nu = (double)u[2] * e2 + ((double)u[1] * e1 + (double)u[0] * e0);
            const double nv = (double)v[2] * e2 + ((double)v[1] * e1 + (double)v[0] * e0);
            const double zp = 0.5 * (nz / det);
            rr.zp[c] = (float)(c == 2 ? zp + (0.5 * (double)pc.zk + 0.5) : zp);
            sr.wp[c] = (float)(n1 / det);
            sr.up[c] = (float)(nu / det);
            sr.vp[c] = (float)(nv / det);
          }
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
          rr.flags = ((t - lv.first_tri) & 0xFFFFFFu) | (tl << 24) | (kind << 27) | (masked << 29);  // primitive id: position in ITS level's draw order
          rr.pad = 0;
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
          const uint32_t tbase = is_flat ? lv.flat_base : (is_decor ? lv.decor_base : lv.wall_base);
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

#ifndef RDOOM_CULL_OCC
#define RDOOM_CULL_OCC 1
#endif
#ifndef RDOOM_SETUP_OCC
#define RDOOM_SETUP_OCC 1
#endif
__global__ __launch_bounds__(256, RDOOM_CULL_OCC) void cull_kernel(DeviceLevelView lv, const PoseConst *__restrict__ poses,
                                                   const ObjectConst *__restrict__ objects, uint32_t n_objects,
                                                   int width, int height, uint32_t kinds_mask,
                                                   uint32_t *__restrict__ visible, uint32_t *__restrict__ counts,
                                                   uint32_t *__restrict__ ghist, uint32_t cap, uint32_t groups) {
  __shared__ uint32_t hist[SORT_BUCKETS];   // this workgroup's share of the pose's depth-bucket histogram
  __shared__ uint16_t clist[CLUSTER_LIST];  // my clusters that survived the coarse cull, ascending
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t chunk_first;          // where this chunk's records go in the pose's staging array
  __shared__ uint4 wstage[4][384];  // per wave: 2 x 32 level triangles (6 x 16 B each)
  // this chunk's visible triangles: written after the barrier that ends the chunk's cull steps and read before the one that
  // ends the chunk, i.e. while no wave uses its stage -- the list shares wave 0's (4 of its 6 KiB); 34 KiB of LDS: four
  // workgroups per CU
  uint32_t *cand = reinterpret_cast<uint32_t *>(wstage[0]);
  static_assert(sizeof(uint32_t) * CULL_CHUNK <= sizeof(uint4) * 384, "cull_kernel: cand does not fit a wave's stage");
  const uint32_t pose = blockIdx.x / groups, group = blockIdx.x - pose * groups;  // (one-dimensional grid: any number of poses)
  const PoseConst &pc = poses[pose];
  const LevelSlice &ls = lv.slices[pc.level];  // (uniform: scalar loads)
  const ObjectConst *objs = objects ? objects + (size_t)pose * n_objects : nullptr;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  uint32_t *pvisible = visible + (size_t)pose * cap;  // the pose's visible triangles, in arrival order
  uint32_t *phist = ghist + (size_t)pose * SORT_BUCKETS;
  for (uint32_t i = tid; i < SORT_BUCKETS; i += 256) hist[i] = 0;
  // my share of the level's clusters
  const uint32_t c_begin = ls.first_cluster + (uint32_t)(((uint64_t)ls.n_clusters * group) / groups),
                 c_end = ls.first_cluster + (uint32_t)(((uint64_t)ls.n_clusters * (group + 1u)) / groups);
  // Phase 0, coarse cull: one lane per cluster, survivors listed in order
  const bool listed = c_end - c_begin <= CLUSTER_LIST;
  uint32_t n_live = c_end - c_begin;  // uniform
  if (listed) {
    n_live = 0;
    for (uint32_t c0 = c_begin; c0 < c_end; c0 += 256u) {
      const uint32_t ci = c0 + (uint32_t)tid;
      bool live = false;
      if (ci < c_end) {
        const Cluster c = lv.clusters[ci];
        const float *pm = objs ? objs[(c.count_object >> 8) & 0xFFFu].pm : pc.pm;
        live = (c.count_object >> 31) != 0u || !cluster_culled(c, pm, width, height);
      }
      const unsigned long long m = __ballot(live);
      if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
      __syncthreads();
      uint32_t off = n_live + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), total = 0;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const uint32_t cnt = wsum[w];
        if (w < wave) off += cnt;
        total += cnt;
      }
      if (live) clist[off] = (uint16_t)(ci - c_begin);
      n_live += total;
      __syncthreads();
    }
  }
  __syncthreads();
  const uint32_t n_slots = n_live * CLUSTER_TRIS;  // every live cluster owns CLUSTER_TRIS triangle slots
  uint4 *stage = wstage[wave];
  for (uint32_t base = 0; base < n_slots; base += CULL_CHUNK) {
    // Phase 1, cull: four steps of one triangle per lane.  In a step the two halves of a wave take the 32 slots of two
    // clusters: the 3 KiB of a cluster's triangles are read with whole-line loads into the wave's LDS stage (a lane
    // reading its own 96 bytes at a 96-byte stride would touch a line per lane and load), and each lane picks its
    // triangle up from there.  Only setup_triangle()'s verdict and nothing it writes is used here, so the compiler drops
    // the planes, divisions and texture parameters from this instantiation; the culls themselves (kind mask, S1, S4, S6)
    // are the same operations as in phase 2.
    uint32_t vis4 = 0, tri_of[4], bucket_of[4];
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) {
      const uint32_t slot = base + j * 256u + (uint32_t)tid;  // lane-contiguous
      const uint32_t half = (uint32_t)lane >> 5, in = (uint32_t)lane & 31u;
      uint32_t first = 0, count = 0;
      if (slot < n_slots) {
        const uint32_t ci = c_begin + (listed ? (uint32_t)clist[slot / CLUSTER_TRIS] : slot / CLUSTER_TRIS);
        first = lv.clusters[ci].first, count = lv.clusters[ci].count_object & 0xFFu;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the stage's previous readers are done
      const uint4 *src = reinterpret_cast<const uint4 *>(lv.tris + first);
#pragma unroll
      for (uint32_t k = 0; k < 6u; k++) {  // 6 x 32 x 16 bytes per half-wave: the cluster's triangles, line by line
        const uint32_t idx = k * 32u + in;
        if (idx < count * 6u) stage[half * 192u + idx] = src[idx];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      tri_of[j] = first + in;
      if (in < count) {
        LevelTri tri;
        uint4 *dst = reinterpret_cast<uint4 *>(&tri);
#pragma unroll
        for (uint32_t k = 0; k < 6u; k++) dst[k] = stage[half * 192u + in * 6u + k];
        RasterRec rr;
        ShadeRec sr;
        float wkey;
        if (setup_triangle(ls, pc, objs, tri_of[j], tri, width, height, kinds_mask, rr, sr, wkey)) vis4 |= 1u << j;
        bucket_of[j] = depth_bucket(wkey);
      }
    }
    // compaction of the survivors (any order will do: the sort below re-orders, and nothing depends on record order):
    // exclusive scan of the per-thread counts over the workgroup
    const uint32_t mine = (uint32_t)__popc(vis4);
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t off = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint32_t c = wsum[w];
      if (w < wave) off += c;
      total += c;
    }
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++)
      if ((vis4 >> j) & 1u) {
        cand[off++] = tri_of[j];
        atomicAdd(&hist[bucket_of[j]], 1u);
      }
    if (tid == 0) chunk_first = total ? atomicAdd(&counts[pose], total) : 0u;  // the pose's workgroups share one list
    __syncthreads();
    for (uint32_t i = tid; i < total; i += 256u) pvisible[chunk_first + i] = cand[i];
    __syncthreads();  // cand, wsum and chunk_first are rewritten by the next chunk
  }
  for (uint32_t i = tid; i < SORT_BUCKETS; i += 256) {
    const uint32_t c = hist[i];
    if (c) atomicAdd(&phist[i], c);
  }
}

// Counting sort, second step: exclusive scan of the pose's bucket histogram, in place (one workgroup per pose).
__global__ __launch_bounds__(256) void sort_scan_kernel(uint32_t *__restrict__ ghist) {
  __shared__ uint32_t scan_tmp[256];
  uint32_t *phist = ghist + (size_t)blockIdx.x * SORT_BUCKETS;
  const int tid = threadIdx.x;
  uint32_t local[8], sum = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    local[k] = sum;
    sum += phist[tid * 8 + k];
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
  for (int k = 0; k < 8; k++) phist[tid * 8 + k] = before + local[k];
}

// Kernel 1c: full set-up (V2..V5, S1..S6), one lane per visible triangle, all lanes busy; the record goes straight to its
// near-to-far position (scanned histogram + one atomic): record index == position in the sorted list from here on (bin /
// raster / fragment gather records by that index).  The order inside a bucket is whatever the atomics hand out (nothing
// depends on it).  A few workgroups per pose, each striding over the pose's list of visible triangles.
__global__ __launch_bounds__(256, RDOOM_SETUP_OCC) void setup_kernel(DeviceLevelView lv, const PoseConst *__restrict__ poses,
                                                    const ObjectConst *__restrict__ objects, uint32_t n_objects,
                                                    int width, int height, uint32_t kinds_mask,
                                                    const uint32_t *__restrict__ visible, TriRec *__restrict__ recs,
                                                    const uint32_t *__restrict__ counts,
                                                    uint32_t *__restrict__ ghist, uint32_t cap,
                                                    uint32_t *__restrict__ mismatch_flag, uint32_t place_groups) {
  const uint32_t pose = blockIdx.x / place_groups, pgroup = blockIdx.x - pose * place_groups, n = counts[pose];
  const PoseConst &pc = poses[pose];
  const LevelSlice &ls = lv.slices[pc.level];
  const ObjectConst *objs = objects ? objects + (size_t)pose * n_objects : nullptr;
  const uint32_t *pvisible = visible + (size_t)pose * cap;
  TriRec *prec = recs + (size_t)pose * cap;
  uint32_t *phist = ghist + (size_t)pose * SORT_BUCKETS;
  // positions: scanned histogram + rank.  The ranks of a chunk of 256 records are counted in LDS and each bucket that
  // occurs claims its run with ONE global atomic (a returning global atomic per record serialises on the few buckets a
  // pose's triangles crowd into)
  __shared__ uint32_t lcount[SORT_BUCKETS], lbase[SORT_BUCKETS];
  __shared__ uint4 wstage[4][32 * 8];  // per wave: 32 records of 128 bytes on their way to memory (below)
  if (pgroup * 256u >= n) return;  // uniform: nothing for this workgroup
  for (uint32_t i = threadIdx.x; i < SORT_BUCKETS; i += 256u) lcount[i] = 0;
  __syncthreads();
  for (uint32_t i0 = pgroup * 256u; i0 < n; i0 += place_groups * 256u) {  // uniform per workgroup
    const uint32_t i = i0 + threadIdx.x;
    const bool valid = i < n;
    TriRec rec;
    uint32_t bucket = 0, lrank = 0;
    if (valid) {
      const uint32_t t = pvisible[i];
      const LevelTri tri = lv.tris[t];
      float wkey;
      // visible: the cull kernel said so, with the same operations (-ffp-contract=off, every deciding product an explicit
      // fmaf), and counted this bucket.  Should a build ever break that, the histogram's runs no longer add up: flagged,
      // and reported by rdoom_batch_finish / the read functions instead of drawing from overwritten records.
      if (!setup_triangle(ls, pc, objs, t, tri, width, height, kinds_mask, rec.r, rec.s, wkey)) *mismatch_flag = 1u;
      bucket = depth_bucket(wkey);
#ifndef RDOOM_NO_EMPTY_CULL
      // A triangle whose bbox holds at most 3 x 3 pixel centres and covers none of them (the rasteriser's own edge
      // functions, operation order and fill rule: R1, R2) draws nothing: its bbox is made empty, so the binning kernel
      // lists it in no tile (far geometry at small frame sizes: most of the visible triangles are of this kind).
      {
        const int bx0 = (int)(rec.r.bb0 & 0xFFFFu), by0 = (int)(rec.r.bb0 >> 16), bx1 = (int)(rec.r.bb1 & 0xFFFFu),
                  by1 = (int)(rec.r.bb1 >> 16);
        if (bx1 - bx0 <= 2 && by1 - by0 <= 2) {
          const bool tl0 = (rec.r.flags & (1u << 24)) != 0u, tl1 = (rec.r.flags & (1u << 25)) != 0u, tl2 = (rec.r.flags & (1u << 26)) != 0u;
          bool any = false;
#pragma unroll
          for (int iy = 0; iy < 3; iy++) {
            const float py = (float)(by0 + iy) + 0.5f;
            const float t0 = fmaf(rec.r.e[1], py, rec.r.e[2]), t1 = fmaf(rec.r.e[4], py, rec.r.e[5]), t2 = fmaf(rec.r.e[7], py, rec.r.e[8]);
#pragma unroll
            for (int ix = 0; ix < 3; ix++) {
              const float px = (float)(bx0 + ix) + 0.5f;
              const float e0 = fmaf(rec.r.e[0], px, t0), e1 = fmaf(rec.r.e[3], px, t1), e2 = fmaf(rec.r.e[6], px, t2);
              const bool in = ((e0 > 0.0f) | ((e0 == 0.0f) & tl0)) & ((e1 > 0.0f) | ((e1 == 0.0f) & tl1)) &
                              ((e2 > 0.0f) | ((e2 == 0.0f) & tl2));
              any |= in & (bx0 + ix <= bx1) & (by0 + iy <= by1);
            }
          }
          if (!any) rec.r.bb0 = 0x0000FFFFu, rec.r.bb1 = 0u;  // x0 = 65535 > x1 = 0
        }
      }
#endif
      lrank = atomicAdd(&lcount[bucket], 1u);
    }
    __syncthreads();
    if (valid && lrank == 0u) lbase[bucket] = atomicAdd(&phist[bucket], lcount[bucket]);
    __syncthreads();
    // The record goes to its near-to-far position as ONE 128-byte line written by EIGHT lanes (round 6): a lane storing its own
    // record issued eight 16-byte stores, each of a wave's store instructions touching 64 different lines.  Half a wave at a time
    // parks its 32 records in the wave's LDS stage (4 KiB; 16-byte words swizzled so that neither side conflicts), then lane l
    // of the whole wave stores word l & 7 of records (l >> 3) + 8 j: a store instruction writes eight whole lines.
    const uint32_t pos = valid ? lbase[bucket] + lrank : NONE;
    {
      const uint32_t lane = threadIdx.x & 63u, word = lane & 7u;
      uint4 *stage = wstage[threadIdx.x >> 6];
      const uint4 *v = reinterpret_cast<const uint4 *>(&rec);
#pragma unroll
      for (uint32_t h = 0; h < 2u; h++) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the stage's previous readers are done
        if (valid && (lane >> 5) == h) {
          const uint32_t r = lane & 31u;
#pragma unroll
          for (uint32_t k = 0; k < 8u; k++) stage[r * 8u + (k ^ ((r >> 1) & 7u))] = v[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
          const uint32_t r = (lane >> 3) + 8u * j;
          const uint32_t pos_r = (uint32_t)__shfl((int)pos, (int)(32u * h + r));
          if (pos_r != NONE) reinterpret_cast<uint4 *>(&prec[pos_r])[word] = stage[r * 8u + (word ^ ((r >> 1) & 7u))];
        }
      }
    }
    __syncthreads();
    if (valid && lrank == 0u) lcount[bucket] = 0u;  // ready for the next chunk (its atomics follow the barrier below)
    __syncthreads();
  }
}

}  // namespace

rdoom_status launch_setup(hipStream_t st, uint32_t n_poses, const DeviceLevelView &lv, const PoseConst *poses,
                          const ObjectConst *objects, uint32_t n_objects, int width, int height, uint32_t kinds_mask,
                          TriRec *recs, uint32_t *visible, uint32_t *counts, uint32_t *ghist, uint32_t cap,
                          uint32_t *mismatch_flag) {
  // (counts and ghist arrive zeroed: the caller clears them together with its other per-render words in one fill)
  // several workgroups per pose on large levels (each takes a share of the clusters): one would walk them serially
  // (of a set of levels: by the largest; a smaller level's workgroups take shorter shares)
  const uint32_t groups = std::min<uint32_t>(std::max<uint32_t>(lv.max_clusters / 96u, 1u), 16u);
  (void)hipGetLastError();  // (a stale error of an earlier, unrelated call must not be blamed on these launches)
  hipLaunchKernelGGL(cull_kernel, dim3(n_poses * groups), dim3(256), 0, st, lv, poses, objects, n_objects, width, height,
                     kinds_mask, visible, counts, ghist, cap, groups);
  hipLaunchKernelGGL(sort_scan_kernel, dim3(n_poses), dim3(256), 0, st, ghist);
  const uint32_t place_groups = std::min<uint32_t>((cap + 1023u) / 1024u, 16u);  // about a fifth of a level is visible: one or two chunks of 256 records each
  hipLaunchKernelGGL(setup_kernel, dim3(n_poses * place_groups), dim3(256), 0, st, lv, poses, objects, n_objects, width, height,
                     kinds_mask, visible, recs, counts, ghist, cap, mismatch_flag, place_groups);
  // a launch that failed (an invalid configuration) must not go unreported: the binning kernel would build its lists from stale
  // records and counts and the render would still return RDOOM_OK
  HIP_TRY(hipGetLastError());
  return RDOOM_OK;
}

size_t setup_histogram_bytes(uint32_t max_poses) { return sizeof(uint32_t) * SORT_BUCKETS * (size_t)max_poses; }

}  // namespace rdoom_dev
