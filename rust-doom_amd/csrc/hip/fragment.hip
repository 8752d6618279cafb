// Kernels 3 + 4: fragment kernel (visibility -> palette index) and the alpha-leak fixup.
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
// Kernel 3: fragment kernel (F1..F6): visibility record -> atlas texel -> COLORMAP row -> 8-bit
// palette index.  One lane per run of 8 (or 4) horizontally adjacent pixels: the record from the rasteriser's
// quadrant table where it has an entry (no visibility words exist there), else one 16-byte visibility
// load; one 8-byte packed store; a wavefront takes a block of 8 runs x 8 rows per iteration (64 x 8 pixels: its
// texel gathers and record loads then fall into a compact patch).  COLORMAP (8 KiB) is staged in LDS once per
// workgroup, whose four waves walk FRAG_CHUNK such blocks each (all workgroups of a pose run on one XCD).
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
// A wave whose 512 pixels all show the same triangle takes the record by scalar loads (SGPR operands) -- the same
// body, instantiated a second time.  A run of sky is shaded from per-batch ndc tables.  A run that fails any precondition (mixed triangles,
// decor, rw outside the verified range, an uncertified mod, a transparent texel) is appended, quad by quad,
// to a per-wave LDS list and shaded afterwards by the general per-pixel body, lane per pixel -- same
// results, one code path for everything unusual.
// =================================================================================================
#ifdef RDOOM_FRAG_STATS  // census build (tools/variant.sh fstats fragment -DRDOOM_FRAG_STATS): where do the runs go?
__device__ unsigned long long g_frag_stats[16];
#endif
#ifndef RDOOM_FRAG_MAGIC
#define RDOOM_FRAG_MAGIC 0  // texel addresses through the round-down magic-number floor (0: four v_cvt_flr_i32_f32 per pixel pair)
#endif
#ifndef RDOOM_FRAG_ADDR2
#define RDOOM_FRAG_ADDR2 1  // texel byte offsets from scaled coordinates (two and-s and an or per texel)
#endif
#ifndef RDOOM_FRAG_CHUNK
#define RDOOM_FRAG_CHUNK 16
#endif
constexpr int FRAG_CHUNK = RDOOM_FRAG_CHUNK;
#ifndef RDOOM_FRAG_WAVES
#define RDOOM_FRAG_WAVES 4
#endif
constexpr uint32_t FRAG_WAVES = RDOOM_FRAG_WAVES;  // waves per workgroup (they share the LDS copy of COLORMAP)
typedef uint32_t TexelWord __attribute__((aligned(2)));
#ifndef RDOOM_FRAG_OCC
#define RDOOM_FRAG_OCC 6  // waves per SIMD the register allocation must allow: at most 80 VGPRs
#endif
#define FRAG_OCCUPANCY __attribute__((amdgpu_waves_per_eu(RDOOM_FRAG_OCC, 8)))
constexpr int FRAG_WLIST = 160;  // per-wave list of unfinished quads: at most 15 carried over + 64 x 2 new

__device__ __forceinline__ uint32_t shade_sky(const LevelSlice &lv, const uint16_t *__restrict__ sky_texels, const uint8_t *cmap, float px, float py,
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
  const uint32_t texel = sky_texels[(size_t)lv.sky_base + (size_t)iy * lv.sky_w + (size_t)ix];
  return cmap[texel & 0xFFu];
}

// returns the palette index, or 0x100 | index when the winning wall fragment's texel is transparent (the
// rasteriser treated a border-masked texture as opaque and the coordinate leaked onto the ring).
// IN_RANGE: the caller guarantees 2^-100 <= rw <= 2^100 (or a sky record): the divisions by rw, dist + 0.9 and dist + 1
// then take fastmath.hpp's exhaustively verified reciprocal forms -- same bits, a third of the instructions.
template <bool IN_RANGE>
__device__ __forceinline__ uint32_t shade_pixel(const DeviceLevelView &lv, const uint8_t *cmap, const ShadeRec &s,
                                                float px, float py, float row_w, float row_u, float row_v,
                                                int width, int height, const PoseConst &pc) {
  const uint32_t kind = s.flags & 3u;
  if (kind == RDOOM_KIND_SKY) return shade_sky(lv.slices[pc.level], lv.sky_texels, cmap, px, py, width, height, s.atlas_u, s.atlas_v);
  const TexelAt t = texel_coords<IN_RANGE>(s, px, row_w, row_u, row_v);
  const uint32_t texel = load_texel(lv.texels, s, t.ix, t.iy);
  if (kind != RDOOM_KIND_FLAT && (texel & 0x8000u)) return 0x100u;
  float light;
  if (kind == RDOOM_KIND_DECOR) {  // sprite.frag:22-25: DIST_SCALE = 1, light = min(v_light, 2 v_light - dist_term)
    const float q = IN_RANGE ? exact_rcp(t.dist + 1.0f) : 1.0f / (t.dist + 1.0f);
    const float dist_term = fminf(1.0f, 1.0f - q);
    light = fminf(s.light, s.light * 2.0f - dist_term);
  } else {
    const float q = IN_RANGE ? exact_div09(t.dist + 0.9f) : 0.9f / (t.dist + 0.9f);
    const float dist_term = fminf(1.0f, 1.0f - q);
    light = s.light * 2.0f - dist_term;
  }
  const float tt = (1.0f - light) * 32.0f;
  const int rowc = tt < 0.0f ? 0 : (tt >= 32.0f ? 31 : (int)floorf(tt));
  return cmap[rowc * 256 + (int)(texel & 0xFFu)];
}

// What only the kernel's rare paths read (the general per-pixel body, sky runs, the alpha-leak queue), in device memory
// rather than in the kernel arguments: kernel arguments are scalar registers for the whole kernel, and the hot loop has
// none to spare (106 of 106 in use).  One per batch, written once (launch_fragment).
struct FragConst {
  DeviceLevelView lv;
  uint32_t *fix_count;
  uint2 *fix_list;
  const float *ndc_tab;
  uint32_t fix_cap, div_m, div_sh, pad;
};

template <int NQ, int DBG, bool VIS16>  // NQ: adjacent quads per lane (1 or 2; the frame width is a multiple of 4 NQ);
                                        // DBG: timing experiments only; VIS16: 16-bit visibility words (0xFFFF = none)
__global__ __launch_bounds__(64 * RDOOM_FRAG_WAVES) FRAG_OCCUPANCY void fragment_kernel(const FragConst *__restrict__ fc,
                                                       const uint16_t *__restrict__ texels, const uint8_t *__restrict__ colormap,
                                                       const TriRec *__restrict__ recs,
                                                       uint32_t cap, const PoseConst *__restrict__ poses,
                                                       const uint32_t *__restrict__ vis, uint32_t n_poses,
                                                       uint32_t chunks_per_pose, uint32_t chunk_iters,
                                                       uint32_t quads_per_pose, uint32_t quads_per_row,
                                                       uint32_t wblocks_per_row, uint32_t wblocks_per_pose, uint32_t bw_log2,
                                                       int width, int height,
                                                       uint8_t *__restrict__ fb,
                                                       uint32_t debug_leak_mod, const uint32_t *__restrict__ qtab,
                                                       uint32_t qtab_mode, uint32_t tiles_x, uint32_t n_tiles) {
  constexpr int NP = 2 * NQ, NPX = 4 * NQ;  // float2 pairs and pixels per lane
  __shared__ uint8_t cmap[32 * 256];
  __shared__ uint32_t wlist[FRAG_WAVES][FRAG_WLIST];
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(colormap);
    uint4 *dst = reinterpret_cast<uint4 *>(cmap);
#pragma unroll
    for (uint32_t k = threadIdx.x; k < 512u; k += 64u * FRAG_WAVES) dst[k] = src[k];
  }
  __syncthreads();
  // blockIdx -> (pose, chunk): all chunks of a pose on one XCD (b % 8), like the rasteriser
#ifdef RDOOM_FRAG_CHUNKS_OUTER
  // (round 5 experiment, NOT the default: the rasteriser gained 11 % from dispatching its heavy tile rows first, raster.hip; the
  // same order here -- a two-dimensional grid, the frame's chunks the slow dimension, the middle ones first -- measured 5 %
  // SLOWER at 1080p and 9 % at 4K: a pose's visibility words and records are no longer walked while they are in its XCD's L2)
  const uint32_t pose = blockIdx.x;
  const uint32_t yk = blockIdx.y, yc = chunks_per_pose >> 1;
  const uint32_t chunk = (yk & 1u) ? yc - ((yk + 1u) >> 1) : yc + (yk >> 1);  // c, c - 1, c + 1, c - 2, ...
#else
  const uint32_t g = blockIdx.x >> 3;
  const uint32_t pose = (g / chunks_per_pose) * 8u + (blockIdx.x & 7u);
  const uint32_t chunk = g % chunks_per_pose;
#endif
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
      const DeviceLevelView &lv = fc->lv;  // (read here, on the rare path, not held in registers through the hot loop)
      const uint32_t div_m = fc->div_m, div_sh = fc->div_sh;
      const uint32_t qi = mylist[first + j];
      const uint32_t row = fast_div(qi, div_m, div_sh), qx = qi - row * quads_per_row;
      // the record of my pixel: from the quadrant table where it has an entry (the rasteriser then wrote no visibility
      // words for that quadrant), else the pixel's visibility word
      uint32_t id = NONE;
      if (qtab_mode != 0u)
        id = qtab_record(qtab[((size_t)pose * n_tiles + ((row >> 6) * tiles_x + (qx >> 4))) * 4u + ((row >> 5) & 1u) * 2u + ((qx >> 3) & 1u)]);
      if (id == NONE) id = VIS16 ? (uint32_t)pvis16[(size_t)qi * 4u + k] : pvis32[(size_t)qi * 4u + k];
      uint32_t c = 0;
      const uint32_t pix = (row * quads_per_row + qx) * 4u + k;
      if (id != NONE_ID) {
        const ShadeRec cur = prec[id].s;
        const float py = (float)row + 0.5f, px = (float)(qx * 4u + k) + 0.5f;
        const float row_w = fmaf(cur.wp[1], py, cur.wp[2]), rw = fmaf(cur.wp[0], px, row_w);
        const bool in_range = ((cur.flags & 3u) == RDOOM_KIND_SKY) | ((rw >= 0x1p-100f) & (rw <= 0x1p100f));
        if (__all(in_range))  // (wave-uniform choice: one copy of the body runs)
          c = shade_pixel<true>(lv, cmap, cur, px, py, row_w, fmaf(cur.up[1], py, cur.up[2]), fmaf(cur.vp[1], py, cur.vp[2]),
                                width, height, pc);
        else
          c = shade_pixel<false>(lv, cmap, cur, px, py, row_w, fmaf(cur.up[1], py, cur.up[2]), fmaf(cur.vp[1], py, cur.vp[2]),
                                 width, height, pc);
        // debug_leak_mod != 0 (tests only): pretend every n-th pixel leaked, so fixup_kernel's general rule
        // is exercised on ordinary pixels too -- the output must not change
        const bool forced = debug_leak_mod != 0u && pix % debug_leak_mod == 0u;
        if ((c & 0x100u) || forced) {  // rare: alpha leak, queue the pixel for exact re-resolution
          const uint32_t slot = atomicAdd(fc->fix_count, 1u);
          if (slot < fc->fix_cap) fc->fix_list[slot] = make_uint2(pose, pix);
        }
      }
      uint32_t v = (c & 0xFFu) << (8u * k);
      v |= __shfl_xor(v, 1);
      v |= __shfl_xor(v, 2);
      if (k == 0) pfb[qi] = v;
    }
  };
  // A unit = the NQ adjacent quads of one lane.  A wavefront takes a block of 8 units x 8 rows (64 x 8 pixels for
  // NQ = 2) per iteration rather than 64 units of one row: the texels (and records) it gathers then come from a compact
  // patch of texture space -- a few cache lines per load instruction instead of one per lane on floors and ceilings.
  const uint32_t units_per_row = quads_per_row / (uint32_t)NQ;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // the lane's share of col / row / quad index is loop-invariant; the block's share is scalar arithmetic
  const uint32_t lane_col = lane & ((1u << bw_log2) - 1u), lane_row = lane >> bw_log2;
  uint32_t lane_q = lane_row * quads_per_row + lane_col * (uint32_t)NQ;
  asm("" : "+v"(lane_q));  // (opaque: otherwise the product with the row is re-derived inside the loop)
  // byte offsets from the pose's visibility words / framebuffer fit 32 bits (quads_per_pose < 2^24): one VGPR each
  const char *pvis_bytes = VIS16 ? reinterpret_cast<const char *>(pvis16) : reinterpret_cast<const char *>(pvis32);
  char *pfb_bytes = reinterpret_cast<char *>(pfb);
  for (uint32_t it = 0; it < chunk_iters; it++) {
    const uint32_t wb = (chunk * chunk_iters + it) * FRAG_WAVES + wave;
    if (wb >= wblocks_per_pose) break;  // wave-uniform: past the end of the frame
    const uint32_t wby = wb / wblocks_per_row, wbx = wb - wby * wblocks_per_row;
    const uint32_t col = (wbx << bw_log2) + lane_col, row = (wby << (6u - bw_log2)) + lane_row;
    const bool valid = (col < units_per_row) & (row < (uint32_t)height);
    const uint32_t qx = col * (uint32_t)NQ;
    const uint32_t q0 = ((wby << (6u - bw_log2)) * quads_per_row + (wbx << bw_log2) * (uint32_t)NQ) + lane_q;
    // visibility words of my NPX pixels: all the same?  (compared as loaded, two 16-bit words at a time; a lane outside
    // the frame reads unit 0 and is treated as background -- no divergent branch, no boolean phi)
    const uint32_t q0l = valid ? q0 : 0u;
    // The rasteriser's quadrant table: when it says that every pixel of the 32 x 32 quadrant(s) this block lies in shows
    // ONE record, the block's visibility words are neither loaded nor compared (all scalar: the block origin is uniform).
    // qtab_mode 1: the block is 32 pixels wide (inside one quadrant); 2: 64 pixels wide (two quadrants of one tile, side
    // by side -- the right one may lie outside the frame, where the rasteriser writes nothing).
    // A quadrant with an entry HAS no visibility words (the rasteriser leaves them out, raster.hip SKIPVIS): wherever an
    // entry exists it is the only source.  Two quadrants with different entries, or one with and one without: each lane takes
    // its own quadrant's entry (tl), and only lanes without one load visibility words.
    uint32_t tq = NONE;
    uint32_t tl = NONE;  // per lane: the entry of the quadrant my pixels lie in, when the block spans two that differ
    if (qtab_mode != 0u) {
      const uint32_t bx0 = (wbx << bw_log2) * (uint32_t)NPX, by0 = wby << (6u - bw_log2);
      const uint32_t *e = qtab + ((size_t)pose * n_tiles + ((by0 >> 6) * tiles_x + (bx0 >> 6))) * 4u + ((by0 >> 5) & 1u) * 2u;
      // A quadrant fragment_quadrant_kernel has shaded carries QTAB_HANDLED: a block that lies in handled quadrants only is
      // done (wave-uniform: the entries arrive by scalar loads).  A block with one handled half shades that half again from
      // the entry like any other described quadrant -- same record, same operations, same bytes.
      if (qtab_mode == 1u) {
        const uint32_t one = e[(bx0 >> 5) & 1u];
#ifdef RDOOM_FRAG_STATS
        if (lane == 0u) atomicAdd(&g_frag_stats[qtab_handled(one) ? 1 : 0], 1ull);
#endif
        if (qtab_handled(one)) continue;
        tq = one;
      } else {
        const uint2 two = *reinterpret_cast<const uint2 *>(e);
        const bool right_out = bx0 + 32u >= (uint32_t)width;
#ifdef RDOOM_FRAG_STATS
        if (lane == 0u) {
          atomicAdd(&g_frag_stats[(qtab_handled(two.x) & (right_out | qtab_handled(two.y))) ? 1 : 0], 1ull);
          if (qtab_handled(two.x) != (right_out | qtab_handled(two.y))) atomicAdd(&g_frag_stats[2], 1ull);
        }
#endif
        if (qtab_handled(two.x) & (right_out | qtab_handled(two.y))) continue;
        const uint32_t left = qtab_record(two.x), right = qtab_record(two.y);
        const bool same = left == right || right_out;
        tq = same ? left : NONE;
#ifndef RDOOM_FRAG_NO_TL  // (A/B builds only: without the per-lane entries every visibility word must be there -- keep_vis)
        tl = same ? NONE : (((lane_col * (uint32_t)NPX) & 32u) ? right : left);
#endif
      }
    }
    tq = (uint32_t)__builtin_amdgcn_readfirstlane((int)tq);
    const bool table_one = (tq != NONE) & (debug_leak_mod == 0u);
    uint32_t id0;
    bool uniform;
    if (table_one) {
      id0 = tq;
      uniform = true;
    } else if ((tl != NONE) & (debug_leak_mod == 0u)) {  // (per lane: my quadrant has an entry, the other one has another or none;
                                                         // under leak_mod every pixel goes through its visibility word: all are there)
      id0 = tl;
      uniform = true;
    } else if (VIS16) {
      if (NQ == 2) {
        const uint4 v = *reinterpret_cast<const uint4 *>(pvis_bytes + q0l * 8u);
        id0 = v.x & 0xFFFFu;
        uniform = (((v.x ^ __builtin_amdgcn_alignbit(v.x, v.x, 16)) | (v.x ^ v.y)) | ((v.y ^ v.z) | (v.z ^ v.w))) == 0u;
      } else {
        const uint2 v = *reinterpret_cast<const uint2 *>(pvis_bytes + q0l * 8u);
        id0 = v.x & 0xFFFFu;
        uniform = ((v.x ^ __builtin_amdgcn_alignbit(v.x, v.x, 16)) | (v.x ^ v.y)) == 0u;
      }
    } else {
      const uint4 v = *reinterpret_cast<const uint4 *>(pvis_bytes + q0l * 16u);
      id0 = v.x;
      uint32_t diff = (v.x ^ v.y) | (v.y ^ v.z) | (v.z ^ v.w);
      if (NQ == 2) {
        const uint4 w = *reinterpret_cast<const uint4 *>(pvis_bytes + (q0l + 1u) * 16u);
        diff |= (w.x ^ id0) | (w.x ^ w.y) | (w.y ^ w.z) | (w.z ^ w.w);
      }
      uniform = diff == 0u;
    }
    id0 = valid ? id0 : NONE_ID;
    uniform |= !valid;
    bool done = false;
    uint32_t out[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) out[q] = 0;
    if (uniform & (id0 == NONE_ID)) done = true;  // background (or past the end: nothing is stored)
    // The packed body for a run that lies in ONE flat / wall triangle with SHADE_FAST.  ONE = the whole wave holds the
    // same record (77 % of the waves of the 1080p E1M1 sweep): its words then arrive by scalar loads and are SGPR
    // operands of the packed instructions -- no per-lane record loads, no register pairs to build for the splats.
    auto fast_run = [&](auto one, const uint4 r0, const uint4 r1, const uint4 r2, const uint4 r3)
                        __attribute__((always_inline)) {
      constexpr bool ONE = decltype(one)::value;
      const uint32_t flags = r3.z, tex = r3.w;
      const float wa = __uint_as_float(r0.x), wb = __uint_as_float(r0.y), wc = __uint_as_float(r0.z),
                  ua = __uint_as_float(r0.w), ub = __uint_as_float(r1.x), uc = __uint_as_float(r1.y),
                  va = __uint_as_float(r1.z), vb = __uint_as_float(r1.w), vc = __uint_as_float(r2.x),
                  atlas_u = __uint_as_float(r2.y), atlas_v = __uint_as_float(r2.z), size_x = __uint_as_float(r2.w),
                  size_y = __uint_as_float(r3.x), light = __uint_as_float(r3.y);
      const float py = (float)row + 0.5f;
      const float px0 = (float)(qx * 4u) + 0.5f;
      const float row_w = fmaf(wb, py, wc), row_u = fmaf(ub, py, uc), row_v = fmaf(vb, py, vc);
      // F2 preparation: q0 = t * RN(1/size) equals the quotient exactly for a power-of-two size; for an integer
      // size it is within |t/size| * 2^-23 of it, and the remainder test below certifies
      // floor(q0) == floor(RN(t / size)) (else the run goes to the general body).
      // (1 / 2^k is one integer subtraction on the exponent field; the reciprocal forms only run in waves that hold
      // a record with an integer, non-power-of-two size)
      const bool any_np2 = ONE ? (flags & SHADE_NP2) != 0u : __any((flags & SHADE_NP2) != 0u);
      f32x2 inv_s = {__uint_as_float(0x7F000000u - __float_as_uint(size_x)), __uint_as_float(0x7F000000u - __float_as_uint(size_y))};
      if (any_np2) inv_s = exact_rcp2(f32x2{size_x, size_y});
      // F3 parameters: one u16 texel store, REPEAT = masks
      const uint32_t wm = tex & 0xFFFFu, hm = tex >> 16, lw = (flags >> 8) & 15u, base = (flags >> 16) << 10;
      const uint32_t base2 = base * 2u;  // byte offsets < 2^27: one 32-bit VGPR offset from the uniform base pointer
      const char *tb = reinterpret_cast<const char *>(texels);
#if RDOOM_FRAG_ADDR2 && !RDOOM_FRAG_MAGIC
      const uint32_t wm2a = wm << 1, hms = hm << (lw + 1u);
      const float ysc = __uint_as_float((128u + lw) << 23);  // 2 W = 2^(lw + 1)
      const float au2 = atlas_u * 2.0f, avs = atlas_v * ysc;
      const char *tba = ONE ? tb + base2 : tb;  // (wave-uniform record: the store's base is scalar pointer arithmetic)
#endif
#if RDOOM_FRAG_MAGIC
      const uint32_t wm2 = wm << 1;
      // the store's base: with a wave-uniform record the pointer arithmetic is scalar; per-lane records keep a 32-bit offset
      const char *tb2 = ONE ? tb + base2 : tb;
      const uint32_t lane_base2 = ONE ? 0u : base2;
#endif
      // Certificate for integer (non-power-of-two) tile sizes, evaluated only in waves that hold such a record
      // (fastmath.hpp, mod_cert): with guard >= 2^-20 * max(|x|, y), guard <= r <= y - guard and |x| < 2^23 imply that
      // no integer lies between x * RN(1/y) and RN(x / y) and that y * floor is exact.  One guard per run and axis:
      // |x_k| = |n_k * w_k| <= max(|n_first|, |n_last|) * max(w_first, w_last) because the numerator plane n and, for
      // rw > 0, w = 1/rw are monotone along the run (and rounding is monotone).  The run's guard is at least every
      // pixel's own guard, so passing here implies mod_cert() for each pixel -- the form the on-device self-test sweeps.
      // Power-of-two axes always pass.
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
      float w_first = 0.0f, w_last = 0.0f;  // 1/rw at the two ends of the run
      uint32_t texel[NPX], any_texel = 0;
      float rw_first = 0.0f, rw_last = 0.0f;
#pragma unroll
      for (int p = 0; p < NP; p++) {  // one pair of pixels at a time, straight through to its two texel loads
        const f32x2 px = {px0 + (float)(2 * p), px0 + (float)(2 * p + 1)};
        const f32x2 rw = pk_fma(splat(wa), px, splat(row_w));  // F1
        if (p == 0) rw_first = rw.x;
        if (p == NP - 1) rw_last = rw.y;
        const f32x2 w = exact_rcp2(rw);
        if (p == 0) w_first = w.x;
        if (p == NP - 1) w_last = w.y;
        const f32x2 tu = pk_fma(splat(ua), px, splat(row_u)) * w;
        const f32x2 tv = pk_fma(splat(va), px, splat(row_v)) * w;
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
#if RDOOM_FRAG_MAGIC
        // floor() of four coordinates with two packed instructions: for 0 <= x < 2^22, x + 2^23 rounded TOWARDS MINUS
        // INFINITY is 2^23 + floor(x) exactly (the sum lies in [2^23, 2^24), where binary32 has unit spacing), i.e. the
        // bits 0x4B000000 + floor(x); the REPEAT masks below strip the exponent.  The horizontal coordinate goes through
        // fma(x, 2, 2^23): floor(2x) = 2 floor(x) + {0, 1}, and the mask (wm << 1) drops the odd bit -- the byte offset of
        // the 16-bit texel without a shift.  The rounding mode is switched for exactly these two instructions (one asm
        // statement: nothing can be scheduled in between).  Coordinates are >= 0 in every lane whose result is used
        // (mod results lie in [0, size], atlas positions are >= 0); other lanes produce masked, in-range garbage as before.
        f32x2 fxb, fyb;
        asm volatile(
            "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\t"
            "v_pk_fma_f32 %0, %2, 2.0, %4 op_sel_hi:[1,0,1]\n\t"
            "v_pk_add_f32 %1, %3, %4\n\t"
            "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
            : "=&v"(fxb), "=v"(fyb)
            : "v"(ux), "v"(uy), "s"(f32x2{0x1p23f, 0x1p23f}));
        const uint32_t o0 = ((__float_as_uint(fyb.x) & hm) << (lw + 1u)) | (__float_as_uint(fxb.x) & wm2);
        const uint32_t o1 = ((__float_as_uint(fyb.y) & hm) << (lw + 1u)) | (__float_as_uint(fxb.y) & wm2);
        texel[2 * p] = (DBG & 2) ? (o0 & 255u) : *reinterpret_cast<const TexelWord *>(tb2 + (ONE ? o0 : o0 + lane_base2));
        texel[2 * p + 1] = (DBG & 2) ? (o1 & 255u) : *reinterpret_cast<const TexelWord *>(tb2 + (ONE ? o1 : o1 + lane_base2));
#elif RDOOM_FRAG_ADDR2
        // the texel's BYTE offset from two and-s and an or (as fragment_quadrant_kernel forms it): the atlas origin is added
        // with fma(r, 2, 2 atlas_u) = 2 RN(r + atlas_u) and fma(r, 2 W, 2 W atlas_v) = 2 W RN(r + atlas_v) -- scaling by a power
        // of two commutes with the rounding --, floor of those is 2 floor(x) + {0, 1} and 2 W floor(y) + {0 .. 2 W - 1}, and the
        // masks (W - 1) << 1 and (H - 1) << (log2 W + 1) drop exactly the surplus bits (coordinates are >= 0)
        (void)ux, (void)uy;
        const f32x2 ux2 = pk_fma(rx, splat(2.0f), splat(au2)), uys = pk_fma(ry, splat(ysc), splat(avs));
        const uint32_t b0 = ((uint32_t)cvt_floor_i32(uys.x) & hms) | ((uint32_t)cvt_floor_i32(ux2.x) & wm2a);
        const uint32_t b1 = ((uint32_t)cvt_floor_i32(uys.y) & hms) | ((uint32_t)cvt_floor_i32(ux2.y) & wm2a);
        texel[2 * p] = (DBG & 2) ? (b0 & 255u) : *reinterpret_cast<const TexelWord *>(tba + (ONE ? b0 : b0 + base2));
        texel[2 * p + 1] = (DBG & 2) ? (b1 & 255u) : *reinterpret_cast<const TexelWord *>(tba + (ONE ? b1 : b1 + base2));
#else
        const uint32_t o0 = (((uint32_t)cvt_floor_i32(uy.x) & hm) << lw) | ((uint32_t)cvt_floor_i32(ux.x) & wm);
        const uint32_t o1 = (((uint32_t)cvt_floor_i32(uy.y) & hm) << lw) | ((uint32_t)cvt_floor_i32(ux.y) & wm);
        // (a 32-bit load at the texel's 2-byte-aligned address: bits 0..7 = palette index and bit 15 = transparent are
        // all that is read from it, the upper half is the next texel -- the array ends with a padding texel.  A 16-bit
        // load would be zero-extended again wherever it is used in another basic block: eight more instructions per run)
        texel[2 * p] = (DBG & 2) ? (o0 & 255u) : *reinterpret_cast<const TexelWord *>(tb + (o0 * 2u + base2));
        texel[2 * p + 1] = (DBG & 2) ? (o1 & 255u) : *reinterpret_cast<const TexelWord *>(tb + (o1 * 2u + base2));
#endif
      }
      // rw is monotone along the run: both ends inside the verified range of the exact reciprocal forms
      // (one unsigned compare per end: the bit patterns of [2^-100, 2^100] are the integers [0x0D800000, 0x71800000];
      // negative numbers and NaNs land above the interval)
      const uint32_t off_first = __float_as_uint(rw_first) - 0x0D800000u, off_last = __float_as_uint(rw_last) - 0x0D800000u;
      const bool in_range = max(off_first, off_last) <= 0x71800000u - 0x0D800000u;
#pragma unroll
      for (int k = 0; k < NPX; k++) any_texel |= texel[k];
      // F4, F5 at the two end pixels; the pixels between them only when the ends disagree
      auto rows_of = [&](f32x2 dist) {
        const f32x2 dterm = splat(1.0f) - exact_div09_2(dist + splat(0.9f));
        const f32x2 lgt = splat(light * 2.0f) - f32x2{fminf(1.0f, dterm.x), fminf(1.0f, dterm.y)};
        const f32x2 tt = (splat(1.0f) - lgt) * splat(32.0f);
        return f32x2{fminf(fmaxf(floorf(tt.x), 0.0f), 31.0f), fminf(fmaxf(floorf(tt.y), 0.0f), 31.0f)};
      };
      const f32x2 rf_ends = rows_of(f32x2{w_first, w_last});
      uint32_t ci[NPX];  // COLORMAP index = row * 256 + texel
#ifdef RDOOM_FRAG_STATS
      {  // census: how often does the per-pixel row evaluation below run (it runs for the whole wave when one run needs it)?
        const unsigned long long dm = __ballot(rf_ends.x != rf_ends.y);
        if (lane == 0u) atomicAdd(&g_frag_stats[14], 1ull);
        if (lane == 0u && dm) atomicAdd(&g_frag_stats[15], 1ull), atomicAdd(&g_frag_stats[2], (unsigned long long)__popcll(dm));
      }
#endif
      if (rf_ends.x == rf_ends.y) {
        const uint32_t r8 = (uint32_t)(int)rf_ends.x << 8;
#pragma unroll
        for (int k = 0; k < NPX; k++) ci[k] = r8 | (texel[k] & 0xFFu);
      } else {
#pragma unroll
        for (int p = 0; p < NP; p++) {
          // (the ends disagree -- a few runs in a hundred: 1/rw of the pixels between them is evaluated again here
          // rather than kept in eight registers through the whole body)
          const f32x2 px = {px0 + (float)(2 * p), px0 + (float)(2 * p + 1)};
          const f32x2 rows = rows_of(exact_rcp2(pk_fma(splat(wa), px, splat(row_w))));
          ci[2 * p] = ((uint32_t)(int)rows.x << 8) | (texel[2 * p] & 0xFFu);
          ci[2 * p + 1] = ((uint32_t)(int)rows.y << 8) | (texel[2 * p + 1] & 0xFFu);
        }
      }
      const bool opaque = (any_texel & 0x8000u) == 0u;
#ifdef RDOOM_FRAG_STATS
      if (!in_range) atomicAdd(&g_frag_stats[11], 1ull);
      if (!mod_ok) atomicAdd(&g_frag_stats[12], 1ull);
      if (!opaque) atomicAdd(&g_frag_stats[13], 1ull);
#endif
      if (in_range & mod_ok & opaque) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
          const uint32_t c0 = (DBG & 8) ? (ci[2 * p] & 0xFFu) : cmap[ci[2 * p]], c1 = (DBG & 8) ? (ci[2 * p + 1] & 0xFFu) : cmap[ci[2 * p + 1]];
          out[p >> 1] |= (c0 | (c1 << 8)) << (16 * (p & 1));
        }
        done = true;
      }
    };
    // wave-uniform record?  (the broadcast is an unconditional initialiser, see DESIGN 5 on v_readlane under branches)
    // (with the table a lane outside the frame may ride along: it stores nothing, its texel offsets are masked into range)
    const uint32_t id_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)id0);
    const uint32_t id_one = table_one ? tq : id_first;
    const bool wave_one = table_one | (__all(uniform & valid & (id0 == id_one)) & (id_one != NONE_ID) & (debug_leak_mod == 0u));
    bool took_one = false;
    if (wave_one) {
      const uint4 *rp = reinterpret_cast<const uint4 *>(&prec[id_one].s);
      const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
      asm volatile("" ::"s"(r0.x), "s"(r1.x), "s"(r2.x));  // (all four scalar loads in flight before the flag is examined)
      if (r3.z & SHADE_FAST) {
        fast_run(std::true_type{}, r0, r1, r2, r3);
        took_one = true;
      }
    }
    if (!took_one & uniform & (id0 != NONE_ID) & (debug_leak_mod == 0u)) {
      const uint4 *rp = reinterpret_cast<const uint4 *>(&prec[id0].s);
      const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
      const uint32_t flags = r3.z;
      // (all four loads are issued before the flag is examined: one memory latency, not two)
      asm volatile("" ::"v"(r0.x), "v"(r1.x), "v"(r2.x));
      if (flags & SHADE_FAST) {
        fast_run(std::false_type{}, r0, r1, r2, r3);
      } else if ((flags & 3u) == RDOOM_KIND_SKY) {
        // a run of sky (sky.frag:12-26): the colour depends on the pixel and the pose only.  ndc_tab holds
        // p / (size / 2) - 1 for every column and row of the frame (computed once per batch with the same two
        // operations), the record carries v_r.y and 4 v_r.x / 3.14159265358; the row part is evaluated once per run.
        const LevelSlice &lv = fc->lv.slices[pc.level];  // the sky of this pose's level
        const float *ndc_tab = fc->ndc_tab;
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
        const uint16_t *srow = fc->lv.sky_texels + ((size_t)lv.sky_base + (size_t)iy * lv.sky_w);
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
    if (done & valid & (!(DBG & 4) || out[0] == 0x12345679u)) {  // (DBG & 4, timing experiment: practically never)
      if (DBG & 16) {  // timing experiment (wrong image): the block's 512 bytes as ONE contiguous run -- what would whole-line stores cost?
        *reinterpret_cast<uint2 *>(pfb_bytes + (size_t)wb * 512u + lane * 8u) = make_uint2(out[0], out[NQ - 1]);
      } else
      if (NQ == 2)
        *reinterpret_cast<uint2 *>(pfb_bytes + q0 * 4u) = make_uint2(out[0], out[NQ - 1]);
      else
        *reinterpret_cast<uint32_t *>(pfb_bytes + q0 * 4u) = out[0];
    }
#ifdef RDOOM_FRAG_STATS
    if (valid) atomicAdd(&g_frag_stats[8], 1ull);
    if (valid && !done) atomicAdd(&g_frag_stats[9], 1ull);
    if (valid && !uniform) atomicAdd(&g_frag_stats[10], 1ull);
#endif
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
// Kernel 3a: the whole-quadrant path.  73 % of the 32 x 32 quadrants of a 1080p sweep show ONE triangle and the rasteriser's
// quadrant table says which (raster.hip).  fragment_kernel's block loop spends more issue cycles AROUND the pixel arithmetic
// of such a block (block index -> position, table look-up, record loads, range / mod / opacity guards, visibility
// bookkeeping, COLORMAP rows at both ends of every run) than in it -- tools/isa_blocks.py: about 1 250 VALU cycles per
// 8-pixel run of which 560 are the four pixel pairs.  Here a wavefront takes a 64 x 64 tile, reads the tile's four table
// entries with one scalar load and shades every described quadrant whose record is SHADE_FAST as two half quadrants of
// 32 x 16 pixels, lane per 8-pixel run:
//   * the record arrives once per quadrant by scalar loads and everything that depends only on it is computed once:
//     reciprocal tile sizes, the texel-store window, masks, the scaled atlas origin; the body is instantiated per record
//     class (power-of-two sizes or certified integer mod; byte texel loads or 16-bit ones with the opacity test);
//   * 1/w is evaluated at the quadrant's four corner pixels: fmaf is monotone in each argument, so when the corners lie in
//     [2^-100, 2^100] every pixel's computed 1/w lies in the range where exact_rcp2 equals the division bit for bit (a
//     quadrant that fails is left to fragment_kernel);
//   * COLORMAP rows once per quadrant instead of twice per run: lane l evaluates F4 / F5 at one end of screen row l / 2 of the
//     quadrant; 1/w is monotone along a row, so equal rows at the two ends are the row of all 32 pixels between them; a lane
//     picks its row's value up with one ds_bpermute.  Only when some row's ends disagree do the runs evaluate their own ends
//     (and the pixels between them when those disagree), as fragment_kernel does;
//   * the texel's byte offset comes from two and-s and an or: the atlas origin is added with fma(r, 2, 2 atlas_u) =
//     2 RN(r + atlas_u) and fma(r, 2 W, 2 W atlas_v) = 2 W RN(r + atlas_v) (scaling by a power of two commutes with the
//     rounding), floor of those is 2 floor(x) + {0, 1} and 2 W floor(y) + {0 .. 2 W - 1}, and the masks (W - 1) << 1 and
//     (H - 1) << (log2 W + 1) drop exactly the surplus bits (coordinates are >= 0: a mod result plus an atlas position);
//   * what fragment_kernel sends to its general per-pixel body -- a run whose integer mod is not certified, a transparent
//     texel under a winner the rasteriser treated as opaque -- is queued for fixup_kernel here, pixel by pixel: the general
//     rule R1..R6 + plain IEEE division, same bytes (a handful of pixels per batch).
// Same operations on the same operands as fragment_kernel's packed body (F1..F6), hence the same bits.  A quadrant shaded
// here gets QTAB_HANDLED in its table entry; fragment_kernel, launched next on the same stream, skips the blocks whose
// quadrants are handled and treats every other entry as before.
// =================================================================================================
#ifndef RDOOM_QUAD_TILES
#define RDOOM_QUAD_TILES 4
#endif
#ifndef RDOOM_QUAD_NP2
#define RDOOM_QUAD_NP2 1  // 0: quadrants whose record has a non-power-of-two tile size are left to fragment_kernel (two instantiations of the body fewer: 64 VGPRs)
#endif
#ifndef RDOOM_QUAD_OCC
#define RDOOM_QUAD_OCC 5  // waves per SIMD the register allocation must allow (96 VGPRs; 8 with RDOOM_QUAD_NP2 = 0: measured alike)
#endif
#define QUAD_OCCUPANCY __attribute__((amdgpu_waves_per_eu(RDOOM_QUAD_OCC, 8)))
constexpr uint32_t QUAD_TILES_PER_WAVE = RDOOM_QUAD_TILES;  // a workgroup = 4 waves x this many tiles shares one LDS copy of COLORMAP

__device__ __forceinline__ f32x2 colormap_rows(f32x2 dist, float light2) {  // F4, F5 for two pixels
  const f32x2 dterm = splat(1.0f) - exact_div09_2(dist + splat(0.9f));
  const f32x2 lgt = splat(light2) - f32x2{fminf(1.0f, dterm.x), fminf(1.0f, dterm.y)};
  const f32x2 tt = (splat(1.0f) - lgt) * splat(32.0f);
  return f32x2{fminf(fmaxf(floorf(tt.x), 0.0f), 31.0f), fminf(fmaxf(floorf(tt.y), 0.0f), 31.0f)};
}
__device__ __forceinline__ float colormap_row(float dist, float light2) {  // the same operations for one pixel
  const float dterm = 1.0f - exact_div09(dist + 0.9f);
  const float lgt = light2 - fminf(1.0f, dterm);
  const float tt = (1.0f - lgt) * 32.0f;
  return fminf(fmaxf(floorf(tt), 0.0f), 31.0f);
}

__device__ __forceinline__ void queue_fixup(const FragConst *fc, uint32_t pose, uint32_t pix) {
  const uint32_t slot = atomicAdd(fc->fix_count, 1u);
  if (slot < fc->fix_cap) fc->fix_list[slot] = make_uint2(pose, pix);  // (beyond the capacity: fixup_kernel raises the error flag)
}

// One described quadrant, two half quadrants of 32 x 16 pixels.  NP2: a tile size is an integer that is not a power of two
// (reciprocal by exact_rcp2, the floor certified per run as in fragment_kernel); MASKED: the texture has transparent texels
// in its rectangle or the ring around it (16-bit texel loads and the opacity test; otherwise the low byte alone is loaded).
template <bool NP2, bool MASKED>
__device__ __forceinline__ void shade_quadrant(const FragConst *__restrict__ fc, const uint8_t *cmap, const char *tb, char *pfb, uint32_t pose,
                                               const uint4 r0, const uint4 r1, const uint4 r2, const uint4 r3, uint32_t qx0, uint32_t qy0,
                                               uint32_t lane, int width, int height) {
  const uint32_t flags = r3.z, tex = r3.w;
  const float wa = __uint_as_float(r0.x), wb = __uint_as_float(r0.y), wc = __uint_as_float(r0.z),
              ua = __uint_as_float(r0.w), ub = __uint_as_float(r1.x), uc = __uint_as_float(r1.y),
              va = __uint_as_float(r1.z), vb = __uint_as_float(r1.w), vc = __uint_as_float(r2.x),
              atlas_u = __uint_as_float(r2.y), atlas_v = __uint_as_float(r2.z), size_x = __uint_as_float(r2.w),
              size_y = __uint_as_float(r3.x), light = __uint_as_float(r3.y);
  // per record: reciprocal tile sizes (an exponent flip for a power of two), texel-store window, scaled origin
  f32x2 inv_s = {__uint_as_float(0x7F000000u - __float_as_uint(size_x)), __uint_as_float(0x7F000000u - __float_as_uint(size_y))};
  if (NP2) inv_s = exact_rcp2(f32x2{size_x, size_y});
  const bool p2x = (flags & SHADE_POW2_X) != 0u, p2y = (flags & SHADE_POW2_Y) != 0u;
  const uint32_t wm = tex & 0xFFFFu, hm = tex >> 16, lw = (flags >> 8) & 15u;
  const uint32_t wm2 = wm << 1, hms = hm << (lw + 1u);
  const char *tbase = tb + (size_t)((flags >> 16) << 11);  // (texel base >> 10) << 10 texels of two bytes
  const float ys = __uint_as_float((128u + lw) << 23);      // 2 W = 2^(lw + 1)
  const float au2 = atlas_u * 2.0f, avs = atlas_v * ys, light2 = light * 2.0f;
  const uint32_t lc8 = (lane & 3u) * 8u, lr = lane >> 2;  // my run inside a half quadrant: column of 8 pixels, row
  const float px0 = (float)(qx0 + lc8) + 0.5f;
  const bool xin = qx0 + lc8 < (uint32_t)width;
  // COLORMAP row of every screen row of the quadrant, from its two end pixels: lane l -> row l / 2, end l & 1
  uint32_t row_word;  // row << 8, bit 31 = the two ends disagree
  {
    const float pye = (float)(qy0 + (lane >> 1)) + 0.5f, pxe = (float)(qx0 + (lane & 1u) * 31u) + 0.5f;
    const float re = colormap_row(exact_rcp(fmaf(wa, pxe, fmaf(wb, pye, wc))), light2);
    const float ro = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(re), 0xB1, 0xF, 0xF, true));  // quad_perm [1, 0, 3, 2]
    row_word = ((uint32_t)(int)re << 8) | (re == ro ? 0u : 0x80000000u);
  }
#pragma unroll 1
  for (uint32_t h = 0; h < 2u; h++) {
    const uint32_t y = qy0 + h * 16u + lr;
    const float py = (float)y + 0.5f;
    const float row_w = fmaf(wb, py, wc), row_u = fmaf(ub, py, uc), row_v = fmaf(vb, py, vc);
    const uint32_t my_row = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((h * 16u + lr) * 8u), (int)row_word);  // from lane 2 (h 16 + lr)
    // NP2: one guard per run and axis, exactly as in fragment_kernel (see there for the argument)
    bool mod_ok = true;
    float lox = 0.0f, hix = 0.0f, loy = 0.0f, hiy = 0.0f;
    if (NP2) {
      const float pxl = px0 + 7.0f;
      const f32x2 w_ends = exact_rcp2(f32x2{fmaf(wa, px0, row_w), fmaf(wa, pxl, row_w)});
      const float w_hi = fmaxf(w_ends.x, w_ends.y);
      const float bu = fmaxf(fabsf(fmaf(ua, px0, row_u)), fabsf(fmaf(ua, pxl, row_u))) * w_hi;
      const float bv = fmaxf(fabsf(fmaf(va, px0, row_v)), fabsf(fmaf(va, pxl, row_v))) * w_hi;
      lox = fmaxf(bu, size_x) * 0x1p-20f, hix = size_x - lox;
      loy = fmaxf(bv, size_y) * 0x1p-20f, hiy = size_y - loy;
      mod_ok = (p2x | (bu < 0x1p23f)) & (p2y | (bv < 0x1p23f));
    }
    uint32_t texel[8], any_texel = 0;
    float w_first = 0.0f, w_last = 0.0f;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const f32x2 px = {px0 + (float)(2 * p), px0 + (float)(2 * p + 1)};
      const f32x2 w = exact_rcp2(pk_fma(splat(wa), px, splat(row_w)));  // F1
      if (p == 0) w_first = w.x;
      if (p == 3) w_last = w.y;
      const f32x2 tu = pk_fma(splat(ua), px, splat(row_u)) * w;
      const f32x2 tv = pk_fma(splat(va), px, splat(row_v)) * w;
      f32x2 fq = tu * splat(inv_s.x);  // F2: mod(t, size) = t - size * floor(t / size)
      fq = f32x2{floorf(fq.x), floorf(fq.y)};
      const f32x2 rx = pk_fma(splat(-size_x), fq, tu);
      f32x2 fh = tv * splat(inv_s.y);
      fh = f32x2{floorf(fh.x), floorf(fh.y)};
      const f32x2 ry = pk_fma(splat(-size_y), fh, tv);
      if (NP2)
        mod_ok = mod_ok & (p2x | ((rx.x >= lox) & (rx.x <= hix) & (rx.y >= lox) & (rx.y <= hix))) &
                 (p2y | ((ry.x >= loy) & (ry.x <= hiy) & (ry.y >= loy) & (ry.y <= hiy)));
      const f32x2 ux2 = pk_fma(rx, splat(2.0f), splat(au2));  // F3: 2 (r + atlas_u), 2 W (r + atlas_v), rounded as the sums round
      const f32x2 uys = pk_fma(ry, splat(ys), splat(avs));
      const uint32_t b0 = ((uint32_t)cvt_floor_i32(uys.x) & hms) | ((uint32_t)cvt_floor_i32(ux2.x) & wm2);
      const uint32_t b1 = ((uint32_t)cvt_floor_i32(uys.y) & hms) | ((uint32_t)cvt_floor_i32(ux2.y) & wm2);
#if defined(RDOOM_TIMING_EXPERIMENTS) && defined(RDOOM_Q_FOLD)  // wrong images by design: what do the gathers cost?
      texel[2 * p] = RDOOM_Q_FOLD ? *reinterpret_cast<const uint8_t *>(tbase + (b0 & 0x1FEu)) : (b0 & 0xFFu);
      texel[2 * p + 1] = RDOOM_Q_FOLD ? *reinterpret_cast<const uint8_t *>(tbase + (b1 & 0x1FEu)) : (b1 & 0xFFu);
#else
#ifdef RDOOM_Q_LOAD32  // (A/B: 32-bit loads at the texel's 2-byte-aligned address, as fragment_kernel issues them)
      if (true) {
        texel[2 * p] = *reinterpret_cast<const TexelWord *>(tbase + b0) & 0xFFFFu;
        texel[2 * p + 1] = *reinterpret_cast<const TexelWord *>(tbase + b1) & 0xFFFFu;
      } else
#endif
      if (MASKED) {
        texel[2 * p] = *reinterpret_cast<const uint16_t *>(tbase + b0);
        texel[2 * p + 1] = *reinterpret_cast<const uint16_t *>(tbase + b1);
      } else {  // little endian: the palette index is the low byte
        texel[2 * p] = *reinterpret_cast<const uint8_t *>(tbase + b0);
        texel[2 * p + 1] = *reinterpret_cast<const uint8_t *>(tbase + b1);
      }
#endif
    }
    if (MASKED) {
#pragma unroll
      for (int k = 0; k < 8; k++) any_texel |= texel[k];
    }
    uint32_t ci[8];  // COLORMAP index = row * 256 + texel
    if (__all((int)my_row >= 0)) {  // (wave-uniform) every screen row of this half quadrant has one COLORMAP row
#pragma unroll
      for (int k = 0; k < 8; k++) ci[k] = MASKED ? (my_row | (texel[k] & 0xFFu)) : (my_row | texel[k]);
    } else {
      // F4, F5 at the run's two end pixels; the pixels between them only when the ends disagree
      const f32x2 rf_ends = colormap_rows(f32x2{w_first, w_last}, light2);
      if (rf_ends.x == rf_ends.y) {
        const uint32_t r8 = (uint32_t)(int)rf_ends.x << 8;
#pragma unroll
        for (int k = 0; k < 8; k++) ci[k] = r8 | (texel[k] & 0xFFu);
      } else {
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const f32x2 px = {px0 + (float)(2 * p), px0 + (float)(2 * p + 1)};
          const f32x2 rows = colormap_rows(exact_rcp2(pk_fma(splat(wa), px, splat(row_w))), light2);
          ci[2 * p] = ((uint32_t)(int)rows.x << 8) | (texel[2 * p] & 0xFFu);
          ci[2 * p + 1] = ((uint32_t)(int)rows.y << 8) | (texel[2 * p + 1] & 0xFFu);
        }
      }
    }
    uint32_t c[8];
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = cmap[ci[k]];
    const uint32_t out0 = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24), out1 = c[4] | (c[5] << 8) | (c[6] << 16) | (c[7] << 24);
    const bool inside = xin & (y < (uint32_t)height);
    if (inside) *reinterpret_cast<uint2 *>(pfb + ((size_t)y * (size_t)width + (size_t)(qx0 + lc8))) = make_uint2(out0, out1);
    // rare: what fragment_kernel hands to its general body goes to fixup_kernel's general rule, pixel by pixel
    const bool leak = MASKED && (any_texel & 0x8000u) != 0u;
    if (inside & (leak | !mod_ok)) {
#pragma unroll 1
      for (uint32_t k = 0; k < 8u; k++)
        if (!mod_ok || (texel[k] & 0x8000u)) queue_fixup(fc, pose, y * (uint32_t)width + qx0 + lc8 + k);
    }
  }
}

__global__ __launch_bounds__(256) QUAD_OCCUPANCY void fragment_quadrant_kernel(
    const FragConst *__restrict__ fc, const uint16_t *__restrict__ texels, const uint8_t *__restrict__ colormap,
    const TriRec *__restrict__ recs, uint32_t cap, uint32_t *__restrict__ qtab, uint32_t n_poses, uint32_t groups_per_pose,
    uint32_t tiles_x, uint32_t n_tiles, int width, int height, uint8_t *__restrict__ fb) {
  __shared__ uint8_t cmap[32 * 256];
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(colormap);
    uint4 *dst = reinterpret_cast<uint4 *>(cmap);
#pragma unroll
    for (uint32_t k = threadIdx.x; k < 512u; k += 256u) dst[k] = src[k];
  }
  __syncthreads();
  // blockIdx -> (pose, group of tiles): all groups of a pose on one XCD (b % 8), like the other kernels
  const uint32_t g = blockIdx.x >> 3;
  const uint32_t pose = (g / groups_per_pose) * 8u + (blockIdx.x & 7u);
  const uint32_t grp = g % groups_per_pose;
  if (pose >= n_poses) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const TriRec *prec = recs + (size_t)pose * cap;
  uint32_t *ptab = qtab + (size_t)pose * n_tiles * 4u;
  char *pfb = reinterpret_cast<char *>(fb) + (size_t)pose * (size_t)width * (size_t)height;
  const char *tb = reinterpret_cast<const char *>(texels);
#pragma unroll 1
  for (uint32_t i = 0; i < QUAD_TILES_PER_WAVE; i++) {
    const uint32_t t = (grp * 4u + wave) * QUAD_TILES_PER_WAVE + i;
    if (t >= n_tiles) break;  // (wave-uniform)
    const uint32_t ty = (uint32_t)(((float)t + 0.5f) / (float)tiles_x), tx = t - ty * tiles_x;  // (t < 2^16: exact)
    const uint4 e4 = *reinterpret_cast<const uint4 *>(ptab + (size_t)t * 4u);  // the tile's four entries: one scalar load
#pragma unroll 1
    for (uint32_t q = 0; q < 4u; q++) {
      const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)(q == 0u ? e4.x : (q == 1u ? e4.y : (q == 2u ? e4.z : e4.w))));
      const uint32_t qx0 = tx * 64u + (q & 1u) * 32u, qy0 = ty * 64u + (q >> 1) * 32u;
#ifdef RDOOM_FRAG_STATS
      if (lane == 0u && qx0 < (uint32_t)width && qy0 < (uint32_t)height) atomicAdd(&g_frag_stats[ent == NONE ? 3 : 4], 1ull);
#endif
      // (the rasteriser writes no entry for a quadrant outside the frame: position first, then the entry)
      if (qx0 >= (uint32_t)width || qy0 >= (uint32_t)height || ent == NONE) continue;
      const uint4 *rp = reinterpret_cast<const uint4 *>(&prec[ent].s);
      const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
      const uint32_t rflags = prec[ent].r.flags;
      const uint32_t flags = r3.z;
#ifdef RDOOM_FRAG_STATS
      if (lane == 0u) {
        if (!(flags & SHADE_FAST)) atomicAdd(&g_frag_stats[5], 1ull);
        else if (flags & SHADE_NP2) atomicAdd(&g_frag_stats[6], 1ull);
        else if (rflags & RASTER_MASKED_ANY) atomicAdd(&g_frag_stats[7], 1ull);
      }
#endif
      if ((flags & SHADE_FAST) == 0u) continue;  // sky, decor, tile sizes the packed arithmetic does not cover
      // 1/w at the four corner pixels, evaluated as the pixels evaluate it
      const float wa = __uint_as_float(r0.x), wb = __uint_as_float(r0.y), wc = __uint_as_float(r0.z);
      const float xl = (float)qx0 + 0.5f, xh = (float)qx0 + 31.5f, yl = (float)qy0 + 0.5f, yh = (float)qy0 + 31.5f;
      const float rwl = fmaf(wb, yl, wc), rwh = fmaf(wb, yh, wc);
      const uint32_t o0 = __float_as_uint(fmaf(wa, xl, rwl)) - 0x0D800000u, o1 = __float_as_uint(fmaf(wa, xh, rwl)) - 0x0D800000u,
                     o2 = __float_as_uint(fmaf(wa, xl, rwh)) - 0x0D800000u, o3 = __float_as_uint(fmaf(wa, xh, rwh)) - 0x0D800000u;
      // (the bit patterns of [2^-100, 2^100] are the integers [0x0D800000, 0x71800000]; negative numbers and NaNs land above)
      if ((uint32_t)__builtin_amdgcn_readfirstlane((int)max(max(o0, o1), max(o2, o3))) > 0x71800000u - 0x0D800000u) continue;
      const bool masked = (rflags & RASTER_MASKED_ANY) != 0u;
#if RDOOM_QUAD_NP2
      if (flags & SHADE_NP2) {
        if (masked)
          shade_quadrant<true, true>(fc, cmap, tb, pfb, pose, r0, r1, r2, r3, qx0, qy0, lane, width, height);
        else
          shade_quadrant<true, false>(fc, cmap, tb, pfb, pose, r0, r1, r2, r3, qx0, qy0, lane, width, height);
      } else
#else
      if (flags & SHADE_NP2) continue;  // integer tile sizes that are not powers of two: fragment_kernel certifies their mod
#endif
      {
        if (masked)
          shade_quadrant<false, true>(fc, cmap, tb, pfb, pose, r0, r1, r2, r3, qx0, qy0, lane, width, height);
        else
          shade_quadrant<false, false>(fc, cmap, tb, pfb, pose, r0, r1, r2, r3, qx0, qy0, lane, width, height);
      }
      if (lane == 0u) ptab[(size_t)t * 4u + q] = ent | QTAB_HANDLED;
    }
  }
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
                                                    const uint32_t *__restrict__ counts, uint32_t cap,
                                                    const PoseConst *__restrict__ poses, int width, int pitch, int height,
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
    const int iy = (int)(pix / (uint32_t)pitch), ix = (int)(pix - (uint32_t)iy * (uint32_t)pitch);  // (queued as row * pitch + column)
    const float px = (float)ix + 0.5f, py = (float)iy + 0.5f;
    const TriRec *prec = recs + (size_t)pose * cap;
    const bool binned = overflow[pose] == 0u;
    const uint32_t T = (uint32_t)(tiles_x * tiles_y), tile = (uint32_t)((iy >> 6) * tiles_x + (ix >> 6));
    uint2 hdr = binned ? tile_hdr[(size_t)pose * T + tile] : make_uint2(0u, counts[pose]);
    if (binned && (hdr.y & TILE_SPLIT)) {  // a tile with a list per quadrant (bin.hip): the list of the pixel's quadrant
      const uint32_t *sh = entries + (size_t)pose * entry_cap + hdr.x + 2u * (uint32_t)(((iy >> 5) & 1) * 2 + ((ix >> 5) & 1));
      hdr = make_uint2(sh[0], sh[1]);
    }
    unsigned long long best = ~0ull;
    uint32_t best_rec = NONE;
    for (uint32_t base = 0; base < hdr.y; base += 64u) {
      const uint32_t e = base + lane;
      unsigned long long key = ~0ull;
      uint32_t rec = NONE;
      if (e < hdr.y) {
        rec = binned ? (entries[(size_t)pose * entry_cap + hdr.x + e] & ENTRY_REC_MASK) : e;  // (no bins: every record of the pose, near to far)
        const RasterRec r = prec[rec].r;
        const float *rwp = prec[rec].s.wp;  // (the 1/w plane lives in the shade part)
        const int x0 = (int)(r.bb0 & 0xFFFFu), y0 = (int)(r.bb0 >> 16), x1 = (int)(r.bb1 & 0xFFFFu),
                  y1 = (int)(r.bb1 >> 16);
        const float e0 = fmaf(r.e[0], px, fmaf(r.e[1], py, r.e[2])), e1 = fmaf(r.e[3], px, fmaf(r.e[4], py, r.e[5])),
                    e2 = fmaf(r.e[6], px, fmaf(r.e[7], py, r.e[8]));
        const bool in0 = (e0 > 0.0f) | ((e0 == 0.0f) & ((r.flags & (1u << 24)) != 0u));
        const bool in1 = (e1 > 0.0f) | ((e1 == 0.0f) & ((r.flags & (1u << 25)) != 0u));
        const bool in2 = (e2 > 0.0f) | ((e2 == 0.0f) & ((r.flags & (1u << 26)) != 0u));
        const float zw = fmaf(r.zp[0], px, fmaf(r.zp[1], py, r.zp[2]));
        const float rw = fmaf(rwp[0], px, fmaf(rwp[1], py, rwp[2]));
        bool pass = (ix >= x0) & (ix <= x1) & (iy >= y0) & (iy <= y1) & in0 & in1 & in2 & (zw >= 0.0f) & (zw <= 1.0f) &
                    (rw > 0.0f);
        if (pass && (r.flags & RASTER_MASKED_ANY) != 0u) {
          const ShadeRec sh = prec[rec].s;
          const TexelAt t = texel_coords(sh, px, fmaf(sh.wp[1], py, sh.wp[2]), fmaf(sh.up[1], py, sh.up[2]),
                                         fmaf(sh.vp[1], py, sh.vp[2]));
          pass = (load_texel(lv.texels, sh, t.ix, t.iy) & 0x8000u) == 0u;
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
      const size_t o = ((size_t)pose * (size_t)height + (size_t)iy) * (size_t)pitch + (size_t)ix;
      uint32_t colour = 0;
      if (best_rec != NONE) {
        const ShadeRec sh = prec[best_rec].s;
        colour = shade_pixel<false>(lv, lv.colormap, sh, px, py, fmaf(sh.wp[1], py, sh.wp[2]), fmaf(sh.up[1], py, sh.up[2]),
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

}  // namespace

size_t fragment_const_bytes() { return sizeof(FragConst); }

FragmentPlan plan_fragment(int width, int pitch, int height, bool have_qtab) {
  const rdoom::DebugOptions dbg = rdoom::debug_options();  // test hooks: equivalent paths, same image
  FragmentPlan p{};
  p.leak_mod = (uint32_t)std::max(0, dbg.leak_mod);
  p.nq = (dbg.frag_nq == 2 && pitch % 8 == 0) ? 2 : 1;  // quads per lane: two when rows (of `pitch` pixels) divide into 8-pixel runs
  // log2(units per block row).  By default a wave's block is 4 runs x 16 rows (32 x 16 pixels, inside ONE quadrant of the
  // rasteriser's table) for frames of 1280 x 720 and up, 8 runs x 8 rows (64 x 8, two quadrants side by side) below: measured on
  // one box in round 5 -- 4K 596 -> 620 Gpixel/s, E1M1..E1M9 at 1080p 508 -> 523, 1080p 487 -> 491, 720p alike, 320 x 200
  // 119 -> 116 (profiles/r05_ab.txt); rounds 3-4 ran 64 x 8 everywhere.
  // (a row pitch that is a multiple of 4 but not of 8 leaves one quad per lane: its 32-pixel block is 8 units wide whatever the frame)
  const int bw_auto = (p.nq == 2 && (size_t)pitch * (size_t)height >= (size_t)1280 * 720) ? 2 : 3;
  p.bwl = (uint32_t)std::min(6, dbg.frag_bw >= 0 ? dbg.frag_bw : bw_auto);
  p.chunk = dbg.frag_chunk > 0 ? (uint32_t)dbg.frag_chunk : (uint32_t)FRAG_CHUNK;
  // the quadrant table serves blocks that lie inside one 32 x 32 quadrant, or inside two side by side
  const uint32_t bw = 1u << p.bwl, bh = 64u >> p.bwl, block_px = bw * 4u * (uint32_t)p.nq;
  p.qtab_mode = (!have_qtab || dbg.no_qtab || bh > 32u) ? 0u : (block_px == 32u ? 1u : (block_px == 64u ? 2u : 0u));
  // With leak_mod (tests) the kernel sends every block through its visibility words, so they must all be there;
  // keep_vis (tests, A/B runs) asks for them outright.
  p.skip_described_vis = p.qtab_mode != 0u && p.leak_mod == 0u && !dbg.keep_vis;
  // the whole-quadrant kernel runs first and marks what it shaded in the table: only where fragment_kernel reads the table
  // (and not under leak_mod, which sends every pixel through the general rule); rows must divide into 8-pixel runs
  // -- an alternative path, OFF by default (the hook "qpath" switches it on): measured in round 4, the two kernels together are
  // 10 % slower than fragment_kernel alone (the quadrant kernel needs 211 VALU instructions per 8-pixel run
  // where fragment_kernel's wave-uniform body needs 275, and what is left for fragment_kernel are the expensive blocks)
  p.quadrant_path = p.qtab_mode != 0u && p.leak_mod == 0u && dbg.qpath && width % 8 == 0 && pitch == width;
  return p;
}

rdoom_status launch_fragment(hipStream_t st, uint32_t n_poses, const DeviceLevelView &lv, const TriRec *recs,
                             const uint32_t *counts, uint32_t cap, const PoseConst *poses,
                             int width, int pitch, int height, int tiles_x, int tiles_y, const uint2 *tile_hdr,
                             const uint32_t *entries, uint32_t entry_cap, const uint32_t *overflow, uint32_t *vis,
                             bool vis16, uint32_t *prim_out, const float *ndc_tab, uint8_t *fb, uint32_t *fix_count,
                             uint2 *fix_list, uint32_t fix_cap, uint32_t *qtab, void *d_frag_const,
                             bool *frag_const_ready, const FragmentPlan &plan) {
  const uint32_t n = n_poses;
  // W: the frame (sky ndc); rows of visibility words and framebuffer bytes are `pitch` pixels apart (a multiple of 4, W <= pitch < W + 8)
  const int W = width, H = height;
  const uint32_t qpr = (uint32_t)pitch / 4u, qpp = qpr * (uint32_t)H;
  if (qpp >= (1u << 24)) return rdoom::fail(RDOOM_BAD_ARG, "frame too large");
  // multiply-high divisor for idx / qpr, idx < 2^24 (checked exhaustively at the only places it can fail)
  if (qpr < 2) return rdoom::fail(RDOOM_BAD_ARG, "width must be at least 8");
  uint32_t div_sh = 0;
  while ((2u << div_sh) <= qpr) div_sh++;   // floor(log2(qpr))
  if ((qpr & (qpr - 1)) == 0) div_sh -= 1;  // power of two: m = 2^31
  const uint32_t div_m = (uint32_t)((((uint64_t)1 << (32 + div_sh)) + qpr - 1) / qpr);
  for (uint32_t k = 1; k * qpr <= qpp; k++) {
    const uint32_t lo = k * qpr - 1, hi = k * qpr;
    if ((uint32_t)(((uint64_t)lo * div_m) >> 32) >> div_sh != k - 1 ||
        (hi < qpp && (uint32_t)(((uint64_t)hi * div_m) >> 32) >> div_sh != k))
      return rdoom::fail(RDOOM_BAD_ARG, "internal: fast_div constants invalid for row pitch %d", pitch);
  }
  const uint32_t debug_leak_mod = plan.leak_mod;
  const int nq = plan.nq;
  const uint32_t bwl = plan.bwl;
  const uint32_t bw = 1u << bwl, bh = 64u >> bwl;
  const uint32_t wbpr = (qpr / (uint32_t)nq + bw - 1u) / bw, wbpp = wbpr * (((uint32_t)H + bh - 1u) / bh);  // bw-unit x bh-row blocks
  const uint32_t frag_chunk = plan.chunk;
  const uint32_t fblocks = (wbpp + frag_chunk * FRAG_WAVES - 1u) / (frag_chunk * FRAG_WAVES);  // a workgroup = FRAG_WAVES waves x frag_chunk blocks
  // (fix_count[0..1] arrive zeroed: the caller's one fill at the start of the render)
  const uint64_t fgrid = (uint64_t)((n + 7) / 8) * 8ull * fblocks;
  if (fgrid > 0x7FFFFFFFull) return rdoom::fail(RDOOM_BAD_ARG, "batch too large for one launch");
  auto frag = nq == 2 ? (vis16 ? fragment_kernel<2, 0, true> : fragment_kernel<2, 0, false>)
                      : (vis16 ? fragment_kernel<1, 0, true> : fragment_kernel<1, 0, false>);
#ifdef RDOOM_TIMING_EXPERIMENTS  // wrong images by design: never in the shipped library
  // RDOOM_FRAG_DBG = bit set: 2 no texel loads, 4 no framebuffer stores, 8 no COLORMAP look-ups in LDS (two quads per lane, 16-bit words only)
  if (getenv("RDOOM_FRAG_DBG") && nq == 2 && vis16) {
    switch (atoi(getenv("RDOOM_FRAG_DBG"))) {
      case 2: frag = fragment_kernel<2, 2, true>; break;
      case 4: frag = fragment_kernel<2, 4, true>; break;
      case 8: frag = fragment_kernel<2, 8, true>; break;
      case 6: frag = fragment_kernel<2, 6, true>; break;
      case 10: frag = fragment_kernel<2, 10, true>; break;
      case 14: frag = fragment_kernel<2, 14, true>; break;
      case 16: frag = fragment_kernel<2, 16, true>; break;
      default: break;
    }
  }
#endif
  const uint32_t qtab_mode = qtab ? plan.qtab_mode : 0u;
  if (!*frag_const_ready) {  // constant for the life of the batch: written once
    FragConst h{};
    h.lv = lv, h.fix_count = fix_count, h.fix_list = fix_list, h.ndc_tab = ndc_tab, h.fix_cap = fix_cap, h.div_m = div_m, h.div_sh = div_sh;
    HIP_TRY(hipMemcpy(d_frag_const, &h, sizeof h, hipMemcpyHostToDevice));
    *frag_const_ready = true;
  }
  if (qtab && plan.quadrant_path) {  // described quadrants first: whole quadrants, one record each (marks them QTAB_HANDLED)
    const uint32_t n_tiles = (uint32_t)(tiles_x * tiles_y), groups = (n_tiles + 4u * QUAD_TILES_PER_WAVE - 1u) / (4u * QUAD_TILES_PER_WAVE);
    const uint64_t qgrid = (uint64_t)((n + 7) / 8) * 8ull * groups;
    if (qgrid > 0x7FFFFFFFull || n_tiles >= (1u << 16)) return rdoom::fail(RDOOM_BAD_ARG, "batch too large for one launch");
    hipLaunchKernelGGL(fragment_quadrant_kernel, dim3((uint32_t)qgrid), dim3(256), 0, st, static_cast<const FragConst *>(d_frag_const), lv.texels,
                       lv.colormap, recs, cap, qtab, n, groups, (uint32_t)tiles_x, n_tiles, W, H, fb);
  }
#ifdef RDOOM_FRAG_CHUNKS_OUTER
  if (fblocks > 65535u) return rdoom::fail(RDOOM_BAD_ARG, "frame too large for one launch");
  const dim3 fgrid_dim(((n + 7u) / 8u) * 8u, fblocks);
#else
  const dim3 fgrid_dim((uint32_t)fgrid);
#endif
  hipLaunchKernelGGL(frag, fgrid_dim, dim3(64 * FRAG_WAVES), 0, st, static_cast<const FragConst *>(d_frag_const), lv.texels,
                     lv.colormap, recs, cap, poses, vis, n, fblocks, frag_chunk, qpp, qpr, wbpr, wbpp, bwl, W, H, fb, debug_leak_mod, qtab,
                     qtab_mode, (uint32_t)tiles_x, (uint32_t)(tiles_x * tiles_y));
  hipLaunchKernelGGL(fixup_kernel, dim3(64), dim3(256), 0, st, lv, recs, counts, cap, poses, W, pitch, H, tiles_x, tiles_y,
                     tile_hdr, entries, entry_cap, overflow, fix_count, fix_list, fix_cap, vis, vis16 ? 1u : 0u, prim_out, fb,
                     fix_count + 1);
#ifdef RDOOM_FRAG_STATS
  {
    unsigned long long h[16];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_frag_stats), sizeof h);
    fprintf(stderr, "[frag stats] quadrants in the frame: undescribed %llu, described %llu (not shaded by the quadrant kernel: sky / decor / ineligible sizes %llu, non-power-of-two size %llu, masked texture %llu) | blocks of fragment_kernel: walked %llu, skipped %llu, walked with one handled half %llu\n",
            h[3], h[4], h[5], h[6], h[7], h[0], h[1], h[2]);
    fprintf(stderr, "[frag stats] packed-body invocations (waves) %llu, of which with some run whose COLORMAP rows differ at its ends %llu (%.1f %%; such runs %llu)\n", h[14], h[15], h[14] ? 100.0 * h[15] / h[14] : 0.0, h[2]);
    fprintf(stderr, "[frag stats] runs %llu: to the general body %llu (%.2f %%): mixed %llu, rw out of range %llu, mod uncertified %llu, transparent texel %llu, other (decor, ineligible sizes) %llu\n", h[8], h[9], 100.0 * h[9] / h[8], h[10], h[11], h[12], h[13], h[9] - h[10] - h[11] - h[12] - h[13]);
  }
#endif
  return RDOOM_OK;
}

}  // namespace rdoom_dev
