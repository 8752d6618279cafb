// Kernel 1b: binning of the near-to-far record list into per-tile lists.
//
// Part of the pose-batch renderer for gfx950 (MI355X) that replaces the reference's GL draw path:
// assets/shaders/static.{vert,frag}, sky.{vert,frag}, sprite.{vert,frag} and the fixed-function state of
// engine/src/renderer.rs:49-57 + engine/src/window.rs:12,40-44.  The arithmetic is specified in DESIGN.md
// "Raster arithmetic"; operation order follows that text, not the oracle's source.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"

#pragma clang fp contract(off)

namespace rdoom_dev {
namespace {

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
// SPLIT LISTS.  A tile whose list is longer than one batch of the rasteriser (LONG_LIST = 64 entries, one per lane) is far
// geometry -- small triangles that touch one 32 x 32 quadrant each -- and the rasteriser, which gathers a tile's list once
// for its four quadrants when it fits the lanes, would gather ALL of it again for every quadrant (at 320 x 200 a quarter of
// the tiles hold 78 % of the entries this way; on the large level a tenth holds 76 %).  Such a tile gets one list PER
// QUADRANT instead (an entry that touches several quadrants is in each of their lists): header word y carries
// TILE_SPLIT, word x points at eight words (first entry, count) x 4 in the entry array, the four lists follow.  The
// rasteriser treats each as the list of a tile that consists of that quadrant: most fit one batch again (one gather, the
// shortcuts apply), the rest re-gathers only its own entries.
// =================================================================================================

// Inclusive prefix sum over the workgroup: shuffles inside a wave, the wave totals through LDS (two barriers; tmp[w] ends
// up holding the inclusive total of waves 0..w, so tmp[N / 64 - 1] is the grand total).  Ends with a barrier.
template <int N>
__device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t *tmp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  __syncthreads();  // earlier readers of tmp are done
  if (lane == 63) tmp[wave] = incl;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (int w = 0; w < N / 64; w++)
    if (w < wave) before += tmp[w];
  __syncthreads();
  if (lane == 63) tmp[wave] = before + incl;
  __syncthreads();
  return before + incl;
}

#ifndef RDOOM_BIN_OCC
#define RDOOM_BIN_OCC 4  // waves per SIMD the register allocation must allow (128 VGPRs: all 1024 workgroups of a 1024-pose batch resident at once)
#endif
template <int BIN_THREADS, int BIN_LOG2>  // threads per workgroup = triangles staged per round (one per thread)
__global__ __launch_bounds__(BIN_THREADS, RDOOM_BIN_OCC) void bin_kernel(const TriRec *__restrict__ recs,
                                                          const uint32_t *__restrict__ counts, uint32_t cap,
                                                          int tiles_x, int tiles_y, uint2 *__restrict__ tile_hdr,
                                                          uint32_t *__restrict__ entries, uint32_t entry_cap,
                                                          uint2 *__restrict__ hits, uint32_t *__restrict__ overflow,
                                                          uint32_t split) {
  constexpr uint32_t BIN_CHUNK = BIN_THREADS;
  static_assert((1 << BIN_LOG2) == BIN_THREADS, "bin_kernel: BIN_LOG2");
  extern __shared__ unsigned long long bin_dyn[];  // split: qc[T] (64 bits per tile), then -- always -- tile_cnt[T] (entries per tile, later the fill pass's cursor)
  __shared__ uint4 coef[BIN_CHUNK][3];   // e[9], zp[3] of the staged triangles
  __shared__ uint2 bbox[BIN_CHUNK];
  __shared__ uint32_t trange[BIN_CHUNK];  // tile rectangle to visit: tx0 | ty0 << 8 | width << 16
  __shared__ uint32_t pref[BIN_CHUNK + 1];
  __shared__ uint32_t scan_tmp[BIN_THREADS];
  __shared__ uint32_t n_hits;  // (triangle, tile) pairs that passed the tile test so far
  const uint32_t pose = blockIdx.x;
  const int tid = threadIdx.x;
  const uint32_t T = (uint32_t)(tiles_x * tiles_y);
  if (T > MAX_TILES) {
    if (tid == 0) overflow[pose] = 1u;
    return;
  }
  // per-quadrant entry counts of every tile, four 16-bit fields in one 64-bit word -- one LDS atomic per pair found (a tile
  // with more than 65 535 entries overflows the pose); for the fill pass the fields become the four lists' cursors
  unsigned long long *qc = bin_dyn;
  uint32_t *tile_cnt = reinterpret_cast<uint32_t *>(bin_dyn + (split ? T : 0u));
  const TriRec *prec = recs + (size_t)pose * cap;
  uint2 *hdr = tile_hdr + (size_t)pose * T;
  uint32_t *pent = entries + (size_t)pose * entry_cap;
  uint2 *phits = hits + (size_t)pose * entry_cap;  // (entry, tile) of every pair that passed: the fill pass only scatters
  const uint32_t n = counts[pose];
  for (uint32_t i = tid; i < T; i += BIN_THREADS) {
    tile_cnt[i] = 0;
    if (split) qc[i] = 0ull;
  }
  if (tid == 0) n_hits = 0;
  {
    for (uint32_t cbase = 0; cbase < n; cbase += BIN_CHUNK) {
      const uint32_t cn = min(BIN_CHUNK, n - cbase);
      __syncthreads();  // previous round's readers of coef/pref are done (and the tile counters are zeroed)
      // Staging, line by line: the first 64 bytes of a record (edges, depth plane, bbox) are four 16-byte words, and FOUR
      // consecutive lanes read the four words of one record -- a wave's load instruction touches 16 half lines, each whole, where a
      // lane reading its own record at the records' 128-byte stride touched 64 lines for 16 bytes each, three times over (round 6;
      // the bbox used to come from a second array, `sorted`, which repeated what the record holds).  A wave stages the 64 records
      // its own lanes will work on: no workgroup barrier between the staging and the reads below.
      {
        const uint32_t lane = (uint32_t)tid & 63u, w64 = (uint32_t)tid & ~63u, word = lane & 3u;
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
          const uint32_t r = w64 + (lane >> 2) + 16u * j;
          if (r < cn) {
            const uint4 v = reinterpret_cast<const uint4 *>(&prec[cbase + r])[word];
            if (word < 3u)
              coef[r][word] = v;
            else
              bbox[r] = make_uint2(v.x, v.y);  // (bb0, bb1, flags, pad)
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      uint32_t nt = 0;
      if ((uint32_t)tid < cn) {
        const uint4 c0 = coef[tid][0], c1 = coef[tid][1], c2 = coef[tid][2];
        const uint2 ent = bbox[tid];
        int tx0 = (int)((ent.x & 0xFFFFu) >> 6), ty0 = (int)((ent.x >> 16) >> 6), tx1 = (int)((ent.y & 0xFFFFu) >> 6),
            ty1 = (int)((ent.y >> 16) >> 6);
        if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > 32) {
          // a big tile rectangle (typically a triangle that crosses the eye plane: bbox = whole frame, S6): shrink
          // it to the tile rows / columns whose full-width / full-height strip can be touched at all.  Exact
          // (rect_may_touch is conservative and hierarchical), so no tile that passes the per-tile test is lost.
          const float fx0 = (float)(tx0 * 64) + 0.5f, fx1 = (float)(tx1 * 64) + 63.5f;
          const float fy0 = (float)(ty0 * 64) + 0.5f, fy1 = (float)(ty1 * 64) + 63.5f;
          // (one loop over the four sides -- first rows from the top, last rows, first columns, last columns -- so that
          // rect_may_touch is inlined once: this rare path must not cost the kernel registers)
          int side = 0;
#pragma unroll 1
          while (side < 4 && tx0 <= tx1 && ty0 <= ty1) {
            const bool rows = side < 2;
            const int t = side == 0 ? ty0 : (side == 1 ? ty1 : (side == 2 ? tx0 : tx1));
            const float lo = (float)(t * 64) + 0.5f, hi = (float)(t * 64) + 63.5f;
            if (rect_may_touch(c0, c1, c2, rows ? fx0 : lo, rows ? fx1 : hi, rows ? lo : fy0, rows ? hi : fy1)) {
              side++;
            } else {
              ty0 += side == 0;
              ty1 -= side == 1;
              tx0 += side == 2;
              tx1 -= side == 3;
            }
          }
        }
        nt = (tx1 >= tx0 && ty1 >= ty0) ? (uint32_t)((tx1 - tx0 + 1) * (ty1 - ty0 + 1)) : 0u;
        trange[tid] = (uint32_t)tx0 | ((uint32_t)ty0 << 8) | ((uint32_t)(tx1 - tx0 + 1) << 16);  // tiles per side <= 128
      }
      const uint32_t incl = block_scan<BIN_THREADS>(nt, scan_tmp);  // inclusive prefix of nt over the workgroup
      pref[tid] = incl - nt;                                        // exclusive; entries past cn repeat the total
      const uint32_t W = scan_tmp[BIN_THREADS / 64 - 1];
      if (tid == 0) pref[BIN_CHUNK] = W;
      __syncthreads();
      // Two pairs per thread and iteration, stage by stage (both binary searches, then both pairs' coefficient reads and tests,
      // then the counters): the workgroup is alone with its latencies -- eight dependent LDS reads of the search, three 16-byte
      // reads, the atomics -- and a second independent pair fills them.  The hit index comes from ONE atomic per wave (ballot +
      // prefix count): a per-pair returning atomic on one LDS word serialised the 64 lanes.
      constexpr uint32_t PAIRS = 2;
      for (uint32_t w0 = (uint32_t)tid; w0 < W; w0 += PAIRS * BIN_THREADS) {
        uint32_t wq[PAIRS], lo[PAIRS], hi[PAIRS];
#pragma unroll
        for (uint32_t u = 0; u < PAIRS; u++) wq[u] = w0 + u * BIN_THREADS, lo[u] = 0u, hi[u] = BIN_CHUNK;
        // largest i with pref[i] <= w (pref non-decreasing, pref[0] = 0, pref[BIN_CHUNK] = W > w); triangles past cn have
        // pref == W and are never selected (a w past W selects nothing: masked below)
#pragma unroll
        for (int step = 0; step < BIN_LOG2; step++) {
#pragma unroll
          for (uint32_t u = 0; u < PAIRS; u++) {
            const uint32_t mid = (lo[u] + hi[u]) >> 1;
            const bool le = pref[mid] <= wq[u];
            lo[u] = le ? mid : lo[u];
            hi[u] = le ? hi[u] : mid;
          }
        }
        uint32_t qm[PAIRS], tile[PAIRS];
#pragma unroll
        for (uint32_t u = 0; u < PAIRS; u++) {
          qm[u] = 0u, tile[u] = 0u;
          if (wq[u] < W) {
            const uint32_t t = wq[u] - pref[lo[u]];
            const uint2 bb = bbox[lo[u]];
            const int x0 = (int)(bb.x & 0xFFFFu), y0 = (int)(bb.x >> 16), x1 = (int)(bb.y & 0xFFFFu), y1 = (int)(bb.y >> 16);
            const uint32_t tr = trange[lo[u]];
            const int tx0 = (int)(tr & 0xFFu), ty0 = (int)((tr >> 8) & 0xFFu), ntx = (int)(tr >> 16);
            // t / ntx for t < 8192, ntx <= 128: (t + 0.5) / ntx lies at least 1 / 256 away from an integer and v_rcp_f32's error
            // (one ulp) moves the product by less than 8192.5 * 2^-22 < 1 / 256: exact without the IEEE division's dozen instructions
            const int ty = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)ntx)), tx = (int)t - ty * ntx;
            const uint4 c0 = coef[lo[u]][0], c1 = coef[lo[u]][1], c2 = coef[lo[u]][2];
            // (no whole-tile test first: a quadrant's corner values ask for no less than the tile's, so the mask is zero wherever
            // the tile test fails -- and with 64 pairs per wave some lane passed it nearly always: 30 instructions per pair for nothing)
            qm[u] = tile_quadrant_mask(c0, c1, c2, x0, y0, x1, y1, (tx0 + tx) * 64, (ty0 + ty) * 64);
            tile[u] = (uint32_t)((ty0 + ty) * tiles_x + tx0 + tx);
          }
        }
#pragma unroll
        for (uint32_t u = 0; u < PAIRS; u++) {
          const unsigned long long hitm = __ballot(qm[u] != 0u);
          if (hitm == 0ull) continue;  // (wave-uniform)
          uint32_t first = 0;
          if ((threadIdx.x & 63u) == 0u) first = atomicAdd(&n_hits, (uint32_t)__popcll(hitm));  // (a pose with more than entry_cap pairs overflows below)
          first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
          if (qm[u]) {
            atomicAdd(&tile_cnt[tile[u]], 1u);
            if (split)
              atomicAdd(&qc[tile[u]], (unsigned long long)((qm[u] & 1u) | ((qm[u] & 2u) << 15)) | ((unsigned long long)(((qm[u] >> 2) & 1u) | ((qm[u] & 8u) << 13)) << 32));
            const uint32_t k = first + (uint32_t)__popcll(hitm & ((1ull << (threadIdx.x & 63u)) - 1ull));
            if (k < entry_cap) phits[k] = make_uint2((cbase + lo[u]) | (qm[u] << 28), tile[u]);
          }
        }
      }
    }
    __syncthreads();
    // exclusive scan of the tiles' list sizes -> first entries (thread t owns T / BIN_THREADS consecutive tiles); headers out
    const uint32_t per = (T + BIN_THREADS - 1u) / BIN_THREADS, lo = min((uint32_t)tid * per, T), hi = min(lo + per, T);
    auto quads_of = [&](uint32_t i, uint32_t (&c)[4]) {
      const unsigned long long v = qc[i];
      c[0] = (uint32_t)v & 0xFFFFu, c[1] = ((uint32_t)v >> 16), c[2] = (uint32_t)(v >> 32) & 0xFFFFu, c[3] = (uint32_t)(v >> 48);
    };
    uint32_t sum = 0;
    int huge = 0;
    for (uint32_t i = lo; i < hi; i++) {
      const uint32_t c = tile_cnt[i];
      uint32_t size = c;
      if (split && c > LONG_LIST) {
        uint32_t qc[4];
        quads_of(i, qc);
        size = 8u + qc[0] + qc[1] + qc[2] + qc[3];
        huge |= (c > 0xFFFFu) | (size > 0xFFFFu);  // (counts and cursors are 16-bit fields)
      }
      sum += size;
    }
    const uint32_t incl_t = block_scan<BIN_THREADS>(sum, scan_tmp);
    const uint32_t total = scan_tmp[BIN_THREADS / 64 - 1];
    // too many entries for the pose's share of the entry array (the pairs found, which the hits array holds, or the entries
    // with the split lists' copies), or a 16-bit quadrant count that may have wrapped: the rasteriser scans the sorted list
    const bool over = __syncthreads_or(huge) != 0 || total > entry_cap || n_hits > entry_cap;
    if (tid == 0) overflow[pose] = over ? 1u : 0u;
    if (over) return;  // uniform
    uint32_t run = incl_t - sum;
    for (uint32_t i = lo; i < hi; i++) {
      const uint32_t c = tile_cnt[i];
      if (split && c > LONG_LIST) {
        uint32_t qc[4];
        quads_of(i, qc);
        uint32_t o = run + 8u;
        unsigned long long cursors = 0ull;  // each list's next free place, relative to the tile's first word
#pragma unroll
        for (int q = 0; q < 4; q++) {
          pent[run + 2u * (uint32_t)q] = o;
          pent[run + 2u * (uint32_t)q + 1u] = qc[q];
          cursors |= (unsigned long long)(o - run) << (16 * q);
          o += qc[q];
        }
        hdr[i] = make_uint2(run, c | TILE_SPLIT);
        tile_cnt[i] = run | TILE_SPLIT;
        run = o;
        bin_dyn[i] = cursors;
      } else {
        hdr[i] = make_uint2(run, c);
        tile_cnt[i] = run;  // the fill pass's cursor
        run += c;
      }
    }
    __syncthreads();  // cursors, and the split tiles' sub-headers (global memory written by this workgroup), are in place
    // fill: the pairs are read back in the order they were found (near to far up to a window of BIN_CHUNK triangles)
    // (four pairs per thread and iteration, their loads and cursor atomics in flight together: the pass is a chain of
    // load -> LDS atomic -> store per pair, and a workgroup has nothing else to hide it behind)
    const uint32_t found = n_hits;
    constexpr uint32_t FILL_UNROLL = 4;
    for (uint32_t k0 = 0; k0 < found; k0 += FILL_UNROLL * BIN_THREADS) {
      uint2 h[FILL_UNROLL];
#pragma unroll
      for (uint32_t u = 0; u < FILL_UNROLL; u++) {
        const uint32_t k = k0 + u * BIN_THREADS + (uint32_t)tid;
        h[u] = k < found ? phits[k] : make_uint2(0u, NONE);
      }
#pragma unroll
      for (uint32_t u = 0; u < FILL_UNROLL; u++) {
        if (h[u].y == NONE) continue;
        const uint32_t cur = tile_cnt[h[u].y];
        if (!(cur & TILE_SPLIT)) {
          pent[atomicAdd(&tile_cnt[h[u].y], 1u)] = h[u].x;
        } else {
          // one copy per quadrant touched: ONE atomic advances the cursors of all of them and returns where each copy goes
          const uint32_t base = cur & ~TILE_SPLIT, qm = h[u].x >> 28;
          const unsigned long long inc = (unsigned long long)((qm & 1u) | ((qm & 2u) << 15)) | ((unsigned long long)(((qm >> 2) & 1u) | ((qm & 8u) << 13)) << 32);
          const unsigned long long at = atomicAdd(&qc[h[u].y], inc);
#pragma unroll
          for (uint32_t q = 0; q < 4u; q++)
            if ((qm >> q) & 1u) pent[base + ((uint32_t)(at >> (16u * q)) & 0xFFFFu)] = h[u].x;
        }
      }
    }
  }
}

}  // namespace

rdoom_status launch_bin(hipStream_t st, uint32_t n_poses, const TriRec *recs, const uint32_t *counts,
                        uint32_t cap, int tiles_x, int tiles_y, uint2 *tile_hdr, uint32_t *entries, uint32_t entry_cap,
                        uint2 *hits, uint32_t *overflow, bool want_split, bool *launched, bool *used_split) {
  *launched = false, *used_split = false;
  const uint32_t bin_tiles = std::min<uint32_t>((uint32_t)(tiles_x * tiles_y), MAX_TILES);
  if ((uint32_t)(tiles_x * tiles_y) > MAX_TILES) return RDOOM_OK;
  // threads per workgroup = triangles staged per round: 256; 128 for small frames (at most 64 tiles: 512 x 512 pixels), where a
  // triangle touches one tile or two -- the smaller window also keeps the lists closer to near-to-far order, which the
  // rasteriser's early-z lives on (320 x 200: set-up + binning 1.56 -> 1.50 ms, rasteriser 2.47 -> 2.33 ms)
  // 512 for a large level rendered for few poses (BASELINE config 5's class: 36 k triangles, 256 poses at 4K): the kernel is one
  // workgroup per pose walking the pose's visible triangles round by round -- with no more workgroups than CUs, twice the
  // threads halve the rounds (set-up + binning 0.99 -> 0.76 ms there; on E1M1-sized levels the wider window costs the
  // rasteriser's early-z more than it saves: measured, profiles/r05_ab.txt)
  const int by_shape = tiles_x * tiles_y <= 64 ? 128 : ((cap >= 16384u && n_poses <= 512u) ? 512 : 256);
  const bool forced = rdoom::debug_options().bin_threads > 0;
  int bin_threads = forced ? rdoom::debug_options().bin_threads : by_shape;
  if (bin_threads != 512 && bin_threads != 128 && bin_threads != 64) bin_threads = 256;
  // (the dynamic segment as it is REQUESTED: a counter per tile, two more with split lists, and 8 bytes that keep the 64-bit
  // words of the split lists' counters aligned -- the same number in the budget test and in the launch)
  auto dyn_bytes = [&](bool with_split) { return (with_split ? 3 : 1) * sizeof(uint32_t) * (size_t)bin_tiles + 8u; };
  // LDS budget: the kernel's static arrays plus a counter per tile must fit the 64 KiB a workgroup may use; frames
  // with more tiles than that (7680x4320 and up) are rasterised from the sorted list instead
  static std::atomic<size_t> static_lds[4];  // per variant, asked once (a constant of the compiled kernel); 0 = not asked yet
  auto variant = [&](int threads, size_t *static_bytes) {
    auto k = threads == 512 ? bin_kernel<512, 9> : (threads == 128 ? bin_kernel<128, 7> : (threads == 64 ? bin_kernel<64, 6> : bin_kernel<256, 8>));
    std::atomic<size_t> &slot = static_lds[threads == 512 ? 2 : (threads == 128 ? 1 : (threads == 64 ? 3 : 0))];
    size_t lds = slot.load(std::memory_order_relaxed);
    if (lds == 0) {
      hipFuncAttributes attr;
      if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(k)) != hipSuccess) {
        *static_bytes = ~(size_t)0 >> 1;
        return k;
      }
      lds = attr.sharedSizeBytes + 1;
      slot.store(lds, std::memory_order_relaxed);
    }
    *static_bytes = lds - 1;
    return k;
  };
  size_t static_bytes = 0;
  auto bk = variant(bin_threads, &static_bytes);
  // The 512-thread variant holds twice the static LDS of the 256-thread one (33 KiB against 17): at very large frames it fails
  // the budget -- or keeps whole-tile lists where 256 threads would still split the long ones -- while the narrower kernel fits.
  // The automatic choice then falls back to 256 threads rather than to the sorted-list rasteriser (a forced choice stays).
  if (!forced && bin_threads == 512 && static_bytes + dyn_bytes(want_split) > 65536u) {
    bin_threads = 256;
    bk = variant(bin_threads, &static_bytes);
  }
  if (static_bytes + dyn_bytes(false) > 65536u) return RDOOM_OK;  // not launched: the caller flags every pose as "bins incomplete"
  // split lists need two more counters per tile (frames beyond ~4 000 tiles -- 5K and up -- keep whole-tile lists)
  const bool split = want_split && static_bytes + dyn_bytes(true) <= 65536u;
  hipLaunchKernelGGL(bk, dim3(n_poses), dim3(bin_threads), dyn_bytes(split), st, recs, counts, cap, tiles_x,
                     tiles_y, tile_hdr, entries, entry_cap, hits, overflow, split ? 1u : 0u);
  // (launch_setup ended with a check of its own launches: an error here is this launch's.  It is REPORTED, not turned into
  // "bins incomplete": the configuration was validated above, so a failure means the device or the queue is in trouble)
  HIP_TRY(hipGetLastError());
  *launched = true, *used_split = split;
  return RDOOM_OK;
}

}  // namespace rdoom_dev
