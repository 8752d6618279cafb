// Level tessellation on the device (gfx950): SSECTOR -> convex floor/ceiling polygon, SEG -> wall quads.
//
// Restates LevelWalker::subsector's implicit-point search and points_to_polygon
// (wad/src/visitor.rs:672-699, 1184-1259; math/src/line.rs:43-84) as one wavefront per sub-sector:
// lanes enumerate the pairs of BSP half-plane lines, intersect them, test the intersection against
// every BSP line (tolerance 1e-3) and every seg line (tolerance 0.1) and append the survivors to an
// LDS point list IN PAIR ORDER (ballot + prefix popcount), so the list equals the host's.  Lane 0
// then runs the pinned insertion sort / area filter / outward bias.  binary32, no contraction; the
// result is bit-identical to the host walk (tests/test_gpu_tessellate.py).
#include <hip/hip_runtime.h>

#include <vector>

#include "../host/game_level.hpp"

#pragma clang fp contract(off)

namespace {

constexpr int MAX_POINTS = 512;  // explicit (2 per seg) + implicit points per sub-sector
constexpr float kBspTol = 1e-3f, kSegTol = 0.1f, kPolyBias = 0.64f * 3e-4f, kEps = 1.1920929e-7f;

struct LeafDesc {
  uint32_t line_first, n_bsp;  // bsp lines   [line_first, line_first + n_bsp)
  uint32_t seg_first, n_seg;   // seg lines   [seg_first, seg_first + n_seg)
  uint32_t pt_first, n_pts;    // explicit points
  uint32_t out_first, pad;     // output slot: MAX_POINTS points per leaf
};

__device__ __forceinline__ float sdist(const float4 l, float px, float py) {  // Line2::signed_distance
  return (px * l.w - py * l.z) + (l.z * l.y - l.w * l.x);                      // l = (ox, oy, dx, dy)
}
__device__ __forceinline__ float mag(float x, float y) { return sqrtf(x * x + y * y); }

__device__ bool poly_less(float2 a, float2 b, float2 c) {  // visitor.rs:1195-1224, true iff Less
  const float acx = a.x - c.x, acy = a.y - c.y, bcx = b.x - c.x, bcy = b.y - c.y;
  if (acx >= 0.0f && bcx < 0.0f) return true;
  if (acx < 0.0f && bcx >= 0.0f) return false;
  if (acx == 0.0f && bcx == 0.0f) {
    if (acy >= 0.0f || bcy >= 0.0f) return a.y > b.y;
    return b.y > a.y;
  }
  return acx * bcy - acy * bcx < 0.0f;
}

__device__ float2 center_of(const float2 *p, int n) {
  float cx = 0.0f, cy = 0.0f;
  for (int i = 0; i < n; i++) {
    cx += p[i].x;
    cy += p[i].y;
  }
  const float fn = (float)n;
  return make_float2(cx / fn, cy / fn);
}

__global__ __launch_bounds__(64) void subsector_polygon_kernel(const LeafDesc *__restrict__ leaves,
                                                               const float4 *__restrict__ lines,
                                                               const float2 *__restrict__ points,
                                                               float2 *__restrict__ out_pts,
                                                               uint32_t *__restrict__ out_n,
                                                               uint32_t *__restrict__ overflow) {
  __shared__ float2 pts[MAX_POINTS];
  __shared__ float2 simp[MAX_POINTS];
  __shared__ int npts;
  const LeafDesc L = leaves[blockIdx.x];
  const int lane = threadIdx.x;
  for (uint32_t i = lane; i < L.n_pts && i < (uint32_t)MAX_POINTS; i += 64) pts[i] = points[L.pt_first + i];
  if (lane == 0) npts = (int)min(L.n_pts, (uint32_t)MAX_POINTS);
  __syncthreads();
  const int nb = (int)L.n_bsp;
  const int npairs = nb * (nb - 1) / 2;
  for (int base = 0; base < npairs; base += 64) {
    const int k = base + lane;
    bool inside = false;
    float2 p = make_float2(0.0f, 0.0f);
    if (k < npairs) {
      int i = 0, rem = k;  // pair k -> (i, j), i < j, in the host's loop order
      while (rem >= nb - 1 - i) {
        rem -= nb - 1 - i;
        i++;
      }
      const int j = i + 1 + rem;
      const float4 l1 = lines[L.line_first + i], l2 = lines[L.line_first + j];
      const float den = l1.z * l2.w - l1.w * l2.z;  // Line2::intersect_offset
      if (!(fabsf(den) < 1e-16f)) {
        const float ex = l2.x - l1.x, ey = l2.y - l1.y;
        const float off = (ex * l2.w - ey * l2.z) / den;
        p = make_float2(l1.x + l1.z * off, l1.y + l1.w * off);
        inside = true;
        for (int q = 0; q < nb && inside; q++) inside = sdist(lines[L.line_first + q], p.x, p.y) >= -kBspTol;
        for (uint32_t q = 0; q < L.n_seg && inside; q++) inside = sdist(lines[L.seg_first + q], p.x, p.y) <= kSegTol;
      }
    }
    const unsigned long long m = __ballot(inside);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    const int at = npts + before;
    if (inside) {
      if (at < MAX_POINTS)
        pts[at] = p;
      else
        atomicAdd(overflow, 1u);
    }
    __syncthreads();
    if (lane == 0) npts = min(npts + (int)__popcll(m), MAX_POINTS);
    __syncthreads();
  }
  if (lane != 0) return;
  // ---- points_to_polygon, serial (n is a few dozen) -------------------------------------------------
  int n = npts;
  uint32_t out_count = 0;
  float2 *dst = out_pts + L.out_first;
  if (n >= 2) {
    const float2 c = center_of(pts, n);
    for (int i = 1; i < n; i++)
      for (int j = i; j > 0 && poly_less(pts[j], pts[j - 1], c); j--) {
        const float2 t = pts[j];
        pts[j] = pts[j - 1];
        pts[j - 1] = t;
      }
    int ns = 0;
    simp[ns++] = pts[0];
    float2 cur = pts[1];
    float area = 0.0f;
    for (int i = 2; i < n; i++) {
      const float2 nxt = pts[i], prev = simp[ns - 1];
      const float na = ((nxt.x - cur.x) * (cur.y - prev.y) - (nxt.y - cur.y) * (cur.x - prev.x)) * 0.5f;
      if (na >= 0.0f) {
        if (area + na > 1.024e-5f) {
          area = 0.0f;
          simp[ns++] = cur;
        } else {
          area += na;
        }
      }
      cur = nxt;
    }
    simp[ns++] = pts[n - 1];
    if (ns >= 3) {
      while (ns > 1 && mag(simp[0].x - simp[ns - 1].x, simp[0].y - simp[ns - 1].y) < 0.0032f) ns--;
      const float2 c2 = center_of(simp, ns);
      for (int i = 0; i < ns; i++) {
        const float vx = simp[i].x - c2.x, vy = simp[i].y - c2.y;
        float m = mag(vx, vy);
        if (!(m > kEps)) m = kEps;
        dst[i] = make_float2(simp[i].x + (vx / m) * kPolyBias, simp[i].y + (vy / m) * kPolyBias);
      }
      out_count = (uint32_t)ns;
    }
  }
  out_n[blockIdx.x] = out_count;
}

// =================================================================================================
// SEG -> wall quads (+ sky quads).  One lane per seg restates LevelWalker::seg (visitor.rs:711-837: which
// of lower / upper / middle / one-sided wall and sky quads exist, their height spans, pegging and owner
// object), wall_quad (visitor.rs:839-937: end points pushed outward by POLY_BIAS along the wall, heights
// / 100 +- POLY_BIAS, s from seg offset + biased length * 100, t by pegging + y offset, fake contrast,
// scroll) and sky_quad (visitor.rs:987-1008).  Every table lookup (sectors, sidedefs, texture heights,
// dynamic-sector ranges) was resolved into SegInput by the host; the arithmetic is here.  binary32, no
// contraction: bit-identical to the host walk (tests/test_gpu_tessellate.py).
// =================================================================================================
using rdoom::wad::SegGeometry;
using rdoom::wad::SegInput;
using rdoom::wad::SegQuadGeometry;
using rdoom::wad::SegSkyGeometry;

enum SegPeg { PEG_TOP, PEG_BOTTOM, PEG_BOTTOM_LOWER, PEG_TOP_FLOAT, PEG_BOTTOM_FLOAT };

__device__ __forceinline__ float from_wad_height_d(int16_t x) { return (float)x / 100.0f; }

__device__ void seg_wall_quad(const SegInput &in, SegQuadGeometry &out, uint32_t object_id, int16_t low_i, int16_t high_i,
                              int slot, SegPeg peg, bool blocker) {
  out.valid = 0;
  if (low_i >= high_i) return;
  const int16_t th = in.tex_h[slot];
  if (th == -2) return;  // "No such wall texture": the quad is skipped
  const bool textured = th >= 0;
  float dx = in.v2x - in.v1x, dy = in.v2y - in.v1y;
  float m = mag(dx, dy);
  if (!(m > kEps)) m = kEps;  // normalize_or_zero
  dx = dx / m;
  dy = dy / m;
  const float bx = dx * kPolyBias, by = dy * kPolyBias;
  const float v1x = in.v1x + (-bx), v1y = in.v1y + (-by), v2x = in.v2x + bx, v2y = in.v2y + by;
  float low, high;
  if (textured && peg == PEG_TOP_FLOAT) {
    low = from_wad_height_d((int16_t)(low_i + in.y_offset));
    high = from_wad_height_d((int16_t)(low_i + th + in.y_offset));
  } else if (textured && peg == PEG_BOTTOM_FLOAT) {
    low = from_wad_height_d((int16_t)(high_i + in.y_offset - th));
    high = from_wad_height_d((int16_t)(high_i + in.y_offset));
  } else {
    low = from_wad_height_d(low_i);
    high = from_wad_height_d(high_i);
  }
  uint32_t contrast = 0;
  if (!(in.flags & rdoom::wad::SEG_LIGHT_EFFECT)) {
    if (fabsf(v1x - v2x) < kEps)
      contrast = 1;
    else if (fabsf(v1y - v2y) < kEps)
      contrast = 2;
  }
  const float height = (high - low) * 100.0f;
  const float s1 = (float)in.seg_offset + (float)in.x_offset;
  const float s2 = s1 + mag(v2x - v1x, v2y - v1y) * 100.0f;
  const float fth = (float)th;
  float t1, t2;
  if (!textured || peg == PEG_TOP) {
    t1 = height;
    t2 = 0.0f;
  } else if (peg == PEG_BOTTOM) {
    t1 = fth;
    t2 = fth - height;
  } else if (peg == PEG_BOTTOM_LOWER) {
    const float sector_height = (float)(int16_t)(in.ceiling - in.floor);
    t1 = fth + sector_height;
    t2 = fth - height + sector_height;
  } else {
    t1 = fth;
    t2 = 0.0f;
  }
  t1 = t1 + (float)in.y_offset;
  t2 = t2 + (float)in.y_offset;
  out.valid = 1;
  out.object_id = object_id;
  out.v1x = v1x, out.v1y = v1y, out.v2x = v2x, out.v2y = v2y;
  out.s1 = s1, out.t1 = t1, out.s2 = s2, out.t2 = t2;
  out.low = low - kPolyBias;
  out.high = high + kPolyBias;
  out.scroll = (in.flags & rdoom::wad::SEG_SCROLL) ? 35.0f : 0.0f;
  out.contrast = contrast;
  out.slot = (uint32_t)slot;
  out.blocker = blocker ? 1u : 0u;
}

__device__ void seg_sky_quad(const SegInput &in, SegSkyGeometry &out, uint32_t object_id, int16_t low, int16_t high) {
  out.valid = 0;
  if (low >= high) return;
  float ex = in.v2x - in.v1x, ey = in.v2y - in.v1y;
  float m = mag(ex, ey);
  if (!(m > kEps)) m = kEps;
  ex = ex / m;
  ey = ey / m;
  const float bx = ex * kPolyBias * 16.0f, by = ey * kPolyBias * 16.0f;
  const float nx = -ey, ny = ex;
  const float nbx = nx * kPolyBias * 16.0f, nby = ny * kPolyBias * 16.0f;
  out.valid = 1;
  out.object_id = object_id;
  out.v1x = in.v1x + (nbx - bx), out.v1y = in.v1y + (nby - by);
  out.v2x = in.v2x + (nbx + bx), out.v2y = in.v2y + (nby + by);
  out.low = from_wad_height_d(low);
  out.high = from_wad_height_d(high);
}

__global__ __launch_bounds__(64) void seg_quads_kernel(const SegInput *__restrict__ inputs, SegGeometry *__restrict__ outputs,
                                                       uint32_t n) {
  using namespace rdoom::wad;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SegInput in = inputs[i];
  SegGeometry g;
  for (int k = 0; k < 3; k++) g.quad[k].valid = 0;
  g.sky[0].valid = g.sky[1].valid = 0;
  if (in.flags & SEG_VALID) {
    const bool unpeg_lower = (in.flags & SEG_UNPEG_LOWER) != 0, unpeg_upper = (in.flags & SEG_UNPEG_UPPER) != 0;
    const int16_t max_height = (int16_t)(in.f_ceil_hi - in.f_floor_lo);  // SectorInfo::max_height
    if (!(in.flags & SEG_HAS_BACK)) {
      seg_wall_quad(in, g.quad[0], unpeg_lower ? in.f_floor_id : in.f_ceil_id,
                    unpeg_lower ? in.floor : (int16_t)(in.ceiling - max_height),
                    unpeg_lower ? (int16_t)(in.floor + max_height) : in.ceiling, 2, unpeg_lower ? PEG_BOTTOM : PEG_TOP, true);
      if (in.flags & SEG_F_CEIL_SKY) seg_sky_quad(in, g.sky[0], in.f_ceil_id, in.ceiling, in.mx);
      if (in.flags & SEG_F_FLOOR_SKY) seg_sky_quad(in, g.sky[1], in.f_floor_id, in.mn, in.floor);
    } else {
      if ((in.flags & SEG_F_CEIL_SKY) && !(in.flags & SEG_B_CEIL_SKY)) seg_sky_quad(in, g.sky[0], in.f_ceil_id, in.ceiling, in.mx);
      if ((in.flags & SEG_F_FLOOR_SKY) && !(in.flags & SEG_B_FLOOR_SKY)) seg_sky_quad(in, g.sky[1], in.f_floor_id, in.mn, in.floor);
      int16_t fl, ce;
      if (in.b_floor_hi > in.f_floor_lo) {
        seg_wall_quad(in, g.quad[0], in.b_floor_id, (int16_t)(in.back_floor - in.b_floor_hi + in.f_floor_lo), in.back_floor, 0,
                      unpeg_lower ? PEG_BOTTOM_LOWER : PEG_TOP, true);
        fl = in.back_floor;
      } else {
        fl = in.floor;
      }
      if (in.back_ceiling < in.ceiling) {
        if (!(in.flags & SEG_B_CEIL_SKY))
          seg_wall_quad(in, g.quad[1], in.b_ceil_id, in.back_ceiling, in.ceiling, 1, unpeg_upper ? PEG_TOP : PEG_BOTTOM, true);
        ce = in.back_ceiling;
      } else {
        ce = in.ceiling;
      }
      SegPeg peg;
      if (unpeg_lower)
        peg = in.tex_h[1] == -1 ? PEG_TOP_FLOAT : PEG_BOTTOM;
      else
        peg = in.tex_h[0] == -1 ? PEG_BOTTOM_FLOAT : PEG_TOP;
      seg_wall_quad(in, g.quad[2], unpeg_lower ? in.f_floor_id : in.f_ceil_id, fl, ce, 2, peg, (in.flags & SEG_IMPASSABLE) != 0);
    }
  }
  outputs[i] = g;
}

struct DevBuf {
  void *p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc_copy(const void *src, size_t bytes) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess && src) e = hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
    return e;
  }
};

}  // namespace

namespace rdoom::game {

std::vector<std::vector<wad::Pnt2f>> tessellate_on_device(const wad::Level &level,
                                                          const std::vector<wad::LevelWalker::LeafInput> &leaves) {
  using wad::WadError;
  std::vector<LeafDesc> descs;
  std::vector<float4> lines;
  std::vector<float2> points;
  for (const auto &leaf : leaves) {
    LeafDesc d{};
    d.line_first = (uint32_t)lines.size();
    d.n_bsp = (uint32_t)leaf.bsp_lines.size();
    for (const wad::Line2f &l : leaf.bsp_lines) lines.push_back(make_float4(l.origin.x, l.origin.y, l.displace.x, l.displace.y));
    const wad::WadSubsector ss = level.subsectors[leaf.subsector];
    d.seg_first = (uint32_t)lines.size();
    d.pt_first = (uint32_t)points.size();
    for (uint32_t s = 0; s < ss.num_segs; s++) {
      const wad::WadSeg &sg = level.segs[ss.first_seg + s];
      const auto v1 = level.vertex(sg.start_vertex), v2 = level.vertex(sg.end_vertex);
      if (!v1 || !v2) throw WadError(RDOOM_BAD_LEVEL, "tessellate: seg without vertices");
      const wad::Line2f l = wad::Line2f::from_two_points(*v1, *v2);
      lines.push_back(make_float4(l.origin.x, l.origin.y, l.displace.x, l.displace.y));
      points.push_back(make_float2(v1->x, v1->y));
      points.push_back(make_float2(v2->x, v2->y));
    }
    d.n_seg = ss.num_segs;
    d.n_pts = 2u * ss.num_segs;
    d.out_first = (uint32_t)descs.size() * MAX_POINTS;
    descs.push_back(d);
  }
  std::vector<std::vector<wad::Pnt2f>> polygons(level.subsectors.size());
  if (descs.empty()) return polygons;
  DevBuf d_desc, d_lines, d_points, d_out, d_n, d_over;
  hipError_t e = d_desc.alloc_copy(descs.data(), descs.size() * sizeof(LeafDesc));
  if (e == hipSuccess) e = d_lines.alloc_copy(lines.data(), lines.size() * sizeof(float4));
  if (e == hipSuccess) e = d_points.alloc_copy(points.data(), points.size() * sizeof(float2));
  if (e == hipSuccess) e = d_out.alloc_copy(nullptr, descs.size() * MAX_POINTS * sizeof(float2));
  if (e == hipSuccess) e = d_n.alloc_copy(nullptr, descs.size() * sizeof(uint32_t));
  const uint32_t zero = 0;
  if (e == hipSuccess) e = d_over.alloc_copy(&zero, sizeof zero);
  if (e != hipSuccess) throw WadError(RDOOM_HIP_ERROR, std::string("tessellate: ") + hipGetErrorString(e));
  hipLaunchKernelGGL(subsector_polygon_kernel, dim3((uint32_t)descs.size()), dim3(64), 0, 0,
                     (const LeafDesc *)d_desc.p, (const float4 *)d_lines.p, (const float2 *)d_points.p,
                     (float2 *)d_out.p, (uint32_t *)d_n.p, (uint32_t *)d_over.p);
  std::vector<uint32_t> counts(descs.size());
  std::vector<float2> out(descs.size() * MAX_POINTS);
  uint32_t over = 0;
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(counts.data(), d_n.p, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out.data(), d_out.p, out.size() * sizeof(float2), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(&over, d_over.p, sizeof over, hipMemcpyDeviceToHost);
  if (e != hipSuccess) throw WadError(RDOOM_HIP_ERROR, std::string("tessellate: ") + hipGetErrorString(e));
  if (over) throw WadError(RDOOM_BAD_LEVEL, "tessellate: sub-sector with more than 512 candidate points");
  for (size_t i = 0; i < leaves.size(); i++) {
    auto &poly = polygons[leaves[i].subsector];
    poly.clear();
    for (uint32_t k = 0; k < counts[i]; k++) poly.push_back({out[i * MAX_POINTS + k].x, out[i * MAX_POINTS + k].y});
  }
  return polygons;
}

// SEG -> wall quads on the device: one lane per recorded seg (see seg_quads_kernel)
std::vector<wad::SegGeometry> tessellate_segs_on_device(const std::vector<wad::SegInput> &inputs) {
  using wad::WadError;
  std::vector<wad::SegGeometry> out(inputs.size());
  if (inputs.empty()) return out;
  DevBuf d_in, d_out;
  hipError_t e = d_in.alloc_copy(inputs.data(), inputs.size() * sizeof(wad::SegInput));
  if (e == hipSuccess) e = d_out.alloc_copy(nullptr, out.size() * sizeof(wad::SegGeometry));
  if (e != hipSuccess) throw WadError(RDOOM_HIP_ERROR, std::string("tessellate segs: ") + hipGetErrorString(e));
  const uint32_t n = (uint32_t)inputs.size();
  hipLaunchKernelGGL(seg_quads_kernel, dim3((n + 63u) / 64u), dim3(64), 0, 0, (const wad::SegInput *)d_in.p,
                     (wad::SegGeometry *)d_out.p, n);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(out.data(), d_out.p, out.size() * sizeof(wad::SegGeometry), hipMemcpyDeviceToHost);
  if (e != hipSuccess) throw WadError(RDOOM_HIP_ERROR, std::string("tessellate segs: ") + hipGetErrorString(e));
  return out;
}

}  // namespace rdoom::game
