// SSECTOR -> convex floor/ceiling polygon on the device (gfx950).
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

}  // namespace rdoom::game
