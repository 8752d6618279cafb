// rdoom_selftest_fastmath: on-device verification of the exact short forms in fastmath.hpp.
//   sweep 0  exact_rcp(x)   == 1.0f / x   for every binary32 x with 2^-100 <= |x| <= 2^100   (2^32 patterns visited)
//   sweep 1  exact_div09(x) == 0.9f / x   likewise
//   sweep 2  mod certificate: for integer sizes y and x placed on / next to multiples of y (+- 0..12 ulps, where
//            floor(x * RN(1/y)) and floor(x / y) can disagree) and at pseudo-random positions:
//            mod_cert(x, r, y) must imply floor(x * RN(1/y)) == floor(x / y)
//   sweep 3  v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 == scalar fmaf / * / + on pseudo-random operands
//   sweep 4  the binning kernel's pair -> tile row: (int)((t + 0.5f) * v_rcp_f32(ntx)) == t / ntx for every t < 8192,
//            1 <= ntx <= 128 (bin.hip; mismatches are added to sweep 3's count)
// Takes a few seconds on an MI355X; used by tests/test_gpu_fastmath.py.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../common.hpp"
#include "fastmath.hpp"

#pragma clang fp contract(off)

namespace {
using namespace rdoom_fm;

__device__ __forceinline__ bool same_bits(float a, float b) {
  return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}

__global__ void sweep_div(unsigned long long *out) {  // out[0] rcp mismatches, out[1] div09 mismatches, out[2] inputs in range
  const uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long bad0 = 0, bad1 = 0, n = 0;
  for (uint64_t b = blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
    const float x = __uint_as_float((uint32_t)b);
    const float ax = fabsf(x);
    if (!(ax >= 0x1p-100f && ax <= 0x1p100f)) continue;
    n++;
    if (!same_bits(exact_rcp(x), 1.0f / x)) bad0++;
    if (!same_bits(exact_div09(x), 0.9f / x)) bad1++;
    const f32x2 r2 = exact_rcp2(f32x2{x, -x}), d2 = exact_div09_2(f32x2{x, -x});
    if (!same_bits(r2.x, 1.0f / x) || !same_bits(r2.y, 1.0f / -x)) bad0++;
    if (!same_bits(d2.x, 0.9f / x) || !same_bits(d2.y, 0.9f / -x)) bad1++;
  }
  atomicAdd(&out[0], bad0);
  atomicAdd(&out[1], bad1);
  atomicAdd(&out[2], n);
}

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// out[3] certificate violations, out[4] samples, out[5] samples that passed the certificate,
// out[6] samples where the two floors differ (the certificate must have rejected every one of them)
__global__ void sweep_mod(unsigned long long *out) {
  // y = blockIdx.x + 1 (1..4096); threads walk multiples n * y and their neighbourhoods
  const float y = (float)(blockIdx.x + 1u);
  const float ry = exact_rcp(y);
  unsigned long long viol = 0, samples = 0, passed = 0, differ = 0;
  auto probe = [&](float x) {
    const float f0 = floorf(x * ry);
    const float r = fmaf(-y, f0, x);
    const float fl = floorf(x / y);
    samples++;
    const bool c = mod_cert(x, r, y, false);
    if (f0 != fl) differ++;
    if (c) {
      passed++;
      if (f0 != fl || !same_bits(r, x - y * fl)) viol++;
    }
  };
  for (int n = (int)threadIdx.x - 8192; n <= 8192; n += (int)blockDim.x) {
    const float base = (float)n * y;  // exact for |n * y| < 2^24, else merely a float near a multiple
    if (fabsf(base) >= 0x1p23f) continue;
    uint32_t bits = __float_as_uint(base);
    for (int k = -12; k <= 12; k++) {
      const uint32_t bk = base == 0.0f ? (k < 0 ? 0x80000000u | (uint32_t)(-k) * 0x00100000u : (uint32_t)k * 0x00100000u)
                                       : (uint32_t)((int32_t)bits + (base < 0.0f ? -k : k));
      probe(__uint_as_float(bk));
    }
    // pseudo-random positions inside the period
    for (int k = 0; k < 8; k++) {
      const uint32_t h = mix((uint32_t)(n + 8192) * 4099u + (uint32_t)k * 31u + blockIdx.x * 977u);
      probe(base + y * ((float)(h >> 8) * 0x1p-24f));
    }
  }
  atomicAdd(&out[3], viol);
  atomicAdd(&out[4], samples);
  atomicAdd(&out[5], passed);
  atomicAdd(&out[6], differ);
}

__global__ void sweep_packed(unsigned long long *out) {  // out[7] mismatches
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long n = 0;
  for (uint32_t i = 0; i < 1024; i++) {
    const uint32_t s = t * 1024u + i;
    auto gen = [&](uint32_t k) {
      uint32_t b = mix(s * 6u + k);
      if ((i & 15u) != 0u) b = (b & 0x807FFFFFu) | ((100u + (mix(b) % 56u)) << 23);  // mostly ordinary magnitudes
      return __uint_as_float(b);
    };
    const f32x2 a = {gen(0), gen(1)}, b = {gen(2), gen(3)}, c = {gen(4), gen(5)};
    const f32x2 f = pk_fma(a, b, c), m = a * b, d = a + b;
    for (int k = 0; k < 2; k++) {
      float ak = a[k], bk = b[k], ck = c[k];
      asm volatile("" : "+v"(ak), "+v"(bk), "+v"(ck));  // keep the scalar forms scalar
      if (!same_bits(fmaf(ak, bk, ck), f[k])) n++;
      if (!same_bits(ak * bk, m[k])) n++;
      if (!same_bits(ak + bk, d[k])) n++;
    }
  }
  atomicAdd(&out[7], n);
}

__global__ void sweep_tile_index(unsigned long long *out) {  // out[7] += mismatches
  const uint32_t ntx = blockIdx.x + 1u;  // 1..128
  unsigned long long n = 0;
  for (uint32_t t = threadIdx.x; t < 8192u; t += blockDim.x) {
    const int ty = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)ntx));
    if ((uint32_t)ty != t / ntx) n++;
  }
  if (n) atomicAdd(&out[7], n);
}

}  // namespace

extern "C" rdoom_status rdoom_selftest_fastmath(uint64_t out_counts[8]) {
  if (!out_counts) return rdoom::fail(RDOOM_BAD_ARG, "out_counts is null");
  unsigned long long *d = nullptr;
  hipError_t e = hipMalloc((void **)&d, 8 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(d, 0, 8 * sizeof(unsigned long long));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(sweep_div, dim3(4096), dim3(256), 0, nullptr, d);
    hipLaunchKernelGGL(sweep_mod, dim3(4096), dim3(256), 0, nullptr, d);
    hipLaunchKernelGGL(sweep_packed, dim3(1024), dim3(256), 0, nullptr, d);
    hipLaunchKernelGGL(sweep_tile_index, dim3(128), dim3(256), 0, nullptr, d);
    e = hipGetLastError();
  }
  unsigned long long h[8] = {};
  if (e == hipSuccess) e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  if (d) (void)hipFree(d);
  if (e != hipSuccess) return rdoom::fail(RDOOM_HIP_ERROR, "selftest failed: %s", hipGetErrorString(e));
  for (int i = 0; i < 8; i++) out_counts[i] = h[i];
  return RDOOM_OK;
}
