// Exact short forms of the IEEE operations the fragment stage needs, for gfx950.  Each is either a theorem
// (stated at the function) or verified exhaustively on the hardware by rdoom_selftest_fastmath
// (csrc/hip/selftest.hip; tests/test_gpu_fastmath.py runs it with -m gpu).
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace rdoom_fm {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 evaluate the same IEEE operation as the scalar forms, per half
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat(float x) { return f32x2{x, x}; }

// Correctly rounded 1/x for 2^-100 <= |x| <= 2^100: v_rcp_f32 plus one Newton step.  Verified for every
// binary32 input in that range (selftest sweep 0).
__device__ __forceinline__ float exact_rcp(float x) {
  const float r0 = __builtin_amdgcn_rcpf(x);
  const float e = fmaf(-x, r0, 1.0f);
  return fmaf(r0, e, r0);
}
__device__ __forceinline__ f32x2 exact_rcp2(f32x2 x) {
  const f32x2 r0 = {__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
  const f32x2 e = pk_fma(-x, r0, splat(1.0f));
  return pk_fma(r0, e, r0);
}
// Correctly rounded 0.9f/x for 2^-100 <= |x| <= 2^100: q0 = 0.9 * rcp(x), one residual correction.
// Verified for every binary32 input in that range (selftest sweep 1).
__device__ __forceinline__ float exact_div09(float x) {
  const float r0 = __builtin_amdgcn_rcpf(x);
  const float q0 = 0.9f * r0;
  const float rem = fmaf(-x, q0, 0.9f);
  return fmaf(rem, r0, q0);
}
__device__ __forceinline__ f32x2 exact_div09_2(f32x2 x) {
  const f32x2 r0 = {__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
  const f32x2 q0 = splat(0.9f) * r0;
  const f32x2 rem = pk_fma(-x, q0, splat(0.9f));
  return pk_fma(rem, r0, q0);
}
__device__ __forceinline__ int cvt_floor_i32(float x) {  // (int)floorf(x) in one instruction
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// mod(x, y) for an integer-valued y in [1, 4096] without a division:
//   f0 = floor(x * RN(1/y)),  r = fma(-y, f0, x)   (y * f0 is an integer < 2^24, hence exact, when |x| < 2^23).
// THEOREM: if mod_cert(x, r, y, false) holds then f0 == floor(RN(x / y)), so r == x - y * floor(x / y) as the
// reference evaluates it.  Proof sketch: q0 = RN(x * RN(1/y)) and q = RN(x / y) both lie within |x/y| * 2^-22
// of Q = x/y; if their floors differed an integer n in {f0, f0 + 1} would lie between them, so the true
// remainder y * (Q - f0) would be within |x| * 2^-22 of 0 or of y.  The test keeps the *computed* remainder
// 2^-20 * max(|x|, y) away from both ends, which dominates its own rounding error (2^-24 relative) with room
// to spare.  Verified on dense near-boundary samples by the selftest (sweep 2).  p2 = the axis is a power
// of two: x * 2^-k is the quotient itself and nothing needs certifying.
__device__ __forceinline__ bool mod_cert(float x, float r, float y, bool p2) {
  const float lo = fmaxf(fabsf(x), y) * 0x1p-20f;
  const float hi = y - lo;
  return p2 | ((r >= lo) & (r <= hi) & (fabsf(x) < 0x1p23f));
}

}  // namespace rdoom_fm
