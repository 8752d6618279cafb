// Exhaustive search (all 2^32 binary32 bit patterns) for short instruction sequences that reproduce IEEE
// correctly-rounded division on gfx950, as candidates for the fragment kernel's 1/rw and 0.9/(dist+0.9).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/fastmath_exhaustive.hip -o /tmp/fm && /tmp/fm
// Prints, per candidate, the number of inputs (all / positive normal in [2^-100, 2^100]) whose result differs
// from the compiler's IEEE division, plus a few offending inputs.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#pragma clang fp contract(off)

constexpr int NCAND = 8;

__device__ __forceinline__ float cand(int c, float x) {
  const float r0 = __builtin_amdgcn_rcpf(x);
  switch (c) {
    case 0: {  // 1/x: one Newton step
      const float e = fmaf(-x, r0, 1.0f);
      return fmaf(r0, e, r0);
    }
    case 1: {  // 1/x: two Newton steps
      float e = fmaf(-x, r0, 1.0f);
      const float r1 = fmaf(r0, e, r0);
      e = fmaf(-x, r1, 1.0f);
      return fmaf(r1, e, r1);
    }
    case 2: {  // 0.9/x: q0 = 0.9*r0, one residual correction with r0
      const float q0 = 0.9f * r0;
      const float rem = fmaf(-x, q0, 0.9f);
      return fmaf(rem, r0, q0);
    }
    case 3: {  // 0.9/x: refined reciprocal r1, q0 = 0.9*r1, one residual correction with r1
      const float e = fmaf(-x, r0, 1.0f);
      const float r1 = fmaf(r0, e, r0);
      const float q0 = 0.9f * r1;
      const float rem = fmaf(-x, q0, 0.9f);
      return fmaf(rem, r1, q0);
    }
    case 4: {  // 0.9/x: two residual corrections with r0
      const float q0 = 0.9f * r0;
      float rem = fmaf(-x, q0, 0.9f);
      const float q1 = fmaf(rem, r0, q0);
      rem = fmaf(-x, q1, 0.9f);
      return fmaf(rem, r0, q1);
    }
    case 5: {  // 1/x: raw v_rcp_f32
      return r0;
    }
    case 6: {  // 0.9/x: raw
      return 0.9f * r0;
    }
    default: {  // 0.9/x: r1 refined, q0 = 0.9*r1, residual correction with r0 (cheaper dependency chain)
      const float e = fmaf(-x, r0, 1.0f);
      const float r1 = fmaf(r0, e, r0);
      const float q0 = 0.9f * r1;
      const float rem = fmaf(-x, q0, 0.9f);
      return fmaf(rem, r0, q0);
    }
  }
}

__device__ __forceinline__ float truth(int c, float x) { return (c == 0 || c == 1 || c == 5) ? 1.0f / x : 0.9f / x; }

__global__ void sweep(unsigned long long *bad_all, unsigned long long *bad_rng, uint32_t *samples) {
  const uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long la[NCAND] = {}, lr[NCAND] = {};
  for (uint64_t b = blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
    const float x = __uint_as_float((uint32_t)b);
    const float ax = fabsf(x);
    const bool in_rng = ax >= 0x1p-100f && ax <= 0x1p100f;
#pragma unroll
    for (int c = 0; c < NCAND; c++) {
      const float got = cand(c, x), want = truth(c, x);
      const bool same = __float_as_uint(got) == __float_as_uint(want) || (got != got && want != want);
      if (!same) {
        la[c]++;
        if (in_rng) {
          lr[c]++;
          const uint32_t slot = atomicAdd(&samples[c * 64], 1u);
          if (slot < 20) samples[c * 64 + 1 + slot] = (uint32_t)b;
        }
      }
    }
  }
  for (int c = 0; c < NCAND; c++) {
    if (la[c]) atomicAdd(&bad_all[c], la[c]);
    if (lr[c]) atomicAdd(&bad_rng[c], lr[c]);
  }
}

// packed-math equivalence: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 == scalar fmaf / * / + on pseudo-random bit patterns
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ void packed_check(unsigned long long *bad) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long n = 0;
  for (uint32_t i = 0; i < 4096; i++) {
    const uint32_t s = t * 4096u + i;
    // exponents restricted to a band around 1 so products/sums are ordinary numbers most of the time,
    // every 16th sample uses raw bits (denormals, inf, nan included)
    auto gen = [&](uint32_t k) {
      uint32_t b = mix(s * 6u + k);
      if ((i & 15u) != 0u) b = (b & 0x807FFFFFu) | ((100u + (mix(b) % 56u)) << 23);
      return __uint_as_float(b);
    };
    f2 a = {gen(0), gen(1)}, b = {gen(2), gen(3)}, c = {gen(4), gen(5)};
    f2 f, m, d;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(f) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    for (int k = 0; k < 2; k++) {
      const float ff = fmaf(a[k], b[k], c[k]), mm = a[k] * b[k], dd = a[k] + b[k];
      auto same = [](float p, float q) { return __float_as_uint(p) == __float_as_uint(q) || (p != p && q != q); };
      if (!same(ff, f[k])) n++;
      if (!same(mm, m[k])) n++;
      if (!same(dd, d[k])) n++;
    }
  }
  if (n) atomicAdd(bad, n);
}

int main() {
  unsigned long long *d_all, *d_rng, *d_pk;
  uint32_t *d_samples;
  hipMalloc(&d_all, NCAND * 8);
  hipMalloc(&d_rng, NCAND * 8);
  hipMalloc(&d_pk, 8);
  hipMalloc(&d_samples, NCAND * 64 * 4);
  hipMemset(d_all, 0, NCAND * 8);
  hipMemset(d_rng, 0, NCAND * 8);
  hipMemset(d_pk, 0, 8);
  hipMemset(d_samples, 0, NCAND * 64 * 4);
  sweep<<<4096, 256>>>(d_all, d_rng, d_samples);
  packed_check<<<1024, 256>>>(d_pk);
  unsigned long long all[NCAND], rng[NCAND], pk;
  uint32_t samples[NCAND * 64];
  hipMemcpy(all, d_all, sizeof all, hipMemcpyDeviceToHost);
  hipMemcpy(rng, d_rng, sizeof rng, hipMemcpyDeviceToHost);
  hipMemcpy(&pk, d_pk, 8, hipMemcpyDeviceToHost);
  hipMemcpy(samples, d_samples, sizeof samples, hipMemcpyDeviceToHost);
  const char *names[NCAND] = {"1/x newton1", "1/x newton2", "0.9/x r0 corr1", "0.9/x r1 corr1(r1)", "0.9/x r0 corr2",
                              "1/x raw rcp", "0.9/x raw", "0.9/x r1 corr1(r0)"};
  for (int c = 0; c < NCAND; c++) {
    printf("%-22s mismatches: all inputs %llu, |x| in [2^-100,2^100] %llu ;", names[c], all[c], rng[c]);
    const uint32_t n = samples[c * 64] < 20 ? samples[c * 64] : 20;
    for (uint32_t i = 0; i < n && i < 8; i++) printf(" %08x", samples[c * 64 + 1 + i]);
    printf("\n");
  }
  printf("packed-vs-scalar mismatches: %llu\n", pk);
  return 0;
}
