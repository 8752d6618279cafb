// Issue-rate micro-benchmark for gfx950: how many cycles does a SIMD spend per wave64 instruction of each kind the
// renderer's hot kernels are made of?  (The two guides disagree with the PMC counters: "SIMD-32, 2 cycles per VALU" vs
// SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.0 quad-cycles.)  Every kernel runs REPS iterations of a body of 8 independent
// chains x UNROLL instructions of ONE kind per lane; grid = 256 CUs x 4 SIMDs x W waves, so that each SIMD holds W waves
// (W = 1, 2, 4, 8).  cycles per instruction per SIMD = kernel cycles (s_memtime around the loop, max over waves is close
// to wall) x 1 / (instructions per wave x W).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_valu tools/ubench_valu.hip && /tmp/ubench_valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e = (x);                                                              \
    if (e != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                         \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int REPS = 2000, UNROLL = 8, CHAINS = 8;

enum Kind { FMA, PK_FMA, MUL, PK_MUL, ADD_U32, AND_OR, CVT_FLR, FLOOR, RCP, MED3, CNDMASK, MIN3, LSHL_ADD, PERM, MAD_U24, FMA_SGPR, PK_FMA_SGPR, READLANE, DS_READ_U8, MIX_FMA_AND, CND_VCCSET, CND_E64, CMP_E64, CMP_VCC, SUB_CO, MIN_U32, MIN3_F32, BFE_I32, CVT_U32_F32, OR3, LSHLREV, FRACT, PK_ADD, MOV, BFI, LSHL_OR, ADD_F32, MAX_F32, CVT_F32_U32, DPP_MAX, AND_B32, OR_B32, FMA_CHAIN1, CMP_CND_PAIR, GLOAD_UBYTE, KINDS };
const char *kind_name[KINDS] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_pk_mul_f32", "v_add_u32", "v_and_or_b32", "v_cvt_flr_i32_f32", "v_floor_f32", "v_rcp_f32", "v_med3_f32", "v_cndmask_b32", "v_min3_u32", "v_lshl_add_u32", "v_perm_b32", "v_mad_u32_u24", "v_fma_f32 (sgpr operand)", "v_pk_fma_f32 (sgpr pair operand)", "v_readlane_b32", "ds_read_u8", "v_fma_f32 + v_and_b32 alternating", "v_cndmask_b32 vcc (vcc written by v_cmp per 8)", "v_cndmask_b32_e64 (sgpr pair mask)", "v_cmp_lt_u32_e64 -> sgpr pair", "v_cmp_lt_f32 -> vcc", "v_sub_co_u32 (carry to vcc)", "v_min_u32", "v_min3_f32", "v_bfe_i32", "v_cvt_u32_f32", "v_or3_b32", "v_lshlrev_b32", "v_fract_f32", "v_pk_add_f32", "v_mov_b32", "v_bfi_b32", "v_lshl_or_b32", "v_add_f32", "v_max_f32", "v_cvt_f32_u32", "v_max_u32 dpp quad_perm", "v_and_b32", "v_or_b32", "v_fma_f32 ONE dependent chain", "v_cmp_lt_u32 vcc + v_cndmask (pair)", "global_load_ubyte gather (64 KiB window) + v_and/v_add"};

template <int K>
__global__ __launch_bounds__(64) void bench(float *out, unsigned long long *cycles, float seed, unsigned useed) {
  __shared__ unsigned char lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned char)(i * 7);
  __syncthreads();
  float a[CHAINS];
  f32x2 p[CHAINS];
  unsigned u[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) {
    a[c] = seed + (float)(threadIdx.x + c);
    p[c] = f32x2{a[c], a[c] + 0.5f};
    u[c] = useed + threadIdx.x * 3u + (unsigned)c;
  }
  const unsigned long long mask64 = ((unsigned long long)useed << 32) | (unsigned long long)(useed * 2654435761u);  // (kernel argument arithmetic: SGPR pair)
  const float s0 = seed * 0.999f, s1 = seed * 1.001f;  // wave-uniform (kernel argument arithmetic): SGPR operands
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < REPS; r++) {
#pragma unroll
    for (int k = 0; k < UNROLL; k++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) {
        if (K == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(a[(c + 1) % CHAINS]), "v"(a[(c + 2) % CHAINS]));
        if (K == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(p[(c + 1) % CHAINS]), "v"(p[(c + 2) % CHAINS]));
        if (K == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(a[(c + 1) % CHAINS]));
        if (K == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[c]) : "v"(p[(c + 1) % CHAINS]));
        if (K == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]));
        if (K == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]), "v"(u[(c + 2) % CHAINS]));
        if (K == CVT_FLR) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(u[c]) : "v"(a[c]));
        if (K == FLOOR) asm volatile("v_floor_f32 %0, %0" : "+v"(a[c]));
        if (K == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[c]));
        if (K == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(a[(c + 1) % CHAINS]), "v"(a[(c + 2) % CHAINS]));
        if (K == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]));
        if (K == MIN3) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]), "v"(u[(c + 2) % CHAINS]));
        if (K == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]));
        if (K == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]), "v"(u[(c + 2) % CHAINS]));
        if (K == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]), "v"(u[(c + 2) % CHAINS]));
        if (K == FMA_SGPR) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "s"(s0), "v"(a[(c + 2) % CHAINS]));
        if (K == PK_FMA_SGPR) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "s"(f32x2{s0, s1}), "v"(p[(c + 2) % CHAINS]));
        if (K == READLANE) {
          unsigned s;
          asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(u[c]));
          asm volatile("" ::"s"(s));
        }
        if (K == DS_READ_U8) u[c] = lds[(u[c] * 0x9E3779B1u) >> 20];
        if (K == CND_VCCSET) {
          if (c == 0) asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(u[0]), "v"(u[1]) : "vcc");
          asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]) : );
        }
        if (K == CND_E64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]), "s"(mask64));
        if (K == CMP_E64) {
          unsigned long long m;
          asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(u[c]), "v"(u[(c + 1) % CHAINS]));
          asm volatile("" ::"s"(m));
        }
        if (K == CMP_VCC) asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(a[c]), "v"(a[(c + 1) % CHAINS]) : "vcc");
        if (K == SUB_CO) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]) : "vcc");
        if (K == MIN_U32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]));
        if (K == MIN3_F32) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(a[(c + 1) % CHAINS]), "v"(a[(c + 2) % CHAINS]));
        if (K == BFE_I32) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(u[c]));
        if (K == CVT_U32_F32) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[c]) : "v"(a[c]));
        if (K == OR3) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]), "v"(u[(c + 2) % CHAINS]));
        if (K == LSHLREV) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[c]));
        if (K == FRACT) asm volatile("v_fract_f32 %0, %0" : "+v"(a[c]));
        if (K == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(p[(c + 1) % CHAINS]));
        if (K == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(u[c]) : "v"(u[(c + 1) % CHAINS]));
        if (K == BFI) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]), "v"(u[(c + 2) % CHAINS]));
        if (K == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]));
        if (K == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(a[(c + 1) % CHAINS]));
        if (K == MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[c]) : "v"(a[(c + 1) % CHAINS]));
        if (K == CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[c]) : "v"(u[c]));
        if (K == DPP_MAX) asm volatile("v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[c]));
        if (K == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]));
        if (K == OR_B32) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]));
        if (K == FMA_CHAIN1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(a[1]), "v"(a[2]));
        if (K == CMP_CND_PAIR) {
          if (c & 1)
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[(c + 2) % CHAINS]));
          else
            asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(u[c]), "v"(u[(c + 2) % CHAINS]) : "vcc");
        }
        if (K == GLOAD_UBYTE) {
          const unsigned char *gp = reinterpret_cast<const unsigned char *>(out);
          u[c] = gp[(u[c] & 0xFFFFu)] + u[(c + 1) % CHAINS];
        }
        if (K == MIX_FMA_AND) {
          if (c & 1)
            asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]));
          else
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(a[(c + 2) % CHAINS]), "v"(a[(c + 4) % CHAINS]));
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float acc = 0;
  unsigned uacc = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc += a[c] + p[c].x + p[c].y, uacc += u[c];
  out[blockIdx.x * 64 + threadIdx.x] = acc + (float)uacc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int K>
void run(int waves_per_simd, float *d_out, unsigned long long *d_cyc) {
  const int grid = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(bench<K>, dim3(grid), dim3(64), 0, 0, d_out, d_cyc, 1.0001f, 12345u);  // warm-up
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(bench<K>, dim3(grid), dim3(64), 0, 0, d_out, d_cyc, 1.0001f, 12345u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> cyc(grid);
  CHECK(hipMemcpy(cyc.data(), d_cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto c : cyc) mean += (double)c;
  mean /= grid;
  const double insts = (double)REPS * UNROLL * CHAINS;  // per wave
  // s_memtime ticks at a constant 100 MHz on gfx9-family parts?  Both figures are printed: per-wave ticks and wall time.
  printf("%-36s W=%d  wall %8.3f ms  -> %6.2f ns per wave-instruction per SIMD = %5.2f cycles at 2.4 GHz | in-kernel ticks per instruction x W: %6.3f\n",
         kind_name[K], waves_per_simd, ms, ms * 1e6 / (insts * waves_per_simd), ms * 1e6 / (insts * waves_per_simd) * 2.4,
         mean / insts / waves_per_simd);
}

template <int K>
void sweep(float *d_out, unsigned long long *d_cyc) {
  for (int w : {1, 2, 4, 8}) run<K>(w, d_out, d_cyc);
}

int main() {
  float *d_out;
  unsigned long long *d_cyc;
  CHECK(hipMalloc((void **)&d_out, sizeof(float) * 64 * 256 * 4 * 8));
  CHECK(hipMalloc((void **)&d_cyc, sizeof(unsigned long long) * 256 * 4 * 8));
  sweep<FMA>(d_out, d_cyc);
  sweep<PK_FMA>(d_out, d_cyc);
  sweep<MUL>(d_out, d_cyc);
  sweep<PK_MUL>(d_out, d_cyc);
  sweep<ADD_U32>(d_out, d_cyc);
  sweep<AND_OR>(d_out, d_cyc);
  sweep<CVT_FLR>(d_out, d_cyc);
  sweep<FLOOR>(d_out, d_cyc);
  sweep<RCP>(d_out, d_cyc);
  sweep<MED3>(d_out, d_cyc);
  sweep<CNDMASK>(d_out, d_cyc);
  sweep<MIN3>(d_out, d_cyc);
  sweep<LSHL_ADD>(d_out, d_cyc);
  sweep<PERM>(d_out, d_cyc);
  sweep<MAD_U24>(d_out, d_cyc);
  sweep<FMA_SGPR>(d_out, d_cyc);
  sweep<PK_FMA_SGPR>(d_out, d_cyc);
  sweep<READLANE>(d_out, d_cyc);
  sweep<DS_READ_U8>(d_out, d_cyc);
  sweep<MIX_FMA_AND>(d_out, d_cyc);
  sweep<CND_VCCSET>(d_out, d_cyc);
  sweep<CND_E64>(d_out, d_cyc);
  sweep<CMP_E64>(d_out, d_cyc);
  sweep<CMP_VCC>(d_out, d_cyc);
  sweep<SUB_CO>(d_out, d_cyc);
  sweep<MIN_U32>(d_out, d_cyc);
  sweep<MIN3_F32>(d_out, d_cyc);
  sweep<BFE_I32>(d_out, d_cyc);
  sweep<CVT_U32_F32>(d_out, d_cyc);
  sweep<OR3>(d_out, d_cyc);
  sweep<LSHLREV>(d_out, d_cyc);
  sweep<FRACT>(d_out, d_cyc);
  sweep<PK_ADD>(d_out, d_cyc);
  sweep<MOV>(d_out, d_cyc);
  sweep<BFI>(d_out, d_cyc);
  sweep<LSHL_OR>(d_out, d_cyc);
  sweep<ADD_F32>(d_out, d_cyc);
  sweep<MAX_F32>(d_out, d_cyc);
  sweep<CVT_F32_U32>(d_out, d_cyc);
  sweep<DPP_MAX>(d_out, d_cyc);
  sweep<AND_B32>(d_out, d_cyc);
  sweep<OR_B32>(d_out, d_cyc);
  sweep<FMA_CHAIN1>(d_out, d_cyc);
  sweep<CMP_CND_PAIR>(d_out, d_cyc);
  sweep<GLOAD_UBYTE>(d_out, d_cyc);
  return 0;
}
