// Issue cost (cycles per instruction per wave) of the VALU ops the step loop is made of, on gfx950.
// One workgroup per CU; W waves per SIMD each run ITERS x 64 independent instances of one opcode (8 register
// chains), timed with s_memtime (shader clock) inside the wave.  Reported: cycles / instruction with 1 wave
// per SIMD (pure issue cost incl. dependency stalls of an 8-deep chain) and with 2 waves per SIMD (what each
// wave sees when the SIMD is shared: 2x means the two waves take turns).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

enum Op { FMA, PK_FMA, PK_MUL, PK_ADD, MUL_LO, MUL_HI, MAD_U64, MUL_U24, MAD_U24, EXP, LOG, SINF, SQRT, RCP, CVT_PK_F16, CVT_PKRTZ, CVT_F32_F16,
          XOR, CNDMASK, BFI, PERM, ALIGNBIT, ADD3, PK_FMA_F16, MFMA_F16, FMAMIX, CVT_F32_U32, LDEXP, MED3, NOPS };
static const char* kNames[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_u32_u24",
                               "v_mad_u32_u24", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_sqrt_f32", "v_rcp_f32", "v_cvt_pk_f16_f32", "v_cvt_pkrtz_f16_f32",
                               "v_cvt_f32_f16", "v_xor_b32", "v_cndmask_b32", "v_bfi_b32", "v_perm_b32", "v_alignbit_b32", "v_add3_u32", "v_pk_fma_f16",
                               "v_mfma_f32_16x16x32_f16", "v_fma_mix_f32", "v_cvt_f32_u32", "v_ldexp_f32", "v_med3_f32"};

template <int OP>
__global__ void __launch_bounds__(512, 1) k(int iters, long long* cyc, float* out) {
  float v[8];
  f32x2 w[8];
  unsigned long long q[8];
  f32x4 acc[8];
  f16x8 ah, bh;
  for (int j = 0; j < 8; ++j) {
    v[j] = 1.0f + threadIdx.x * 1e-3f + j;
    w[j] = f32x2{v[j], v[j] + 0.5f};
    q[j] = threadIdx.x * 977u + j;
    acc[j] = f32x4{0, 0, 0, 0};
    ah[j] = (_Float16)(v[0] + j); bh[j] = (_Float16)(v[1] - j);
  }
  const float c1 = 1.0001f, c2 = 0.5f;
  const f32x2 p1 = {1.0001f, 0.9999f}, p2 = {0.5f, 0.25f};
  const unsigned m1 = 0xD2511F53u, m2 = 0x9E3779B9u;
  __syncthreads();
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#define I1(j) if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c1), "v"(c2)); \
      else if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[j]) : "v"(p1), "v"(p2)); \
      else if (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[j]) : "v"(p1)); \
      else if (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[j]) : "v"(p2)); \
      else if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[j]) : "v"(m1)); \
      else if (OP == MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[j]) : "v"(m1)); \
      else if (OP == MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(q[j]) : "v"(v[j]), "v"(m1) : "vcc"); \
      else if (OP == MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[j]) : "v"(m1)); \
      else if (OP == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m1), "v"(m2)); \
      else if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j])); \
      else if (OP == LOG) asm volatile("v_log_f32 %0, %0" : "+v"(v[j])); \
      else if (OP == SINF) asm volatile("v_sin_f32 %0, %0" : "+v"(v[j])); \
      else if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[j])); \
      else if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[j])); \
      else if (OP == CVT_PK_F16) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c1)); \
      else if (OP == CVT_PKRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c1)); \
      else if (OP == CVT_F32_F16) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[j])); \
      else if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[j]) : "v"(m1)); \
      else if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c1) : "vcc"); \
      else if (OP == BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(v[j]) : "v"(m1), "v"(m2)); \
      else if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m1), "v"(m2)); \
      else if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(v[j])); \
      else if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m1), "v"(m2)); \
      else if (OP == PK_FMA_F16) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c1), "v"(c2)); \
      else if (OP == MFMA_F16) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[j], 0, 0, 0); \
      else if (OP == FMAMIX) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,0]" : "+v"(v[j]) : "v"(c1), "v"(c2)); \
      else if (OP == CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[j])); \
      else if (OP == LDEXP) asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(v[j])); \
      else if (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c1), "v"(c2));
      REP8(I1)
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int j = 0; j < 8; ++j) s += v[j] + w[j][0] + w[j][1] + (float)q[j] + acc[j][0] + acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(int iters, long long* cyc, float* out, long long* hcyc) {
  double res[2];
  float msv[2];
  for (int wv = 1; wv <= 2; ++wv) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256 * wv), 0, 0, iters, cyc, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256 * wv), 0, 0, iters, cyc, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&msv[wv - 1], a, b);
    hipMemcpy(hcyc, cyc, 256 * 8 * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) for (int w = 0; w < 4 * wv; ++w) s += (double)hcyc[i * 8 + w];
    res[wv - 1] = s / (256.0 * 4 * wv) / ((double)iters * 64);
  }
  // s_memtime ticks at a fixed 100 MHz on this part; wall time per instruction is the clock-independent figure
  printf("%-26s ticks/instr 1w %.4f 2w %.4f | ns/instr/wave 1w %.3f  2w %.3f  (x2.4 GHz = %.2f / %.2f cycles)\n", kNames[OP], res[0], res[1],
         msv[0] * 1e6 / ((double)iters * 64), msv[1] * 1e6 / ((double)iters * 64), msv[0] * 1e6 / ((double)iters * 64) * 2.4,
         msv[1] * 1e6 / ((double)iters * 64) * 2.4);
}

template <int OP>
void run_all(int iters, long long* cyc, float* out, long long* hcyc) {
  run<OP>(iters, cyc, out, hcyc);
  if constexpr (OP + 1 < NOPS) run_all<OP + 1>(iters, cyc, out, hcyc);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  long long* hcyc = (long long*)malloc(256 * 8 * 8);
  run_all<0>(4000, cyc, out, hcyc);
  return 0;
}
