// Do FP32 MFMA and FP32 VALU overlap on gfx950?  Four waves per SIMD (1024 threads, one block per CU):
// `n_mfma` waves per SIMD run an MFMA loop, the others run a dependent-free VALU fma loop.  If the pipes are
// independent, time(both) ~ max(time(mfma only), time(valu only)); if shared, ~ sum.
// build: hipcc --offload-arch=gfx950 -O3 -o pipe_share pipe_share.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define SFMA(v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c1), "v"(c2))
template <int KIND, int PK>  // KIND 0: fp32 16x16x4 MFMA, 1: f16 16x16x32 MFMA; PK 1: let the compiler pack (v_pk_fma_f32), 0: scalar v_fma_f32
__global__ void __launch_bounds__(1024, 4) k(int iters, int mfma_waves, int valu_waves, float* out) {
  const float c1 = 1.0001f, c2 = 0.5f;
  const int wave = threadIdx.x >> 6;       // 16 waves; wave w sits on SIMD (w % 4) in slot w / 4
  const int slot = wave >> 2;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
  float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  if (slot < mfma_waves) {
    const float a = v0, b = v1;
    f16x8 ah, bh;
    for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(v0 + j); bh[j] = (_Float16)(v1 - j); }
    for (int i = 0; i < iters; ++i) {
      if (KIND == 0) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc3, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc3, 0, 0, 0);
      }
    }
  } else if (slot < mfma_waves + valu_waves) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (PK) {
          v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f);
          v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f);
          v4 = __builtin_fmaf(v4, 1.0001f, 0.5f); v5 = __builtin_fmaf(v5, 1.0001f, 0.5f);
          v6 = __builtin_fmaf(v6, 1.0001f, 0.5f); v7 = __builtin_fmaf(v7, 1.0001f, 0.5f);
        } else {
          SFMA(v0); SFMA(v1); SFMA(v2); SFMA(v3); SFMA(v4); SFMA(v5); SFMA(v6); SFMA(v7);
        }
      }
    }
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

template <int KIND, int PK>
float run(int iters, int m, int v, float* out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<KIND, PK>), dim3(256), dim3(1024), 0, 0, iters, m, v, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<KIND, PK>), dim3(256), dim3(1024), 0, 0, iters, m, v, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 20000;
  printf("iters=%d per wave: 4 MFMA/iter or 32 v_fma/iter; 256 blocks x 16 waves\n", iters);
  for (int kind = 0; kind < 4; ++kind) {
    const char* name = kind == 0 ? "fp32 MFMA + v_fma   " : kind == 1 ? "f16 MFMA  + v_fma   " : kind == 2 ? "fp32 MFMA + v_pk_fma" : "f16 MFMA  + v_pk_fma";
    auto R = [&](int m, int v) {
      return kind == 0 ? run<0, 0>(iters, m, v, out) : kind == 1 ? run<1, 0>(iters, m, v, out) : kind == 2 ? run<0, 1>(iters, m, v, out) : run<1, 1>(iters, m, v, out);
    };
    printf("%s mfma waves/SIMD=2 alone      : %.3f ms\n", name, R(2, 0));
    printf("%s valu waves/SIMD=2 alone      : %.3f ms\n", name, R(0, 2));
    printf("%s 2 mfma + 2 valu waves/SIMD   : %.3f ms\n", name, R(2, 2));
    printf("%s mfma waves/SIMD=1 alone      : %.3f ms\n", name, R(1, 0));
    printf("%s 1 mfma + 3 valu waves/SIMD   : %.3f ms\n", name, R(1, 3));
    printf("%s valu waves/SIMD=3 alone      : %.3f ms\n", name, R(0, 3));
  }
  return 0;
}
