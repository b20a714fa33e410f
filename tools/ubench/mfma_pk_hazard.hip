// Is the wait hipcc leaves between v_mfma_f32_16x16x32_f16 and a packed-fp32 (v_pk_fma_f32) or scalar consumer of
// its result enough when TWO waves share a SIMD?  Each wave repeats: acc = MFMA(1, s, 0) = 32 s exactly (s cycles
// through 1..8, so a stale accumulator is visibly different), FILL pinned independent VALU instructions, then the
// consumer reads acc and the result is compared.  Mismatches are counted per (lane group g = lane/16, register r).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_pk_hazard mfma_pk_hazard.hip   (inspect the s_nop hipcc inserts
// with -save-temps)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define FILL1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(junk) : "v"(c1), "v"(c2));

template <int FILL, int PK>
__global__ void __launch_bounds__(512, 2) k(int iters, unsigned* bad) {
  const int lane = threadIdx.x & 63;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)1.0f;
  unsigned nbad[4] = {0, 0, 0, 0};
  float junk = lane, c1 = 1.0001f, c2 = 0.5f;
  const f32x2 k1 = {2.0f, 2.0f}, k2 = {1.0f, 1.0f};
  for (int it = 0; it < iters; ++it) {
    const float s = (float)((it & 7) + 1);
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)s;
    asm volatile("" ::: "memory");
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (FILL >= 1) { FILL1 } if (FILL >= 2) { FILL1 } if (FILL >= 3) { FILL1 } if (FILL >= 4) { FILL1 }
    if (FILL >= 5) { FILL1 } if (FILL >= 6) { FILL1 } if (FILL >= 7) { FILL1 } if (FILL >= 8) { FILL1 }
    if (FILL >= 9) { FILL1 } if (FILL >= 10) { FILL1 } if (FILL >= 11) { FILL1 } if (FILL >= 12) { FILL1 }
    float o[4];
    if (PK) {
      const f32x2 lo = __builtin_elementwise_fma(f32x2{acc[0], acc[1]}, k1, k2);
      const f32x2 hi = __builtin_elementwise_fma(f32x2{acc[2], acc[3]}, k1, k2);
      o[0] = lo[0]; o[1] = lo[1]; o[2] = hi[0]; o[3] = hi[1];
    } else {
      for (int r = 0; r < 4; ++r) o[r] = __builtin_fmaf(acc[r], 2.0f, 1.0f);
    }
    const float want = 64.0f * s + 1.0f;
    for (int r = 0; r < 4; ++r) nbad[r] += (o[r] != want);
  }
  for (int r = 0; r < 4; ++r)
    if (nbad[r]) atomicAdd(&bad[(lane >> 4) * 4 + r], nbad[r]);
  if (junk == 1234.5f) bad[0] = 0xFFFFFFFFu;
}

template <int FILL, int PK>
void run(unsigned* bad, int waves) {
  unsigned h[16];
  hipMemset(bad, 0, 64);
  hipLaunchKernelGGL((k<FILL, PK>), dim3(256), dim3(64 * waves), 0, 0, 200000, bad);
  hipDeviceSynchronize();
  hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost);
  unsigned tot = 0;
  for (int i = 0; i < 16; ++i) tot += h[i];
  printf("fill=%2d %s consumer, %d waves/WG (%d per SIMD): mismatches %u  by [g][r]:", FILL, PK ? "v_pk_fma_f32" : "v_fma_f32   ", waves, waves / 4, tot);
  for (int i = 0; i < 16; ++i) printf(" %u", h[i]);
  printf("\n");
}

int main() {
  unsigned* bad; hipMalloc(&bad, 64);
  for (int waves : {4, 8}) {
    run<0, 1>(bad, waves); run<4, 1>(bad, waves); run<8, 1>(bad, waves); run<12, 1>(bad, waves);
    run<0, 0>(bad, waves); run<4, 0>(bad, waves); run<8, 0>(bad, waves); run<12, 0>(bad, waves);
  }
  return 0;
}
