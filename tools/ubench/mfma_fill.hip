// What do vector instructions cost when they sit BETWEEN a wave's own MFMAs (v_mfma_f32_16x16x32_f16, 16 cycles each) on gfx950?
// MI355X_MICROARCH.md (constants table, 'vector-instruction ISSUE cost'): an MFMA holds the SIMD's vector issue for 8 of its 16 cycles,
// so up to ~2 plain VALU per MFMA should be nearly free in the SAME wave -- while profiles/r01_ubench_mfma_valu_serialize.log shows that
// the VALU stream of ANOTHER wave of the SIMD is blocked for ~12 of the 16 cycles.  This measures both placements with exact instruction
// order (every instruction is an asm volatile statement; fillers write registers no MFMA reads, so no software wait states are needed).
//
//   per iteration: 4 x [ MFMA (4 rotating accumulators) ; F fillers ]          W waves per SIMD all run this stream, or
//                                                                              (PARTNER) wave slot 0 runs it and slot 1 runs 4 F fillers only
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_fill mfma_fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define MFMA(acc) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah), "v"(bh))
#define FMA(v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c1), "v"(c2))
#define EXP(v) asm volatile("v_exp_f32 %0, %0" : "+v"(v))

template <int F, int TRANS, int MODE>  // MODE 0: every wave runs MFMA + fillers; 1: slot 0 MFMAs only, slot 1 the fillers only; 2: fillers only (all waves)
__global__ void __launch_bounds__(1024) k(int iters, float* out) {
  const float c1 = 1.0001f, c2 = 0.5f;
  const int slot = (threadIdx.x >> 6) >> 2;  // wave w sits on SIMD w % 4, slot w / 4
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3f + j;
  f16x8 ah, bh;
  for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(v[0] + j); bh[j] = (_Float16)(v[1] - j); }
  const bool do_mfma = MODE == 0 || (MODE == 1 && slot == 0);
  const bool do_fill = MODE == 0 || MODE == 2 || (MODE == 1 && slot != 0);
  if (do_mfma && do_fill) {
    for (int i = 0; i < iters; ++i) {
#define GAP(acc, base)                                                   \
  MFMA(acc);                                                             \
  _Pragma("unroll") for (int f = 0; f < F; ++f) {                        \
    if (TRANS && f == 0) EXP(v[(base + f) & 7]); else FMA(v[(base + f) & 7]); \
  }
      GAP(a0, 0) GAP(a1, 2) GAP(a2, 4) GAP(a3, 6)
    }
  } else if (do_mfma) {
    for (int i = 0; i < iters; ++i) { MFMA(a0); MFMA(a1); MFMA(a2); MFMA(a3); }
  } else if (do_fill) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int f = 0; f < F; ++f) {
          if (TRANS && f == 0) EXP(v[(2 * q + f) & 7]); else FMA(v[(2 * q + f) & 7]);
        }
    }
  }
  float s = a0[0] + a1[1] + a2[2] + a3[3];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int F, int TRANS, int MODE>
float run(int iters, int waves_per_simd, float* out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int threads = 256 * waves_per_simd;
  hipLaunchKernelGGL((k<F, TRANS, MODE>), dim3(256), dim3(threads), 0, 0, iters, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<F, TRANS, MODE>), dim3(256), dim3(threads), 0, 0, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e6f / iters / 4.0f;  // ns per [MFMA + F fillers] group and wave
}

template <int F, int TRANS>
void row(int iters, float* out) {
  printf("F=%d%s | same wave, 1 wave/SIMD: %6.2f ns/gap | same wave, 2 waves/SIMD: %6.2f | partner wave (1 MFMA wave + 1 filler wave): %6.2f | "
         "fillers alone 1 wave: %6.2f, 2 waves: %6.2f\n", F, TRANS ? " (first filler v_exp_f32)" : "",
         run<F, TRANS, 0>(iters, 1, out), run<F, TRANS, 0>(iters, 2, out), run<F, TRANS, 1>(iters, 2, out),
         run<F, TRANS, 2>(iters, 1, out), run<F, TRANS, 2>(iters, 2, out));
}

int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 20000;
  printf("ns per gap = one v_mfma_f32_16x16x32_f16 + F v_fma_f32 fillers (per wave; with 2 waves/SIMD the SIMD executes two gaps in that time)\n");
  row<0, 0>(iters, out); row<1, 0>(iters, out); row<2, 0>(iters, out); row<3, 0>(iters, out); row<4, 0>(iters, out);
  row<6, 0>(iters, out); row<8, 0>(iters, out); row<12, 0>(iters, out); row<16, 0>(iters, out);
  row<2, 1>(iters, out); row<4, 1>(iters, out); row<8, 1>(iters, out);
  return 0;
}
