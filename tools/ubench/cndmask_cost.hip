// What does v_cndmask_b32 cost on gfx950?  valu_cost.hip measured 4x a plain VALU op for the vcc form; this separates the forms:
//   0 v_xor_b32 (baseline)   1 v_cndmask vcc, vcc never written   2 v_cndmask e64 with an SGPR-pair mask
//   3 v_cmp_gt_f32 vcc + v_cndmask vcc (pair)   4 v_cmp e64 -> sgpr pair + v_cndmask e64 (pair)   5 v_max_f32   6 v_med3_f32   7 v_cmp only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
static const char* kNames[] = {"v_xor_b32", "v_cndmask vcc (static)", "v_cndmask e64 sgpr (static)", "v_cmp vcc + v_cndmask vcc", "v_cmp e64 + v_cndmask e64",
                               "v_max_f32", "v_med3_f32", "v_cmp_gt_f32 vcc only", "v_cmp + 2 cndmask (vcc)", "v_cmp + 4 cndmask (vcc)",
                               "v_cmp vcc, 1 xor, v_cndmask vcc", "v_cmp vcc, 2 xor, v_cndmask vcc", "v_cmp vcc, 4 xor, v_cndmask vcc", "v_cmp e64, 2 xor, v_cndmask e64",
                               "v_cndmask_e64 ..., vcc (static)", "s_mov vcc + v_cndmask vcc", "v_cmp vcc, s_nop 3, v_cndmask vcc", "v_cmp + cndmask vcc + cndmask_e64 vcc",
                               "v_cmp vcc, s_nop 0, v_cndmask e32", "v_cmp vcc, s_nop 1, v_cndmask e32", "v_cmp vcc, s_nop 1, v_cndmask_e64 vcc", "v_cmp e64 sgpr, s_nop 1, v_cndmask_e64 sgpr",
                               "v_cmp vcc, xor, s_nop 0, v_cndmask e32", "v_cmp vcc, s_nop 1, 2x v_cndmask_e64 vcc"};
constexpr int NOPS = 24;
#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
template <int OP>
__global__ void __launch_bounds__(512, 1) k(int iters, float* out, unsigned long long mask) {
  float v[8], z[8];
  for (int j = 0; j < 8; ++j) { v[j] = 1.0f + threadIdx.x * 1e-3f + j; z[j] = v[j] * 0.5f; }
  const float c1 = 1.0001f, c2 = 0.5f;
  const unsigned m1 = 0xD2511F53u;
  unsigned long long sm = __builtin_amdgcn_readfirstlane((unsigned)mask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mask >> 32)) << 32);
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#define I1(j) if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[j]) : "v"(m1)); \
      else if (OP == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c1)); \
      else if (OP == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c1), "s"(sm)); \
      else if (OP == 3) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c1) : "vcc"); \
      else if (OP == 4) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[j]) : "v"(c1) : "s20", "s21"); \
      else if (OP == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c1)); \
      else if (OP == 6) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c1), "v"(c2)); \
      else if (OP == 7) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[j]), "v"(c1) : "vcc"); \
      else if (OP == 8) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(v[j]), "+v"(z[j]) : "v"(c1) : "vcc"); \
      else if (OP == 9) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(v[j]), "+v"(z[j]) : "v"(c1) : "vcc"); \
      else if (OP == 10) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_xor_b32 %1, %1, %3\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(v[j]), "+v"(z[j]) : "v"(c1), "v"(m1) : "vcc"); \
      else if (OP == 11) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %3\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(v[j]), "+v"(z[j]) : "v"(c1), "v"(m1) : "vcc"); \
      else if (OP == 12) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %3\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(v[j]), "+v"(z[j]) : "v"(c1), "v"(m1) : "vcc"); \
      else if (OP == 13) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %2\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %3\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(v[j]), "+v"(z[j]) : "v"(c1), "v"(m1) : "s20", "s21"); \
      else if (OP == 14) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c1)); \
      else if (OP == 15) asm volatile("s_mov_b64 vcc, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c1), "s"(sm) : "vcc"); \
      else if (OP == 16) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\ts_nop 3\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c1) : "vcc"); \
      else if (OP == 17) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32_e64 %1, %1, %2, vcc" : "+v"(v[j]), "+v"(z[j]) : "v"(c1) : "vcc"); \
      else if (OP == 18) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\ts_nop 0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c1) : "vcc"); \
      else if (OP == 19) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c1) : "vcc"); \
      else if (OP == 20) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c1) : "vcc"); \
      else if (OP == 21) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[j]) : "v"(c1) : "s20", "s21"); \
      else if (OP == 22) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_xor_b32 %1, %1, %3\n\ts_nop 0\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(v[j]), "+v"(z[j]) : "v"(c1), "v"(m1) : "vcc"); \
      else if (OP == 23) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %2, vcc\n\tv_cndmask_b32_e64 %1, %1, %2, vcc" : "+v"(v[j]), "+v"(z[j]) : "v"(c1) : "vcc");
      REP8(I1)
    }
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) s += v[j] + z[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run_all(int iters, float* out) {
  float ms[2];
  for (int wv = 1; wv <= 2; ++wv) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256 * wv), 0, 0, iters, out, 0x5555aaaa5555aaaaull);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256 * wv), 0, 0, iters, out, 0x5555aaaa5555aaaaull);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms[wv - 1], a, b);
  }
  printf("%-30s ns per asm statement per wave: 1 wave/SIMD %.3f   2 waves/SIMD %.3f (SIMD time per statement %.3f)\n", kNames[OP],
         ms[0] * 1e6 / ((double)iters * 64), ms[1] * 1e6 / ((double)iters * 64), ms[1] * 1e6 / ((double)iters * 64) / 2);
  if constexpr (OP + 1 < NOPS) run_all<OP + 1>(iters, out);
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  run_all<0>(4000, out);
  return 0;
}
