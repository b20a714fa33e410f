// Do packed-fp32 vector instructions of one wave get wrong results while ANOTHER wave of the same SIMD issues MFMAs
// (or anything else)?  DESIGN 4a: the packed-fp32 build of the step loop is not reproducible with two waves per SIMD, padding
// with s_nop makes it worse (more interleaving of the two waves), and no single-wave hazard explains it.
//
// 8 waves per workgroup, one workgroup per CU: waves w and w+4 share SIMD w%4.  Waves 0-3 run role A, waves 4-7 role B.
//   role A (checker): a pinned chain of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on small integers (exact in fp32), each
//       result compared with the same arithmetic done by plain v_fma_f32; mismatches counted per (lane group, dword of the pair)
//   role B (aggressor), template AGG: 0 idle (s_sleep), 1 MFMA stream, 2 plain VALU stream, 3 packed VALU stream,
//       4 MFMA + packed mix, 5 LDS reads
// NOPS: s_nop between A's instructions (lets B's instructions in between).
// build: hipcc --offload-arch=gfx950 -O3 -o xwave_pk xwave_pk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define PK_FMA(d, a, b, c) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c))
#define PK_MUL(d, a, b) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define PK_ADD(d, a, b) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))

template <int AGG, int NOPS, int SWAP>
__global__ void __launch_bounds__(512, 2) k(int iters, unsigned* bad, float* sink) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (float)(i & 7);
  __syncthreads();
  if ((wave >= 4) == (SWAP != 0)) {  // ---- role A
    unsigned nbad[2] = {0, 0};
    for (int it = 0; it < iters; ++it) {
      const float s = (float)((it & 15) + 1 + (lane & 3));
      f32x2 a = {s, s + 1.0f}, b = {2.0f, 3.0f}, c = {1.0f, -1.0f};
      asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
      f32x2 r0, r1, r2, r3;
      // the operand forms the step loop's packed build uses most: low-dword broadcast through op_sel_hi, an SGPR operand, negated
      // operands, an inline constant
      const float sc = 3.0f;
      f32x2 sc2 = {3.0f, 5.0f};
      asm volatile("" : "+s"(sc2));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r0) : "v"(a), "v"(b), "v"(c));
      if (NOPS) asm volatile("s_nop %0" ::"n"(NOPS > 0 ? NOPS - 1 : 0));
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r1) : "v"(r0), "s"(sc2));
      if (NOPS) asm volatile("s_nop %0" ::"n"(NOPS > 0 ? NOPS - 1 : 0));
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r2) : "v"(r1), "v"(a));
      if (NOPS) asm volatile("s_nop %0" ::"n"(NOPS > 0 ? NOPS - 1 : 0));
      asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(r3) : "v"(r2), "v"(c));
      if (NOPS) asm volatile("s_nop %0" ::"n"(NOPS > 0 ? NOPS - 1 : 0));
      {
        const float q0l = __builtin_fmaf(a[0], b[0], c[0]), q0h = __builtin_fmaf(a[0], b[1], c[1]);
        const float q1l = q0l * sc, q1h = q0h * sc;
        const float q2l = q1l - a[0], q2h = q1h - a[1];
        const float q3l = __builtin_fmaf(q2l, c[0], 0.0f), q3h = __builtin_fmaf(q2l, c[1], 0.0f);
        nbad[0] += (r3[0] != q3l) + (r2[0] != q2l);
        nbad[1] += (r3[1] != q3h) + (r2[1] != q2h);
      }
    }
    for (int h = 0; h < 2; ++h)
      if (nbad[h]) atomicAdd(&bad[(lane >> 4) * 2 + h], nbad[h]);
  } else {  // ---- role B
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)1.0f; b[j] = (_Float16)(float)(lane & 3); }
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    f32x2 p = {1.0f, 2.0f}, q = {1.0f, 1.0f};
    float v = lane, t = 0.0f;
    for (int it = 0; it < iters; ++it) {
      if (AGG == 0) __builtin_amdgcn_s_sleep(2);
      if (AGG == 1 || AGG == 4) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc3, 0, 0, 0);
      }
      if (AGG == 2) {
        asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(t));
      }
      if (AGG == 3 || AGG == 4) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(q));
      }
      if (AGG == 6 || AGG == 7) {  // transcendental stream (quarter rate): exp2, rcp, sin, log
        asm volatile("v_exp_f32 %0, %1\n\tv_rcp_f32 %0, %1\n\tv_sin_f32 %0, %1\n\tv_log_f32 %0, %1" : "=&v"(v) : "v"(t));
      }
      if (AGG == 7) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc1, 0, 0, 0);
        asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(p), "+v"(t) : "v"(q));
      }
      if (AGG == 5) {
        const f32x4 l0 = *reinterpret_cast<const f32x4*>(&lds[((it * 64 + lane) * 4) & 4095]);
        t += l0[0] + l0[3];
      }
    }
    // role B checks itself too: 4 accumulators of iters * 32 * (lane&3)
    unsigned nb = 0;
    if (AGG == 1 || AGG == 4) {
      const float want = (float)iters * 32.0f * (float)(lane & 3);
      for (int r = 0; r < 4; ++r) nb += (acc0[r] != want) + (acc1[r] != want) + (acc2[r] != want) + (acc3[r] != want);
    }
    if (nb) atomicAdd(&bad[8 + (lane >> 4)], nb);
    sink[threadIdx.x] = v + t + p[0] + p[1] + acc0[0];
  }
}

template <int AGG, int NOPS, int SWAP = 0>
void run(unsigned* bad, float* sink, const char* what) {
  unsigned h[12];
  hipMemset(bad, 0, 48);
  hipLaunchKernelGGL((k<AGG, NOPS, SWAP>), dim3(256), dim3(512), 0, 0, 100000, bad, sink);
  hipDeviceSynchronize();
  hipMemcpy(h, bad, 48, hipMemcpyDeviceToHost);
  unsigned tot = 0;
  for (int i = 0; i < 12; ++i) tot += h[i];
  printf("checker in waves %s, aggressor %-22s nops %d: checker mismatches by [lane group][dword]:", SWAP ? "4-7" : "0-3", what, NOPS);
  for (int i = 0; i < 8; ++i) printf(" %u", h[i]);
  printf("   aggressor self-check by lane group:");
  for (int i = 8; i < 12; ++i) printf(" %u", h[i]);
  printf("   %s\n", tot ? "<-- WRONG RESULTS" : "clean");
}

int main() {
  unsigned* bad; float* sink;
  hipMalloc(&bad, 48); hipMalloc(&sink, 512 * 4);
  run<0, 0>(bad, sink, "idle");
  run<1, 0>(bad, sink, "mfma");          run<1, 1>(bad, sink, "mfma");          run<1, 4>(bad, sink, "mfma");
  run<2, 0>(bad, sink, "plain valu");    run<2, 2>(bad, sink, "plain valu");
  run<3, 0>(bad, sink, "packed valu");   run<3, 2>(bad, sink, "packed valu");
  run<4, 0>(bad, sink, "mfma + packed"); run<4, 1>(bad, sink, "mfma + packed"); run<4, 4>(bad, sink, "mfma + packed");
  run<5, 0>(bad, sink, "lds reads");     run<5, 2>(bad, sink, "lds reads");
  run<6, 0>(bad, sink, "transcendentals"); run<6, 2>(bad, sink, "transcendentals"); run<7, 0>(bad, sink, "trans+mfma+pk"); run<7, 2>(bad, sink, "trans+mfma+pk");
  run<0, 0, 1>(bad, sink, "idle");
  run<1, 0, 1>(bad, sink, "mfma");          run<1, 2, 1>(bad, sink, "mfma");
  run<2, 0, 1>(bad, sink, "plain valu");    run<3, 0, 1>(bad, sink, "packed valu");
  run<4, 0, 1>(bad, sink, "mfma + packed"); run<4, 2, 1>(bad, sink, "mfma + packed");
  run<5, 0, 1>(bad, sink, "lds reads");
  run<6, 0, 1>(bad, sink, "transcendentals"); run<6, 2, 1>(bad, sink, "transcendentals"); run<7, 0, 1>(bad, sink, "trans+mfma+pk"); run<7, 2, 1>(bad, sink, "trans+mfma+pk");
  return 0;
}
